import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in range(0, int(sys.argv[3]) if len(sys.argv) > 3 else 26):
    xa, xb = a["x%d" % k], b["x%d" % k]; la, lb = a["l%d" % k], b["l%d" % k]
    print("max_iter %2d | a flag/iter/kkt/fval %s | b %s | dx %.3e dlam %.3e" % (k, np.array2string(a["f%d" % k], precision=3), np.array2string(b["f%d" % k], precision=3), np.abs(xa - xb).max(), np.abs(la - lb).max()))
