import sys, numpy as np
n = int(sys.argv[3]) if len(sys.argv) > 3 else 129
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
def seg(d, k): return d[n * n + k * n: n * n + (k + 1) * n]
Ma, Mb = a[: n * n].reshape(n, n), b[: n * n].reshape(n, n)
dM = np.abs(Ma - Mb); i, j = np.unravel_index(dM.argmax(), dM.shape)
print("M max diff %.3e at (%d,%d) rel to |M|max %.3e | nan a %d b %d" % (dM.max(), i, j, np.abs(Mb).max(), np.isnan(Ma).sum(), np.isnan(Mb).sum()))
rows = np.where(dM.max(axis=1) > 1e-9 * np.abs(Mb).max())[0]; cols = np.where(dM.max(axis=0) > 1e-9 * np.abs(Mb).max())[0]
print("rows with diff:", rows[:40], "cols:", cols[:40])
for k, nm in enumerate(("p1", "p2", "p3", "Hx", "R1 rhs", "R2 rhs", "R1 sol", "R2 sol")):
    d = np.abs(seg(a, k) - seg(b, k)); print("%-7s max diff %.3e (max |b| %.3e) at %d" % (nm, d.max(), np.abs(seg(b, k)).max(), d.argmax()), np.where(d > 1e-9 * max(1e-300, np.abs(seg(b, k)).max()))[0][:20])
