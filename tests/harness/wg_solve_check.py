#!/usr/bin/env python3
"""Development check (dump build of the workgroup kernel, -DQP_DEBUG_DUMP): the normal matrix of iteration `it` of instance 0 and the
two step directions the kernel solved for, against numpy.  usage: wg_solve_check.py [model=1] [N=60] [it=0]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
import fsae_mpc_amd as fm
model = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
it = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tr = fm.Track.load("fsg2019")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(2))
q = fm.LtvBatch(model, N, 0.05, tr, 2).build_qp(dev(x0), dev(xr), dev(xl), dev(ul))
nu = q["g"].shape[1]
n = 16 * ((nu - 4 + 15) // 16) + 4 if (nu % 16 == 0 or nu % 16 > 4) else nu      # the solver's n with the slack border policy (dynamic model)
print("nu", nu, "solver n", n)
dump = torch.zeros(4 * n * n + 64 * n, dtype=torch.float64, device="cuda")
fm.lib().fsaempc_debug_set_dump(C.c_void_p(dump.data_ptr()), 1 | (it << 8))
o = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")), want_aux=True)
torch.cuda.synchronize()
fm.lib().fsaempc_debug_set_dump(None, 0)
d = dump.cpu().numpy()
M = d[:n * n].reshape(n, n)
r1, r2 = d[n * n + 4 * n:n * n + 5 * n], d[n * n + 5 * n:n * n + 6 * n]
s1, s2 = d[n * n + 6 * n:n * n + 7 * n], d[n * n + 7 * n:n * n + 8 * n]
print("M symmetric:", np.abs(M - M.T).max(), " diag min", np.diag(M).min(), "flags", o["exitflag"].cpu().numpy(), "iters", o["iter"].cpu().numpy())
for name, r, sol in (("R1", r1, s1), ("R2", r2, s2)):
    ref = np.linalg.solve(M, r)
    print(name, "|rhs| %.3e |sol| %.3e  max|sol - numpy| / |numpy| = %.3e   residual |M sol - rhs|/|rhs| = %.3e" % (np.abs(r).max(), np.abs(sol).max(), np.abs(sol - ref).max() / np.abs(ref).max(), np.abs(M @ sol - r).max() / np.abs(r).max()))
    bad = np.argsort(-np.abs(sol - ref))[:8]
    print("   worst entries", bad, (sol - ref)[bad])

np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r3h", "solve_dump.npz"), M=M, r1=r1, r2=r2, s1=s1, s2=s2)
