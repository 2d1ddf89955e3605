#!/usr/bin/env python3
"""Diagnostic: dump solver internals of instance 0 (stage 1: M,p1..p3,Hx at iteration IT; stage 2: both solutions) to a file."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm
tag, it = sys.argv[1], int(sys.argv[2])
model = fm.DYNAMIC if (len(sys.argv) > 3 and sys.argv[3] == "dyn") else fm.KINEMATIC
N = int(sys.argv[4]) if len(sys.argv) > 4 else 40
B = 2
tr = fm.Track.load("fsg2019")
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
q = fm.LtvBatch(model, N, 0.05, tr, B).build_qp(up(x0), up(xr), up(xl), up(ul))
n, m = q["g"].shape[1], q["lbA"].shape[1]
res = {}
for stage in (1, 2, 4):
    dump = torch.zeros(4 * n * n + 8 * (n + m), dtype=torch.float64, device="cuda")
    fm.lib().fsaempc_debug_set_dump(C.c_void_p(dump.data_ptr()), stage | (it << 8))
    # dump_iter is fixed at 0 unless the library exposes it; stage 1/2 use P.dump_iter
    out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")))
    torch.cuda.synchronize()
    fm.lib().fsaempc_debug_set_dump(None, 0)
    res["s%d" % stage] = dump.cpu().numpy()
res["iter"] = out["iter"].cpu().numpy(); res["flag"] = out["exitflag"].cpu().numpy()
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/dump_%s_%d.npz" % (tag, it), **res)
print(tag, "iters", res["iter"], "flags", res["flag"])
