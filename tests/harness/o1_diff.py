#!/usr/bin/env python3
"""Dump (stage 1: M, p1..p3, Hx, R1, R2, the two step directions) of one iteration of one instance, to diff two builds.
usage: FSAEMPC_LIB=... o1_diff.py out.npy [instance=2] [iteration=0] [model=0] [N=64]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
import fsae_mpc_amd as fm
out, inst, it = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2, int(sys.argv[3]) if len(sys.argv) > 3 else 0
model = int(sys.argv[4]) if len(sys.argv) > 4 else 0
N = int(sys.argv[5]) if len(sys.argv) > 5 else 64
tr = fm.Track.load("fsg2019")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 515, np.array([inst]))
q = fm.LtvBatch(model, N, 0.05, tr, 1).build_qp(dev(x0), dev(xr), dev(xl), dev(ul))
n = 16 * 9 + 8
dump = torch.zeros(4 * n * n + 16 * n, dtype=torch.float64, device="cuda")
fm.lib().fsaempc_debug_set_dump(C.c_void_p(dump.data_ptr()), 1 | (it << 8))
o = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")), want_aux=True)
torch.cuda.synchronize()
fm.lib().fsaempc_debug_set_dump(None, 0)
print(os.path.basename(fm._lib.LIB_PATH), "flag", o["exitflag"].item(), "iter", o["iter"].item())
np.save(out, dump.cpu().numpy())
