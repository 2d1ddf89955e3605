#!/usr/bin/env python3
"""Diagnostic (QP_PROBE build): cycles of pass 1 alone vs inside the solver (compare with tools/phase_profile.py)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tr = fm.Track.load("fsg2019")
x0, xl, ul, xr = fm.instances(fm.KINEMATIC, 40, 0.05, tr.L, 20190, range(B))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
q = fm.LtvBatch(fm.KINEMATIC, 40, 0.05, tr, B).build_qp(up(x0), up(xr), up(xl), up(ul))
dump = torch.zeros(2 * B + 64, dtype=torch.float64, device="cuda")
fm.lib().fsaempc_debug_set_dump(C.c_void_p(dump.data_ptr()), 8)
for _ in range(2):
    out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")))
    torch.cuda.synchronize()
fm.lib().fsaempc_debug_set_dump(None, 0)
d = dump.cpu().numpy()
print("pass 1 alone: %.0f cycles/pass (mean over %d QPs), min %.0f max %.0f ; checksum[0] %.6e" % (d[:B].mean(), B, d[:B].min(), d[:B].max(), d[B]))
