#!/usr/bin/env python3
"""One shape of the O1 guard (tests/opt_compare_child.py: seed 515, 6 instances) against the oracle, for the library in FSAEMPC_LIB.
usage: t8_check.py [model=0] [N=64] [B=6]"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
import fsae_mpc_amd as fm
import oracle as orc
model = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 6
otr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, xl, ul, xr = fm.instances(model, N, 0.05, otr.L, 515, range(B))
q = orc.build_qp_batch(model, otr, N, 0.05, x0, xr, xl, ul)
K = ("H", "g", "A", "lb", "ub", "lbA", "ubA")
ref = orc.qp_solve_batch_aux(*(q[k] for k in K))
for rep in range(3):
    o = fm.qp_solve_batch_device(*(dev(q[k]) for k in K), want_aux=True)
    torch.cuda.synchronize()
    ex = np.abs(o["x"].cpu().numpy() - ref["x"]).max(axis=1) / np.maximum(1, np.abs(ref["x"]).max(axis=1))
    print(os.path.basename(fm._lib.LIB_PATH), "rep", rep, "flags", o["exitflag"].cpu().numpy(), "iters", o["iter"].cpu().numpy(), "oracle", ref["exitflag"], ref["iter"], "x err", np.array2string(ex, precision=1), flush=True)
