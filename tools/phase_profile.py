#!/usr/bin/env python3
"""Per-phase cycle shares of qp_solve_kernel from the diagnostic build (bash tools/build_exp.sh 5 -DQP_STAMPS=1; FSAEMPC_QP_V1=1 selects the phase names of the one-wavefront kernel).
Never quote this build's run time: read the SHARES."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("FSAEMPC_LIB", os.path.join(ROOT, "fsae-mpc_amd", "lib", "libfsaempc_exp.so"))   # bash tools/build_exp.sh 5 -DQP_STAMPS=1
import torch  # noqa: E402
import fsae_mpc_amd as fm  # noqa: E402

NAMES_V1 = ["setup", "row1", "hx", "syrk", "resid+toLDS", "chol", "solve2", "passAv2", "row2", "passAtw", "solve1", "passAv1", "row3+update", "epilogue"]
NAMES_WG = ["setup", "row1", "hx", "acc_init+syrk", "resid+rhs", "diag add+dmax", "chol (K loop)", "border+backward", "pass2 (fused)", "row2", "corrector solve", "pass3", "row3a+alpha", "update sweep", "epilogue + refinement", "(of syrk: per-trip wait + barrier)"]
NAMES = NAMES_V1 if os.environ.get("FSAEMPC_QP_V1") else NAMES_WG


def main():
    model = fm.KINEMATIC if (len(sys.argv) < 2 or sys.argv[1] == "kin") else fm.DYNAMIC
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    tr = fm.Track.load("fsg2019")
    x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    q = fm.LtvBatch(model, N, 0.05, tr, B).build_qp(up(x0), up(xr), up(xl), up(ul))
    dump = torch.zeros(B * 16, dtype=torch.float64, device="cuda")
    fm.lib().fsaempc_debug_set_dump(C.c_void_p(dump.data_ptr()), 9)
    out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")))
    torch.cuda.synchronize()
    d = dump.cpu().numpy().reshape(B, 16)
    it = out["iter"].cpu().numpy()
    tot = d.sum(axis=1)
    print("instances %d  mean iters %.2f  mean cycles/QP %.3e  cycles/iter %.3e" % (B, it.mean(), tot.mean(), tot.sum() / max(1, it.sum())))
    sh = d.sum(axis=0) / d.sum()
    for i, nme in enumerate(NAMES):
        print("  %-14s %6.2f %%   %.3e cycles/iter" % (nme, 100 * sh[i], d[:, i].sum() / max(1, it.sum())))


if __name__ == "__main__":
    main()
