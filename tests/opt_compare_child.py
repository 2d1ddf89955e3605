#!/usr/bin/env python3
"""Child process of tests/test_gpu_parity.py::test_shipped_build_matches_O1_build: solves one small batch per kernel
instantiation (tile count T = 1..12, border width 0 / 1 / 4) with the library named by FSAEMPC_LIB and stores exit flags,
iteration counts and x.  usage: opt_compare_child.py <tag> <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def shapes():
    """(key, model, N) per instantiation: nV = 2N+1 (kinematic) / 2N+4 (dynamic); T = nV // 16 with a border of nV % 16 in
    1..4, else ceil(nV / 16) without border -- or, with the slack-border policy of qp_make_dims, the slack columns as the border
    behind a core padded with dummy variables (the _pad keys)."""
    out = []
    for T in range(1, 13):
        out.append(("T%d_nb1" % T, 0, 8 * T))                       # nV = 16T + 1
        if T <= 10:
            out.append(("T%d_nb4" % T, 1, 8 * T))                   # nV = 16T + 4
        out.append(("T%d_nb0" % T, 0, 8 * T - 1) if T > 1 else ("T1_nb0", 0, 4))   # nV = 16T - 1 (T = 1: nV = 9); solved with the slack-border policy off
        if T > 1:
            out.append(("T%d_pad1" % T, 0, 8 * T - 1))              # the same shape with the policy on: slack column as the border, 1 dummy variable
        if 2 <= T <= 9:
            out.append(("T%d_pad4" % T, 1, 8 * T - 3))              # nV = 16T - 2, dynamic: 4 slack columns as the border, 6 dummy variables
    return out


def main():
    tag, out_path = sys.argv[1], sys.argv[2]
    import torch
    import fsae_mpc_amd as fm
    import oracle as orc
    otr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    res = {}
    B = 6
    for key, model, N in shapes():
        x0, xl, ul, xr = fm.instances(model, N, 0.05, otr.L, 515, range(B))
        q = orc.build_qp_batch(model, otr, N, 0.05, x0, xr, xl, ul)
        os.environ["FSAEMPC_SLACK_BORDER"] = "0" if key.endswith("_nb0") else "2"   # (read by qp_make_dims at every call; 2 = also where it does not pay)
        o = fm.qp_solve_batch_device(*(dev(q[k]) for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")))
        torch.cuda.synchronize()
        res[key + "_fl"] = o["exitflag"].cpu().numpy(); res[key + "_it"] = o["iter"].cpu().numpy(); res[key + "_x"] = o["x"].cpu().numpy()
    # T = 11, 12 with a 4-column border (nV = 180, 196): the dynamic horizons of that size have more rows than the LDS budget
    # of the workgroup kernel allows, use random SPD data
    rng = np.random.default_rng(5)
    for key, n, m in (("T11_nb4", 180, 240), ("T12_nb4", 196, 260)):
        Q = rng.normal(size=(n, n)); H = Q @ Q.T + n * np.eye(n); g = rng.normal(size=n) * 10
        A = rng.normal(size=(m, n)); xs = rng.normal(size=n)
        lbA = A @ xs - rng.uniform(0.1, 1, m); ubA = A @ xs + rng.uniform(0.1, 1, m)
        o = fm.qp_solve_batch_device(dev(H[None]), dev(g[None]), dev(A.T[None].copy()), dev((xs - 1)[None]), dev((xs + 1)[None]), dev(lbA[None]), dev(ubA[None]))
        torch.cuda.synchronize()
        res[key + "_fl"] = o["exitflag"].cpu().numpy(); res[key + "_it"] = o["iter"].cpu().numpy(); res[key + "_x"] = o["x"].cpu().numpy()
    np.savez(out_path, **res)
    print(tag, "ok", len(res) // 3, "instantiations")


if __name__ == "__main__":
    main()
