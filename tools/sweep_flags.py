#!/usr/bin/env python3
"""Robustness sweep: exit-flag / iteration tally of the solve over instance ids [lo, hi) of one shape (generic mode, product options).
usage: tools/sweep_flags.py <kin|dyn> <N> <lo> <hi> [chunk=4096]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm


def main():
    model = fm.KINEMATIC if sys.argv[1] == "kin" else fm.DYNAMIC
    N, lo, hi = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
    tr = fm.Track.load("fsg2019")
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    st = fm.LtvBatch(model, N, 0.05, tr, chunk)
    tally, its, vert, bad, nonfin = {}, [], [], [], 0
    for a in range(lo, hi, chunk):
        ids = np.arange(a, a + chunk)
        x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, ids)
        q = st.build_qp(up(x0), up(xr), up(xl), up(ul))
        o = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")), want_aux=True)
        fl = o["exitflag"].cpu().numpy(); n = min(chunk, hi - a)
        for f in np.unique(fl[:n]):
            tally[int(f)] = tally.get(int(f), 0) + int((fl[:n] == f).sum())
        its.append(o["iter"].cpu().numpy()[:n]); vert.append(o["polished"].cpu().numpy()[:n] > 0)
        bad += [int(i) for i in ids[:n][fl[:n] != 0]]
        nonfin += int((~torch.isfinite(o["x"][:n]).all(1)).sum().item())   # (a flag-0 instance must never carry a NaN)
    its = np.concatenate(its); vert = np.concatenate(vert)
    print(json.dumps({"shape": "%s N=%d ids %d..%d" % (sys.argv[1], N, lo, hi - 1), "exitflags": tally, "nonzero_ids": bad[:20],
                      "mean_iter": float(its.mean()), "max_iter": int(its.max()), "on_vertex": float(vert.mean()), "instances_with_nonfinite_x": nonfin}))


if __name__ == "__main__":
    main()
