#!/usr/bin/env python3
"""A BASELINE.json configuration at its full size on one GPU: `total` DISTINCT instances in chunks through the fused entry
(x0, x_ref, x_lin, u_lin -> u_opt, x_opt: linearise + condense + QP build + solve + post-solve per chunk).  All chunk inputs are
resident in HBM before the timed region; the stepper's QP tensors and workspace are reused from chunk to chunk.
Default = configs[2]: 65,536 dynamic-model QPs, N = 60 (nV = 124, nC = 1200).  Prints one JSON line.
usage: tools/config_at_size.py [--model dynamic] [--horizon 60] [--total 65536] [--chunk 4096]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dynamic", choices=["kinematic", "dynamic"])
    ap.add_argument("--horizon", type=int, default=60)
    ap.add_argument("--total", type=int, default=65536)
    ap.add_argument("--chunk", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=20190)
    a = ap.parse_args()
    model = fm.KINEMATIC if a.model == "kinematic" else fm.DYNAMIC
    N, dt = a.horizon, 0.05
    nx, ns, nV, nC = fm.dims(model, N)
    tr = fm.Track.load("fsg2019")
    dev = torch.device("cuda", 0)
    up = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    nch = (a.total + a.chunk - 1) // a.chunk
    chunks = []
    for c in range(nch):
        ids = np.arange(c * a.chunk, min(a.total, (c + 1) * a.chunk))
        if len(ids) < a.chunk:                                   # ragged tail: pad with repeats, counted out below
            ids = np.concatenate([ids, np.full(a.chunk - len(ids), ids[-1])])
        x0, xl, ul, xr = fm.instances(model, N, dt, tr.L, a.seed, ids)
        chunks.append((up(x0), up(xr), up(xl), up(ul)))
    st = fm.LtvBatch(model, N, dt, tr, a.chunk, device=dev)
    st.step(*chunks[0]); torch.cuda.synchronize(dev)             # warm-up (code load, allocations)
    flags, iters = [], []
    t0 = time.perf_counter()
    for ch in chunks:
        o = st.step(*ch)
        flags.append(o["exitflag"].clone()); iters.append(o["iter"].clone())
    torch.cuda.synchronize(dev)
    t = time.perf_counter() - t0
    fl = torch.cat(flags).cpu().numpy()[:a.total]; it = torch.cat(iters).cpu().numpy()[:a.total]
    solved = int((fl == 0).sum())
    print(json.dumps({"metric": "QP solves/sec (fused mode, %s N=%d, fp64)" % (a.model, N), "value": solved / t, "unit": "QP solves/s", "n_gpus": 1,
                      "seconds_total": t, "ms_per_chunk": 1e3 * t / nch, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": "%d distinct instances (ids 0..%d, seed %d, fsg2019) in %d chunks of %d, nV=%d nC=%d; fused entry "
                                             "(construction + solve + post-solve), chunk inputs resident in HBM" % (a.total, a.total - 1, a.seed, nch, a.chunk, nV, nC),
                                 "solved": solved, "exitflag_histogram": {int(k): int(v) for k, v in zip(*np.unique(fl, return_counts=True))},
                                 "mean_ipm_iterations": float(it.mean()), "max_ipm_iterations": int(it.max())}}))


if __name__ == "__main__":
    main()
