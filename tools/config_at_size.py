#!/usr/bin/env python3
"""A BASELINE.json configuration at its full size on one GPU: `total` DISTINCT instances in chunks through the fused entry
(x0, x_ref, x_lin, u_lin -> u_opt, x_opt: linearise + condense + QP build + solve + post-solve per chunk).  All chunk inputs are
resident in HBM before the timed region; the stepper's QP tensors and workspace are reused from chunk to chunk.
--streams 2 (default): chunks alternate between two steppers, each on its own HIP stream with its own workspace, so the construction
and prep kernels of chunk c+1 fill the SIMDs the tail of chunk c's solve leaves idle (one wavefront / workgroup per QP: a batch ends
with its slowest instances).  --streams 1 = one chunk after the other (the round-2 / early round-3 figures).
Default = configs[2]: 65,536 dynamic-model QPs, N = 60 (nV = 124, nC = 1200).  Prints one JSON line.
usage: tools/config_at_size.py [--model dynamic] [--horizon 60] [--total 65536] [--chunk 16384] [--streams 2]"""
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
    ap.add_argument("--chunk", type=int, default=16384, help="instances per launch (sized for 288 GB of HBM: 16,384 dynamic N = 60 QPs take 53 GB per stepper)")
    ap.add_argument("--seed", type=int, default=20190)
    ap.add_argument("--streams", type=int, default=2, help="HIP streams (and steppers / workspaces) the chunks alternate between")
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
    S = max(1, a.streams)
    sts = [fm.LtvBatch(model, N, dt, tr, a.chunk, device=dev) for _ in range(S)]
    streams = [torch.cuda.Stream(dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    for st, sm in zip(sts, streams):                             # warm-up (code load, allocations)
        with torch.cuda.stream(sm):
            st.step(*chunks[0], stream=sm.cuda_stream)
    torch.cuda.synchronize(dev)
    outs = []
    t0 = time.perf_counter()
    for c, ch in enumerate(chunks):
        sm = streams[c % S]
        with torch.cuda.stream(sm):                              # (the step's output tensors belong to this stream)
            outs.append(sts[c % S].step(*ch, stream=sm.cuda_stream))
    torch.cuda.synchronize(dev)
    t = time.perf_counter() - t0
    flags = [o["exitflag"] for o in outs]; iters = [o["iter"] for o in outs]
    fl = torch.cat(flags).cpu().numpy()[:a.total]; it = torch.cat(iters).cpu().numpy()[:a.total]
    solved = int((fl == 0).sum())
    print(json.dumps({"metric": "QP solves/sec (fused mode, %s N=%d, fp64)" % (a.model, N), "value": solved / t, "unit": "QP solves/s", "n_gpus": 1,
                      "seconds_total": t, "ms_per_chunk": 1e3 * t / nch, "streams": S, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": "%d distinct instances (ids 0..%d, seed %d, fsg2019) in %d chunks of %d, nV=%d nC=%d; fused entry "
                                             "(construction + solve + post-solve), chunk inputs resident in HBM" % (a.total, a.total - 1, a.seed, nch, a.chunk, nV, nC),
                                 "solved": solved, "exitflag_histogram": {int(k): int(v) for k, v in zip(*np.unique(fl, return_counts=True))},
                                 "mean_ipm_iterations": float(it.mean()), "max_ipm_iterations": int(it.max())}}))


if __name__ == "__main__":
    main()
