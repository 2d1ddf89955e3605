#!/usr/bin/env python3
"""Generic-mode solves of the same batch double-buffered over S HIP streams (each stream its own workspace and outputs): step k+1's
prep kernel and first solve workgroups fill the SIMDs that the tail of step k's solve leaves idle (one wavefront / workgroup per QP:
a launch ends with its slowest instances).  What a service that always has the next batch at hand gets from one GPU; bench.py's
`value` is the plain one-step-after-the-other rate.  Prints one JSON line.
usage: pipelined_steps.py [kinematic|dynamic] [N=40] [B=4096] [steps=20] [streams=2]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm

model = fm.DYNAMIC if (len(sys.argv) > 1 and sys.argv[1].startswith("dyn")) else fm.KINEMATIC
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
S = int(sys.argv[5]) if len(sys.argv) > 5 else 2
dev = torch.device("cuda", 0)
tr = fm.Track.load("fsg2019")
up = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
q = fm.LtvBatch(model, N, 0.05, tr, B, device=dev).build_qp(up(x0), up(xr), up(xl), up(ul))
args = [q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")]
streams = [torch.cuda.Stream(dev) for _ in range(S)]
ws = [None] * S
torch.cuda.synchronize(dev)


def run(nsteps, nstreams):
    outs = []
    t0 = time.perf_counter()
    for k in range(nsteps):
        j = k % nstreams
        with torch.cuda.stream(streams[j]):
            o = fm.qp_solve_batch_device(*args, workspace=ws[j], stream=streams[j].cuda_stream)
        ws[j] = o["workspace"]; outs.append(o)
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0, outs


run(2 * S, S)                                   # warm-up: code load, workspaces
t1, o1 = run(steps, 1)
tS, oS = run(steps, S)
same = all(torch.equal(o1[0][k], o[k]) for o in oS for k in ("x", "exitflag", "iter"))
solved = int((o1[0]["exitflag"] == 0).sum().item())
print(json.dumps({"what": "generic-mode steps of one batch, one after the other vs double-buffered over %d HIP streams" % S,
                  "model": "dynamic" if model == fm.DYNAMIC else "kinematic", "N": N, "batch": B, "steps": steps, "solved_per_step": solved,
                  "one_stream_qp_per_s": solved * steps / t1, "one_stream_ms_per_step": 1e3 * t1 / steps,
                  "pipelined_qp_per_s": solved * steps / tS, "pipelined_ms_per_step": 1e3 * tS / steps, "streams": S,
                  "results_identical": bool(same)}))
