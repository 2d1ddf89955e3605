#!/usr/bin/env python3
"""Per-phase timing of the fused step (construction, prep, solve, post-solve) by the library's HIP events
(fsaempc_ltv_get_timing); run under `rocprofv3 --marker-trace --kernel-trace` it also shows the roctx ranges the library
opens around each phase's launches.  usage: fused_phases.py [kinematic|dynamic] [N=40] [B=4096] [steps=5]"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm

model = fm.DYNAMIC if (len(sys.argv) > 1 and sys.argv[1].startswith("dyn")) else fm.KINEMATIC
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
tr = fm.Track.load("fsg2019")
up = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
a = [up(v) for v in (x0, xr, xl, ul)]
st = fm.LtvBatch(model, N, 0.05, tr, B)
st.step(*a); torch.cuda.synchronize()
L = fm.lib(); L.fsaempc_qp_set_timing(1)
rows = []
for _ in range(steps):
    o = st.step(*a)
    ph = [C.c_double(0) for _ in range(4)]
    assert L.fsaempc_ltv_get_timing(*[C.byref(p) for p in ph]) == 0
    rows.append([p.value for p in ph])
L.fsaempc_qp_set_timing(0)
m = np.mean(rows, 0)
print(json.dumps({"what": "fused step phases, HIP events on the launch stream, mean of %d steps" % steps, "model": "dynamic" if model == fm.DYNAMIC else "kinematic",
                  "N": N, "batch": B, "build_ms": m[0], "prep_ms": m[1], "solve_ms": m[2], "post_ms": m[3], "step_ms": float(m.sum()),
                  "qp_per_s": float((o["exitflag"] == 0).sum().item()) / (1e-3 * float(m.sum()))}))
