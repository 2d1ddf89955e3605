import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch, fsae_mpc_amd as fm
tr = fm.Track.load("fsg2019")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for model, N, B in ((1, 60, 4096), (0, 40, 4096)):
    x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
    st = fm.LtvBatch(model, N, 0.05, tr, B)
    a = [dev(x0), dev(xr), dev(xl), dev(ul)]
    st.build_qp(*a); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record(); q = st.build_qp(*a); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(os.path.basename(fm._lib.LIB_PATH), "model", model, "N", N, "B", B, "build_qp ms", np.round(ts, 2), flush=True)
