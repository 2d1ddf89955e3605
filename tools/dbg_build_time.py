import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch, fsae_mpc_amd as fm
tr = fm.Track.load("fsg2019")
for model, B in ((fm.KINEMATIC, 4096), (fm.DYNAMIC, 2048)):
    x0, xl, ul, xr = fm.instances(model, 40, 0.05, tr.L, 20190, range(B))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    st = fm.LtvBatch(model, 40, 0.05, tr, B)
    a = [up(x0), up(xr), up(xl), up(ul)]
    st.build_qp(*a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): q = st.build_qp(*a)
    torch.cuda.synchronize(); print("model", model, "build_qp %.3f ms" % (1e3 * (time.perf_counter() - t0) / 5), end=" | ")
print()
