import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch, fsae_mpc_amd as fm
tr = fm.Track.load("fsg2019")
B = 4096
x0, xl, ul, xr = fm.instances(fm.KINEMATIC, 40, 0.05, tr.L, 20190, range(B))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
st = fm.LtvBatch(fm.KINEMATIC, 40, 0.05, tr, B)
a = [up(x0), up(xr), up(xl), up(ul)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = st.step(*a); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("fused step %.2f ms" % (1e3 * (t1 - t0)), "iters", o["iter"].double().mean().item(), "flags0", int((o["exitflag"] == 0).sum()))
q = st.build_qp(*a); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); q = st.build_qp(*a); torch.cuda.synchronize(); print("build_qp %.2f ms" % (1e3 * (time.perf_counter() - t0)))
for rep in range(2):
    t0 = time.perf_counter(); out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA"))); torch.cuda.synchronize(); print("solve %.2f ms" % (1e3 * (time.perf_counter() - t0)))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    fo = st.step(*a)
torch.cuda.synchronize(); print("5 async fused steps: %.2f ms each" % (1e3 * (time.perf_counter() - t0) / 5))
ws = None
for _ in range(3):
    out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")), workspace=ws); ws = out["workspace"]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    fo = st.step(*a)
torch.cuda.synchronize(); print("after generic solves with a second workspace alive: %.2f ms each" % (1e3 * (time.perf_counter() - t0) / 5))
print("mem allocated GB", torch.cuda.memory_allocated() / 1e9, "reserved", torch.cuda.memory_reserved() / 1e9)
