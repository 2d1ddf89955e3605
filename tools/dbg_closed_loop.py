import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, fsae_mpc_amd as fm
from closed_loop_bench import initial_carts
model = fm.DYNAMIC if sys.argv[1] == "dyn" else fm.KINEMATIC
tr = fm.Track.load("fss2019")
B = 2048
cart0, s_init = initial_carts(tr, B, 20190)
cl = fm.ClosedLoop(model, 40, 0.05, tr, cart0)
cl.x_opt[:, :, 0] += torch.from_numpy(s_init).cuda()[:, None]
cl.x_opt[:, :, 3] += torch.from_numpy(cart0[:, 3]).cuda()[:, None]
seen = set()
for step in range(60):
    out = cl.step(); torch.cuda.synchronize()
    fl = out["exitflag"].cpu().numpy()
    x0 = cl.x0.cpu().numpy(); cart = cl.cart.cpu().numpy()
    bad = np.nonzero(~np.isfinite(x0).all(axis=1) | ~np.isfinite(cart).all(axis=1))[0]
    new = [b for b in bad if b not in seen]
    if new:
        b = new[0]
        print("step", step, "new non-finite cars", len(new), "e.g.", b, "flag", fl[b], "x0", x0[b], "cart", cart[b])
        seen.update(new)
    if step in (0, 1, 5, 20, 59):
        print("step", step, "flags", dict(zip(*np.unique(fl, return_counts=True))), "nonfinite", len(bad))
# what do failing-but-finite cars look like
bad = np.nonzero((fl != 0) & np.isfinite(x0).all(axis=1))[0]
for b in bad[:6]:
    print("car", b, "flag", fl[b], "x0", np.round(x0[b], 3))

# cross-check the failing QPs of the last step with the CPU oracle's solver (same H,g,A,bounds)
import oracle as orc
cl.pre(); torch.cuda.synchronize()
q = cl.mpc.build_qp(cl.x0, cl.x_ref, cl.x_opt, cl.u_opt); torch.cuda.synchronize()
out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA"))); torch.cuda.synchronize()
fl = out["exitflag"].cpu().numpy()
x0 = cl.x0.cpu().numpy()
sel = [b for b in np.nonzero(fl != 0)[0] if np.isfinite(x0[b]).all()][:12]
agree = 0
for b in sel:
    g = lambda k: q[k][b].cpu().numpy()
    xo, fo, flo, ito, lamo = orc.qp_solve(g("H").T, g("g"), g("A").T, g("lb"), g("ub"), g("lbA"), g("ubA"))
    print("car", b, "gpu flag", fl[b], "oracle flag", flo, "oracle iters", ito)
    agree += int((flo != 0) == (fl[b] != 0))
print("agreement on solvable/unsolvable:", agree, "of", len(sel))

# save a few failing QPs for offline inspection with the oracle's verbose trace
import os
os.makedirs("gpurun_out", exist_ok=True)
keep = {}
for b in sel[:6]:
    for k_ in ("H", "g", "A", "lb", "ub", "lbA", "ubA"):
        keep["%s_%d" % (k_, b)] = q[k_][b].cpu().numpy()
keep["cars"] = np.array(sel[:6]); keep["flags"] = fl[sel[:6]]
np.savez_compressed("gpurun_out/failing_qps.npz", **keep)
