import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import fsae_mpc_amd as fm
tr = fm.Track.load("fsg2019")
for (model, N) in [(0, 20), (0, 12), (1, 10), (0, 24)]:
    B = 128
    x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    q = fm.LtvBatch(model, N, 0.05, tr, B).build_qp(up(x0), up(xr), up(xl), up(ul))
    for sb in ("1", "0"):
        os.environ["FSAEMPC_SLACK_BORDER"] = sb
        out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")), want_lambda=True, want_aux=True)
        torch.cuda.synchronize()
        x = out["x"].cpu().numpy(); lam = out["lam"].cpu().numpy(); fl = out["exitflag"].cpu().numpy()
        nx = np.isnan(x); nl = np.isnan(lam)
        print(model, N, "policy", sb, "flags", np.unique(fl, return_counts=True), "nan x inst", np.where(nx.any(1))[0][:8], "cols", np.where(nx.any(0))[0][:10],
              "nan lam inst", np.where(nl.any(1))[0][:8], "cols", np.where(nl.any(0))[0][:10], "iters", out["iter"].float().mean().item(), flush=True)
