import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import fsae_mpc_amd as fm
tr = fm.Track.load("fsg2019")
model, N, B = 0, 20, 128
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
q = fm.LtvBatch(model, N, 0.05, tr, B).build_qp(up(x0), up(xr), up(xl), up(ul))
for sb in ("1", "0"):
    for pol in (1, 0):
        for kern in ("", "wg"):
            os.environ["FSAEMPC_SLACK_BORDER"] = sb
            if kern: os.environ["FSAEMPC_QP_KERNEL"] = kern
            else: os.environ.pop("FSAEMPC_QP_KERNEL", None)
            out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")), options=fm.default_opts(polish=pol), want_lambda=True, want_aux=True)
            torch.cuda.synchronize()
            x = out["x"].cpu().numpy(); lam = out["lam"].cpu().numpy(); fl = out["exitflag"].cpu().numpy()
            b = 58
            print("policy", sb, "polish", pol, "kernel", kern or "auto", "| inst 58: flag", fl[b], "iter", out["iter"][b].item(), "polished", out["polished"][b].item(), "kkt", out["kkt"][b].item(), "fval", out["fval"][b].item(),
                  "nan x", int(np.isnan(x[b]).sum()), "nan lam", int(np.isnan(lam[b]).sum()), "| batch nan-x instances", np.where(np.isnan(x).any(1))[0], flush=True)
