import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, fsae_mpc_amd as fm, oracle as orc
tr = fm.Track.load("fsg2019"); otr = orc.Track.load(os.path.join('fsae-mpc_amd','tracks','fsg2019.json'))
model = fm.KINEMATIC if (len(sys.argv) < 2 or sys.argv[1] == 'kin') else fm.DYNAMIC
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
LO = int(sys.argv[4]) if len(sys.argv) > 4 else 0
print("lib:", os.environ.get("FSAEMPC_LIB"))
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, np.arange(LO, LO + B))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
q = fm.LtvBatch(model, N, 0.05, tr, B).build_qp(up(x0), up(xr), up(xl), up(ul))
args = [q[k] for k in ("H","g","A","lb","ub","lbA","ubA")]
for pol in (1, 0):
    o = fm.qp_solve_batch_device(*args, options=fm.default_opts(polish=pol), want_aux=True)
    fl = o["exitflag"].cpu().numpy(); it = o["iter"].cpu().numpy(); kk = o["kkt"].cpu().numpy()
    bad = np.where(fl != 0)[0]
    print("polish", pol, "bad ids", bad, "flags", fl[bad], "iters", it[bad], "kkt", kk[bad])
for b in bad[:2]:
    a = [t[b].cpu().numpy() for t in args]
    xo, fo, flo, ito, lamo = orc.qp_solve(a[0].T, a[1], a[2].T, a[3], a[4], a[5], a[6])
    print("oracle on id", b, "flag", flo, "iters", ito)
    for mi in (14, 16, 18, 20, 22, 24, 30, 60):
        o = fm.qp_solve_batch_device(*[t[b:b+1].contiguous() for t in args], options=fm.default_opts(polish=1, max_iter=mi), want_aux=True)
        print("  max_iter", mi, "flag", int(o["exitflag"][0]), "iter", int(o["iter"][0]), "kkt %.2e" % float(o["kkt"][0]))
