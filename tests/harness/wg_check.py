#!/usr/bin/env python3
"""Development check of the workgroup kernel on one shape: solve parity with the oracle (flags, iterations, fval, x, vertex rate,
numpy KKT certificate) on B instances and the solve time of a 4096 batch.  usage: wg_check.py [model=1] [N=60] [B=24] [bench_batch=4096]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fsae_mpc_amd as fm
import oracle as orc
from kkt_numpy import kkt_certificate

model = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B = int(sys.argv[3]) if len(sys.argv) > 3 else 24
BB = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
print("lib:", fm._lib.LIB_PATH)
otr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
tr = fm.Track.load("fsg2019")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, xl, ul, xr = fm.instances(model, N, 0.05, otr.L, 20190, range(B))
q = orc.build_qp_batch(model, otr, N, 0.05, x0, xr, xl, ul, threads=8)
KEYS = ("H", "g", "A", "lb", "ub", "lbA", "ubA")
for pol in (0, 1):
    o = fm.qp_solve_batch_device(*(dev(q[k]) for k in KEYS), options=fm.default_opts(polish=pol), want_lambda=True, want_aux=True)
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in o.items() if v is not None and k != "workspace"}
    ref = orc.qp_solve_batch_aux(*(q[k] for k in KEYS), opts=orc.default_opts(polish=pol), threads=8)
    c = kkt_certificate(*(q[k] for k in KEYS), out["x"], out["lam"])
    ex = np.abs(out["x"] - ref["x"]).max(axis=1) / np.maximum(1, np.abs(ref["x"]).max(axis=1))
    both = (out["polished"] > 0) & (ref["polished"] > 0)
    print("polish", pol, "flags", dict(zip(*np.unique(out["exitflag"], return_counts=True))), "iters gpu %.2f oracle %.2f" % (out["iter"].mean(), ref["iter"].mean()),
          "| dfval max %.2e" % (np.abs(out["fval"] - ref["fval"]) / np.maximum(1, np.abs(ref["fval"]))).max(), "| x err max %.2e med %.2e" % (ex.max(), np.median(ex)),
          "| on vertex gpu %.3f oracle %.3f" % ((out["polished"] > 0).mean(), (ref["polished"] > 0).mean()), "| x err where both on vertex %.2e" % (ex[both].max() if both.any() else -1),
          "| numpy cert max %.2e (kernel's kkt max %.2e)" % (c["max"].max(), out["kkt"].max()), "| polished codes", dict(zip(*np.unique(out["polished"], return_counts=True))))
if BB > 0:
    x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(BB))
    st = fm.LtvBatch(model, N, 0.05, tr, BB)
    qq = st.build_qp(dev(x0), dev(xr), dev(xl), dev(ul))
    args = [qq[k] for k in KEYS]
    ws = None
    L = fm.lib()
    for pol in (1, 0):
        opts = fm.default_opts(polish=pol)
        o = fm.qp_solve_batch_device(*args, options=opts, workspace=ws, want_aux=True); ws = o["workspace"]
        torch.cuda.synchronize()
        L.fsaempc_qp_set_timing(1)
        ts = []
        for _ in range(3):
            o = fm.qp_solve_batch_device(*args, options=opts, workspace=ws, want_aux=True)
            a, b_ = C.c_double(0), C.c_double(0)
            L.fsaempc_qp_get_timing(C.byref(a), C.byref(b_)); ts.append((a.value, b_.value))
        L.fsaempc_qp_set_timing(0)
        fl = o["exitflag"].cpu().numpy(); it = o["iter"].cpu().numpy(); pl = o["polished"].cpu().numpy()
        print("bench B=%d polish=%d: prep %.2f ms solve %.2f ms -> %.1f k QP/s (solve only %.1f k) | flags %s mean it %.2f on vertex %.3f" % (
            BB, pol, np.mean([t[0] for t in ts]), np.mean([t[1] for t in ts]), BB / np.mean([t[0] + t[1] for t in ts]), BB / np.mean([t[1] for t in ts]),
            dict(zip(*np.unique(fl, return_counts=True))), it.mean(), (pl > 0).mean()))
