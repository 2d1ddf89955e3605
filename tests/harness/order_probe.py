#!/usr/bin/env python3
"""Does the ORDER of the instances in a batch matter?  (One wavefront / workgroup per QP, dispatched in index order: the batch ends
with its last-started, slowest instances.)  Solve time of the same batch in the given order, sorted by a cheap difficulty predictor
(rows violated at A x = 0) descending / ascending, sorted by the true iteration count (the bound of what a predictor can give), and
in the given order with the library's own launch order on (QpParams::order: what ships).
usage: order_probe.py [model=0] [N=40] [B=4096]"""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch, fsae_mpc_amd as fm
model = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
tr = fm.Track.load("fsg2019")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
q = fm.LtvBatch(model, N, 0.05, tr, B).build_qp(dev(x0), dev(xr), dev(xl), dev(ul))
K = ("H", "g", "A", "lb", "ub", "lbA", "ubA")
L = fm.lib()
def run(order, tag):
    a = [q[k] if order is None else q[k].index_select(0, order).contiguous() for k in K]
    o = fm.qp_solve_batch_device(*a, want_aux=True); ws = o["workspace"]; torch.cuda.synchronize()
    L.fsaempc_qp_set_timing(1); ts = []
    for _ in range(4):
        o = fm.qp_solve_batch_device(*a, workspace=ws, want_aux=True)
        p, s = C.c_double(0), C.c_double(0); L.fsaempc_qp_get_timing(C.byref(p), C.byref(s)); ts.append(s.value)
    L.fsaempc_qp_set_timing(0)
    print("%-34s solve %.2f ms (min %.2f) | all solved %s" % (tag, np.mean(ts), np.min(ts), bool((o["exitflag"] == 0).all())), flush=True)
    return o
os.environ["FSAEMPC_QP_ORDER"] = "0"     # the library's own launch order off: the orders below are the caller's
o = run(None, "given order")
viol = ((q["lbA"] > 0) & (q["lbA"] > -1e9)) | ((q["ubA"] < 0) & (q["ubA"] < 1e9))
score = viol.sum(1)
run(torch.argsort(score, descending=True), "rows violated at Ax=0, descending")
run(torch.argsort(score, descending=False), "rows violated at Ax=0, ascending")
run(torch.argsort(o["iter"], descending=True), "true iteration count, descending")
del os.environ["FSAEMPC_QP_ORDER"]
o2 = run(None, "given order, library's launch order on")
print("same results with and without the launch order:", all(bool(torch.equal(o[k], o2[k])) for k in ("x", "fval", "exitflag", "iter")), flush=True)
