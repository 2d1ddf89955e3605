#!/usr/bin/env python3
"""Which cheap quantity predicts the interior-point iteration count well enough to order a batch by?  (CPU study behind
QpParams::order / fsaempc_qp_aux.difficulty; results in profiles/round3/launch_order.txt.)  The synthetic QPs of the bench are built
and solved with the CPU oracle (its iteration counts agree with the HIP path's to +-1), candidate scores are ranked by Spearman
correlation with the iteration count and by the makespan of a list schedule -- `slots` identical machines, jobs started in score
order, each on the first free machine, cost = iterations + 1.5 -- which is what the hardware dispatcher does with one wavefront /
workgroup per QP.  usage: order_predictors.py [model=0] [N=40] [B=4096] [slots=1024]"""
import heapq, os, sys
import numpy as np
from scipy.stats import rankdata, spearmanr
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import oracle as orc

model = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
tr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, tr.L, 20190, range(B))
q = orc.build_qp_batch(model, tr, N, 0.05, x0, xr, xl, ul, threads=8)
o = orc.qp_solve_batch_aux(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"], threads=8)
it = o["iter"]; cost = it + 1.5
g, A, lb, ub, lbA, ubA = (q[k] for k in ("g", "A", "lb", "ub", "lbA", "ubA"))


def makespan(order):
    h = [0.0] * S; heapq.heapify(h)
    for i in order:
        heapq.heappush(h, heapq.heappop(h) + cost[i])
    return max(h)


rown = np.abs(A).max(1) + 1e-300
viol = (np.maximum(lbA, 0) + np.maximum(-ubA, 0)) / rown
vc = ((lbA > 0) | (ubA < 0)).sum(1) + ((lb > 0) | (ub < 0)).sum(1)
R = lambda a: rankdata(a) / B
margin = np.minimum(np.where(lbA > -1e9, -lbA, np.inf), np.where(ubA < 1e9, ubA, np.inf)) / rown
cands = {
    "rows+bounds excluding x=0 (shipped)": vc.astype(float),
    "largest scaled violation at x=0": viol.max(1),
    "sum of scaled violations at x=0": viol.sum(1),
    "|g|": np.linalg.norm(g, axis=1),
    "count + 0.5 rank(largest violation)": R(vc) + 0.5 * R(viol.max(1)),
    "count + 0.5 rank(|g|)": R(vc) + 0.5 * R(np.linalg.norm(g, axis=1)),
    "rows within 0.05 of a bound at x=0": (margin < 0.05).sum(1).astype(float),
}
ideal = cost.sum() / S
print("model %d N %d B %d slots %d: iterations mean %.2f min %d max %d" % (model, N, B, S, it.mean(), it.min(), it.max()))
print("makespan (iteration units): no tail %.1f | index order %.1f | true iteration count descending %.1f" % (ideal, makespan(range(B)), makespan(np.argsort(-cost, kind="stable"))))
for k, s in cands.items():
    print("  %-40s Spearman %.3f  makespan %.1f" % (k, spearmanr(s, it).correlation, makespan(np.argsort(-s, kind="stable"))))
