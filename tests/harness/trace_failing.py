#!/usr/bin/env python3
"""Iteration traces (dbg build, dump stage 5) of the QPs the shipped build misses, next to what the oracle does on them:
kinematic N = 40 id 6585 and the regression fixtures tests/golden/regress_*.npz.
usage: FSAEMPC_LIB=fsae-mpc_amd/lib/libfsaempc_dbg.so python tests/harness/trace_failing.py out.json"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
import fsae_mpc_amd as fm
import oracle as orc

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
KEYS = ("H", "g", "A", "lb", "ub", "lbA", "ubA")


def trace(q, **optkw):
    """q: dict of single-instance arrays in the batch-major device layout (H (n,n), A (n,m)); returns the trace dict"""
    dump = torch.zeros(16 * 124, dtype=torch.float64, device="cuda")
    fm.lib().fsaempc_debug_set_dump(C.c_void_p(dump.data_ptr()), 5 | (0 << 8))
    try:
        o = fm.qp_solve_batch_device(*(dev(q[k][None]) for k in KEYS), options=fm.default_opts(**optkw), want_aux=True, want_lambda=True)
        torch.cuda.synchronize()
    finally:
        fm.lib().fsaempc_debug_set_dump(None, 0)
    d = dump.cpu().numpy().reshape(124, 16)
    it = int(o["iter"][0])
    rows = [dict(it=i, merit=d[i, 0], rd=d[i, 1], rp=d[i, 2], gap=d[i, 3], mu=d[i, 4], saved=int(d[i, 6]), a_aff=d[i, 8], sigma=d[i, 9], alpha=d[i, 10], cw=d[i, 11], stall=int(d[i, 13])) for i in range(min(it + 1, 120))]
    fin = d[120:].ravel()
    att = [dict(m_rd=fin[8 + 5 * a + 1], m_rp=fin[8 + 5 * a + 2], m_sg=fin[8 + 5 * a + 3], m_cp=fin[8 + 5 * a + 4]) for a in range(8) if fin[8 + 5 * a] != 0]
    return dict(flag=int(o["exitflag"][0]), iter=it, kkt=float(o["kkt"][0]), polished=int(o["polished"][0]), flag_before_refinement=int(fin[4]),
                attempts=att, rows=rows, x=o["x"][0].cpu().numpy())


def show(name, q, out):
    t = trace(q)
    xo, fo, flo, ito, lamo = orc.qp_solve(q["H"].T, q["g"], q["A"].T, q["lb"], q["ub"], q["lbA"], q["ubA"])
    t0 = trace(q, polish=0)
    print("==", name, "gpu flag %d iter %d kkt %.2e polished %d (flag before refinement %d) | polish=0: flag %d iter %d | oracle flag %d iter %d | max|x-x_orc| %.2e"
          % (t["flag"], t["iter"], t["kkt"], t["polished"], t["flag_before_refinement"], t0["flag"], t0["iter"], flo, ito, np.abs(t["x"] - xo).max()))
    for r in t["rows"][-12:]:
        print("   it %2d merit %.2e rd %.2e rp %.2e gap %.2e mu %.2e saved %d | a_aff %.3f sigma %.2e alpha %.4f cw %.0f stall %d" % (
            r["it"], r["merit"], r["rd"], r["rp"], r["gap"], r["mu"], r["saved"], r["a_aff"], r["sigma"], r["alpha"], r["cw"], r["stall"]))
    for a in t["attempts"]:
        print("   attempt: m_rd %.2e m_rp %.2e m_sg %.2e m_cp %.2e" % (a["m_rd"], a["m_rp"], a["m_sg"], a["m_cp"]))
    t.pop("x"); t["oracle"] = dict(flag=flo, iter=ito)
    out[name] = t


out = {}
tr = fm.Track.load("fsg2019")
x0, xl, ul, xr = fm.instances(fm.KINEMATIC, 40, 0.05, tr.L, 20190, np.array([6585]))
qq = fm.LtvBatch(fm.KINEMATIC, 40, 0.05, tr, 1).build_qp(dev(x0), dev(xr), dev(xl), dev(ul))
show("kin40_id6585", {k: qq[k][0].cpu().numpy() for k in KEYS}, out)
import glob
for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "regress_*.npz"))):
    z = np.load(p)
    show(os.path.basename(p)[:-4], {k_: z[k_] for k_ in KEYS}, out)
with open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w") as f:
    json.dump(out, f, default=float)
