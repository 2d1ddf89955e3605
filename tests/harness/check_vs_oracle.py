#!/usr/bin/env python3
"""Development check of the workgroup solve kernel against the CPU oracle (no polish on either side) on several shapes,
plus a timing A/B against the round-1 kernel (FSAEMPC_QP_V1=1 in a child process)."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import fsae_mpc_amd as fm
import oracle as orc


def shape(model, N, B, polish=0, timing=True):
    otr = orc.Track.load(os.path.join(os.path.dirname(fm.tracks.__file__), "tracks", "fsg2019.json"))
    x0, xl, ul, xr = fm.instances(model, N, 0.05, otr.L, 20190, range(B))
    q = orc.build_qp_batch(model, otr, N, 0.05, x0, xr, xl, ul)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = [dev(q[k]) for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")]
    o = fm.default_opts(polish=polish)
    out = fm.qp_solve_batch_device(*args, options=o, want_lambda=True)
    torch.cuda.synchronize()
    x = out["x"].cpu().numpy(); fl = out["exitflag"].cpu().numpy(); it = out["iter"].cpu().numpy(); lam = out["lam"].cpu().numpy()
    nchk = min(B, int(os.environ.get('NCHK', '64')))
    xo, fo, flo, ito, lamo, _ = orc.qp_solve_batch(*[q[k][:nchk] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")], orc.default_opts(polish=polish))
    ok = (fl[:nchk] == 0) & (flo == 0)
    ex = np.abs(x[:nchk] - xo).max(axis=1) / np.maximum(1, np.abs(xo).max(axis=1))
    kk = np.array([orc.qp_kkt(q["H"][b].T, q["g"][b], q["A"][b].T, q["lb"][b], q["ub"][b], q["lbA"][b], q["ubA"][b], x[b], lam[b])[0] if fl[b] == 0 else np.nan for b in range(nchk)])
    msg = "model %d N %2d B %4d: flags %s | iters mean %.2f (oracle %.2f, same %d/%d) | x err vs oracle max %.1e med %.1e p90 %.1e | kkt max %.1e" % (
        model, N, B, dict(zip(*np.unique(fl, return_counts=True))), it.mean(), ito.mean(), int((it[:nchk] == ito).sum()), nchk,
        np.nanmax(np.where(ok, ex, np.nan)) if ok.any() else np.nan, np.nanmedian(np.where(ok, ex, np.nan)) if ok.any() else np.nan, np.nanpercentile(np.where(ok, ex, np.nan), 90) if ok.any() else np.nan, np.nanmax(kk) if np.isfinite(kk).any() else np.nan)
    if timing:
        ws = out["workspace"]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            out = fm.qp_solve_batch_device(*args, options=o, workspace=ws)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        msg += " | %.3f ms/solve-batch = %.0f QP/s" % (1e3 * dt, B / dt)
    print(msg, flush=True)


if __name__ == "__main__":
    which = os.environ.get("FSAEMPC_QP_V1") and "v1" or "wg"
    print("kernel:", which, flush=True)
    shapes = [(0, 40, 4096), (0, 20, 4096), (1, 40, 2048), (0, 12, 256), (1, 8, 256), (0, 9, 256), (1, 7, 256), (0, 5, 64), (1, 60, 512)]
    if which == "wg":
        shapes += [(1, 80, 256), (0, 90, 128)]
    if os.environ.get("SHAPES"):
        shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
    for sh in shapes:
        mo, N, B = sh[:3]
        try:
            shape(mo, N, B, polish=sh[3] if len(sh) > 3 else 0)
        except Exception as e:
            print("model %d N %d FAILED: %s" % (mo, N, e), flush=True)
