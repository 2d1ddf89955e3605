#!/usr/bin/env python3
"""Investigation of the build-variant fragility: libraries whose solve kernel was cut short in the assembly (build/frag/inject.py:
LDS dumped to P.dump, s_endpgm at a chosen basic block) are run on ONE instance; LDS image and workspace are stored and compared.
usage: frag_dump.py child <out.npz>   |   frag_dump.py cmp <libA> <libB> [label]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..")
sys.path.insert(0, ROOT)
MODEL, N = 0, int(os.environ.get("FRAG_N", "39"))
VEC = ["X", "G", "HX", "R1", "R2", "P1", "P2", "P3", "DX", "E"]
ROW = ["L", "U", "TL", "TU", "ZL", "ZU", "V", "D", "W1", "W2", "W3", "VA", "VC", "RPL", "RPU", "CB1", "CC1", "CB2", "CC2"]


def layout(n, m):
    rem = n % 16
    T, nb = (n // 16, rem) if (n >= 16 and 1 <= rem <= 4) else ((n + 15) // 16, 0)
    nc = 16 * T; np_ = nc + (16 if nb else 0); Kq = (m + 3) // 4; J = (Kq + 15) // 16; JB = (np_ + 63) // 64
    rowlen = (J + JB) * 64; ntr = (Kq + 3) // 4
    off = 0; o = {}
    o["Aw"] = off; off += 2 * ntr * T * 128
    o["meta"] = off; off += (max(J, 1) * 64 + 2 * ntr + 1 + 16 + 1) // 2 + 1
    off = (off + 1) & ~1
    o["Hw"] = off; off += T * T * 4 * 64
    o["gw"] = off; off += np_
    o["E"] = off; off += np_
    o["F"] = off; off += max(J, 1) * 64
    o["Ab"] = off; off += 4 * max(J, 1) * 64
    o["Hb"] = off; off += 4 * np_
    o["rows"] = off; off += len(ROW) * rowlen
    o["save"] = off; off += np_ + rowlen
    lds = {v: (i * np_, (i + 1) * np_) for i, v in enumerate(VEC)}
    p = len(VEC) * np_
    lds["MB"] = (p, p + 4 * np_); p += 4 * np_
    lds["SCR"] = (p, p + 288); p += 288
    lds["YL"] = (p, p + T * 272); p += T * 272
    lds["ring"] = (p, p + 2 * T * 128); p += (2 * T + 2) * 128
    lds["cof"] = (p, p + 6 * 64)
    return o, lds, rowlen, np_


def child(out):
    import torch
    import fsae_mpc_amd as fm
    import oracle as orc
    otr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
    x0, xl, ul, xr = fm.instances(MODEL, N, 0.05, otr.L, 20190, range(1))
    q = orc.build_qp_batch(MODEL, otr, N, 0.05, x0, xr, xl, ul)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = [dev(q[k]) for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")]
    dump = torch.zeros(32768, dtype=torch.float64, device="cuda")
    fm.lib().fsaempc_debug_set_dump(C.c_void_p(dump.data_ptr()), 0)
    o = fm.qp_solve_batch_device(*args, options=fm.default_opts(polish=0, max_iter=int(os.environ.get("FRAG_MAXIT", "1"))), want_lambda=True)
    torch.cuda.synchronize()
    np.savez(out, dump=dump.cpu().numpy(), ws=o["workspace"].cpu().numpy().view(np.float64).ravel(), x=o["x"].cpu().numpy(), n=q["g"].shape[1], m=q["lbA"].shape[1])


def main():
    if sys.argv[1] == "child":
        return child(sys.argv[2])
    outs = []
    for i, lib in enumerate(sys.argv[2:4]):
        out = "/tmp/frag_dump_%d.npz" % i
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", out], env=dict(os.environ, FSAEMPC_LIB=os.path.abspath(lib)))
        outs.append(np.load(out))
    a, b = outs
    o, lds, rowlen, np_ = layout(int(a["n"]), int(a["m"]))
    print("== %s: %s vs %s" % (sys.argv[4] if len(sys.argv) > 4 else "", sys.argv[2], sys.argv[3]))
    def rep(name, va, vb):
        d = np.abs(va - vb); bad = ~(np.isfinite(va) & np.isfinite(vb)) & ~((va == vb) | (np.isnan(va) & np.isnan(vb)))
        d = np.where(np.isfinite(d), d, 0)
        sc = max(1e-300, np.abs(np.where(np.isfinite(va), va, 0)).max())
        flag = "  <<<<" if (d.max() > 1e-9 * sc or bad.any()) else ""
        print("  %-6s max|a| %.3e  max|a-b| %.3e (rel %.1e) at %d  nonfinite-mismatch %d%s" % (name, sc, d.max(), d.max() / sc, int(d.argmax()), int(bad.sum()), flag))
    print(" LDS image:")
    for name, (lo, hi) in lds.items():
        rep(name, a["dump"][lo:hi], b["dump"][lo:hi])
    print(" workspace rows:")
    for i, name in enumerate(ROW):
        lo = o["rows"] + i * rowlen
        rep(name, a["ws"][lo:lo + rowlen], b["ws"][lo:lo + rowlen])
    for name in ("Aw", "Hw", "gw", "E", "F"):
        nxt = sorted(v for v in o.values() if v > o[name])[0]
        rep(name, a["ws"][o[name]:nxt], b["ws"][o[name]:nxt])
    if os.environ.get("FRAG_SAVE"):
        np.savez(os.environ["FRAG_SAVE"], a_dump=a["dump"], b_dump=b["dump"], a_ws=a["ws"], b_ws=b["ws"])


if __name__ == "__main__":
    main()
