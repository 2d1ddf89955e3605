import os, sys, numpy as np
sys.argv = [sys.argv[0], "/dev/null"]
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import importlib.util, types
src = open(os.path.join(ROOT, "tests", "harness", "trace_failing.py")).read().split("out = {}\ntr = fm.Track")[0]
exec(compile(src, "trace_failing_head", "exec"))
tr = fm.Track.load("fsg2019")
x0, xl, ul, xr = fm.instances(fm.KINEMATIC, 20, 0.05, tr.L, 20190, np.array([58]))
qq = fm.LtvBatch(fm.KINEMATIC, 20, 0.05, tr, 1).build_qp(dev(x0), dev(xr), dev(xl), dev(ul))
q = {k: qq[k][0].cpu().numpy() for k in KEYS}
for sb in ("1", "0"):
    os.environ["FSAEMPC_SLACK_BORDER"] = sb
    t = trace(q, polish=0)
    print("policy", sb, "flag", t["flag"], "iter", t["iter"], "kkt", t["kkt"])
    for r in t["rows"]:
        print("   it %2d merit %.2e rd %.2e rp %.2e gap %.2e mu %.2e saved %d | a_aff %.3f sigma %.2e alpha %.4f cw %.0f stall %d" % (
            r["it"], r["merit"], r["rd"], r["rp"], r["gap"], r["mu"], r["saved"], r["a_aff"], r["sigma"], r["alpha"], r["cw"], r["stall"]))
