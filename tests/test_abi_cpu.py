"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol include/fsaempc.h
declares (no compute without a GPU), argument validation mirrors the MEX gateway, host-side generators agree
with the oracle's, and the N>1 sharding/gather path works over gloo."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    import fsae_mpc_amd as fm
    hdr = open(os.path.join(ROOT, "include", "fsaempc.h")).read()
    declared = set(re.findall(r"\b(fsaempc_[A-Za-z_0-9]+)\s*\(", hdr))
    declared -= {"fsaempc_qp_opts", "fsaempc_qp_desc", "fsaempc_spline", "fsaempc_ltv_desc"}
    assert declared == set(fm._lib.EXPORTS)
    L = fm.lib()
    for s in sorted(declared):
        assert hasattr(L, s), s


def test_default_opts_and_dims():
    import fsae_mpc_amd as fm
    o = fm.default_opts()
    assert o.tol == 1e-8 and o.tol_loose == 1e-6 and o.inf_bound == 1e9 and o.max_iter == 100 and o.polish == 1
    assert fm.dims(fm.KINEMATIC, 40) == (5, 1, 81, 240) and fm.dims(fm.DYNAMIC, 60) == (7, 4, 124, 1200)
    d = fm._lib.QpDesc(81, 240, 4096, 0)
    assert fm.lib().fsaempc_qp_workspace_bytes(C.byref(d)) > 4096 * 240 * 81 * 8
    d = fm._lib.QpDesc(200, 10, 1, 0)
    assert fm.lib().fsaempc_qp_workspace_bytes(C.byref(d)) == -2   # FSAEMPC_ERR_DIM


def test_gateway_argument_validation():
    import fsae_mpc_amd as fm
    H = np.eye(2); g = np.zeros(2)
    with pytest.raises(fm.FsaempcError, match="NaN"):
        fm.qpOASES(H, np.array([np.nan, 0]), [0, 0], [1, 1])
    with pytest.raises(ValueError, match="dimension mismatch"):
        fm.qpOASES(H, g, np.ones((1, 3)), [0, 0], [1, 1], [0], [1])
    with pytest.raises(TypeError):
        fm.qpOASES(H, g, [0, 0])


def test_sequence_handle_errors():
    import fsae_mpc_amd as fm
    # main.m:193 calls qpOASES_sequence('c', QP) with QP == 0: the gateway answers "Invalid handle to QP instance!"
    with pytest.raises(fm.FsaempcError, match="Invalid handle"):
        fm.qpOASES_sequence("c", 0)
    with pytest.raises(fm.FsaempcError, match="Invalid handle"):
        fm.qpOASES_sequence("h", 7, [0.0], [0.0], [1.0], [], [])


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import fsae_mpc_amd as fm
    with pytest.raises(fm.FsaempcError):
        fm.qpOASES(np.eye(2), np.zeros(2), [0, 0], [1, 1])
    assert fm.lib().fsaempc_selftest_mfma() < 0


def test_product_generators_match_oracle(orc, otrack):
    import fsae_mpc_amd as fm
    tr = fm.Track.load("fsg2019")
    assert tr.M == otrack.M and tr.dl == otrack.dl and np.array_equal(tr.xP, otrack.xP)
    for model in (fm.KINEMATIC, fm.DYNAMIC):
        a = fm.instances(model, 12, 0.05, tr.L, 20190, np.arange(100, 164))
        b = orc.synth_instances(model, 12, 0.05, tr.L, 20190, np.arange(100, 164))
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def _dist_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fsae_mpc_amd import shard
    B = 37
    lo, hi = shard.shard_range(B, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float64)[:, None] * torch.ones(1, 3, dtype=torch.float64)  # stand-in per-instance results
    full = shard.gather_rows(local, B, rank, world)
    ok = bool(torch.equal(full[:, 0], torch.arange(B, dtype=torch.float64)))
    tmax = shard.max_over_ranks(float(rank + 1))
    q.put((rank, lo, hi, ok, tmax))
    dist.destroy_process_group()


def test_sharding_and_gather_over_gloo():
    import torch.multiprocessing as mp
    import fsae_mpc_amd  # noqa: F401
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps: p.join(60)
    assert res[0][1:3] == (0, 19) and res[1][1:3] == (19, 37)
    assert all(r[3] for r in res) and all(r[4] == 2.0 for r in res)


def test_bench_self_starts_two_ranks_over_gloo():
    """`python bench.py --gpus 2` with no launcher around it starts its own ranks: the dry run takes the same
    launch -> init_process_group -> shard -> gather_rows -> max_over_ranks -> one JSON line route as the GPU run, on gloo."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--batch", "37", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["gathered_rows"] == 74 and out["config"]["gather_ok"]
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline", "cpu_baseline"):
        assert k in out


@pytest.mark.parametrize("src", ["qpOASES.cpp", "qpOASES_sequence.cpp"])
def test_mex_gateways_compile(src):
    """the MEX gateways type-check against a stand-in mex.h and the real include/fsaempc.h (no MATLAB in this image)"""
    import subprocess
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "stub_mex"),
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "mex", src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
