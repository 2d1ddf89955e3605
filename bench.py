#!/usr/bin/env python3
"""bench.py -- QP solves/sec of the batched LTV-MPC QP hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): batch of 4096 independent condensed QPs per GPU, curvilinear kinematic
bicycle, N=40 (nV=81, nC=240), fp64, generic mode of SURVEY 8(d): the dense (H,g,A,lb,ub,lbA,ubA) tensors are
resident in HBM when the timed region starts; one step = one batched solve (+ the RCCL gather of x when N>1).
  python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU over RCCL.  Started under torch.distributed.run (RANK / WORLD_SIZE in the environment) this
process is one rank; started plainly it spawns the N ranks itself (a child `python -m torch.distributed.run ... bench.py`
-- before anything here touches the GPU) and relays rank 0's line.  Rank 0 prints ONE JSON line.
cpu_baseline = the CPU oracle timed on the host cores (a reported baseline).  --dry-run exercises the launch, sharding,
gather and reporting path without a GPU (gloo, zeros instead of solves; used by tests/test_abi_cpu.py)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (the microarch guide lists no fp64 row)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU (weak scaling)")
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--model", default="kinematic", choices=["kinematic", "dynamic"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: launch / shard / gather / report path only (CPU, gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-start: one child launcher, N ranks; this parent never initialises the GPU
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    import fsae_mpc_amd as fm
    from fsae_mpc_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N>1 path on a one-GPU box: every rank on cuda:0, gloo instead of RCCL (RCCL refuses two ranks on
    # one device); never used for reported numbers
    rehearsal = os.environ.get("FSAEMPC_BENCH_REHEARSAL") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if (rehearsal or args.dry_run) else "nccl", rank=rank, world_size=world)
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    if args.dry_run:
        return dry_run(args, rank, world)
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)

    model = fm.KINEMATIC if args.model == "kinematic" else fm.DYNAMIC
    N, dt, Bl = args.horizon, 0.05, args.batch
    Btot = Bl * world
    nx, ns, nV, nC = fm.dims(model, N)
    tr = fm.Track.load("fsg2019")
    lo, hi = rank * Bl, (rank + 1) * Bl                      # index-pure shard of the global instance ids
    x0, xl, ul, xr = fm.instances(model, N, dt, tr.L, 20190, np.arange(lo, hi))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    stepper = fm.LtvBatch(model, N, dt, tr, Bl, device=dev)
    q = stepper.build_qp(up(x0), up(xr), up(xl), up(ul))      # product construction kernels (untimed setup)
    torch.cuda.synchronize(dev)
    qp_args = [q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")]
    ws = None
    L = fm.lib()
    # the only exchange: every rank's (x, fval, exitflag, iter) as ONE preallocated (B/G) x (nV + 2) block, one
    # all_gather_into_tensor per step over RCCL/xGMI; buffers are allocated here, outside the timed region
    gather = shard.ResultGather(Bl, nV, world, "cpu" if rehearsal else dev)
    gather_ms = []

    def step():
        nonlocal ws
        out = fm.qp_solve_batch_device(*qp_args, workspace=ws, want_aux=True)
        ws = out["workspace"]
        if world > 1:
            g0 = time.perf_counter()
            if rehearsal:
                gather.pack(out["x"].cpu(), out["fval"].cpu(), out["exitflag"].cpu(), out["iter"].cpu())
            else:
                gather.pack(out["x"], out["fval"], out["exitflag"], out["iter"])
            out["all"] = gather.gather()
            if not rehearsal:
                torch.cuda.synchronize(dev)
            gather_ms.append(1e3 * (time.perf_counter() - g0))   # host-side: includes waiting for this rank's solve
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    out = None
    for _ in range(args.warmup):
        out = step()
    barrier()
    L.fsaempc_qp_set_timing(1)
    solve_ms, prep_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        a, b = C.c_double(0), C.c_double(0)
        L.fsaempc_qp_get_timing(C.byref(a), C.byref(b))       # HIP events on the launch stream; syncs this step
        prep_ms.append(a.value); solve_ms.append(b.value)
    barrier()
    t_local = time.perf_counter() - t0
    L.fsaempc_qp_set_timing(0)
    rdev = "cpu" if rehearsal else dev
    t_job = shard.max_over_ranks(t_local, device=rdev)        # MAX over ranks

    flags = out["exitflag"].cpu().numpy()
    iters = out["iter"].cpu().numpy()
    kkt_dev = out["kkt"].cpu().numpy()
    polished = out["polished"].cpu().numpy()
    TOL_KKT = 1e-6                                            # the specified tolerance (BASELINE.json north_star)
    solved = int(((flags == 0) & (kkt_dev <= TOL_KKT)).sum())   # an instance counts only if it is solved to that tolerance
    n_ok = shard.max_over_ranks(float(-solved), device=rdev)  # min over ranks via max of negatives
    solved_total = shard.sum_over_ranks(float(solved), device=rdev)
    solved_per_rank = [solved]
    if world > 1:   # per-rank solved counts from the gathered block itself (exit flag 0), as every rank sees them
        _, _, fl_all, _ = shard.ResultGather.unpack(out["all"], nV)
        solved_per_rank = [int((fl_all[r * Bl:(r + 1) * Bl] == 0).sum().item()) for r in range(world)]
    # fused mode of SURVEY 8(d) (x0, x_ref, x_lin, u_lin -> u_opt, x_opt: construction + solve + post-solve), timed
    # separately after the headline loop; reported in `config`, never as `value`
    fx0, fxr, fxl, ful = up(x0), up(xr), up(xl), up(ul)
    stepper.step(fx0, fxr, fxl, ful); torch.cuda.synchronize(dev)
    tf0 = time.perf_counter()
    for _ in range(max(2, min(args.steps, 5))):
        fo = stepper.step(fx0, fxr, fxl, ful)
    torch.cuda.synchronize(dev)
    t_fused = time.perf_counter() - tf0
    fused_rate = float((fo["exitflag"] == 0).sum().item()) * max(2, min(args.steps, 5)) / t_fused
    # the fused step's four phases by HIP events on the launch stream (one more step, outside every timed region)
    L.fsaempc_qp_set_timing(1)
    stepper.step(fx0, fxr, fxl, ful)
    ph = [C.c_double(0) for _ in range(4)]
    fused_phases = None
    if L.fsaempc_ltv_get_timing(*[C.byref(p_) for p_ in ph]) == 0:
        fused_phases = dict(zip(("build", "prep", "solve", "post"), (round(p_.value, 4) for p_ in ph)))
    L.fsaempc_qp_set_timing(0)
    if os.environ.get("FSAEMPC_BENCH_DEBUG"):
        print("fused: %.2f ms per step, solved %d" % (1e3 * t_fused / max(2, min(args.steps, 5)), int((fo["exitflag"] == 0).sum().item())), file=sys.stderr)
    mean_it = float(iters.mean())
    # SURVEY 8(d) extras, rank 0's shard, outside the timed region: iteration / status histograms and the worst relative
    # KKT residual of the returned (x, lambda), evaluated with torch on the device (plumbing, not the product path)
    it_hist = {int(k_): int(c_) for k_, c_ in zip(*np.unique(iters, return_counts=True))}
    fl_hist = {int(k_): int(c_) for k_, c_ in zip(*np.unique(flags, return_counts=True))}
    chk = fm.qp_solve_batch_device(*qp_args, workspace=ws, want_lambda=True)
    torch.cuda.synchronize(dev)
    Hm, gv, Am, lbv, ubv, lbAv, ubAv = qp_args
    xs, lam = chk["x"], chk["lam"]
    AT = Am.transpose(1, 2)                                   # (B, nC, nV): A is stored column-major nC x nV per instance
    Hx = torch.bmm(Hm, xs.unsqueeze(2)).squeeze(2)
    Ax = torch.bmm(AT, xs.unsqueeze(2)).squeeze(2)
    r_d = Hx + gv - lam[:, :nV] - torch.bmm(Am, lam[:, nV:].unsqueeze(2)).squeeze(2)
    sc_d = torch.clamp(torch.maximum(gv.abs().amax(1), Hx.abs().amax(1)), min=1.0)
    v = torch.cat([xs, Ax], 1); lo_ = torch.cat([lbv, lbAv], 1); hi_ = torch.cat([ubv, ubAv], 1)
    fin_lo, fin_hi = lo_ > -1e9, hi_ < 1e9
    viol = torch.maximum(torch.where(fin_lo, lo_ - v, torch.zeros_like(v)), torch.where(fin_hi, v - hi_, torch.zeros_like(v))).clamp(min=0)
    sc_p = torch.clamp(v.abs().amax(1), min=1.0)
    fv = chk["fval"].abs().clamp(min=1.0)
    comp = torch.where(lam > 0, lam * torch.where(fin_lo, v - lo_, torch.zeros_like(v)), -lam * torch.where(fin_hi, hi_ - v, torch.zeros_like(v))).abs()
    kkt = torch.stack([r_d.abs().amax(1) / sc_d, viol.amax(1) / sc_p, comp.amax(1) / fv], 1)
    okm = chk["exitflag"] == 0
    kkt_max = [float(t) for t in kkt[okm].amax(0).cpu()] if bool(okm.any()) else [float("nan")] * 3
    value = solved_total * args.steps / t_job                  # an instance counts only if its exit flag is 0
    flops_iter = 2.0 * nC * nV * nV + nV ** 3 / 3.0 + 4.0 * nC * nV + 2.0 * nV * nV
    flops_launch = flops_iter * mean_it * Bl
    k_ms = float(np.mean(solve_ms))
    achieved = flops_launch / (k_ms * 1e-3) / 1e12
    bytes_solve = 8.0 * (nV * nV + nV + nC * nV + 2 * nV + 2 * nC) + 8.0 * (nV + 2)

    # tile count / border width as qp_make_dims chooses them: nV mod 16 in 1..4 is a border; else LTV-shaped QPs keep their trailing
    # slack columns as the border behind a core padded to a multiple of 16
    rem = nV % 16
    ns_ = 4 if nC == 10 * (nV - 4) else (1 if nC == 3 * (nV - 1) else 0)
    if nV >= 16 and 1 <= rem <= 4:
        Tt, NBt = nV // 16, (1 if rem == 1 else 4)
    elif ns_ and nV >= 20:   # (every LTV-shaped QP since the final sweeps of round 3, qp_make_dims)
        Tt, NBt = (nV - ns_ + 15) // 16, (1 if ns_ == 1 else 4)
    else:
        Tt, NBt = (nV + 15) // 16, 0
    kernel_name = ("qp_solve_kernel<%d, %d>" % (Tt, NBt)) if Tt <= 7 else "qp_wg_kernel<%d, %d, 8>" % (Tt, 4 if NBt else 0)   # qp_launch's choice
    res = {
        "metric": "QP solves/sec (batched LTV-MPC, N=%d nx=%d nu=2 fp64)" % (N, nx),
        "value": value, "unit": "QP solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * t_job / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: batch=%d independent condensed QPs per GPU, %s model, N=%d, nV=%d, nC=%d, generic mode (dense H,g,A,bounds resident in HBM)"
                               % ("BASELINE configs[1]" if (args.model == "kinematic" and N == 40 and Bl == 4096) else "non-headline shape", Bl, args.model, N, nV, nC),
                   "batch_per_gpu": Bl, "global_batch": Btot, "track": "fsg2019", "seed": 20190,
                   "mean_ipm_iterations": mean_it, "solved_min_per_rank": int(-n_ok), "solved_total": int(solved_total),
                   "tol_kkt": TOL_KKT, "kkt_reported_by_kernel_max_rank0": float(kkt_dev[flags == 0].max()) if (flags == 0).any() else None,
                   "on_vertex_fraction_rank0": float((polished > 0).mean()),
                   "prep_kernel_ms": float(np.mean(prep_ms)), "solve_kernel_ms": k_ms,
                   "fused_mode_qp_per_s_rank0": fused_rate, "fused_mode_phase_ms_rank0": fused_phases,
                   "iteration_histogram_rank0": it_hist, "exitflag_histogram_rank0": fl_hist,
                   "max_rel_kkt_rank0": {"stationarity": kkt_max[0], "primal": kkt_max[1], "complementarity": kkt_max[2]},
                   "exitflag0_per_rank": solved_per_rank,
                   "gather_ms_per_step_rank0": float(np.mean(gather_ms[-args.steps:])) if gather_ms else 0.0,
                   "parallelism": "instances sharded index-pure over %d GPU(s); one all_gather_into_tensor of the (B/G) x (nV+2) result block (x, fval, exitflag/iter) per step" % world},
        "roofline": {"bound": "mfma", "kernel": kernel_name, "achieved": achieved,
                     "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "mfma_busy": None,
                     "flops_per_launch": flops_launch, "algorithmic_bytes_per_solve": bytes_solve,
                     "hbm_frac_one_pass": bytes_solve * Bl / (k_ms * 1e-3) / 8e12},
    }

    # Counter-based figures of the dominant kernel are ARCHIVED measurements (rocprofv3 --pmc needs its own passes and cannot run
    # inside this process): HBM-side traffic (tools/pmc_traffic.py: FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections) and
    # matrix-core busy share (SQ_VALU_MFMA_BUSY_CYCLES pass), committed under profiles/round3/ for exactly this workload (headline and configs[2] shape)
    tag = "%sN%d_B%d" % ("kin" if args.model == "kinematic" else "dyn", N, Bl)
    for key, fname, field in (("traffic", "pmc_traffic_%s.json" % tag, "traffic_bytes_per_launch"), ("mfma_busy", "pmc_mfma_%s.json" % tag, "mfma_busy_frac")):
        path = os.path.join(ROOT, "profiles", "round3", fname)
        if os.path.exists(path):
            with open(path) as f:
                res["roofline"][key] = json.load(f)[field]
            res["roofline"][key + "_source"] = "archived PMC pass of this workload: profiles/round3/" + fname
    if res["roofline"]["traffic"] is not None:
        res["roofline"]["traffic_unit"] = "bytes per launch (L2-miss side; Infinity-Cache hits are counted, so an upper bound on HBM bytes)"

    res["cpu_baseline"] = None    # timed on rank 0 at N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle as orc
        otr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
        avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        cores = min(avail, 16)   # a one-GPU job's CPU share on the GPU box is 16 cores (the affinity mask shows all 256 of the host;
                                 # 256 OpenMP threads on a 16-core share measured 1.6k QP/s against 3.2k with 16)
        H, g, A, lb, ub, lbA, ubA = (t.cpu().numpy() for t in qp_args)
        o = orc.default_opts()                                  # same options as the GPU leg (refinement on)
        pilot = min(Bl, 4 * cores)
        t1 = time.perf_counter()
        orc.qp_solve_batch(H[:pilot], g[:pilot], A[:pilot], lb[:pilot], ub[:pilot], lbA[:pilot], ubA[:pilot], o, threads=cores, want_lambda=False)
        rate = pilot / (time.perf_counter() - t1)
        sample = int(max(pilot, min(Bl, rate * 15.0)))        # ~15 s of CPU work, bounded by the batch
        t1 = time.perf_counter()
        _, _, fl_c, it_c, _, used = orc.qp_solve_batch(H[:sample], g[:sample], A[:sample], lb[:sample], ub[:sample], lbA[:sample], ubA[:sample], o,
                                                        threads=cores, want_lambda=False)
        tc = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": sample / tc, "unit": "QP solves/s", "cores": int(used), "kind": "port",
                               "sample": "first %d instances of the same batch, same (H,g,A,bounds) and the same options (interior-point method + "
                                         "active-set refinement), plain-C oracle with OpenMP over instances on %d threads = the 16-core CPU share a one-GPU job gets on "
                                         "this box (affinity mask: %d); reference-equivalent CPU path (MATLAB+qpOASES cannot run here)" % (sample, cores, avail),
                               "mean_ipm_iterations": float(it_c.mean()), "solved": int((fl_c == 0).sum())}
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def dry_run(args, rank, world):
    """Launch / shard / gather / report path without a GPU (tests/test_abi_cpu.py): every rank contributes a zero block of x for its
    index-pure shard; the gather, the max-over-ranks timing and rank 0's JSON line are the real code."""
    import torch
    import torch.distributed as dist
    from fsae_mpc_amd import shard
    Bl = args.batch
    lo, hi = shard.shard_range(Bl * world, rank, world)
    nV = 3
    x_local = torch.full((hi - lo, nV), float(rank), dtype=torch.float64)
    gather = shard.ResultGather(Bl, nV, world, "cpu")      # the same packed (B/G) x (nV + 2) exchange as the GPU run
    zf = torch.zeros(Bl, dtype=torch.float64); zi = torch.full((Bl,), rank, dtype=torch.int32)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gather.pack(x_local, zf, torch.zeros(Bl, dtype=torch.int32), zi)
        blk = gather.gather()
    t_job = shard.max_over_ranks(time.perf_counter() - t0, device="cpu")
    x_all, _, fl_all, it_all = shard.ResultGather.unpack(blk, nV)
    ok = bool((x_all[lo:hi] == float(rank)).all()) and x_all.shape[0] == Bl * world and bool((fl_all == 0).all()) and bool((it_all[lo:hi] == rank).all())
    if rank == 0:
        print(json.dumps({"metric": "QP solves/sec (dry run: no solves)", "value": 0.0, "unit": "QP solves/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * t_job / max(1, args.steps), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "dry run", "gathered_rows": int(x_all.shape[0]), "gather_ok": ok},
                          "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
