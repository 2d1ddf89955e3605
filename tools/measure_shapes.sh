# Measurement session 2: headline line again (with the archived counters in place), closed loop, configs at size, other shapes, warm-start A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/meas_r3; mkdir -p $O
python bench.py --steps 10 --warmup 2 > $O/bench_line_kinN40_B4096.json 2> $O/bench_err.log; echo "bench rc=$?"
python bench.py --model dynamic --horizon 40 --batch 2048 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_line_dynN40_B2048.json 2>> $O/bench_err.log; echo "bench dyn rc=$?"
python bench.py --model kinematic --horizon 20 --batch 4096 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_line_kinN20_B4096.json 2>> $O/bench_err.log; echo "bench kin20 rc=$?"
timeout -k 10 300 python tools/closed_loop_bench.py --model kinematic > $O/closed_loop_config4_kinematic_N40_B2048.json 2> $O/cl_kin_err.log; echo "cl kin rc=$?"
timeout -k 10 400 python tools/closed_loop_bench.py --model dynamic > $O/closed_loop_config4_dynamic_N40_B2048.json 2> $O/cl_dyn_err.log; echo "cl dyn rc=$?"
timeout -k 10 300 python tools/config_at_size.py > $O/config3_dynamic_N60_65536.json 2> $O/c3_err.log; echo "config3 rc=$?"
timeout -k 10 300 python tools/config_at_size.py --horizon 80 --total 4096 --chunk 512 > $O/config5_shape_dynamic_N80_4096.json 2> $O/c5_err.log; echo "config5 rc=$?"
python bench.py --model dynamic --horizon 60 --batch 4096 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_line_dynN60_B4096.json 2>> $O/bench_err.log; echo "bench dyn60 rc=$?"
python bench.py --model dynamic --horizon 80 --batch 2048 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_line_dynN80_B2048.json 2>> $O/bench_err.log; echo "bench dyn80 rc=$?"
timeout -k 10 600 python tools/warm_start_ab.py 1024 60 > $O/warm_start_ab.json 2> $O/ws_err.log; echo "warm start rc=$?"
for f in $O/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d.get("config",{})
print({k:d.get(k) for k in ("metric","value","ms_per_step","seconds_total")}, {k:c[k] for k in c if k in ("solved_total","abnormal_exit_pct","exitflag_histogram_driving_cars","exitflag_histogram","mean_ipm_iterations","on_vertex_fraction_rank0","solve_kernel_ms","qps_of_driving_cars","cars_lost")}, d.get("roofline",{}) and {k:d["roofline"].get(k) for k in ("kernel","frac","traffic","mfma_busy")})
PY
done
