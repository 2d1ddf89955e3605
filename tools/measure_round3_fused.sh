# Re-measurement of the fused figures after the construction kernel moved H onto the matrix cores (round 3, late)
cd $GRAFT_REPO_ROOT
O=gpurun_out/meas_r3c; mkdir -p $O
python bench.py --steps 10 --warmup 2 > $O/bench_line_kinN40_B4096.json 2> $O/bench_err.log; echo "bench rc=$?"
python bench.py --model dynamic --horizon 60 --steps 5 --warmup 1 > $O/bench_line_dynN60_B4096.json 2>> $O/bench_err.log; echo "bench dyn60 rc=$?"
timeout -k 10 300 python tools/config_at_size.py > $O/config3_dynamic_N60_65536.json 2> $O/c3_err.log; echo "config3 rc=$?"
timeout -k 10 300 python tools/config_at_size.py --horizon 80 --total 4096 --chunk 512 > $O/config5_shape_dynamic_N80_4096.json 2> $O/c5_err.log; echo "config5 rc=$?"
timeout -k 10 300 python tools/closed_loop_bench.py --model kinematic > $O/closed_loop_config4_kinematic_N40_B2048.json 2> $O/cl_kin_err.log; echo "cl kin rc=$?"
timeout -k 10 400 python tools/closed_loop_bench.py --model dynamic > $O/closed_loop_config4_dynamic_N40_B2048.json 2> $O/cl_dyn_err.log; echo "cl dyn rc=$?"
timeout -k 10 400 python tools/closed_loop_bench.py --model dynamic --warm > $O/closed_loop_config4_dynamic_N40_B2048_warm_start.json 2> $O/cl_dynw_err.log; echo "cl dyn warm rc=$?"
timeout -k 10 300 python tools/closed_loop_bench.py --model kinematic --warm > $O/closed_loop_config4_kinematic_N40_B2048_warm_start.json 2> $O/cl_kinw_err.log; echo "cl kin warm rc=$?"
timeout -k 10 200 python tests/harness/build_time.py 2>&1 | grep build_qp
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d.get("config",{})
print(sys.argv[1].split("/")[-1], {k:d.get(k) for k in ("value","ms_per_step","seconds_total")}, {k:c[k] for k in c if k in ("fused_mode_qp_per_s_rank0","abnormal_exit_pct","mean_ipm_iterations","solve_kernel_ms","prep_kernel_ms","cars_lost","warm_start")})
PY
done
