cd $GRAFT_REPO_ROOT
O=gpurun_out/meas_r3b; mkdir -p $O
timeout -k 10 600 python tools/warm_start_ab.py 1024 60 > $O/warm_start_ab.json 2> $O/ws_err.log; echo "warm start rc=$?"
for sb in 0 2; do FSAEMPC_SLACK_BORDER=$sb python bench.py --model kinematic --horizon 20 --batch 4096 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_line_kinN20_B4096_policy$sb.json 2>> $O/bench_err.log; echo "kin20 policy $sb rc=$?"; done
python bench.py --model kinematic --horizon 20 --batch 4096 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_line_kinN20_B4096.json 2>> $O/bench_err.log
for sb in 0 1; do FSAEMPC_SLACK_BORDER=$sb python bench.py --model dynamic --horizon 60 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_line_dynN60_B4096_policy$sb.json 2>> $O/bench_err.log; echo "dyn60 policy $sb rc=$?"; done
timeout -k 10 500 python tools/cl_failures_check.py --model kinematic --cars 2048 --steps 200 --keep 24 --save $O/cl_fail_kin.npz > $O/cl_fail_kin.json 2> $O/clf_kin_err.log; echo "cl fail kin rc=$?"
timeout -k 10 500 python tools/cl_failures_check.py --model dynamic --cars 2048 --steps 100 --keep 40 --save $O/cl_fail_dyn.npz > $O/cl_fail_dyn.json 2> $O/clf_dyn_err.log; echo "cl fail dyn rc=$?"
bash tools/robustness_sweeps.sh > $O/sweeps.log 2>&1; echo "sweeps rc=$?"
for f in $O/bench_line*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]
print(sys.argv[1].split("/")[-1], "%.1f"%d["value"], "ms/step %.2f"%d["ms_per_step"], "solve %.2f prep %.2f"%(c["solve_kernel_ms"],c["prep_kernel_ms"]), d["roofline"]["kernel"], "frac %.3f"%d["roofline"]["frac"], "vertex %.3f"%c["on_vertex_fraction_rank0"])
PY
done
cat $O/warm_start_ab.json | head -60; cat $O/cl_fail_kin.json $O/cl_fail_dyn.json; cat gpurun_out/sweeps/robustness_sweeps.jsonl | tail -6
