mkdir -p gpurun_out/r3U
python -m pytest tests -m gpu -x -q -k "launch_order or closed_loop" > gpurun_out/r3U/pytest_sel.txt 2>&1; echo pytest rc=$?; tail -3 gpurun_out/r3U/pytest_sel.txt
for i in 1 2; do
  FSAEMPC_QP_ORDER=0 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r3U/bench_off_$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r3U/bench_on_$i.json 2>/dev/null
done
FSAEMPC_QP_ORDER=0 python bench.py --no-cpu-baseline --model dynamic --horizon 60 --steps 5 > gpurun_out/r3U/bench_dyn60_off.json 2>/dev/null
python bench.py --no-cpu-baseline --model dynamic --horizon 60 --steps 5 > gpurun_out/r3U/bench_dyn60_on.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3U/bench*.json")):
    d=json.load(open(f)); print(f, round(d["value"]), d["config"]["solve_kernel_ms"], d["config"]["prep_kernel_ms"], d["roofline"]["frac"])
PY
for m in kinematic dynamic; do
  FSAEMPC_QP_ORDER=0 python tools/closed_loop_bench.py --model $m --batch 2048 --steps 100 --no-launch-hint > gpurun_out/r3U/cl_${m}_order_off.json 2>/dev/null
  python tools/closed_loop_bench.py --model $m --batch 2048 --steps 100 --no-launch-hint > gpurun_out/r3U/cl_${m}_order_cold.json 2>/dev/null
  python tools/closed_loop_bench.py --model $m --batch 2048 --steps 100 > gpurun_out/r3U/cl_${m}_order_hint.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3U/cl_*.json")):
    d=json.load(open(f)); print(f, {k: d[k] for k in d if k in ("value","metric","ms_per_step")}, d.get("config",{}).get("abnormal_exit_pct"))
PY
