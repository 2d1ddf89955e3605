# Measurement session (one gpurun call): for the headline workload (kinematic N = 40, configs[1]) and the configs[2] shape (dynamic
# N = 60, workgroup kernel): bench line, rocprofv3 kernel trace, the PMC passes (each in its own run, as MI355X_MICROARCH.md
# prescribes), reducers.  Output under gpurun_out/prof_r3/; the summaries are then copied into profiles/round3/.
# usage: bash tools/collect_profiles.sh [kin|dyn|both]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r3
mkdir -p $O
cd $R
what=${1:-both}
one() {   # tag, kernel substring, bench arguments...
  tag=$1; kern=$2; shift 2
  python bench.py --steps 10 --warmup 2 "$@" > $O/bench_line_$tag.json 2> $O/bench_err_$tag.log
  echo "bench $tag done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_$tag -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $O/bench_under_rocprof_$tag.json 2> $O/rocprof_err_$tag.log
  echo "ktrace $tag done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $O/pmc1_err_$tag.log
  echo "pmc fetch $tag done"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $O/pmc2_err_$tag.log
  echo "pmc write $tag done"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $O/pmc3_err_$tag.log
  echo "pmc mfma $tag done"
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc_wait_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $O/pmc4_err_$tag.log
  echo "pmc wait $tag done"
  python tools/pmc_traffic.py $O/pmc_fetch_$tag $O/pmc_write_$tag $O/pmc_traffic_$tag.json $kern
  python tools/pmc_mfma.py $O/pmc_mfma_$tag $O/pmc_mfma_$tag.json $kern $O/pmc_wait_$tag
  f=$(find $O/ktrace_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_$tag.csv
}
if [ "$what" = kin ] || [ "$what" = both ]; then one kinN40_B4096 qp_solve_kernel; fi
if [ "$what" = dyn ] || [ "$what" = both ]; then one dynN60_B4096 qp_wg_kernel --model dynamic --horizon 60; fi
ls $O/*.json $O/*.csv
