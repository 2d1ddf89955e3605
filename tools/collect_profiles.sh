# Round-2 measurement session (one gpurun call): bench line, rocprofv3 kernel trace, the PMC passes (each in its own run, as
# MI355X_MICROARCH.md prescribes), reducers.  Output under gpurun_out/prof_r2/; the summaries are copied into profiles/round2/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd $R
python bench.py --steps 10 --warmup 2 > $O/bench_line_kinN40_B4096.json 2> $O/bench_err.log
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_err.log
echo "ktrace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc1_err.log
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc2_err.log
echo "pmc write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc3_err.log
echo "pmc mfma done"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc_wait -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc4_err.log
echo "pmc wait done"
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic_kinN40_B4096.json
python tools/pmc_mfma.py $O/pmc_mfma $O/pmc_mfma_kinN40_B4096.json qp_solve_kernel $O/pmc_wait
find $O -name "*stats*.csv" | head
