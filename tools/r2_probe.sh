# round-2 first probe of the round-1 kernel: available counters, MFMA-busy PMC pass, GPU test sanity
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
cd $R
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"
python bench.py --steps 10 --warmup 2 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_mfma.json 2> $O/pmc_mfma_err.log; echo "pmc mfma rc=$?"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc_wait -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_wait.json 2> $O/pmc_wait_err.log; echo "pmc wait rc=$?"
tail -3 $O/tests.log
