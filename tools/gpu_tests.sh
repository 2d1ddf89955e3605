cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r2t/tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r2t/tests.log | cut -c1-300
if grep -q "Memory access fault" gpurun_out/r2t/tests.log; then echo FAULT; exit 1; fi
