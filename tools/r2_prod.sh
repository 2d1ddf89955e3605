cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
export SHAPES="1,80,256;0,90,128;1,70,128;0,72,128;0,40,4096"
timeout -k 10 400 python tools/r2_wg_check.py > gpurun_out/r2h/prod.log 2>&1; echo "prod rc=$?"
tail -6 gpurun_out/r2h/prod.log
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r2h/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r2h/tests.log
