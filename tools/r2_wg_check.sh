cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout -k 10 420 python tools/r2_wg_check.py > gpurun_out/r2b/wg.log 2>&1; echo "wg rc=$?"
tail -15 gpurun_out/r2b/wg.log
FSAEMPC_QP_V1=1 timeout -k 10 300 python tools/r2_wg_check.py > gpurun_out/r2b/v1.log 2>&1; echo "v1 rc=$?"
tail -12 gpurun_out/r2b/v1.log
