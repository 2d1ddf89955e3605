cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout -k 10 420 python tools/r2_wg_check.py > gpurun_out/r2d/wg.log 2>&1; echo "wg rc=$?"
tail -12 gpurun_out/r2d/wg.log
timeout -k 10 300 python tools/phase_profile.py kin 40 4096 > gpurun_out/r2d/phase_kin40.txt 2>&1; echo "rc=$?"
tail -17 gpurun_out/r2d/phase_kin40.txt
timeout -k 10 300 python tools/phase_profile.py dyn 40 1024 > gpurun_out/r2d/phase_dyn40.txt 2>&1; echo "rc=$?"
tail -17 gpurun_out/r2d/phase_dyn40.txt
