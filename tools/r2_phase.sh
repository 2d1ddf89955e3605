cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout -k 10 300 python tools/phase_profile.py kin 40 4096 > gpurun_out/r2c/phase_kin40.txt 2>&1; echo "rc=$?"
cat gpurun_out/r2c/phase_kin40.txt | tail -18
timeout -k 10 300 python tools/phase_profile.py dyn 40 1024 > gpurun_out/r2c/phase_dyn40.txt 2>&1; echo "rc=$?"
cat gpurun_out/r2c/phase_dyn40.txt | tail -18
