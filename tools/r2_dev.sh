cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
export SHAPES="0,40,4096;1,40,2048;0,38,1024"
FSAEMPC_LIB=fsae-mpc_amd/lib/libfsaempc_dev.so timeout -k 10 300 python tools/r2_wg_check.py > gpurun_out/r2f/wg.log 2>&1; echo "wg rc=$?"
tail -5 gpurun_out/r2f/wg.log
FSAEMPC_LIB=fsae-mpc_amd/lib/libfsaempc_devst.so timeout -k 10 300 python tools/phase_profile.py kin 40 4096 > gpurun_out/r2f/phase_kin40.txt 2>&1; echo "rc=$?"
tail -17 gpurun_out/r2f/phase_kin40.txt
FSAEMPC_LIB=fsae-mpc_amd/lib/libfsaempc_devst.so timeout -k 10 300 python tools/phase_profile.py dyn 40 1024 > gpurun_out/r2f/phase_dyn40.txt 2>&1; echo "rc=$?"
tail -17 gpurun_out/r2f/phase_dyn40.txt
