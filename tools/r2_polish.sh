cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
FSAEMPC_LIB=fsae-mpc_amd/lib/libfsaempc_dev.so timeout -k 10 500 python tools/polish_stats.py > gpurun_out/r2i/polish.log 2>&1; echo "rc=$?"
tail -8 gpurun_out/r2i/polish.log
export SHAPES="0,40,4096,1;1,40,2048,1"
FSAEMPC_LIB=fsae-mpc_amd/lib/libfsaempc_dev.so timeout -k 10 300 python tools/r2_wg_check.py 2>&1 | grep model
