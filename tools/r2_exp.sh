cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
export FSAEMPC_LIB=fsae-mpc_amd/lib/libfsaempc_dev.so
for cfg in "8,1" "8,0" "4,1" "4,0" "2,0"; do
  export SHAPES="0,40,4096"
  echo "== FSAEMPC_WG=$cfg"
  FSAEMPC_WG=$cfg timeout -k 10 200 python tools/r2_wg_check.py 2>&1 | grep "model" 
done
for cfg in "8,0" "4,0"; do
  export SHAPES="1,40,2048"
  echo "== dyn FSAEMPC_WG=$cfg"
  FSAEMPC_WG=$cfg timeout -k 10 200 python tools/r2_wg_check.py 2>&1 | grep "model"
done
