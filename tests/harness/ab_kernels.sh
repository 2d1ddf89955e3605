# A/B of the two solve kernels on the shapes where both exist (FSAEMPC_QP_KERNEL=v1|wg), against the oracle; product options
cd $GRAFT_REPO_ROOT
out=gpurun_out/${AB_TAG:-ab}; mkdir -p $out
S="${AB_SHAPES:-0,40,4096,1;1,40,2048,1;1,38,1024,1;0,39,1024,1;1,46,1024,1;0,48,1024,1;1,48,1024,1;0,56,1024,1;1,56,512,1;0,60,512,1;1,60,512,1;0,64,512,1;1,64,512,1}"
for k in v1 wg; do
  FSAEMPC_QP_KERNEL=$k SHAPES="$S" timeout -k 10 600 python tests/harness/check_vs_oracle.py > $out/$k.log 2>&1 || echo "$k rc=$?"
  echo "== $k"; grep "^model" $out/$k.log | cut -c1-260
done
