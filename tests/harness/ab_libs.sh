# A/B of whole libraries (FSAEMPC_LIB) on the same shapes against the oracle.  usage: AB_LIBS="a.so b.so" AB_SHAPES="..." bash tests/harness/ab_libs.sh
cd $GRAFT_REPO_ROOT
out=gpurun_out/${AB_TAG:-abl}; mkdir -p $out
S="${AB_SHAPES:-0,40,4096,1;1,40,2048,1;0,39,1024,1;0,20,4096,1}"
for rep in 1 2; do
for lib in $AB_LIBS; do
  n=$(basename $lib .so)
  FSAEMPC_LIB=$PWD/$lib SHAPES="$S" timeout -k 10 300 python tests/harness/check_vs_oracle.py > $out/$n.$rep.log 2>&1 || echo "$n rc=$?"
  echo "== $n (run $rep)"; grep "^model" $out/$n.$rep.log | cut -c1-40,95-260
done
done
