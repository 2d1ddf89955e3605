# investigation of the build-variant fragility: runs the retired unbordered instantiation of each library in build/frag against the oracle
cd $GRAFT_REPO_ROOT
out=gpurun_out/${FRAG_TAG:-frag}; mkdir -p $out
for lib in build/frag/libfrag_*.so; do
  n=$(basename $lib .so)
  FSAEMPC_LIB=$PWD/$lib SHAPES="${FRAG_SHAPES:-1,38,64,0;0,39,64,0}" timeout -k 10 120 python tests/harness/check_vs_oracle.py > $out/$n.log 2>&1 || echo "$n rc=$?"
  echo "== $n"; grep "^model" $out/$n.log | cut -c1-200
done
