cd $GRAFT_REPO_ROOT
out=gpurun_out/${FRAG_TAG:-fd}; mkdir -p $out
FRAG_SHAPES="0,39,64,0" 
for n in ${FRAG_CTL:-goodg}; do FSAEMPC_LIB=$PWD/build/frag/libfrag_$n.so SHAPES="0,39,64,0" timeout -k 10 120 python tests/harness/check_vs_oracle.py 2>&1 | grep "^model" | cut -c1-160; done
for p in ${FRAG_POINTS:-A B C}; do
  FRAG_SAVE=$PWD/$out/dump_$p.npz timeout -k 10 120 python tests/harness/fragility/frag_dump.py cmp build/frag/libfrag_good$p.so build/frag/libfrag_bad$p.so $p > $out/cmp_$p.log 2>&1 || echo "cmp $p rc=$?"
  grep -v "amdgpu.ids" $out/cmp_$p.log
done
