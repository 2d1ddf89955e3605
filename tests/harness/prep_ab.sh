cd $GRAFT_REPO_ROOT
export FSAEMPC_LIB=$PWD/fsae-mpc_amd/lib/libfsaempc_prep.so
for thr in 256 64 0; do
  for spec in "0 40 4096" "0 20 4096" "1 40 2048"; do
    set -- $spec
    FSAEMPC_PREP_THR=$thr timeout -k 10 200 python tests/harness/wg_check.py $1 $2 2 $3 2>&1 | grep "polish=1" | sed "s/^/thr $thr model $1 N $2: /" | cut -c1-140
  done
done
