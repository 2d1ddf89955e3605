cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweeps
for spec in "kin 20 0 16384" "kin 40 0 16384" "dyn 40 0 8192 2048" "dyn 60 0 4096 1024" "kin 64 0 4096" "dyn 64 0 2048 1024"; do
  timeout -k 10 400 python tools/sweep_flags.py $spec 2>/dev/null | tee -a gpurun_out/sweeps/robustness_sweeps.jsonl
done
