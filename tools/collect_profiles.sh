set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof2
mkdir -p $O
cd $R
python bench.py --steps 10 --warmup 2 > $O/bench_line.json 2> $O/bench_err.log
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_err.log
echo "ktrace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc1_err.log
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc2_err.log
echo "pmc write done"
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json
FSAEMPC_LIB=fsae-mpc_amd/lib/libfsaempc_stamps.so python tools/phase_profile.py > $O/phase_shares.txt 2>&1
find $O -name "*.csv" | head -20
