cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3Y; mkdir -p $O
timeout -k 10 240 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/marker -- python3 $GRAFT_REPO_ROOT/tools/fused_phases.py kinematic 40 4096 3 > $O/fused_under_rocprof.json 2> $O/marker_err.log
echo rc=$?
find $O/marker -name "*.csv" > $O/files.txt; cat $O/files.txt
f=$(grep -m1 "marker_api_trace\|marker" $O/files.txt || true)
if [ -n "$f" ]; then head -20 "$f"; fi
d=$(grep -m1 domain_stats $O/files.txt || true); if [ -n "$d" ]; then cat "$d"; fi
