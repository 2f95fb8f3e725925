cd "$GRAFT_REPO_ROOT"; T=r07r; O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
Q="--cpu-rows 0 --overlap-lanes 0 --pcie-steps 0"
for w in json debezium; do
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/$O/prof_$w" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --workload $w --steps 5 --passes 1 --warmup 2 $Q > "$GRAFT_REPO_ROOT/$O/prof_$w.json" 2> "$GRAFT_REPO_ROOT/$O/prof_$w.err" )
f=$(find "$O/prof_$w" -name "*kernel_stats.csv" | head -1); echo "== rocprof $w"; [ -n "$f" ] && cp "$f" "$O/kernel_stats_$w.csv" && head -9 "$f" | cut -c1-120
rm -rf "$O/prof_$w"
done
echo "== debezium words=0"; ( cd /tmp && TFGPU_DBZ_COPY_WORDS=0 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/$O/prof_d0" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --workload debezium --steps 5 --passes 1 --warmup 2 $Q > /dev/null 2>&1 ); f=$(find "$O/prof_d0" -name "*kernel_stats.csv" | head -1); head -7 "$f" | cut -c1-120; rm -rf "$O/prof_d0"
