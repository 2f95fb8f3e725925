#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats.
# usage: gpurun -- 'bash tools/gpu_check.sh [tag]'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -60 > "$OUT/pytest_gpu.log"
tail -3 "$OUT/pytest_gpu.log"
timeout 300 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
timeout 900 python bench.py --steps 5 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 3000 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --hip-trace --stats -f csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --cpu-rows 0 --overlap-lanes 0 --pcie-steps 0 > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
find "$OUT/prof" -name '*kernel_stats*' | head; find "$OUT/prof" -name '*kernel_trace*' -size +20M -delete
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"; f=$(find "$OUT/prof" -name "*hip_api_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f"
