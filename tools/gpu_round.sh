#!/bin/bash
# One GPU-box visit at the end of a session: the new parity tests first, then the whole -m gpu suite, smoke, the bench
# line, the SR / JSON / collapse side benchmarks, and the rocprofv3 kernel stats of the bench line.  Every step has its own timeout and
# writes under gpurun_out/$TAG so a cut-off visit still leaves what it reached.
# usage: gpurun -- 'bash tools/gpu_round.sh [tag]'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r01n}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
date +%s > "$OUT/t0"
timeout 300 python -m pytest tests/test_queue_serializers.py tests/test_confluent_sr.py -m gpu -q --tb=short 2>&1 | tail -60 > "$OUT/pytest_new.log"
echo "== new tests"; tail -4 "$OUT/pytest_new.log"
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -60 > "$OUT/pytest_gpu.log"
echo "== all gpu tests"; tail -4 "$OUT/pytest_gpu.log"
timeout 200 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "== smoke"; tail -2 "$OUT/smoke.log"
timeout 600 python bench.py --steps 5 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "== bench"; tail -c 1500 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
timeout 300 python bench.py --workload sr --steps 5 --warmup 1 > "$OUT/sr_bench.json" 2> "$OUT/sr_bench.err"; echo "== sr bench"; tail -c 2500 "$OUT/sr_bench.json"; tail -3 "$OUT/sr_bench.err"
timeout 300 python bench.py --workload json --steps 5 --warmup 1 > "$OUT/json_bench.json" 2> "$OUT/json_bench.err"; echo "== json bench"; tail -c 2000 "$OUT/json_bench.json"; tail -3 "$OUT/json_bench.err"
timeout 300 python bench.py --workload collapse --steps 5 --warmup 1 > "$OUT/collapse_bench.json" 2> "$OUT/collapse_bench.err"; echo "== collapse bench"; tail -c 1500 "$OUT/collapse_bench.json"; tail -3 "$OUT/collapse_bench.err"
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --cpu-rows 0 --overlap-lanes 0 --pcie-steps 0 > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
find "$OUT/prof" -name '*kernel_trace*' -size +20M -delete
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); echo "== rocprof"; [ -n "$f" ] && head -12 "$f"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/$OUT/prof_sr" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --workload sr --steps 3 --warmup 1 --cpu-rows 0 > "$GRAFT_REPO_ROOT/$OUT/prof_sr_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_sr.err" )
find "$OUT/prof_sr" -name '*kernel_trace*' -size +20M -delete
f=$(find "$OUT/prof_sr" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -10 "$f"
echo "elapsed $(( $(date +%s) - $(cat $OUT/t0) )) s"
