#!/bin/bash
# One short GPU visit for the CSV ingest: its parity tests, the bench line with per-kernel times, rocprofv3 kernel stats.
# usage: gpurun -- 'bash tools/gpu_csv.sh tag'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-csv}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_csv.py tests/test_gpu_fullsize.py tests/test_gpu_transformers.py -m gpu -q --tb=short -x 2>&1 | tail -30 > "$OUT/pytest_csv.log"; tail -5 "$OUT/pytest_csv.log"
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
TFGPU_CSV_DEBUG=1 timeout 600 python bench.py --steps 10 --warmup 2 --cpu-rows 0 --pcie-steps 0 > "$OUT/bench.json" 2> "$OUT/bench.err"; grep "tfgpu csv" "$OUT/bench.err" | head -2
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"])
for k,v in d["kernels"].items(): print("  %-28s %8.4f ms x%.0f" % (k, v["avg_ms"], v["launches_per_step"]))
PY
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --cpu-rows 0 --overlap-lanes 0 --pcie-steps 0 > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
find "$OUT/prof" -name '*kernel_trace*' -size +20M -delete
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f"
