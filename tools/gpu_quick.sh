#!/bin/bash
# quick perf check of the CSV bench line: per-kernel times + VALU/SALU counts of one kernel.  usage: gpurun -- 'bash tools/gpu_quick.sh tag [kernel-regex] [pytest-files]'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-q}; K=${2:-csv_parse_regular}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
if [ -n "$3" ]; then timeout 600 python -m pytest $3 -m gpu -q --tb=short -x 2>&1 | tail -15; fi
TFGPU_CSV_DEBUG=1 timeout 600 python bench.py --steps 10 --warmup 2 --cpu-rows 0 --pcie-steps 0 --overlap-lanes 0 > "$OUT/bench.json" 2> "$OUT/bench.err"; grep "tfgpu csv" "$OUT/bench.err" | head -1
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
for k,v in d["kernels"].items(): print("  %-28s %8.4f ms x%.0f" % (k, v["avg_ms"], v["launches_per_step"]))
PY
bash tools/gpu_pmc2.sh "$K" "$TAG" csv 1 | awk '{print $2, $3, $4}'
