#!/bin/bash
# csv A/B on one box: the csv parity tests, then the bench line's kernels with csv_parse_lanes (default) and csv_parse_regular (TFGPU_CSV_LANES=0)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_csv.py -m gpu -q -x --tb=short 2>&1 | tail -5
for v in 0 1; do
  TFGPU_CSV_LANES=$v timeout 300 python bench.py --steps 5 --warmup 3 --cpu-rows 0 --overlap-lanes 0 --pcie-steps 0 > "$OUT/bench_lanes$v.json" 2> "$OUT/bench_lanes$v.err"
  echo "== lanes=$v rc=$?"; tail -c 300 "$OUT/bench_lanes$v.err"
  python - "$OUT/bench_lanes$v.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]/d["passes_per_step"], {k: round(v["ms_per_step"],4) for k,v in d["kernels"].items()})
PY
done
