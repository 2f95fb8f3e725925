#!/bin/bash
# json / sr A/B on one box: parity tests of the touched files, then `bench.py --workload W` with the new tile kernel and with the old one
# usage: gpurun -- 'bash tools/gpu_ab_json.sh tag "json sr" ENVVAR'   (ENVVAR=0 selects the old kernel: TFGPU_JSON_QUICK / TFGPU_SR_QUICK)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-abj}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; WLS=${2:-json}; VAR=${3:-TFGPU_JSON_QUICK}
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -k "json or generic or parser or sr or confluent" 2>&1 | tail -5
for w in $WLS; do for v in 1 0; do
  env $VAR=$v timeout 300 python bench.py --workload $w --steps 5 --passes 1 --warmup 2 --cpu-rows 0 > "$OUT/bench_${w}_$v.json" 2> "$OUT/bench_${w}_$v.err"
  echo "== $w $VAR=$v rc=$?"; tail -c 300 "$OUT/bench_${w}_$v.err"
  python - "$OUT/bench_${w}_$v.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: round(v["ms_per_step"],4) for k,v in d["kernels"].items()})
PY
done; done
