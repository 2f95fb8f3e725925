#!/bin/bash
# serializer tile order A/B (TFGPU_SER_ORDER: 0 = a row group's chunks neighbours, 1 = a chunk's row groups neighbours, 2 = 0 + XCD-aware) on configs3,
# then configs2 with its pull+push leg
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-serord}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
timeout 600 python -m pytest tests -m gpu -q -x --tb=short -k "serial or marshal or parquet" 2>&1 | tail -3
for o in 2 1 0; do
  TFGPU_SER_ORDER=$o timeout 300 python bench.py --workload configs3 --steps 5 --passes 1 --warmup 2 --cpu-rows 0 --pcie-steps 0 > "$OUT/configs3_order$o.json" 2> "$OUT/configs3_order$o.err"
  python - "$OUT/configs3_order$o.json" $o <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("order", sys.argv[2], d["value"], d["ms_per_step"], {k: round(v["ms_per_step"],4) for k,v in d["kernels"].items() if k.startswith("ser_")})
PY
done
timeout 400 python bench.py --workload configs2 --steps 5 --passes 1 --warmup 2 --cpu-rows 0 > "$OUT/configs2.json" 2> "$OUT/configs2.err"; tail -c 300 "$OUT/configs2.err"
python - "$OUT/configs2.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d.get("pull_push_concurrent"))[:1500])
PY
