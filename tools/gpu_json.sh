#!/bin/bash
# JSON tile parser: parity tests, then the json / configs2-free benches with and without the tile path
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-json}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_json.py -m gpu -q --tb=short 2>&1 | tail -15 > $OUT/pytest_json.log; grep -E "passed|failed" $OUT/pytest_json.log
for v in 1 0; do
  TFGPU_JSON_TILES=$v TFGPU_JSON_TILE_DEBUG=1 timeout 300 python bench.py --workload json --cpu-rows 0 --overlap-lanes 0 --pcie-steps 0 --steps 20 > $OUT/json_tiles$v.json 2> $OUT/json_tiles$v.err
  grep -m1 "json tiles" $OUT/json_tiles$v.err
  python - $OUT/json_tiles$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), {k:(v["avg_ms"], v.get("alg_gb_s")) for k,v in d["kernels"].items() if k.startswith("json")})
PY
done
