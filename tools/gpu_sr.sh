#!/bin/bash
# SR tile parser: parity tests, then sr / configs2 with and without the tile path
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-sr}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_confluent_sr.py tests/test_gpu_json.py -m gpu -q --tb=short 2>&1 | tail -15 > $OUT/pytest.log; grep -E "passed|failed" $OUT/pytest.log
for w in sr configs2; do
for v in 1 0; do
  TFGPU_SR_TILES=$v TFGPU_JSON_TILE_DEBUG=1 timeout 300 python bench.py --workload $w --cpu-rows 0 --overlap-lanes 0 --pcie-steps 0 --steps 20 > $OUT/${w}_tiles$v.json 2> $OUT/${w}_tiles$v.err
  grep -m1 "sr tiles" $OUT/${w}_tiles$v.err
  python - $OUT/${w}_tiles$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), {k:(v["avg_ms"], v.get("alg_gb_s")) for k,v in d["kernels"].items() if k.startswith("sr_")})
PY
done
done
