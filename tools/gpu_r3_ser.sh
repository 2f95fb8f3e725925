#!/bin/bash
# round 3 visit for the text serializers: parity tests, then configs3 / configs2 with the chunk-walk variants
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3ser}; OUT=gpurun_out/$TAG; mkdir -p $OUT
[ -z "$SKIP_TESTS" ] && { timeout 900 python -m pytest tests/test_serializers.py tests/test_queue_serializers.py tests/test_gpu_fullsize.py -m gpu -q -x --tb=short 2>&1 | tail -8 > $OUT/pytest.log; grep -E "passed|failed|error" $OUT/pytest.log; }
run() {  # label, workload, env...
  local label=$1 w=$2; shift 2
  env "$@" timeout 200 python bench.py --workload $w --steps 10 --warmup 2 --cpu-rows 0 --prof-steps 5 --pcie-steps 0 2>$OUT/bench_$label.err > $OUT/bench_$label.json
  python - $OUT/bench_$label.json $label <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels"]
    print(sys.argv[2], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), {n: round(v["avg_ms"],4) for n,v in k.items() if n.startswith("ser_")})
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
}
run c3_lds configs3 A=0
for t in ${2:-}; do run c3_t$t configs3 TFGPU_SER_CHUNK_BYTES=$t; done
for e in ${3:-}; do run c3_$e configs3 ${e//,/ }; done
run c2_lds configs2 A=0
