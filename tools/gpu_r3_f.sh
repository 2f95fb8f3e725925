#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3f}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_parquet.py tests/test_gpu_transformers.py tests/test_strictify.py -m gpu -q --tb=short 2>&1 | tail -12 > $OUT/pytest.log; tail -6 $OUT/pytest.log
timeout 400 python bench.py --workload configs3 --cpu-rows 0 2>$OUT/bench_configs3.err > $OUT/bench_configs3.json; python - $OUT/bench_configs3.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("configs3 value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), json.dumps(d.get("parquet_source"))[:900])
PY
tail -3 $OUT/bench_configs3.err
