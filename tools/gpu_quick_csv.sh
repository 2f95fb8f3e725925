#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-quick}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_transformers.py tests/test_gpu_csv.py tests/test_gpu_collapse.py -m gpu -q --tb=short 2>&1 | tail -6 > $OUT/pytest.log; grep -E "passed|failed" $OUT/pytest.log
for w in csv configs2 configs4; do
timeout 300 python bench.py --workload $w --cpu-rows 0 --cpu-all-rows 0 --overlap-lanes 0 --pcie-steps 0 > $OUT/$w.json 2> $OUT/$w.err
python - $OUT/$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), {k:v["avg_ms"] for k,v in d["kernels"].items() if "gather" in k or "compact" in k})
PY
done
