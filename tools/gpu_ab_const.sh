#!/bin/bash
# A/B of the constant-table cache (TFGPU_NO_CONST_CACHE=1 = every descriptor table uploaded again), after the -m gpu suite
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-ab_const}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15 > $OUT/pytest_gpu.log; tail -2 $OUT/pytest_gpu.log
for w in csv configs3 configs2 debezium; do
  for v in 0 1; do
    TFGPU_NO_CONST_CACHE=$v timeout 300 python bench.py --workload $w --cpu-rows 0 --cpu-all-rows 0 --overlap-lanes 0 --pcie-steps 0 > $OUT/${w}_nocache$v.json 2> $OUT/${w}_nocache$v.err
    python - $OUT/${w}_nocache$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]))
PY
  done
done
