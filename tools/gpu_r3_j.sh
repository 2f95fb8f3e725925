#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3j}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_transformers.py tests/test_gpu_fullsize.py -m gpu -q --tb=short 2>&1 | tail -3
for w in configs0 csv; do
timeout 300 python bench.py --workload $w --cpu-rows 0 --cpu-all-rows 0 --pcie-steps 0 --overlap-lanes 0 2>$OUT/bench_$w.err > $OUT/bench_$w.json; python - $OUT/bench_$w.json $w <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[2], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), "mask", k.get("mask_hmac_sha256"), d.get("int_roofline", {}).get("frac"))
PY
done
