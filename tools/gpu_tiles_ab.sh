#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-tiles_ab}; mkdir -p $OUT
for w in json sr; do
  timeout 300 python bench.py --workload $w --cpu-rows 0 --overlap-lanes 0 --pcie-steps 0 --steps 20 > $OUT/$w.json 2> $OUT/$w.err
  python - $OUT/$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), {k:v["avg_ms"] for k,v in d["kernels"].items() if "parse" in k})
PY
done
