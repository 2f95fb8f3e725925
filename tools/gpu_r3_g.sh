#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3g}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_confluent_sr.py tests/test_sql.py -m gpu -q --tb=short 2>&1 | tail -4
for w in ${2:-configs2 sr}; do
timeout 300 python bench.py --workload $w --cpu-rows 0 2>$OUT/bench_$w.err > $OUT/bench_$w.json; python - $OUT/bench_$w.json $w <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[2], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), "kernel sum %.3f" % sum(v["ms_per_step"] for v in k.values()), {n: v["ms_per_step"] for n,v in sorted(k.items(), key=lambda kv:-kv[1]["ms_per_step"])[:6]})
PY
done
