#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05o; mkdir -p $OUT
timeout 300 python bench.py --workload configs4 --sink debezium > $OUT/bench_configs4_debezium.json 2> $OUT/bench_configs4_debezium.err; echo rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05o/bench_configs4_debezium.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"])
PY
