#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05m; mkdir -p $OUT
timeout 300 python bench.py --workload configs4 --steps 20 --warmup 3 > $OUT/bench_configs4.json 2> $OUT/bench_configs4.err; echo "rc=$?"; tail -c 300 $OUT/bench_configs4.err
timeout 300 python bench.py --workload configs4 --sink debezium --steps 20 --warmup 3 > $OUT/bench_configs4_debezium.json 2> $OUT/bench_configs4_debezium.err; echo "rc=$?"; tail -c 300 $OUT/bench_configs4_debezium.err
python - <<'PY'
import json
for f in ("bench_configs4", "bench_configs4_debezium"):
    d = json.loads(open("gpurun_out/r05m/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("messages_out_per_step"))
PY
