#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05o; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_dbz_emit.py -m gpu -x -q > $OUT/pytest_dbzemit.log 2>&1; tail -2 $OUT/pytest_dbzemit.log
TFGPU_DBZ_WALK=0 timeout 300 python -m pytest tests/test_gpu_dbz_emit.py -m gpu -x -q > $OUT/pytest_dbzemit_cellmajor.log 2>&1; tail -2 $OUT/pytest_dbzemit_cellmajor.log
timeout 300 python bench.py --workload configs4 --sink debezium --steps 20 --warmup 3 --cpu-rows 0 > $OUT/bench_walk.json 2> $OUT/bench_walk.err
TFGPU_DBZ_WALK=0 timeout 300 python bench.py --workload configs4 --sink debezium --steps 20 --warmup 3 --cpu-rows 0 > $OUT/bench_cellmajor.json 2> $OUT/bench_cellmajor.err
python - <<'PY'
import json
for f in ("bench_walk", "bench_cellmajor"):
    d = json.loads(open("gpurun_out/r05o/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], {k: (v["ms_per_step"], v.get("alg_gb_s")) for k, v in d["kernels"].items() if k.startswith("dbz")}, d["roofline"]["kernel"], d["roofline"]["frac"])
PY
