#!/bin/bash
# round 3 visit b: sql + csv parity on the device, the bench lines of csv / configs2 / configs3 with their side legs
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_sql.py tests/test_gpu_csv.py tests/test_gpu_transformers.py -m gpu -q -x --tb=short 2>&1 | tail -15 > $OUT/pytest.log; tail -5 $OUT/pytest.log
for w in csv configs2 configs3; do
  timeout 400 python bench.py --workload $w 2>$OUT/bench_$w.err > $OUT/bench_$w.json; echo "$w exit $?"; tail -c 3000 $OUT/bench_$w.json; tail -5 $OUT/bench_$w.err
done
