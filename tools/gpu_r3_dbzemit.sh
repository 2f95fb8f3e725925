#!/bin/bash
# the Debezium emitter on the MI355X: its parity tests, the configs4 bench line with --sink debezium, rocprof kernel stats of the same command
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05h}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_dbz_emit.py -m gpu -x -q > $OUT/pytest_dbzemit.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_dbzemit.log
tail -5 $OUT/pytest_dbzemit.log
timeout 600 python bench.py --workload configs4 --sink debezium --steps 10 --warmup 2 > $OUT/bench_configs4_debezium.json 2> $OUT/bench_configs4_debezium.err; echo "bench rc=$?"
tail -c 1500 $OUT/bench_configs4_debezium.json; tail -3 $OUT/bench_configs4_debezium.err
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o dbz -- python $GRAFT_REPO_ROOT/bench.py --workload configs4 --sink debezium --steps 10 --warmup 2 --cpu-rows 0 > $OUT/bench_rocprof.json 2> $OUT/bench_rocprof.err
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -25 "$f" > $OUT/configs4_debezium_kernel_stats.csv && head -12 "$f"
rm -rf $OUT/prof
