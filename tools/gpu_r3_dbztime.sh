#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $OUT
TFGPU_DBZ_TIMING=1 timeout 300 python bench.py --workload configs4 --sink debezium --steps 5 --warmup 2 --cpu-rows 0 --prof-steps 1 > $OUT/t.json 2> $OUT/timing.err
tail -28 $OUT/timing.err
