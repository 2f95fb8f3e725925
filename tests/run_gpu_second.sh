#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_transformers.py tests/test_gpu_csv.py -m gpu -q --tb=short 2>&1 | tail -150 > gpurun_out/second.log
python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err
tail -5 gpurun_out/smoke.log; tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
