#!/bin/bash
# GPU check of tfgpu_collapse: parity tests, the side benchmark, a rocprofv3 kernel trace of it.  Run through gpurun.
mkdir -p gpurun_out/collapse
cd "$(dirname "$0")/.." || exit 1
timeout 300 python -m pytest tests/test_gpu_collapse.py -x -q < /dev/null > gpurun_out/collapse/test.log 2>&1; echo "exit $?" >> gpurun_out/collapse/test.log
timeout 200 python bench.py --workload collapse --steps 10 --warmup 3 < /dev/null > gpurun_out/collapse/bench.json 2> gpurun_out/collapse/bench.err
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/collapse/prof -o collapse -- python bench.py --workload collapse --steps 5 --warmup 2 --cpu-rows 0 < /dev/null > gpurun_out/collapse/prof.log 2>&1
f=$(ls gpurun_out/collapse/prof/*/*kernel_stats.csv gpurun_out/collapse/prof/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then head -25 "$f" > gpurun_out/collapse/kernel_stats.csv; fi
find gpurun_out/collapse/prof -name "*.csv" ! -name "*stats*" -delete 2>/dev/null
find gpurun_out/collapse/prof -name "*.db" -delete 2>/dev/null
tail -5 gpurun_out/collapse/test.log; cat gpurun_out/collapse/bench.json; tail -3 gpurun_out/collapse/bench.err; cat gpurun_out/collapse/kernel_stats.csv 2>/dev/null | cut -c1-160
