#!/bin/bash
# round 3 visit c: the shard / several-devices tests on the device, the one-process bench mode over two lanes of device 0
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_shard.py tests/test_serializers.py tests/test_gpu_transformers.py -m gpu -q -x --tb=short 2>&1 | tail -15 > $OUT/pytest.log; tail -5 $OUT/pytest.log
timeout 300 python bench.py --devices 0,0 --steps 20 --cpu-rows 0 --cpu-all-rows 0 2>$OUT/bench_devices.err > $OUT/bench_devices.json; echo "devices exit $?"; tail -c 1500 $OUT/bench_devices.json; tail -3 $OUT/bench_devices.err
timeout 300 python bench.py --steps 20 --cpu-rows 0 --cpu-all-rows 0 --pcie-steps 0 --overlap-lanes 0 2>$OUT/bench_csv.err > $OUT/bench_csv.json; echo "csv exit $?"; head -c 600 $OUT/bench_csv.json
