#!/bin/bash
# the round's new tests on the device
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3e}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_strictify.py tests/test_pipeline.py tests/test_shard.py tests/test_sql.py tests/test_gpu_csv.py tests/test_serializers.py -m gpu -q --tb=short 2>&1 | tail -25 > $OUT/pytest.log; tail -12 $OUT/pytest.log
