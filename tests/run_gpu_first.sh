#!/bin/bash
# first GPU contact: transformer parity
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_transformers.py -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/first.log
