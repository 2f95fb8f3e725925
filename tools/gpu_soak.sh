#!/bin/bash
# Soak: the randomized -m gpu parity tests under other seeds (TFGPU_TEST_SEED shifts every committed seed).
# usage: gpurun -- 'bash tools/gpu_soak.sh 101 102 103'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/soak
for s in "$@"; do
  TFGPU_TEST_SEED=$s timeout 240 python -m pytest tests -m gpu -q --tb=short -k "random or matches_oracle or mutations or fuzz or large or cdc or stream or deepsizeof or exchange or raw or tile_path or flat_lines" 2>&1 | tail -25 > gpurun_out/soak/seed_$s.log
  echo "seed $s: $(tail -1 gpurun_out/soak/seed_$s.log)"
done
