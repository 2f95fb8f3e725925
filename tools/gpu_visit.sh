#!/bin/bash
# A mid-round GPU visit: the whole -m gpu suite, smoke, then the named bench workloads (short).
# usage: gpurun -- 'bash tools/gpu_visit.sh tag "workloads"'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-visit}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
date +%s > "$OUT/t0"
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -80 > "$OUT/pytest_gpu.log"
echo "== all gpu tests"; tail -15 "$OUT/pytest_gpu.log"
timeout 200 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "== smoke"; tail -2 "$OUT/smoke.log"
bash tools/gpu_bench_all.sh "$TAG" "${2:-csv configs4}"
echo "elapsed $(( $(date +%s) - $(cat $OUT/t0) )) s"
