#!/bin/bash
# phase ablation of the quick JSON / SR tile kernels (TFGPU_JT_ABLATE=n: leave after phase n; the lines then take the per-line parser, so only the named kernel's time means anything)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-jqabl}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for w in ${2:-json sr}; do
  K=json_parse_quick; [ $w = sr ] && K=sr_parse_quick
  for a in ${3:-1 2 3 4 0}; do
    TFGPU_JT_ABLATE=$a timeout 120 python bench.py --workload $w --steps 2 --passes 1 --warmup 1 --cpu-rows 0 --prof-steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('$w ablate=$a', {n:round(k[n]['avg_ms'],4) for n in k if n=='$K'})" | tee -a "$OUT/ablate_$w.txt"
  done
done
