#!/bin/bash
# device time of csv_parse_regular when it leaves after phase n / runs only one group of cell kinds (TFGPU_CSV_ABLATE)
cd "$GRAFT_REPO_ROOT" || exit 1
for a in ${1:-1 2 3 4 10 11 12 13 0}; do
  TFGPU_CSV_ABLATE=$a timeout 60 python bench.py --steps 3 --warmup 1 --cpu-rows 0 --prof-steps 3 --overlap-lanes 0 --pcie-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('ablate=$a', {n:k[n]['avg_ms'] for n in k if 'csv_parse_tiles'==n})"
done
