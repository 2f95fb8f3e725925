#!/bin/bash
# csv parse kernel under the microscope: phase ablation (TFGPU_CSV_ABLATE) and PMC sets, for the kernel named by $2
# usage: gpurun -- 'bash tools/gpu_csv_probe.sh tag csv_parse_lanes "1 2 3 4 10 11 12 13 0" nsets'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-probe}; K=${2:-csv_parse_lanes}; ABL=${3:-"1 2 3 4 10 11 12 13 0"}; NSETS=${4:-3}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for a in $ABL; do
  TFGPU_CSV_ABLATE=$a timeout 60 python bench.py --steps 3 --passes 1 --warmup 1 --cpu-rows 0 --prof-steps 5 --overlap-lanes 0 --pcie-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('ablate=$a', {n:round(k[n]['avg_ms'],4) for n in k if n=='$K'})" | tee -a "$OUT/ablate_$K.txt"
done
[ "$NSETS" -gt 0 ] && bash tools/gpu_pmc2.sh "$K" "$TAG/pmc_$K" csv "$NSETS" > "$OUT/pmc_$K.log" 2>&1
cat "$OUT/pmc_$K/summary.txt" 2>/dev/null
