#!/bin/bash
# instruction counts of the chunk kernels with and without text cells: one PMC pass per TFGPU_SER_ABLATE value
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3serpmc}
for a in ${2:-0 1 2}; do
  echo "== ablate $a"
  TFGPU_SER_ABLATE=$a bash tools/gpu_pmc2.sh "ser_chunk_(len|write)" $TAG/a$a configs3 1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" | awk '{print $1, $3, $4}' | sed 's/void//' 
  TFGPU_SER_ABLATE=$a timeout 200 python bench.py --workload configs3 --steps 10 --warmup 2 --cpu-rows 0 --prof-steps 5 --pcie-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({n: round(v['avg_ms'],4) for n,v in d['kernels'].items() if n.startswith('ser_')})"
done
