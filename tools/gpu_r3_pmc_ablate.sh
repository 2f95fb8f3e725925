#!/bin/bash
# instruction counts of csv_parse_regular per phase: one PMC pass per TFGPU_CSV_ABLATE value
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3pmc}
for a in ${2:-1 2 3 4 10 11 12 13 0}; do
  echo "== ablate $a"
  TFGPU_CSV_ABLATE=$a bash tools/gpu_pmc2.sh csv_parse_regular $TAG/a$a csv 1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE" | awk '{print $2, $3, $4}' | tr '\n' ' '
  echo
done
