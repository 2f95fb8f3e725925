#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS of the csv parse kernel ($2) when it leaves after phase n (TFGPU_CSV_ABLATE): where the instructions are
# usage: gpurun -- 'bash tools/gpu_csv_pmc_ablate.sh tag csv_parse_lanes "2 4 10 11 12 13 0"'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-pmcabl}; K=${2:-csv_parse_lanes}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for a in ${3:-2 4 10 11 12 13 0}; do
  export TFGPU_CSV_ABLATE=$a
  bash tools/gpu_pmc2.sh "$K" "$TAG/a$a" csv 1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" > "$OUT/a$a.log" 2>&1
  echo "== ablate=$a"; grep per_dispatch "$OUT/a$a/summary.txt" | awk '{print "   ", $2, $3}' | tr '\n' ' '; echo
done | tee "$OUT/pmc_ablation_$K.txt"
