#!/bin/bash
# measurement only: where ser_cell_write / ser_cell_len spend their time (text cells vs the rest), configs3 shape
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-serab}; mkdir -p "$OUT"
for a in ${2:-0 1 2 3}; do
  TFGPU_SER_ABLATE=$a timeout 120 python bench.py --workload configs3 --steps 10 --warmup 2 --cpu-rows 0 > "$OUT/ab$a.json" 2> "$OUT/ab$a.err"
  python - "$OUT/ab$a.json" $a <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ablate", sys.argv[2], {k: v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("ser_")})
PY
done
