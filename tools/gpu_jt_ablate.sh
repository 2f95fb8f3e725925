#!/bin/bash
# phase by phase cost of the tile front on the SR workload (TFGPU_JT_ABLATE: leave after phase n)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-jt_ablate}; mkdir -p $OUT
for a in 1 2 3 4 5 0; do
  TFGPU_JT_ABLATE=$a timeout 200 python bench.py --workload sr --cpu-rows 0 --overlap-lanes 0 --pcie-steps 0 --steps 5 --warmup 2 > $OUT/a$a.json 2> $OUT/a$a.err
  python - $OUT/a$a.json $a <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ablate", sys.argv[2], {k:v["avg_ms"] for k,v in d["kernels"].items() if k in ("sr_parse_tiles","sr_parse_frames")})
PY
done
