#!/bin/bash
# A/B of an env toggle on one bench workload.  usage: gpurun -- 'bash tools/gpu_ab.sh tag workload "ENV=a ENV=b ..."'
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-ab}; mkdir -p "$OUT"; WL=${2:-configs3}
i=0
for e in ${3:-X=0}; do
  i=$((i+1))
  env $e timeout 200 python bench.py --workload $WL --steps 10 --warmup 2 --cpu-rows 0 > "$OUT/$WL.$i.json" 2> "$OUT/$WL.$i.err"
  python - "$OUT/$WL.$i.json" "$e" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], {k: v["ms_per_step"] for k,v in d["kernels"].items()})
except Exception as ex:
    print(sys.argv[2], "failed", ex)
PY
done
