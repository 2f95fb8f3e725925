#!/bin/bash
# tools/gpu_tune.sh TAG WORKLOAD "ENV1=a ENV2=b" "ENV1=c" …  — one short bench run per environment setting (kernel table printed): knob sweeps on one box.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; WL=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting timeout 300 python bench.py --workload $WL --steps 5 --passes 1 --warmup 2 --cpu-rows 0 --overlap-lanes 0 --pcie-steps 0 --other-configs 0 > "$OUT/tune_${WL}_$i.json" 2> "$OUT/tune_${WL}_$i.err"
  python - "$OUT/tune_${WL}_$i.json" "$setting" <<'PY' | tee -a "$OUT/tune_$WL.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = {n: round(v["ms_per_step"], 4) for n, v in d["kernels"].items() if v["ms_per_step"] >= 0.02}
    print("%-60s ms/pass %.3f  %s" % (sys.argv[2], d["ms_per_step"] / d["passes_per_step"], k))
except Exception as e:
    print("%-60s FAILED %s" % (sys.argv[2], e))
PY
done
