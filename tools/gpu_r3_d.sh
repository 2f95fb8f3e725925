#!/bin/bash
# csv bench line (default flags, no CPU legs) + the step timeline
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python bench.py --cpu-rows 0 --cpu-all-rows 0 --pcie-steps 0 2>$OUT/bench_csv.err > $OUT/bench_csv.json; python - $OUT/bench_csv.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("csv value %.4g ms/step %.3f steps %d" % (d["value"], d["ms_per_step"], d["steps"]), "overlapped", (d.get("overlapped_lanes") or {}).get("ms_per_step"), "kernel sum %.3f" % sum(v["ms_per_step"] for v in d["kernels"].values()))
PY
bash tools/gpu_r3_trace.sh $TAG/trace
