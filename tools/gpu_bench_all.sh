#!/bin/bash
# every bench workload once (short), outputs under gpurun_out/$TAG.  usage: gpurun -- 'bash tools/gpu_bench_all.sh tag [workloads]'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-bench}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for w in ${2:-csv configs2 configs3 configs4 json sr collapse}; do
  extra="--steps 10 --warmup 2"
  [ "$w" = csv ] && extra=""
  timeout 400 python bench.py --workload $w $extra > "$OUT/$w.json" 2> "$OUT/$w.err"
  echo "== $w rc=$?"; tail -c 600 "$OUT/$w.err"
  python - "$OUT/$w.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no json", e); sys.exit(0)
print("value %.4g %s steps=%d ms/step=%.3f" % (d["value"], d["unit"], d["steps"], d["ms_per_step"]), "roofline", d.get("roofline") and (d["roofline"]["kernel"], d["roofline"]["frac"]), "int", d.get("int_roofline") and d["int_roofline"]["frac"])
print("  cpu", d.get("cpu_baseline") and {k: d["cpu_baseline"].get(k) for k in ("value","cores","nproc","cpu_model","all_cores")})
for k in ("overlapped_lanes","pcie_inclusive"):
    if d.get(k): print("  ", k, json.dumps(d[k])[:400])
print("  kernels", {k: v["ms_per_step"] for k,v in d["kernels"].items()})
PY
done
