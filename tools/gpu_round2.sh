#!/bin/bash
# One evidence visit: the whole -m gpu suite, smoke, every bench workload (JSON lines kept), rocprofv3 kernel stats of the
# headline line and of configs3 / debezium, and the HBM-traffic PMC passes of the headline kernel.  Every step has its own
# timeout and writes under gpurun_out/$TAG.   usage: gpurun -- 'bash tools/gpu_round2.sh tag'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r02}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
date +%s > "$OUT/t0"
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 > "$OUT/pytest_gpu.log"
echo "== gpu tests"; tail -3 "$OUT/pytest_gpu.log"
timeout 200 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "== smoke"; tail -1 "$OUT/smoke.log"
timeout 600 python bench.py > "$OUT/bench_csv.json" 2> "$OUT/bench_csv.err"; echo "== csv rc=$?"
for w in configs0 configs2 configs3 configs4 configs4d json sr collapse debezium; do
  timeout 400 python bench.py --workload $w > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; echo "== $w rc=$?"
done
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "no json", e); continue
    r=d.get("roofline") or {}
    print(os.path.basename(f), "value %.4g steps=%d ms/step=%.3f" % (d["value"], d["steps"], d["ms_per_step"]), "roofline", r.get("kernel"), r.get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
export TMPDIR=/tmp
for w in csv configs3 debezium; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/$OUT/prof_$w" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --workload $w --steps 5 --warmup 2 --cpu-rows 0 --overlap-lanes 0 --pcie-steps 0 > "$GRAFT_REPO_ROOT/$OUT/prof_$w.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_$w.err" )
  find "$OUT/prof_$w" -name '*kernel_trace*' -delete
  f=$(find "$OUT/prof_$w" -name "*kernel_stats.csv" | head -1); echo "== rocprof $w"; [ -n "$f" ] && head -8 "$f" | cut -c1-150
done
bash tools/gpu_pmc2.sh "csv_parse_regular" $TAG/pmc_csv csv 5 > "$OUT/pmc_csv.log" 2>&1; tail -12 "$OUT/pmc_csv.log"
if [ -n "$ICACHE" ]; then  # instruction-cache behaviour of the largest kernel (one pass)
  bash tools/gpu_pmc2.sh "ser_tile_write" $TAG/pmc_icache configs3 1 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" > "$OUT/pmc_icache.log" 2>&1; tail -8 "$OUT/pmc_icache.log"
fi
echo "elapsed $(( $(date +%s) - $(cat $OUT/t0) )) s"
