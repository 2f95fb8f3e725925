#!/bin/bash
# ONE script for every kind of GPU visit (replaces the per-visit one-offs of rounds 1-3).  Everything is written under
# gpurun_out/$TAG; copy what is to be judged into profiles/ afterwards (tools/collect_profiles.py TAG does that).
#
#   gpurun --timeout T -- 'bash tools/gpu_visit.sh MODE TAG [args…]'
#
#   tests    TAG [pytest -k expression]           the -m gpu suite (or a part of it) + smoke
#   bench    TAG "w1 w2 …" [extra bench.py args]   bench.py --workload w for each (default steps), one JSON line per workload
#   ab       TAG ENVVAR "w1 w2 …"                 each workload with ENVVAR=1 and ENVVAR=0 (short runs, kernel tables printed)
#   ablate   TAG ENVVAR KERNEL WORKLOAD "n1 n2 …"  KERNEL's time with ENVVAR=n (TFGPU_CSV_ABLATE / TFGPU_JT_ABLATE / TFGPU_SER_ABLATE)
#   pmc      TAG KERNEL WORKLOAD NSETS ["C1 C2 …"]  rocprofv3 --pmc passes for one kernel (sets of tools/gpu_pmc2.sh, or one custom set)
#   stats    TAG "w1 w2 …"                        rocprofv3 --kernel-trace --stats per workload (the timed region only, no side legs)
#   traffic  TAG                                  the FETCH_SIZE / WRITE_SIZE / instruction-counter passes of the stamped kernels → gpurun_out/TAG/pmc_traffic.json
#   timeline TAG WORKLOAD ANCHOR ["bench args" [pass]]  one pass of WORKLOAD as a timeline of dispatches (tools/timeline.py): offsets, durations, idle gaps
#   evidence TAG                                  tests + smoke + HBM-traffic PMC (pmc_traffic.json) + every bench line + stats
cd "$GRAFT_REPO_ROOT" || exit 1
MODE=${1:-tests}; TAG=${2:-visit}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; shift 2
export TMPDIR=/tmp
QUIET="--cpu-rows 0 --overlap-lanes 0 --pcie-steps 0"

line() {  # one bench JSON line, summarised
python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("   no json:", e); sys.exit(0)
r=d.get("roofline") or {}
print("   value %.4g %s  ms/step %.3f (passes %s)  roofline %s %.4f consistent=%s traffic=%s  parity=%s" % (d["value"], d["unit"], d["ms_per_step"], d.get("passes_per_step"), r.get("kernel"), r.get("frac") or 0, r.get("consistent"), r.get("traffic"), d.get("parity_checked_rows")))
print("   kernels", {k: round(v["ms_per_step"],4) for k,v in d["kernels"].items()})
for k in ("cpu_baseline","overlapped_lanes","pcie_inclusive","pull_push_concurrent","d2h_inclusive"):
    if d.get(k): print("  ", k, json.dumps(d[k])[:500])
PY
}
keepjson() {  # a bench output file holds ONE JSON object: what a library printed to stdout in front of it (RCCL's banner in configs4) goes to FILE.log
python - "$1" <<'PY'
import json, sys
p = sys.argv[1]
try:
    lines = [l for l in open(p).read().splitlines() if l.strip()]
except OSError:
    sys.exit(0)
for i in range(len(lines) - 1, -1, -1):
    try:
        json.loads(lines[i])
    except ValueError:
        continue
    rest = lines[:i] + lines[i + 1:]
    if rest:
        open(p + ".log", "w").write("\n".join(rest) + "\n")
    open(p, "w").write(lines[i] + "\n")
    break
PY
}
tests() {
  timeout 1500 python -m pytest tests -m gpu -q --tb=short ${1:+-k "$1"} 2>&1 | tail -40 > "$OUT/pytest_gpu.log"; echo "== gpu tests"; tail -4 "$OUT/pytest_gpu.log"
  timeout 200 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "== smoke"; tail -1 "$OUT/smoke.log"
}
bench() {  # workloads, extra args
  for w in $1; do
    name=${w// /_}
    timeout 600 python bench.py --workload $w $2 > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "== bench $w rc=$?"; tail -c 200 "$OUT/bench_$name.err" | grep -v amdgpu.ids
    keepjson "$OUT/bench_$name.json"; line "$OUT/bench_$name.json"
  done
}
stats() {
  for w in $1; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/$OUT/prof_$w" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --workload $w --steps 5 --passes 1 --warmup 2 $QUIET > "$GRAFT_REPO_ROOT/$OUT/prof_$w.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_$w.err" )
    find "$OUT/prof_$w" -name '*kernel_trace*' -delete
    f=$(find "$OUT/prof_$w" -name "*kernel_stats.csv" | head -1); echo "== rocprof $w"; [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$w.csv" && head -7 "$f" | cut -c1-150
    rm -rf "$OUT/prof_$w"
  done
}
traffic() {  # HBM bytes per launch of the dominant kernels: FETCH_SIZE / WRITE_SIZE in their own passes, stamped with the source's sha256
  for spec in "csv_parse_regular csv tf_csv.hip" "json_parse_quick json tf_jsonquick.inc" "ser_chunk_write configs3 tf_serialize.hip"; do
    set -- $spec
    bash tools/gpu_pmc2.sh "$1" $TAG/pmc_$2 $2 1 "FETCH_SIZE" > "$OUT/pmc_$2_fetch.log" 2>&1
    bash tools/gpu_pmc2.sh "$1" $TAG/pmc_$2 $2 1 "WRITE_SIZE" > "$OUT/pmc_$2_write.log" 2>&1
  done
  bash tools/gpu_pmc2.sh "mask_hmac" $TAG/pmc_mask csv 1 "SQ_INSTS_VALU SQ_WAVES" > "$OUT/pmc_mask.log" 2>&1
  bash tools/gpu_pmc2.sh "csv_parse_regular" $TAG/pmc_csv_valu csv 1 "SQ_INSTS_VALU SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > "$OUT/pmc_csv_valu.log" 2>&1
  #  the busy side of the same kernel: active VALU, waits, thread-cycles — a second pass (eight SQ counters a pass)
  bash tools/gpu_pmc2.sh "csv_parse_regular" $TAG/pmc_csv_busy csv 1 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" > "$OUT/pmc_csv_busy.log" 2>&1
  python tools/pmc_traffic.py "$OUT" && cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json
}
case $MODE in
  tests) tests "$1" ;;
  bench) bench "${1:-csv}" "$2" ;;
  ab) for w in $2; do for v in 1 0; do echo "== $w $1=$v"; env $1=$v timeout 300 python bench.py --workload $w --steps 5 --passes 1 --warmup 2 $QUIET > "$OUT/ab_${w}_$v.json" 2>/dev/null; line "$OUT/ab_${w}_$v.json"; done; done ;;
  ablate) V=""; [ "$1" = TFGPU_CSV_ABLATE ] && V="TFGPU_LIB_VARIANT=ablate"   # the CSV kernels' profiling branches exist only in the ablate build: tools/build_variant.sh ablate tf_csv.hip -DTF_CSV_ABLATE_BUILD=1 (before gpurun)
    for a in $4; do env $V $1=$a timeout 120 python bench.py --workload $3 --steps 2 --passes 1 --warmup 1 --prof-steps 5 $QUIET 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('$1=$a', {n:round(k[n]['avg_ms'],4) for n in k if n=='$2'})" | tee -a "$OUT/ablate_$2.txt"; done ;;
  pmc) bash tools/gpu_pmc2.sh "$1" "$TAG/pmc_$1" "${2:-csv}" "${3:-2}" "$4"; cp "$OUT/pmc_$1/summary.txt" "$OUT/pmc_$1.txt" ;;
  stats) stats "${1:-csv}" ;;
  traffic) traffic ;;   # the PMC passes behind profiles/pmc_traffic.json alone (after a change to a stamped source file): copy gpurun_out/TAG/pmc_traffic.json to profiles/
  timeline)
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d "$GRAFT_REPO_ROOT/$OUT/tl_$1" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --workload $1 --steps 4 --passes 1 --warmup 2 --prof-steps 1 ${3:-$QUIET} > "$GRAFT_REPO_ROOT/$OUT/tl_$1.json" 2> "$GRAFT_REPO_ROOT/$OUT/tl_$1.err" )
    f=$(find "$OUT/tl_$1" -name '*kernel_trace.csv' | head -1)
    [ -n "$f" ] && python tools/timeline.py "$f" "$2" ${4:-3} | tee "$OUT/timeline_$1.txt"
    rm -rf "$OUT/tl_$1" ;;
  evidence)
    date +%s > "$OUT/t0"
    tests
    traffic
    timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_csv.json" 2> "$OUT/bench_csv.err"; echo "== csv (the driver's command line) rc=$?"; keepjson "$OUT/bench_csv.json"; line "$OUT/bench_csv.json"
    bench "configs0" "--from-rows"
    bench "configs2 configs3 configs4 configs4d json sr sr_proto collapse debezium debezium_sr"
    timeout 400 python bench.py --workload configs4 --sink debezium > "$OUT/bench_configs4_debezium.json" 2> "$OUT/bench_configs4_debezium.err"; echo "== configs4 --sink debezium rc=$?"; keepjson "$OUT/bench_configs4_debezium.json"; line "$OUT/bench_configs4_debezium.json"
    timeout 400 python bench.py --workload collapse --toast 0.5 > "$OUT/bench_collapse_toast.json" 2> "$OUT/bench_collapse_toast.err"; echo "== collapse --toast 0.5 rc=$?"; keepjson "$OUT/bench_collapse_toast.json"; line "$OUT/bench_collapse_toast.json"
    stats "csv configs2 configs3 json sr debezium debezium_sr configs4d"
    echo "elapsed $(( $(date +%s) - $(cat $OUT/t0) )) s" ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
