#!/bin/bash
# One evidence visit: the whole -m gpu suite, smoke, every bench workload (JSON lines kept), rocprofv3 kernel stats of every
# workload whose fraction README / DESIGN quote, and the HBM-traffic PMC passes of the dominant kernels — from which
# pmc_traffic.json is WRITTEN here, stamped with the sha256 of the kernel's source file (bench.py prints roofline.traffic only
# when that stamp matches the source it runs).  Every step has its own timeout and writes under gpurun_out/$TAG.
# usage: gpurun -- 'bash tools/gpu_round3.sh tag'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
date +%s > "$OUT/t0"
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 > "$OUT/pytest_gpu.log"
echo "== gpu tests"; tail -3 "$OUT/pytest_gpu.log"
timeout 200 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "== smoke"; tail -1 "$OUT/smoke.log"
# traffic first: the bench lines below then carry the measured number of this very build
for spec in "csv_parse_regular csv tf_csv.hip" "ser_chunk_write configs3 tf_serialize.hip"; do
  set -- $spec
  bash tools/gpu_pmc2.sh "$1" $TAG/pmc_$2 $2 1 "FETCH_SIZE" > "$OUT/pmc_$2_fetch.log" 2>&1
  bash tools/gpu_pmc2.sh "$1" $TAG/pmc_$2 $2 1 "WRITE_SIZE" > "$OUT/pmc_$2_write.log" 2>&1
done
bash tools/gpu_pmc2.sh "mask_hmac" $TAG/pmc_mask csv 1 "SQ_INSTS_VALU SQ_WAVES" > "$OUT/pmc_mask.log" 2>&1
python - "$OUT" <<'PY'
import hashlib, json, re, sys
out = sys.argv[1]
res = {}
try:  # the mask kernel's measured VALU instructions per masked value (one lane = one value): the int roofline's own count
    mv = {m.group(1): float(m.group(2)) for m in (re.search(r"(SQ_INSTS_VALU|SQ_WAVES)\s+per_dispatch=([0-9.e+]+)", l) for l in open(out + "/pmc_mask.log") if "mask_hmac" in l) if m}
    if "SQ_INSTS_VALU" in mv and mv.get("SQ_WAVES"):
        res["mask_hmac_sha256"] = {"workload": "csv", "rows_per_launch": 1 << 20, "valu_wave_instructions_per_launch": mv["SQ_INSTS_VALU"], "waves": mv["SQ_WAVES"],
                                   "valu_instructions_per_value": round(mv["SQ_INSTS_VALU"] / mv["SQ_WAVES"], 1), "source_file": "tf_transform.hip",
                                   "source_sha256": hashlib.sha256(open("transferia_amd/csrc/tf_transform.hip", "rb").read()).hexdigest(),
                                   "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES of this build (a wave-instruction is one instruction for each of the wave's 64 values)"}
except OSError:
    pass
for kern, wl, src, rows in (("csv_parse_regular", "csv", "tf_csv.hip", 1 << 20), ("ser_chunk_write", "configs3", "tf_serialize.hip", 1 << 20)):
    vals = {}
    for leg in ("fetch", "write"):
        try:
            for line in open(f"{out}/pmc_{wl}_{leg}.log"):
                m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+per_dispatch=([0-9.e+]+)", line)
                if m and kern[:12] in line:
                    vals[m.group(1)] = float(m.group(2))
        except OSError:
            pass
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        # both counters are in KiB; FETCH_SIZE x2 = the gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, "HBM")
        b = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
        res[kern] = {"rows_per_launch": rows, "workload": wl, "bytes_per_launch": b, "fetch_kib": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"],
                     "source_file": src, "source_sha256": hashlib.sha256(open("transferia_amd/csrc/" + src, "rb").read()).hexdigest(),
                     "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB, separate passes) of this build, tools/gpu_round3.sh; FETCH_SIZE x2 per MI355X_MICROARCH.md"}
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
print("== traffic", {k: v["bytes_per_launch"] for k, v in res.items()})
PY
[ -s "$OUT/pmc_traffic.json" ] && cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json
timeout 600 python bench.py > "$OUT/bench_csv.json" 2> "$OUT/bench_csv.err"; echo "== csv rc=$?"
for w in configs0 configs2 configs3 configs4 configs4d json sr collapse debezium; do
  timeout 400 python bench.py --workload $w > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; echo "== $w rc=$?"
done
timeout 400 python bench.py --workload configs4 --sink debezium > "$OUT/bench_configs4_debezium.json" 2> "$OUT/bench_configs4_debezium.err"; echo "== configs4 --sink debezium rc=$?"
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "no json", e); continue
    r=d.get("roofline") or {}
    print(os.path.basename(f), "value %.4g steps=%d ms/step=%.3f" % (d["value"], d["steps"], d["ms_per_step"]), "roofline", r.get("kernel"), r.get("frac"), "traffic", r.get("traffic"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "parity", d.get("parity_checked_rows"))
PY
export TMPDIR=/tmp
for w in csv configs2 configs3 json sr debezium; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/$OUT/prof_$w" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --workload $w --steps 5 --warmup 2 --cpu-rows 0 --cpu-all-rows 0 --overlap-lanes 0 --pcie-steps 0 > "$GRAFT_REPO_ROOT/$OUT/prof_$w.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_$w.err" )
  find "$OUT/prof_$w" -name '*kernel_trace*' -delete
  f=$(find "$OUT/prof_$w" -name "*kernel_stats.csv" | head -1); echo "== rocprof $w"; [ -n "$f" ] && head -6 "$f" | cut -c1-150
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/$OUT/prof_configs4_debezium" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --workload configs4 --sink debezium --steps 5 --warmup 2 --cpu-rows 0 > "$GRAFT_REPO_ROOT/$OUT/prof_configs4_debezium.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_configs4_debezium.err" )
find "$OUT/prof_configs4_debezium" -name '*kernel_trace*' -delete
f=$(find "$OUT/prof_configs4_debezium" -name "*kernel_stats.csv" | head -1); echo "== rocprof configs4 --sink debezium"; [ -n "$f" ] && head -6 "$f" | cut -c1-150
echo "elapsed $(( $(date +%s) - $(cat $OUT/t0) )) s"
