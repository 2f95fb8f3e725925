#!/bin/bash
# PMC counters for one kernel (regex $1) of a bench workload ($3, default csv), each counter set in its own pass (never combined with
# trace domains).  usage: gpurun -- 'bash tools/gpu_pmc2.sh csv_parse_regular tag [workload] [sets] ["COUNTER COUNTER ..."]'
# (a fifth argument replaces the built-in sets by that one set, e.g. "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH")
cd "$GRAFT_REPO_ROOT" || exit 1
K=${1:-csv_parse_regular}; TAG=${2:-pmc}; WL=${3:-csv}; NSETS=${4:-2}; CUSTOM=${5:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE")
[ -n "$CUSTOM" ] && SETS=("$CUSTOM") && NSETS=1
for set in "${SETS[@]}"; do
  i=$((i+1)); [ $i -gt $NSETS ] && break
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "$K" -f csv -d "$OUT/p$i" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --workload $WL --steps 1 --passes 1 --warmup 1 --cpu-rows 0 --prof-steps 1 --overlap-lanes 0 --pcie-steps 0 > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a "$OUT/summary.txt"
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:40], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (v, n) in sorted(acc.items()):
    print(f"{kn:40s} {cn:28s} per_dispatch={v/max(1,n):.5g} n={n}")
PY
  rm -rf "$OUT/p$i"
done
