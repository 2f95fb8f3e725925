#!/bin/bash
# PMC counters for one kernel (regex $1) of `bench.py --workload sr`, one counter set per pass (never combined with traces).
cd "$GRAFT_REPO_ROOT" || exit 1
K=${1:-ser_cell_write}; TAG=${2:-pmc_sr}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-include-regex "$K" -f csv -d "$OUT/p$i" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --workload sr --rows 65536 --steps 1 --warmup 0 --cpu-rows 0 > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:40], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (v, n) in sorted(acc.items()):
    print(f"{kn:40s} {cn:28s} per_dispatch={v/max(1,n):.4g} n={n}")
PY
done
