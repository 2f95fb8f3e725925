#!/bin/bash
# launch counts per step of a workload (rocprofv3 kernel stats), and its bench line
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3h}; W=${2:-configs2}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python bench.py --workload $W --cpu-rows 0 2>$OUT/bench_$W.err > $OUT/bench_$W.json; python - $OUT/bench_$W.json $W <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[2], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), "kernel sum %.3f" % sum(v["ms_per_step"] for v in k.values()))
PY
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 5 --warmup 2 --cpu-rows 0 --overlap-lanes 0 --pcie-steps 0 > $OUT/prof.json 2> $OUT/prof.err
find $OUT/prof -name '*kernel_trace*' -delete
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r:-int(r['Calls']))[:14]:
    print("  %-64s calls/step %6.1f  us/step %8.1f" % (r['Name'][:64], int(r['Calls'])/11, float(r['TotalDurationNs'])/11/1e3))
PY
