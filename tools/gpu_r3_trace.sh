#!/bin/bash
# kernel timeline of the csv bench step: where the device idles between kernels (host read-backs, launches)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3trace}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/t -o tr -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --cpu-rows 0 --cpu-all-rows 0 --prof-steps 1 --overlap-lanes 0 --pcie-steps 0 > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:28]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
# steps start at csv_count_newlines
idx = [i for i, r in enumerate(rows) if "count_newlines" in r[2]]
idx = idx[10:40]
spans = []
for a, b in zip(idx, idx[1:]):
    st = rows[a:b]
    total = rows[b][0] - st[0][0]
    busy = sum(e - s for s, e, _ in st)
    gaps = [(st[k + 1][0] - st[k][1], st[k][2], st[k + 1][2]) for k in range(len(st) - 1)] + [(rows[b][0] - st[-1][1], st[-1][2], "next step")]
    spans.append((total, busy, gaps))
n = len(spans)
print("steps", n, "avg step us %.1f busy us %.1f idle us %.1f" % (sum(s[0] for s in spans) / n / 1e3, sum(s[1] for s in spans) / n / 1e3, sum(s[0] - s[1] for s in spans) / n / 1e3))
acc = collections.defaultdict(float)
for s in spans:
    for g, a, b in s[2]:
        acc[(a, b)] += g
for (a, b), g in sorted(acc.items(), key=lambda kv: -kv[1])[:12]:
    print("  gap %-28s -> %-28s %.1f us/step" % (a, b, g / n / 1e3))
PY
rm -rf $OUT/t
