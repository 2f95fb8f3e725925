#!/bin/bash
# the -m gpu suite + smoke + the headline bench line + the two JSON component lines (a last look at a committed state)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-final}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -12 > $OUT/pytest_gpu.log; grep -E "passed|failed" $OUT/pytest_gpu.log
timeout 200 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 400 python bench.py > $OUT/bench_csv.json 2> $OUT/bench_csv.err; echo "csv rc=$?"
for w in json sr configs3; do timeout 300 python bench.py --workload $w --cpu-rows 0 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
python - $OUT <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print(os.path.basename(f), "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), r.get("kernel"), r.get("frac"))
PY
