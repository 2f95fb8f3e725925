#!/bin/bash
# round 3 quick visit for the CSV path: parity tests of the csv files, the bench line's kernels, optional ablation of csv_parse_regular
# usage: gpurun -- 'bash tools/gpu_r3_csv.sh TAG [ablate list] [tile KB list]'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3csv}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_csv.py tests/test_gpu_fullsize.py -m gpu -q -x --tb=short 2>&1 | tail -8 > $OUT/pytest.log; grep -E "passed|failed|error" $OUT/pytest.log
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 120 python bench.py --steps 10 --warmup 2 --cpu-rows 0 --cpu-all-rows 0 --prof-steps 5 --overlap-lanes 0 --pcie-steps 0 2>$OUT/bench_$label.err > $OUT/bench_$label.json
  python - $OUT/bench_$label.json $label <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels"]
    print(sys.argv[2], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), "row_errors", d.get("row_errors"), {n: round(v["avg_ms"],4) for n,v in k.items() if v["avg_ms"] > 0.004})
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
}
run base A=0
for a in ${2:-}; do run ablate$a TFGPU_CSV_ABLATE=$a; done
for t in ${3:-}; do run tile$t TFGPU_CSV_TILE_KB=$t; done
