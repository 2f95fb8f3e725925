cd "$GRAFT_REPO_ROOT"; T=r07t; O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
Q="--cpu-rows 0 --overlap-lanes 0 --pcie-steps 0"
sum() { python - "$1" <<'PY'
import json,sys
try: d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e: print("no json", e); sys.exit(0)
print("  ms/step %.3f" % d["ms_per_step"], {k: round(v["ms_per_step"],4) for k,v in d["kernels"].items() if k=="csv_parse_regular"})
PY
}
for v in "0 16" "1 16" "2 16" "2 8" "2 32" "2 4"; do set -- $v; echo "== csv col_lanes=$1 piece=$2"; TFGPU_CSV_COL_LANES=$1 TFGPU_CSV_COL_PIECE=$2 timeout 300 python bench.py --workload csv --steps 10 --passes 1 --warmup 3 $Q > $O/ab_$1_$2.json 2>$O/err.log; sum $O/ab_$1_$2.json; done
for a in 10 11 12 13; do echo "== ablate $a mode 2"; TFGPU_CSV_ABLATE=$a TFGPU_CSV_COL_LANES=2 timeout 300 python bench.py --workload csv --steps 5 --passes 1 --warmup 2 $Q > $O/abl_$a.json 2>$O/err.log; sum $O/abl_$a.json; TFGPU_CSV_ABLATE=$a TFGPU_CSV_COL_LANES=0 timeout 300 python bench.py --workload csv --steps 5 --passes 1 --warmup 2 $Q > $O/abl0_$a.json 2>$O/err.log; sum $O/abl0_$a.json; done
