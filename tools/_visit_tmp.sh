cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
Q="--cpu-rows 0 --overlap-lanes 0 --pcie-steps 0"
TFGPU_CSV_PHASES=1 timeout 300 python bench.py --workload csv --steps 3 --passes 1 --warmup 1 $Q 2>&1 >/dev/null | grep "csv phases" | tail -3
for a in 1 2 3 4 10; do echo "== ablate $a"; TFGPU_CSV_ABLATE=$a timeout 300 python bench.py --workload csv --steps 5 --passes 1 --warmup 2 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['kernels']['csv_parse_regular']['ms_per_step'],4))"; done
