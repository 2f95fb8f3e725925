cd "$GRAFT_REPO_ROOT"; T=r07a; O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
Q="--cpu-rows 0 --overlap-lanes 0 --pcie-steps 0"
sum() { python - "$1" <<'PY'
import json,sys
try: d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e: print("no json", e); sys.exit(0)
print("  value %.4g ms/step %.3f passes %s" % (d["value"], d["ms_per_step"], d.get("passes_per_step")), {k: round(v["ms_per_step"],4) for k,v in d["kernels"].items()})
PY
}
for o in 1 3; do for cb in 112 96; do echo "== configs3 order=$o chunk=$cb"; TFGPU_SER_ORDER=$o TFGPU_SER_CHUNK_BYTES=$cb timeout 300 python bench.py --workload configs3 --steps 5 --passes 1 --warmup 2 $Q > $O/c3_o${o}_c$cb.json 2>$O/err.log; sum $O/c3_o${o}_c$cb.json; done; done
for o in 1 3; do echo "== json order=$o"; TFGPU_SER_ORDER=$o timeout 300 python bench.py --workload json --steps 5 --passes 1 --warmup 2 $Q > $O/json_o$o.json 2>$O/err.log; sum $O/json_o$o.json; done
for r in 0 1; do echo "== csv reorder=$r"; TFGPU_CHAIN_REORDER=$r timeout 300 python bench.py --workload csv --steps 10 --passes 1 --warmup 3 $Q > $O/csv_reorder$r.json 2>$O/err.log; sum $O/csv_reorder$r.json; done
for o in 1 3; do TFGPU_SER_ORDER=$o bash tools/gpu_pmc2.sh ser_chunk_write $T/pmc_o$o configs3 1 "WRITE_SIZE" > $O/pmc_w$o.log 2>&1; TFGPU_SER_ORDER=$o bash tools/gpu_pmc2.sh ser_chunk_write $T/pmc_o$o configs3 1 "FETCH_SIZE" > $O/pmc_f$o.log 2>&1; echo "== pmc order $o"; cat $O/pmc_o$o/summary.txt; done
