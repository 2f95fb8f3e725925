#!/usr/bin/env python3
"""Phase breakdown of csv_parse_tiles: run tfgpu_csv_parse alone on the bench's 2^20-row hits CSV under
TFGPU_CSV_ABLATE=n (the kernel returns after phase n; results are NOT valid) and print the kernel's HIP-event time.
    for n in 0 1 2 3 6 5; do TFGPU_CSV_ABLATE=$n python tools/ablate_csv.py; done
1 = stage tile in LDS, 2 = + classify, 3 = + scans and field index, 6 = + per-cell addressing (no value work),
5 = everything but the integer stores, 0 = the full kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from transferia_amd import lib, workload  # noqa: E402

# the profiling branches are compiled out of the product: this script runs the ablate build of tf_csv.hip
#   tools/build_variant.sh ablate tf_csv.hip -DTF_CSV_ABLATE_BUILD=1
lib._LIBPATH = os.path.join(os.path.dirname(lib._LIBPATH), "variants", "libtfgpu_ablate.so")

rows = int(os.environ.get("ROWS", 1 << 20))
lib.init(0)
dbuf, total, hs = bench.stage_shard(lib, workload, 0, rows)
schema, opts = workload.hits_schema(), workload.hits_csv_options()
for _ in range(2):
    db, _, _ = lib.csv_parse(opts, schema, dbuf)
    db.free()
lib.prof_reset(); lib.prof_enable(True)
for _ in range(5):
    db, _, _ = lib.csv_parse(opts, schema, dbuf)
    db.free()
lib.prof_enable(False)
k = {n: round(ms / l, 4) for n, l, ms in lib.prof_get() if l}
print(json.dumps({"ablate": int(os.environ.get("TFGPU_CSV_ABLATE", "0")), "csv_parse_tiles_ms": k.get("csv_parse_tiles"), "all": k}))
