# measure pq_inflate / read time of the hits parquet object (snappy, lz4_raw) under ring sizes
import io, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pyarrow as pa, pyarrow.parquet as pq
from transferia_amd import abi, lib, workload
lib.init(0)
n = 1 << 20
import bench
dbuf, total, hs = bench.stage_shard(lib, workload, 0, n)
db, _, _ = lib.csv_parse(workload.hits_csv_options(), workload.hits_schema(), dbuf)
h = db.download()
arrays, names = [], []
for c in h.cols:
    valid = c.validity
    vbuf = pa.py_buffer(np.packbits(valid, bitorder="little").tobytes()) if valid is not None else None
    if c.repr in abi.VAR_REPRS:
        off = c.offsets.astype(np.int32)
        arr = pa.Array.from_buffers(pa.string(), h.nrows, [vbuf, pa.py_buffer(off.tobytes()), pa.py_buffer(bytes(c.data[: int(off[-1])]))])
    elif c.repr == abi.R_TIME:
        arr = pa.array(c.values.astype(np.int64), pa.int64(), mask=None if valid is None else ~valid)
    else:
        arr = pa.array(c.values, mask=None if valid is None else ~valid)
    arrays.append(arr); names.append(c.name)
t = pa.table(arrays, names=names)
for codec in ("snappy", "LZ4_RAW"):
    buf = io.BytesIO(); pq.write_table(t, buf, compression=codec, row_group_size=n); data = buf.getvalue()
    pinned = data
    for ring in (64, 32, 16):
        os.environ["TFGPU_PQ_RING_KB"] = str(ring)
        lib.parquet_read(pinned).free(); lib.synchronize()
        lib.prof_reset(); lib.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(3):
            lib.parquet_read(pinned).free()
        lib.synchronize()
        dt = (time.perf_counter() - t0) / 3
        lib.prof_enable(False)
        k = {nm: round(ms / l, 3) for nm, l, ms in lib.prof_get() if l and nm.startswith("pq_inflate")}
        print(codec, "ring", ring, "KB: read %.1f ms" % (dt * 1e3), k, flush=True)
