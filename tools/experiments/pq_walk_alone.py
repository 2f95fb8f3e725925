"""The Parquet decode kernels timed with the object already in HBM (tfgpu_parquet_read_staged): no upload in front of them on the stream.
    python tools/experiments/pq_walk_alone.py [rows]"""
import io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pyarrow as pa, pyarrow.parquet as pq
from transferia_amd import lib, abi, workload
lib.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
db, _, _ = lib.csv_parse(workload.hits_csv_options(), workload.hits_schema(), lib.DeviceBuffer.upload(workload.hits_csv(n)))
h = db.download()
arrays, names = [], []
for c in h.cols:
    valid = c.validity
    vbuf = pa.py_buffer(np.packbits(valid, bitorder="little").tobytes()) if valid is not None else None
    if c.repr in abi.VAR_REPRS:
        off = c.offsets.astype(np.int32)
        arr = pa.Array.from_buffers(pa.string() if c.repr == abi.R_STRING else pa.binary(), h.nrows, [vbuf, pa.py_buffer(off.tobytes()), pa.py_buffer(bytes(c.data[: int(off[-1])]))])
    elif c.repr == abi.R_TIME:
        arr = pa.array(c.values.astype(np.int64), pa.int64(), mask=None if valid is None else ~valid)
    else:
        arr = pa.array(c.values, mask=None if valid is None else ~valid)
    arrays.append(arr); names.append(c.name)
buf = io.BytesIO()
pq.write_table(pa.table(arrays, names=names), buf, compression="NONE", row_group_size=h.nrows)
data = buf.getvalue()
need = lib.parquet_staging_size(data)
st = lib.DeviceBuffer.alloc(need)
st.write(0, np.frombuffer(data, np.uint8), len(data))
lib.synchronize()
schema = abi.Schema.of([[c.name, c.dtype] for c in h.cols])
for _ in range(2):
    lib.parquet_read_staged(data, st, schema, "", "hits").free()
lib.synchronize()
lib.prof_reset(); lib.prof_enable(True)
for _ in range(3):
    lib.parquet_read_staged(data, st, schema, "", "hits").free()
lib.prof_enable(False)
print({k: round(ms / max(l, 1), 4) for k, l, ms in lib.prof_get() if k.startswith("pq_")}, "object", len(data))
