# pq_inflate on single-column objects of one page each: which shape of page is slow?
import io, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pyarrow as pa, pyarrow.parquet as pq
from transferia_amd import abi, lib
lib.init(0)
rng = np.random.default_rng(1)
n = 40000
words = ["alpha", "beta-gamma", "https://example.org/path/", "delta", "?q=", "epsilon_zeta", "0123456789"]
shapes = {
    "noise(binary 24B)": pa.array([bytes(rng.integers(0, 256, 24).astype(np.uint8)) for _ in range(n)], pa.binary()),
    "words(text)": pa.array(["".join(words[int(k)] for k in rng.integers(0, len(words), int(rng.integers(1, 9)))) for _ in range(n)]),
    "urls(text)": pa.array(["http://host%d.example.com/%s?id=%d" % (i % 97, "x" * int(rng.integers(0, 40)), int(rng.integers(0, 10**9))) for i in range(n)]),
    "zeros(int64)": pa.array([0] * (n * 4), pa.int64()),
    "random(int64)": pa.array(rng.integers(0, 1 << 62, n * 4), pa.int64()),
    "smallints(int64)": pa.array(rng.integers(0, 1000, n * 4), pa.int64()),
}
for codec in ("snappy", "LZ4_RAW"):
    for name, arr in shapes.items():
        buf = io.BytesIO(); pq.write_table(pa.table({"c": arr}), buf, compression=codec, use_dictionary=False, data_page_size=1 << 26); data = buf.getvalue()
        md = pq.ParquetFile(io.BytesIO(data)).metadata.row_group(0).column(0)
        lib.parquet_read(data).free(); lib.synchronize()
        lib.prof_reset(); lib.prof_enable(True)
        for _ in range(3):
            lib.parquet_read(data).free()
        lib.synchronize(); lib.prof_enable(False)
        k = [ms / l for nm, l, ms in lib.prof_get() if l and nm.startswith("pq_inflate")]
        ms = k[0] if k else float("nan")
        print("%-8s %-20s comp %8d -> %8d B  pq_inflate %8.3f ms  = %7.1f MB/s out" % (codec, name, md.total_compressed_size, md.total_uncompressed_size, ms, md.total_uncompressed_size / ms / 1e3), flush=True)
