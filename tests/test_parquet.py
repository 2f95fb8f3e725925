"""tfgpu_parquet_read (tf_parquet.hip) against an independent reader: files written by pyarrow in the shapes real writers produce —
dictionary pages with RLE / bit-packed indices, PLAIN fall-back pages, optional columns (definition levels), data pages v1 and v2,
several row groups, tiny and large pages — and read back by pyarrow.  PARITY UNPINNED against the reference (no .parquet input in
its tree, parquet-go not vendored): what is checked is that every value and every null equals what pyarrow reads."""
import io
import os
import struct

import numpy as np
import pytest

from transferia_amd import abi

pa = pytest.importorskip("pyarrow")
pq = pytest.importorskip("pyarrow.parquet")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


def table(n, seed, nulls=True):
    rng = np.random.default_rng(seed)

    def opt(vals, every):
        return [None if (nulls and i % every == 0) else v for i, v in enumerate(vals)]
    return pa.table({
        "i32": pa.array(opt([int(x) for x in rng.integers(-1000, 1000, n)], 7), pa.int32()),
        "i64": pa.array(opt([int(x) for x in rng.integers(-(1 << 60), 1 << 60, n)], 11), pa.int64()),
        "few": pa.array([int(x) for x in rng.integers(0, 3, n)], pa.int64()),
        "f32": pa.array(opt([float(np.float32(x)) for x in rng.standard_normal(n)], 13), pa.float32()),
        "f64": pa.array(opt([float(x) for x in rng.standard_normal(n)], 5), pa.float64()),
        "b": pa.array(opt([bool(x) for x in rng.integers(0, 2, n)], 9), pa.bool_()),
        "cat": pa.array(opt(["category-%d" % (i % 37) for i in range(n)], 6), pa.string()),
        "url": pa.array(opt(["http://host/%d/%s" % (i, "x" * int(rng.integers(0, 90))) for i in range(n)], 4), pa.string()),
        "bin": pa.array(opt([bytes(rng.integers(0, 256, int(rng.integers(0, 12))).astype(np.uint8)) for _ in range(n)], 8), pa.binary()),
        "day": pa.array(opt([int(x) for x in rng.integers(-5000, 20000, n)], 10), pa.date32()),
    })


def check(tf, t, compression="NONE", **write_kw):
    buf = io.BytesIO()
    pq.write_table(t, buf, compression=compression, **write_kw)
    data = buf.getvalue()
    ref = pq.read_table(io.BytesIO(data))
    out = tf.parquet_read(data, None, "ns", "tbl").download()
    assert out.nrows == ref.num_rows and (out.table_ns, out.table_name) == ("ns", "tbl")
    assert [c.name for c in out.cols] == ref.column_names
    for c in out.cols:
        want = ref.column(c.name).to_pylist()
        valid = c.validity if c.validity is not None else np.ones(out.nrows, bool)
        assert [bool(v) for v in valid] == [w is not None for w in want], c.name
        for i, w in enumerate(want):
            if w is None:
                continue
            if c.repr in abi.VAR_REPRS:
                got = c.get_bytes(i)
                assert got == (w.encode() if isinstance(w, str) else w), (c.name, i)
            elif c.repr == abi.R_TIME:
                assert int(c.values[i]) == (w - __import__("datetime").date(1970, 1, 1)).days * 86400 and (c.nanos is None or int(c.nanos[i]) == 0), (c.name, i)
            elif c.repr in (abi.R_FLOAT32, abi.R_FLOAT64):
                assert float(c.values[i]) == w or (w != w and c.values[i] != c.values[i]), (c.name, i)
            else:
                assert c.values[i] == w, (c.name, i, c.values[i], w)
    reprs = {c.name: c.repr for c in out.cols}
    assert reprs == {"i32": abi.R_INT32, "i64": abi.R_INT64, "few": abi.R_INT64, "f32": abi.R_FLOAT32, "f64": abi.R_FLOAT64, "b": abi.R_BOOL, "cat": abi.R_STRING,
                     "url": abi.R_STRING, "bin": abi.R_STRING, "day": abi.R_TIME}  # ([]byte under a `string` column is string(v): restore.go:222-229)
    return out


@pytest.mark.parametrize("n", [0, 1, 7, 64, 1000, 20011])
def test_dictionary_pages_as_pyarrow_writes_them(tf, n):
    check(tf, table(n, 100 + n))


def test_plain_pages_small_pages_v2_pages_and_row_groups(tf):
    t = table(6000, 5)
    check(tf, t, use_dictionary=False)                                       # PLAIN values, length-prefixed byte arrays
    check(tf, t, use_dictionary=False, data_page_size=512)                    # hundreds of pages per chunk
    check(tf, t, data_page_size=256, dictionary_pagesize_limit=2048)          # dictionary first, PLAIN after the dictionary outgrows its page
    check(tf, t, data_page_version="2.0")                                     # v2 headers: level lengths in the header, RLE booleans
    check(tf, t, data_page_version="2.0", use_dictionary=False, data_page_size=1024)
    check(tf, t, row_group_size=1000)                                         # six row groups: a dictionary per chunk
    check(tf, table(3000, 6, nulls=False))                                    # required-like data (still optional in the schema)
    req = pa.schema([pa.field("k", pa.int64(), nullable=False), pa.field("s", pa.string(), nullable=False)])
    tt = pa.table({"k": pa.array(range(500), pa.int64()), "s": pa.array(["s%d" % i for i in range(500)])}, schema=req)
    buf = io.BytesIO()
    pq.write_table(tt, buf, compression="NONE")
    out = tf.parquet_read(buf.getvalue()).download()                          # REQUIRED leaves: no definition levels at all
    assert [int(x) for x in out.col("k").values] == list(range(500)) and out.col("k").validity is None
    assert [out.col("s").get_bytes(i) for i in (0, 499)] == [b"s0", b"s499"]


@pytest.mark.parametrize("codec", ["SNAPPY", "GZIP", "ZSTD"])
def test_compressed_pages_as_real_writers_produce_them(tf, codec):
    """pyarrow's default is SNAPPY; the reference's own writer offers SNAPPY / GZIP / ZSTD (serializer/parquet.go CodecFromString)."""
    t = table(5000, 31)
    check(tf, t, compression=codec)                                            # dictionary pages + indices, whole pages compressed (v1)
    check(tf, t, compression=codec, use_dictionary=False, data_page_size=2048)  # PLAIN pages, many of them
    check(tf, t, compression=codec, data_page_version="2.0", row_group_size=1500)  # v2: levels stay uncompressed in front of the values
    check(tf, table(0, 1), compression=codec)
    hits = pa.table({"s": pa.array(["x" * 300 + str(i % 7) for i in range(4000)]), "z": pa.array([0] * 4000, pa.int32())})  # long matches, overlapping copies
    buf = io.BytesIO()
    pq.write_table(hits, buf, compression=codec)
    out = tf.parquet_read(buf.getvalue()).download()
    assert [out.col("s").get_bytes(i) for i in (0, 3999)] == [b"x" * 300 + b"0", b"x" * 300 + str(3999 % 7).encode()] and not np.asarray(out.col("z").values).any()


def _corrupt(data, find, repl, which=0):
    """the object with the `which`-th occurrence of `find` replaced by `repl` (same length)"""
    assert len(find) == len(repl)
    at = -1
    for _ in range(which + 1):
        at = data.index(find, at + 1)
    return data[:at] + repl + data[at + len(find):]


def test_corrupt_objects_are_errors_not_out_of_bounds_reads(tf):
    """Every length, offset and index a file states is untrusted (the reference's parquet-go returns an error for such files)."""
    import struct
    n = 400
    # (a) a PLAIN byte-array length prefix that points far past its page
    t = pa.table({"s": pa.array(["value-%04d" % i for i in range(n)])})
    buf = io.BytesIO()
    pq.write_table(t, buf, compression="NONE", use_dictionary=False)
    good = buf.getvalue()
    assert tf.parquet_read(good).download().col("s").get_bytes(7) == b"value-0007"
    bad = _corrupt(good, struct.pack("<I", 10) + b"value-0100", struct.pack("<I", 0x7FFFFFF0) + b"value-0100")
    with pytest.raises(tf.TfgpuError) as ei:
        tf.parquet_read(bad)
    assert ei.value.code == tf.ERR_INVALID, str(ei.value)
    # (b) dictionary indices past their dictionary: the stated bit width of a two-entry dictionary's indices widened in place
    t = pa.table({"d": pa.array(["aa", "bb"] * (n // 2))})
    buf = io.BytesIO()
    pq.write_table(t, buf, compression="NONE")
    good = buf.getvalue()
    ref = tf.parquet_read(good).download()
    assert ref.col("d").get_bytes(1) == b"bb"
    e = struct.pack("<I", 2) + b"aa" + struct.pack("<I", 2) + b"bb"
    at = good.index(e) + len(e)
    # behind the dictionary page comes the data page; its indices are bit-packed at width 1 (0x55 / 0xAA bytes): widen the stated width
    page = good[at:]
    # try every plausible width byte position: the one that makes the reader fail with ERR_INVALID (and never crash) is the proof
    failures = 0
    for off in range(0, min(len(page) - 1, 80)):
        if page[off] != 1:
            continue
        bad = good[:at + off] + bytes([8]) + good[at + off + 1:]
        try:
            out = tf.parquet_read(bad)
            out.free()
        except tf.TfgpuError as ex:
            assert ex.code in (tf.ERR_INVALID, tf.ERR_UNSUPPORTED), str(ex)
            failures += 1
    assert failures >= 1
    # (e) a footer of structs nested a hundred thousand deep (every byte a field header "struct, id + 1"): refused, not followed down the stack
    deep = b"PAR1" + b"\x1c" * 100000
    deep = deep + struct.pack("<I", 100000) + b"PAR1"
    with pytest.raises(tf.TfgpuError) as ei:
        tf.parquet_read(deep)
    assert ei.value.code in (tf.ERR_INVALID, tf.ERR_UNSUPPORTED), str(ei.value)
    # (c) a dictionary whose last entry runs past its page; (d) truncated objects
    bad = _corrupt(good, struct.pack("<I", 2) + b"bb", struct.pack("<I", 0x00FFFFFF) + b"bb")
    with pytest.raises(tf.TfgpuError) as ei:
        tf.parquet_read(bad)
    assert ei.value.code == tf.ERR_INVALID, str(ei.value)
    for cut in (len(good) // 2, len(good) - 9):
        with pytest.raises(tf.TfgpuError):
            tf.parquet_read(good[:cut] + good[-8:])
    # (e) a PLAIN fixed-width page that states more values than it holds
    t = pa.table({"k": pa.array(range(n), pa.int64())})
    buf = io.BytesIO()
    pq.write_table(t, buf, compression="NONE", use_dictionary=False)
    good = buf.getvalue()
    # corrupt compressed data: a snappy page with a copy that reaches in front of the output
    buf = io.BytesIO()
    pq.write_table(pa.table({"s": pa.array(["abcabcabcabcabcabc%d" % (i % 3) for i in range(n)])}), buf, compression="SNAPPY", use_dictionary=False)
    sn = bytearray(buf.getvalue())
    hit = 0
    for i in range(40, min(len(sn) - 40, 4000)):  # flip bytes across the first page: every outcome must be a clean error or a clean read
        b2 = bytes(sn[:i]) + bytes([sn[i] ^ 0xFF]) + bytes(sn[i + 1:])
        try:
            tf.parquet_read(b2).free()
        except tf.TfgpuError as ex:
            assert ex.code in (tf.ERR_INVALID, tf.ERR_UNSUPPORTED), str(ex)
            hit += 1
    assert hit > 0


def test_schema_selects_columns_and_missing_ones_are_nil(tf):
    t = table(300, 8)
    buf = io.BytesIO()
    pq.write_table(t, buf, compression="NONE")
    schema = abi.Schema.of([["url", "utf8"], ["nope", "int64"], ["i32", "int32", True]])
    out = tf.parquet_read(buf.getvalue(), schema).download()
    assert [c.name for c in out.cols] == ["url", "nope", "i32"]
    assert not out.col("nope").validity.any()
    assert out.col("i32").dtype == "int32" and out.col("url").dtype == "utf8"


def test_what_stays_with_the_stock_reader_is_refused_by_name(tf):
    t = table(100, 9)
    for kw, word in [(dict(compression="BROTLI"), "codec")]:   # (BYTE_STREAM_SPLIT is read since round 6: test_byte_stream_split)
        buf = io.BytesIO()
        pq.write_table(t, buf, **kw)
        with pytest.raises(tf.TfgpuError) as ei:
            tf.parquet_read(buf.getvalue())
        assert ei.value.code == tf.ERR_UNSUPPORTED and word in str(ei.value), str(ei.value)
    nested = pa.table({"l": pa.array([[1, 2], [3]], pa.list_(pa.int32()))})
    buf = io.BytesIO()
    pq.write_table(nested, buf, compression="NONE")
    with pytest.raises(tf.TfgpuError) as ei:
        tf.parquet_read(buf.getvalue())
    assert ei.value.code == tf.ERR_UNSUPPORTED
    with pytest.raises(tf.TfgpuError) as ei:
        tf.parquet_read(b"PAR1 this is not a parquet file PAR1")
    assert ei.value.code == tf.ERR_INVALID


@pytest.mark.parametrize("codec", ["NONE", "SNAPPY", "ZSTD"])
def test_byte_stream_split(tf, codec):
    """BYTE_STREAM_SPLIT (Encodings.md: byte j of every value in stream j, the streams back to back) over FLOAT / DOUBLE / INT32 / INT64 — required and
    optional columns, several pages and row groups, compressed pages (the device's inflate and the host's) — value for value what pyarrow reads."""
    t = table(5000, 41)
    enc = {"f32": "BYTE_STREAM_SPLIT", "f64": "BYTE_STREAM_SPLIT", "i32": "BYTE_STREAM_SPLIT", "i64": "BYTE_STREAM_SPLIT", "few": "BYTE_STREAM_SPLIT"}
    check(tf, t, compression=codec, use_dictionary=False, column_encoding=enc)
    check(tf, t, compression=codec, use_dictionary=False, column_encoding=enc, data_page_size=2048, row_group_size=1700)
    check(tf, table(1, 2), compression=codec, use_dictionary=False, column_encoding=enc)
    check(tf, t, compression=codec, use_dictionary=False, column_encoding=enc, data_page_version="2.0")


@pytest.mark.parametrize("n", [1, 2, 33, 129, 5000, 20011])
def test_delta_encodings_and_lz4_raw(tf, n):
    """DELTA_BINARY_PACKED (int32 / int64 / dates, any miniblock width incl. 0 and 64-bit wrap-around deltas), DELTA_LENGTH_BYTE_ARRAY and
    DELTA_BYTE_ARRAY text, optional and required, several pages and row groups, under LZ4_RAW and uncompressed."""
    t = table(n, 300 + n)
    rng = np.random.default_rng(n)
    wild = pa.array([int(x) for x in rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64)], pa.int64())   # deltas that wrap
    t = t.append_column("wild", wild).append_column("const", pa.array([7] * n, pa.int32()))
    enc = {"i32": "DELTA_BINARY_PACKED", "i64": "DELTA_BINARY_PACKED", "few": "DELTA_BINARY_PACKED", "day": "DELTA_BINARY_PACKED", "wild": "DELTA_BINARY_PACKED", "const": "DELTA_BINARY_PACKED",
           "cat": "DELTA_BYTE_ARRAY", "url": "DELTA_LENGTH_BYTE_ARRAY", "bin": "DELTA_LENGTH_BYTE_ARRAY"}
    for comp in ("NONE", "LZ4"):
        for kw in (dict(), dict(data_page_size=300, row_group_size=max(n // 3, 1)), dict(data_page_version="2.0")):
            buf = io.BytesIO()
            pq.write_table(t, buf, compression=comp, use_dictionary=False, column_encoding=enc, **kw)
            data = buf.getvalue()
            ref = pq.read_table(io.BytesIO(data))
            out = tf.parquet_read(data).download()
            assert out.nrows == n
            for c in out.cols:
                want = ref.column(c.name).to_pylist()
                valid = c.validity if c.validity is not None else np.ones(n, bool)
                assert [bool(v) for v in valid] == [w is not None for w in want], (c.name, comp, kw)
                for i, w in enumerate(want):
                    if w is None:
                        continue
                    if c.repr in abi.VAR_REPRS:
                        assert c.get_bytes(i) == (w.encode() if isinstance(w, str) else w), (c.name, i, comp, kw)
                    elif c.repr == abi.R_TIME:
                        assert int(c.values[i]) == (w - __import__("datetime").date(1970, 1, 1)).days * 86400, (c.name, i)
                    elif c.repr in (abi.R_FLOAT32, abi.R_FLOAT64):
                        assert float(c.values[i]) == w, (c.name, i)
                    else:
                        assert c.values[i] == w, (c.name, i, comp, kw)


def test_staged_object_reads_like_an_uploaded_one(tf):
    """tfgpu_parquet_read_staged: the object already in HBM (a pull / decode pipeline's puller brought it), room for the decoded tail behind it —
    the same batch as tfgpu_parquet_read_object uploads and decodes, for plain, dictionary, DELTA and INT96 columns; a buffer that is too
    short is refused; a compressed object ignores the staging buffer."""
    import datetime
    n = 3000
    t = table(n, 41)
    t = t.append_column("ts96", pa.array([None if i % 11 == 0 else datetime.datetime(2001 + i % 20, 1 + i % 12, 1 + i % 28, i % 24) for i in range(n)], pa.timestamp("ns")))
    for comp, kw in (("NONE", dict(use_dictionary=True, use_deprecated_int96_timestamps=True)),
                     ("NONE", dict(use_dictionary=False, column_encoding={"i32": "DELTA_BINARY_PACKED", "i64": "DELTA_BINARY_PACKED", "url": "DELTA_LENGTH_BYTE_ARRAY"}, data_page_size=2000, row_group_size=1100, use_deprecated_int96_timestamps=True)),
                     ("SNAPPY", dict(use_deprecated_int96_timestamps=True))):
        buf = io.BytesIO()
        pq.write_table(t, buf, compression=comp, **kw)
        data = buf.getvalue()
        want = tf.parquet_read(data).download()
        need = tf.parquet_staging_size(data)
        assert need >= len(data) + 64
        staged = tf.DeviceBuffer.alloc(need)
        staged.write(0, np.frombuffer(data, np.uint8), len(data))
        got = tf.parquet_read_staged(data, staged).download()
        staged.free()
        assert got.nrows == want.nrows == n and [c.name for c in got.cols] == [c.name for c in want.cols]
        for a, b in zip(got.cols, want.cols):
            assert a.repr == b.repr and a.dtype == b.dtype, a.name
            va = a.validity if a.validity is not None else np.ones(n, bool)
            vb = b.validity if b.validity is not None else np.ones(n, bool)
            assert np.array_equal(va, vb), a.name
            if a.repr in abi.VAR_REPRS:
                assert np.array_equal(a.offsets, b.offsets) and bytes(a.data[: int(a.offsets[-1])]) == bytes(b.data[: int(b.offsets[-1])]), (a.name, comp)
            else:
                assert np.array_equal(a.values[va], b.values[vb]), (a.name, comp)
        if comp == "NONE":
            short = tf.DeviceBuffer.alloc(len(data) + 16)
            short.write(0, np.frombuffer(data, np.uint8), len(data))
            with pytest.raises(tf.TfgpuError) as ei:
                tf.parquet_read_staged(data, short)
            assert ei.value.code == tf.ERR_INVALID
            short.free()


def test_wave_walk_equals_one_lane_walk():
    """TFGPU_PQ_WALK_SPEC=1 (read once per process): pq_walk_text's windows by all 64 lanes of a wave — guessed slice entries settled against
    the neighbours' exits; built, measured no faster than the one-lane walk on the MI355X and off by default — must read the same values."""
    import os
    import subprocess
    import sys
    if os.environ.get("TFGPU_PQ_WALK_SPEC"):
        pytest.skip("already inside the wave-walk run")
    env = dict(os.environ, TFGPU_PQ_WALK_SPEC="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                        "plain_text_walk or plain_pages_small or delta_encodings or damaged_objects or corrupt_objects"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]


def test_upload_in_pieces_equals_one_copy():
    """An object of 32 MiB and more goes up in pieces on the lane's copy stream, the page kernels of a piece queued behind its event
    (UploadPieces in tf_parquet.hip).  TFGPU_PQ_PIECES=3 (read once per process) cuts the small objects of these tests the same way."""
    import os
    import subprocess
    import sys
    if os.environ.get("TFGPU_PQ_PIECES"):
        pytest.skip("already inside the pieces run")
    env = dict(os.environ, TFGPU_PQ_PIECES="3")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                        "delta_encodings or staged_object or int96_fixed or plain_text_walk or dictionary_pages or plain_pages_small or damaged_objects"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]


def test_int96_fixed_len_and_restore_conversions(tf):
    """INT96 → the decimal text of its 96 bits (also before 1970 and dictionary-coded); FIXED_LEN_BYTE_ARRAY → string; what Restore makes of a
    value under the resolver's DataType: INT(8/16/32, signed) → int64, unsigned → uint64, float under `double`, TIMESTAMP → microseconds."""
    import datetime
    from oracle import ora_parquet as op
    n = 700
    rng = np.random.default_rng(5)
    ts = [None if i % 9 == 0 else datetime.datetime(1960 + i % 90, 1 + i % 12, 1 + i % 28, i % 24, i % 60, i % 60, i * 1000 % 1000000) for i in range(n)]
    t = pa.table({"ts96": pa.array(ts, pa.timestamp("ns")),
                  "flba": pa.array([None if i % 7 == 0 else bytes(rng.integers(0, 256, 5).astype(np.uint8)) for i in range(n)], pa.binary(5)),
                  "i8": pa.array([int(x) for x in rng.integers(-128, 127, n)], pa.int8()), "u16": pa.array([int(x) for x in rng.integers(0, 65535, n)], pa.uint16()),
                  "u64": pa.array([int(x) for x in rng.integers(0, 1 << 62, n)], pa.uint64()),
                  "tsus": pa.array([None if i % 5 == 0 else int(x) for i, x in enumerate(rng.integers(-(1 << 50), 1 << 50, n))], pa.timestamp("us")),
                  "tsms": pa.array([int(x) for x in rng.integers(0, 1 << 40, n)], pa.timestamp("ms"))})
    for kw in (dict(use_dictionary=False), dict(use_dictionary=True), dict(use_dictionary=False, compression="ZSTD", data_page_size=256)):
        # (pyarrow's INT96 switch is per file: the INT96 column is written in a file of its own shape, the rest beside it)
        buf = io.BytesIO()
        pq.write_table(t.select(["ts96", "flba"]), buf, use_deprecated_int96_timestamps=True, store_schema=False, **kw)
        buf2 = io.BytesIO()
        pq.write_table(t.drop(["ts96"]), buf2, store_schema=False, **kw)
        for data in (buf.getvalue(), buf2.getvalue()):
            sch = tf.parquet_resolve_schema(data)
            osch = op.resolve_schema(data)
            assert [[c.name, c.dtype, c.original_type or "", bool(c.key), bool(c.required)] for c in sch.cols] == [list(c) for c in osch]
            out = tf.parquet_read(data, sch, "ns", "t", file_name="f.parquet").download()
            ref = op.read(data, osch, "f.parquet")
            import test_parquet_canon as tc
            got = tc.device_cells(out)
            assert len(got) == len(ref) == n
            for r in range(n):
                assert [tc.exact_cell(c) for c in got[r]] == [tc.exact_cell(c) for c in ref[r]], r


def test_wide_object_whose_segment_tables_outgrow_the_pinned_ring(tf):
    """Hundreds of small pages per chunk and dozens of columns: the host's segment tables (megabytes in all) travel through the lane's 8 MB
    pinned ring more than once around — the text columns' sizes must not be read back through it (a bug the bench shape found)."""
    rng = np.random.default_rng(77)
    n, ncols = 60000, 48
    cols = {}
    for j in range(ncols):
        if j % 4 == 3:
            cols["s%d" % j] = pa.array(["v%d-%d" % (j, int(x)) for x in rng.integers(0, 1000, n)])
        else:
            vals = rng.integers(0, 1 << 40, n)
            cols["c%d" % j] = pa.array(vals, pa.int64(), mask=(rng.integers(0, 9, n) == 0))
    t = pa.table(cols)
    buf = io.BytesIO()
    pq.write_table(t, buf, compression="NONE", data_page_size=512, use_dictionary=[k for k in cols if k.startswith("s")])
    data = buf.getvalue()
    out = tf.parquet_read(data).download()
    ref = pq.read_table(io.BytesIO(data))
    for c in out.cols:
        col = ref.column(c.name).combine_chunks()
        if c.repr == abi.R_INT64:
            want = col.to_numpy(zero_copy_only=False)
            valid = ~np.isnan(want) if want.dtype.kind == "f" else np.ones(n, bool)
            got_valid = c.validity if c.validity is not None else np.ones(n, bool)
            assert np.array_equal(got_valid, valid), c.name
            assert np.array_equal(c.values[valid], np.asarray(col.drop_null().to_numpy())), c.name
        else:
            want = col.to_pylist()
            assert int(c.offsets[-1]) == sum(len(w) for w in want), c.name
            for i in (0, 1, n // 2, n - 1):
                assert c.get_bytes(i) == want[i].encode(), (c.name, i)


def test_plain_text_walk_windows(tf):
    """PLAIN byte arrays are a chain of length prefixes, walked 16 KiB at a time (pq_walk_text): runs of empty values (4096 links a window),
    values longer than a window (the next window starts behind them), lengths that look like text and text that looks like lengths, pages
    that end inside a window — all against pyarrow's reading."""
    rng = np.random.default_rng(12)
    n = 30000
    def val(i):
        k = i % 11
        if k < 5:
            return b""
        if k == 5:
            return bytes(rng.integers(0, 256, int(rng.integers(1, 9))).astype(np.uint8))            # any bytes, zeros included
        if k == 6:
            return b"\x00\x00\x00\x00" * int(rng.integers(1, 40))                                     # looks like a run of empty values
        if k == 7:
            return (b"\x03\x00\x00\x00abc") * int(rng.integers(1, 30))                                # looks like values
        if k == 8:
            return bytes(rng.integers(32, 127, int(rng.integers(9000, 30000))).astype(np.uint8)) if i % 331 == 8 else b"x" * int(rng.integers(0, 300))
        return ("v%d" % i).encode()
    vals = [None if i % 97 == 0 else val(i) for i in range(n)]
    t = pa.table({"b": pa.array(vals, pa.binary()), "k": pa.array(range(n), pa.int64())})
    for kw in (dict(), dict(data_page_size=3000), dict(data_page_size=100000, data_page_version="2.0"), dict(row_group_size=7001)):
        buf = io.BytesIO()
        pq.write_table(t, buf, compression="NONE", use_dictionary=False, **kw)
        out = tf.parquet_read(buf.getvalue()).download()
        c = out.col("b")
        valid = c.validity if c.validity is not None else np.ones(n, bool)
        assert [bool(v) for v in valid] == [v is not None for v in vals], kw
        lens = np.diff(c.offsets.astype(np.int64))
        assert [int(x) for x in lens] == [len(v) if v is not None else 0 for v in vals], kw
        assert bytes(c.data[: int(c.offsets[-1])]) == b"".join(v for v in vals if v is not None), kw
        assert np.array_equal(out.col("k").values, np.arange(n)), kw


def test_damaged_objects_of_the_round_5_encodings(tf):
    """Bytes flipped across the pages of objects that use what round 5 added — DELTA_BINARY_PACKED / DELTA_LENGTH_BYTE_ARRAY /
    DELTA_BYTE_ARRAY pages, dictionary-index streams expanded on the device, FIXED_LEN_BYTE_ARRAY, INT96, LZ4_RAW: every outcome is a
    clean error or a clean read of the same shape, never a crash (under tools/hipemu/run_asan.sh: never an out-of-bounds access)."""
    import datetime
    n = 600
    rng = np.random.default_rng(77)
    t = pa.table({"d32": pa.array([int(x) for x in rng.integers(-1000, 1000, n)], pa.int32()),
                  "d64": pa.array([None if i % 7 == 0 else int(x) for i, x in enumerate(rng.integers(-(1 << 40), 1 << 40, n))], pa.int64()),
                  "dl": pa.array([None if i % 5 == 0 else "len-%d" % i * (i % 4) for i in range(n)], pa.string()),
                  "db": pa.array(["prefix-shared-%04d" % (i // 3) for i in range(n)], pa.string()),
                  "dict": pa.array(["cat-%d" % (i % 9) for i in range(n)], pa.string()),
                  "flba": pa.array([bytes(rng.integers(0, 256, 6).astype(np.uint8)) for _ in range(n)], pa.binary(6)),
                  "ts": pa.array([datetime.datetime(2001 + i % 20, 1 + i % 12, 1 + i % 28) for i in range(n)], pa.timestamp("ns"))})
    enc = {"d32": "DELTA_BINARY_PACKED", "d64": "DELTA_BINARY_PACKED", "dl": "DELTA_LENGTH_BYTE_ARRAY", "db": "DELTA_BYTE_ARRAY"}
    for comp in ("NONE", "LZ4"):
        buf = io.BytesIO()
        pq.write_table(t, buf, compression=comp, use_dictionary=["dict"], column_encoding=enc, use_deprecated_int96_timestamps=True, data_page_size=700, store_schema=False)
        good = buf.getvalue()
        ref = tf.parquet_read(good)
        assert ref.nrows == n
        ref.free()
        body_end = len(good) - 8 - struct.unpack("<I", good[-8:-4])[0]   # the footer stays intact: the pages are what is damaged
        errors = reads = 0
        emu = bool(os.environ.get("TFGPU_TEST_EMU_LIB"))   # the CPU pre-flight runs every kernel in lockstep on the host: a sparser sweep there
        for i in range(4, body_end, max(1, body_end // (60 if emu else 400))):
            for flip in ((0xFF, 0x01) if emu else (0xFF, 0x01, 0x80)):
                bad = good[:i] + bytes([good[i] ^ flip]) + good[i + 1:]
                try:
                    out = tf.parquet_read(bad)
                    assert out.nrows == n
                    out.free()
                    reads += 1
                except tf.TfgpuError as ex:
                    assert ex.code in (tf.ERR_INVALID, tf.ERR_UNSUPPORTED), str(ex)
                    errors += 1
        assert errors > 0 and reads > 0, (comp, errors, reads)


def _kernels_of(tf, fn):
    tf.prof_reset(); tf.prof_enable(True)
    try:
        out = fn()
    finally:
        names = {n for n, l, ms in tf.prof_get() if l}
        tf.prof_enable(False)
    return out, names


def _inflate_cases(tf, codec, n, check_kws, big_kws, damage_step, forced=True):
    t = table(min(n, 5000) // 2, 33)
    for kw in check_kws:
        _, names = _kernels_of(tf, lambda: check(tf, t, compression=codec, **kw))
        assert not forced or "pq_inflate" in names, (kw, names)
    rng = np.random.default_rng(77)
    words = ["alpha", "beta-gamma", "https://example.org/path/", "delta", "?q=", "epsilon_zeta", "0123456789"]
    big = pa.table({
        "text": pa.array(["".join(words[int(k)] for k in rng.integers(0, len(words), int(rng.integers(1, 9)))) for _ in range(n)]),      # one PLAIN page of n * 30 bytes: matches all over the ring
        "noise": pa.array([bytes(rng.integers(0, 256, 24).astype(np.uint8)) for _ in range(n)], pa.binary()),                            # incompressible: literals of tens of KB
        "zeros": pa.array([0] * n, pa.int64()),                                                                                           # period-1 copies, 64 bytes an element
        "opt": pa.array([None if i % 5 == 0 else int(i) for i in range(n)], pa.int64()),                                                  # levels in front of the values
        "flag": pa.array([bool(i & 4) for i in range(n)], pa.bool_()),
    })
    for kw in big_kws:
        buf = io.BytesIO()
        pq.write_table(big, buf, compression=codec, **kw)
        data = buf.getvalue()
        ref = pq.read_table(io.BytesIO(data))
        dev, names = _kernels_of(tf, lambda: tf.parquet_read(data).download())
        assert "pq_inflate" in names
        for c in dev.cols:
            want = ref.column(c.name).to_pylist()
            valid = c.validity if c.validity is not None else np.ones(n, bool)
            assert [bool(v) for v in valid] == [w is not None for w in want], c.name
            if c.repr in abi.VAR_REPRS:
                off = np.asarray(c.offsets, np.int64)
                assert bytes(c.data[:int(off[-1])]) == b"".join((w.encode() if isinstance(w, str) else w) for w in want if w is not None), c.name
            else:
                got = np.asarray(c.values)
                assert all(got[i] == w for i, w in enumerate(want) if w is not None), c.name
    # a damaged page: a byte of the compressed body flipped — an element's operand (a copy from in front of the page, a literal past its end, a wrong
    # stated length) fails the call; a flip inside literal bytes changes a value, not the structure.  Never an out-of-bounds access.
    buf = io.BytesIO()
    pq.write_table(pa.table({"s": pa.array(["abcdefgh" * 40 + str(i) for i in range(300)])}), buf, compression=codec, use_dictionary=False)
    data = bytearray(buf.getvalue())
    at = pq.ParquetFile(io.BytesIO(bytes(data))).metadata.row_group(0).column(0).data_page_offset
    failed = 0
    for delta in range(40, 400, damage_step):
        bad = bytearray(data)
        bad[at + delta] ^= 0xFF
        try:
            tf.parquet_read(bytes(bad)).download()
        except tf.TfgpuError:
            failed += 1
    assert failed > 0


@pytest.mark.parametrize("mode", ["1", "auto"])
@pytest.mark.parametrize("codec", ["SNAPPY", "LZ4_RAW"])
def test_pages_inflated_on_the_device(tf, codec, mode, monkeypatch):
    """pq_inflate (one wave a page; SNAPPY and LZ4_RAW): the shapes of `check` (dictionary pages + indices, PLAIN pages, v2 pages with their levels outside
    the compressed part, optional columns whose levels the host inflates as a prefix), pages larger than the 64 KiB LDS ring (repetitive text whose
    copies reach back over the ring, incompressible bytes = long literals, period-1 overlapping copies) and damaged pages — equal to pyarrow's reading."""
    # mode "1": every eligible page on the device (the kernel meets text, copies over the whole ring, damaged elements); "auto" (the default): pages that
    # barely compress on the device — the noise column here — and the others on the host's cores, inflated side by side ahead of the walk
    monkeypatch.setenv("TFGPU_PQ_DEVICE_INFLATE", mode)
    _inflate_cases(tf, codec, 2600, ({}, {"use_dictionary": False, "data_page_version": "2.0", "row_group_size": 1500}),
                   ({"use_dictionary": False, "data_page_size": 1 << 24},), 11, forced=mode == "1")


@pytest.mark.parametrize("codec", ["SNAPPY", "LZ4_RAW"])
def test_pages_inflated_on_the_device_fullsize(tf, codec, monkeypatch):
    """the same with pages of 1.8 MB (60 000 rows: the ring wraps 28 times), every page shape, a denser sweep of damaged bytes (MI355X only: minutes on the emulator)"""
    monkeypatch.setenv("TFGPU_PQ_DEVICE_INFLATE", "1")
    _inflate_cases(tf, codec, 60000, ({}, {"use_dictionary": False, "data_page_size": 2048}, {"data_page_version": "2.0", "row_group_size": 1500}, {"use_dictionary": False, "data_page_size": 1 << 22}),
                   ({"use_dictionary": False, "data_page_size": 1 << 24}, {"use_dictionary": False, "data_page_size": 100000, "data_page_version": "2.0"}, {}), 3)
