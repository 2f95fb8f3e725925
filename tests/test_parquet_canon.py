"""The Parquet source against the REFERENCE's own reader canon (tests/golden/parquet_reader.json ← tests/canon/s3/parquet/canondata:
30 files of the apache/parquet-testing corpus read by reader_parquet.go / parquet_schema_resolver.go), on inputs re-created from the
canon's values (tests/parquet_canon.py).

CPU (not gpu): the oracle's restatement of the resolver and of parseParquetField / abstract.Restore reproduces the canon — TableSchema
(names, types, OriginalType, keys) and every kept row's Go types and values — for the 20 flat files.
GPU: tfgpu_parquet_resolve_schema + tfgpu_parquet_read_object give the same TableSchema and the same cells (the canon rows AND every
other row, through the oracle); the 10 files with nested columns are refused naming the column."""
import numpy as np
import pytest

import parquet_canon as pc
from transferia_amd import abi

pa = pytest.importorskip("pyarrow")
FILES = pc.golden()
FLAT = sorted(n for n in FILES if n not in pc.NESTED)


def canon_cell(cell):
    """an oracle / device cell as the canon file shows it"""
    t, v = cell
    if t == "string" and isinstance(v, (bytes, bytearray)):
        return [t, bytes(v).decode("utf-8", errors="replace")]
    if t == "float32":  # (the canon prints the shortest float32 text, 1.1: compared as the float32 it names)
        return [t, float(np.float32(v))]
    return [t, v]


def exact_cell(cell):
    """device and oracle cells in one form, bytes kept as bytes (no lossy rendering)"""
    t, v = cell
    if t == "string":
        return [t, v.encode() if isinstance(v, str) else bytes(v)]
    if t == "float32":
        return [t, float(np.float32(v))]
    if t == "float64" and v != v:
        return [t, "NaN"]
    return [t, tuple(v) if isinstance(v, (tuple, list)) else v]


def test_canon_covers_thirty_files():
    assert len(FILES) == 30 and len(FLAT) == 20 and set(pc.NESTED) <= set(FILES)


@pytest.mark.parametrize("name", FLAT)
def test_oracle_reproduces_canon(name):
    from oracle import ora_parquet as op
    g = FILES[name]
    data, fname = pc.build(name)
    sch = op.resolve_schema(data)
    assert [list(c) for c in sch] == g["table_schema"], name
    assert [c[0] for c in sch] == g["names"]
    rows = op.read(data, sch, fname)
    for counter, want in zip(g["counters"], g["rows"]):
        got = [canon_cell(c) for c in rows[counter - 1]]
        assert got == [canon_cell(c) for c in want], (name, counter)


@pytest.mark.parametrize("name", sorted(pc.NESTED))
def test_oracle_names_nested_columns(name):
    from oracle import ora_parquet as op
    data, fname = pc.build_nested(name)
    sch = op.resolve_schema(data)
    anys = [c[0] for c in sch if c[1] == "any"]
    want_any = [c[0] for c in FILES[name]["table_schema"] if c[1] == "any"]
    assert anys and set(anys) <= set(want_any)
    with pytest.raises(op.Nested):
        op.read(data, sch, fname)


# ---- device ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


def device_cells(out):
    """rows of [go type, value] cells of a downloaded batch"""
    cols = []
    for c in out.cols:
        valid = c.validity if c.validity is not None else np.ones(out.nrows, bool)
        cells = []
        for i in range(out.nrows):
            if not valid[i]:
                cells.append(["nil", None])
            elif c.repr in abi.VAR_REPRS:
                cells.append([{abi.R_STRING: "string", abi.R_BYTES: "[]uint8"}[c.repr], c.get_bytes(i)])
            elif c.repr == abi.R_TIME:
                cells.append(["time", (int(c.values[i]), int(c.nanos[i]) if c.nanos is not None else 0)])
            else:
                gt = {abi.R_BOOL: "bool", abi.R_INT32: "int32", abi.R_INT64: "int64", abi.R_UINT64: "uint64", abi.R_FLOAT32: "float32", abi.R_FLOAT64: "float64"}[c.repr]
                v = c.values[i]
                cells.append([gt, bool(v) if gt == "bool" else float(v) if gt.startswith("float") else int(v)])
        cols.append(cells)
    return [[c[r] for c in cols] for r in range(out.nrows)]


@pytest.mark.gpu
@pytest.mark.parametrize("name", FLAT)
def test_device_reproduces_canon(tf, name):
    from oracle import ora_parquet as op
    g = FILES[name]
    data, fname = pc.build(name, total_rows=1500)
    sch = tf.parquet_resolve_schema(data)
    assert [[c.name, c.dtype, c.original_type or "", bool(c.key), bool(c.required)] for c in sch.cols] == g["table_schema"], name
    out = tf.parquet_read(data, sch, "s3_source_parquet", name, file_name=fname).download()
    assert [c.name for c in out.cols] == g["names"]
    got = device_cells(out)
    for counter, want in zip(g["counters"], g["rows"]):            # the reference's canon rows
        assert [canon_cell(c) for c in got[counter - 1]] == [canon_cell(c) for c in want], (name, counter)
    ref = op.read(data, op.resolve_schema(data), fname)               # every other row through the oracle
    assert len(ref) == len(got)
    for r, (a, b) in enumerate(zip(got, ref)):
        assert [exact_cell(c) for c in a] == [exact_cell(c) for c in b], (name, r)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(pc.NESTED))
def test_device_names_nested_columns(tf, name):
    """the resolver types a group `any` with the reference's OriginalType; reading it is refused naming the column"""
    data, fname = pc.build_nested(name)
    sch = tf.parquet_resolve_schema(data)
    want = {c[0]: c for c in FILES[name]["table_schema"]}
    anys = [c.name for c in sch.cols if c.dtype == "any"]
    assert anys and all(want[n][1] == "any" for n in anys if n in want)
    with pytest.raises(tf.TfgpuError) as ei:
        tf.parquet_read(data, sch, "s3_source_parquet", name, file_name=fname)
    assert ei.value.code == tf.ERR_UNSUPPORTED and any(n in str(ei.value) for n in anys)
    # the flat leaves next to it are read when the caller's schema leaves the groups out
    flat = abi.Schema([c for c in sch.cols if c.dtype != "any"])
    out = tf.parquet_read(data, flat, "s3_source_parquet", name, file_name=fname).download()
    assert [c.name for c in out.cols] == [c.name for c in flat.cols]
