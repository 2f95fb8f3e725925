"""tfgpu_parquet_write (tf_parquetw.hip) — pkg/serializer/parquet.go + parquet_format.go on the device — read back by an independent
reader.  PINNED to the one Parquet object the reference holds (TestBatchSerializer/parquet:default's canon `result`, a parquet-go file:
test_the_reference_canon_object below — field order, physical and logical types, repetition, row count, every value of the same 118 items);
the BYTE layout (encodings, page and footer bytes) is parquet-go's choice and stays unpinned.  Beyond the canon, what is checked is
what the reference's schema builder and value conversion prescribe — fields in name order, the logical types of primitiveTypesMap,
OPTIONAL unless Required, nil → null, float64 as its decimal text, `any` as JSON text, dates as days, times as nanoseconds — through
pyarrow's reading of the object, for every codec CodecFromString names, and a round trip through tfgpu_parquet_read."""
import datetime
import io

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


SCHEMA = [["id", "int64", True, "", "", True], ["i8", "int8"], ["i16", "int16"], ["i32", "int32"], ["u8", "uint8"], ["u16", "uint16"], ["u32", "uint32"], ["u64", "uint64"],
          ["flag", "boolean"], ["f", "float"], ["price", "double"], ["name", "utf8"], ["blob", "string"], ["doc", "any"], ["day", "date"], ["at", "timestamp"],
          ["seen", "datetime"], ["missing", "utf8"]]


def rows(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        nil = lambda k: (i + k) % 7 == 0  # noqa: E731
        sec = int(rng.integers(-10**9, 2 * 10**9))
        out.append([
            ["int64", int(rng.integers(-(1 << 62), 1 << 62))],
            None if nil(1) else ["int8", int(rng.integers(-128, 128))],
            None if nil(2) else ["int16", int(rng.integers(-32768, 32768))],
            None if nil(3) else ["int32", int(rng.integers(-(1 << 31), 1 << 31))],
            None if nil(4) else ["uint8", int(rng.integers(0, 256))],
            None if nil(5) else ["uint16", int(rng.integers(0, 65536))],
            None if nil(6) else ["uint32", int(rng.integers(0, 1 << 32))],
            None if nil(1) else ["uint64", int(rng.integers(0, 1 << 63)) * 2 + 1],
            None if nil(2) else ["bool", bool(rng.integers(0, 2))],
            None if nil(3) else ["float32", float(np.float32(rng.standard_normal()))],
            None if nil(4) else ["jsonnum", "%d.%02d" % (int(rng.integers(0, 10**6)), int(rng.integers(0, 100)))],
            None if nil(5) else ["string", "имя-%d-%s" % (i, "x" * int(rng.integers(0, 40)))],
            None if nil(6) else ["bytes", bytes(rng.integers(0, 256, int(rng.integers(0, 9))).astype(np.uint8))],
            None if nil(1) else ["json", '{"k":%d,"tags":["a","b"]}' % i],
            None if nil(2) else ["time", (sec - sec % 86400, 0)],
            None if nil(3) else ["time", (sec, int(rng.integers(0, 10**9)))],
            None if nil(4) else ["time", (sec, 0)],
        ])
    return out


def make_batch(tf, n, seed):
    schema = abi.Schema.of(SCHEMA)
    names = [c[0] for c in SCHEMA[:-1]]   # the batch lacks "missing": null in every row
    rs = rows(n, seed)
    cells = [[["nil", None] if c is None else c for c in r] for r in rs]
    return schema, rs, tf.DeviceBatch.upload(abi.batch_from_rows(schema, names, cells, "db", "t"))


def expect(r, k):
    c = r[k]
    if c is None:
        return None
    return c[1]


@pytest.mark.parametrize("codec", ["", "SNAPPY", "GZIP", "ZSTD"])
def test_object_reads_back_value_for_value(tf, codec):
    n = 3000
    schema, rs, db = make_batch(tf, n, 7)
    data = tf.parquet_write(db, schema, codec, row_group_max_rows=1024)
    t = pq.read_table(io.BytesIO(data))
    md = pq.ParquetFile(io.BytesIO(data)).metadata
    assert md.num_rows == n and md.num_row_groups == 3
    assert md.row_group(0).column(0).compression == (codec or "UNCOMPRESSED")
    assert t.column_names == sorted(c[0] for c in SCHEMA)          # parquet.Group is a map: fields by name
    sch = pq.ParquetFile(io.BytesIO(data)).schema
    types = {sch.column(i).name: (sch.column(i).physical_type, str(sch.column(i).logical_type)) for i in range(len(sch))}
    assert types["i8"] == ("INT32", "Int(bitWidth=8, isSigned=true)") and types["u64"] == ("INT64", "Int(bitWidth=64, isSigned=false)")
    assert types["price"] == ("BYTE_ARRAY", "String") and types["doc"] == ("BYTE_ARRAY", "JSON") and types["blob"] == ("BYTE_ARRAY", "None")
    assert types["day"] == ("INT32", "Date") and "Timestamp" in types["at"][1] and "nanoseconds" in types["at"][1] and types["flag"][0] == "BOOLEAN"
    assert [sch.column(i).max_definition_level for i in range(len(sch)) if sch.column(i).name == "id"] == [0]   # Required → REQUIRED
    cols = {name: t.column(name).to_pylist() for name in t.column_names}
    import json
    for i, r in enumerate(rs):
        for k, (name, *_) in enumerate(SCHEMA[:-1]):
            want, got = expect(r, k), cols[name][i]
            if want is None:
                assert got is None, (name, i)
            elif name == "doc":
                assert json.loads(got) == json.loads(want), (name, i)
            elif name == "day":
                assert got == datetime.date(1970, 1, 1) + datetime.timedelta(days=want[0] // 86400), (name, i)
            elif name in ("at", "seen"):
                ns = t.column(name).cast(pa.int64())[i].as_py()
                assert ns == want[0] * 10**9 + want[1], (name, i)
            elif name == "f":
                assert np.float32(got) == np.float32(want), (name, i)
            elif name == "name":
                assert got == want, (name, i)
            else:
                assert got == want, (name, i, got, want)
        assert cols["missing"][i] is None
    db.free()


def test_round_trip_through_the_reader_and_edge_sizes(tf):
    for n in (0, 1, 8, 9, 1000):
        schema, rs, db = make_batch(tf, n, 100 + n)
        data = tf.parquet_write(db, schema, "SNAPPY")
        assert pq.read_table(io.BytesIO(data)).num_rows == n
        back = tf.parquet_read(data).download()
        assert back.nrows == n
        if n:
            ids = [int(x) for x in back.col("id").values]
            assert ids == [r[0][1] for r in rs]
            nm = back.col("name")
            for i, r in enumerate(rs):
                if r[11] is None:
                    assert not nm.validity[i]
                else:
                    assert nm.get_bytes(i) == r[11][1].encode()
        db.free()


def test_what_the_device_does_not_write_is_refused_by_name(tf):
    schema = abi.Schema.of([["x", "double"]])
    db = tf.DeviceBatch.upload(abi.batch_from_rows(schema, ["x"], [[["float64", 1.5]]], "db", "t"))
    with pytest.raises(tf.TfgpuError) as ei:
        tf.parquet_write(db, schema)
    assert ei.value.code == tf.ERR_UNSUPPORTED and "Strictify" in str(ei.value)
    db.free()
    # a Required column with a nil: parquet-go writes the type's zero value (pinned by the reference's canon object, below) — every physical type
    schema = abi.Schema.of([["k", "int64", True, "", "", True], ["s", "utf8", False, "", "", True], ["b", "boolean", False, "", "", True], ["f", "float", False, "", "", True]])
    db = tf.DeviceBatch.upload(abi.batch_from_rows(schema, ["k", "s", "b", "f"], [[["int64", 1], ["string", "x"], ["bool", True], ["float32", 1.5]], [["nil", None]] * 4,
                                                                                [["int64", 3], ["nil", None], ["bool", True], ["nil", None]]], "db", "t"))
    t = pq.read_table(io.BytesIO(tf.parquet_write(db, schema)))
    assert t.to_pylist() == [{"b": True, "f": 1.5, "k": 1, "s": "x"}, {"b": False, "f": 0.0, "k": 0, "s": ""}, {"b": True, "f": 0.0, "k": 3, "s": ""}]
    db.free()



def test_the_reference_canon_object(tf):
    """pkg/serializer/reference/canondata/reference.reference.TestBatchSerializer_parquet_default/result (tests/golden/parquet_writer_canon.json is
    pyarrow's reading of it, tools/extract_golden.py): the 118 items of ReadChangeItems(10) — 33 tables — written with the FIRST item's TableSchema,
    every value looked up by field name in item.AsMap() (parquet_format.go:82-100).  The same rows through tfgpu_parquet_write must say the same:
    fields in the same order with the same physical / logical types and repetition, 118 rows, every value — including the 0 a Required field
    gets for a nil (the items of tables without `__primary_key`)."""
    from util import golden
    g = golden("parquet_writer_canon.json")
    schema = abi.Schema.of([[n, t, k, "", "", req] for n, t, k, req in g["table_schema"]])
    names = [c[0] for c in g["table_schema"]]
    rows_in = []
    for tb in golden("serializers_canon.json")["tables"]:
        have = {n: i for i, n in enumerate(tb["common"]["names"])}
        types = {c[0]: c[1] for c in tb["common"]["schema"]}
        for r in tb["rows"]:
            row = []
            for n, t, _, _ in g["table_schema"]:
                cell = r[have[n]] if n in have else ["nil", None]
                assert cell[0] == "nil" or types[n] == t, (tb["name"], n)     # a same-named column of another DataType would be parquet-go's conversion: the corpus has none
                row.append(cell)
            rows_in.append(row)
    assert len(rows_in) == g["num_rows"] == 118
    db = tf.DeviceBatch.upload(abi.batch_from_rows(schema, names, rows_in, "public", "wtf_types"))
    data = tf.parquet_write(db, schema, "")
    f = pq.ParquetFile(io.BytesIO(data))
    sch, md = f.schema, f.metadata
    assert md.num_rows == 118 and md.num_row_groups == g["num_row_groups"] == 1
    got_fields = [{"name": sch.column(i).name, "physical": sch.column(i).physical_type, "logical": str(sch.column(i).logical_type), "converted": str(sch.column(i).converted_type),
                   "required": sch.column(i).max_definition_level == 0, "max_repetition_level": sch.column(i).max_repetition_level} for i in range(len(sch.names))]
    assert got_fields == g["fields"]
    rg = md.row_group(0)
    assert [(rg.column(i).path_in_schema, rg.column(i).compression, rg.column(i).num_values) for i in range(rg.num_columns)] == [(c["name"], c["compression"], c["num_values"]) for c in g["chunks"]]
    t = f.read()
    got_rows = [[r[n] for n in sch.names] for r in t.to_pylist()]
    # BYTE_ARRAY (JSON) values come back as bytes or str depending on the annotation pyarrow sees: compare as text
    norm = lambda v: v.decode("utf-8") if isinstance(v, (bytes, bytearray)) else v  # noqa: E731
    for i, (a, b) in enumerate(zip(got_rows, g["rows"])):
        assert [norm(x) for x in a] == b, (i, a, b)
    assert sum(1 for r in got_rows if r[0] == 0) == sum(1 for r in g["rows"] if r[0] == 0) > 0   # the Required nils
