"""tfgpu_strictify = strictify.Strictify over a device batch (pkg/abstract/changeitem/strictify/strictify.go:17-157; the first step of
the strictifying serializers, pkg/serializer/strictify.go:24-36) against the oracle's ora_strictify: Go strings and json.Numbers under
every DataType, the integer kinds under integer / bool / float / time / interval DataTypes, the reference's all-or-nothing error
(the first failing value in row-then-column order), and the pairs that stay on the host refused by name."""
import numpy as np
import pytest

from transferia_amd import abi
from test_gpu_transformers import assert_batches_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as ora
    ora.build()
    return ora


SCHEMA = abi.Schema.of([["i8", "int8"], ["i32", "int32"], ["i64", "int64"], ["u16", "uint16"], ["u64", "uint64"], ["b", "boolean"], ["f", "float"],
                        ["d", "double"], ["ts", "timestamp"], ["dt", "date"], ["iv", "interval"], ["s", "utf8"], ["y", "string"], ["k", "int64"]])
NAMES = [c.name for c in SCHEMA.cols]


def text_rows(n, seed, bad_at=None):
    rng = np.random.default_rng(seed)
    rows = []
    for r in range(n):
        neg = -1 if rng.integers(0, 2) else 1
        row = [["string", str(int(neg * rng.integers(0, 128)))], ["string", str(int(neg * rng.integers(0, 1 << 31))) + (".000" if rng.integers(0, 4) == 0 else "")],
               ["string", ("0x%x" % rng.integers(0, 1 << 40)) if rng.integers(0, 5) == 0 else str(int(neg * rng.integers(0, 1 << 62)))],
               ["string", str(int(rng.integers(0, 65536)))], ["string", str(int(rng.integers(0, 1 << 63)) * 2 + 1)],
               ["string", ["true", "F", "1", "0", "TRUE", "false", "t"][int(rng.integers(0, 7))]],
               ["string", "%d.%de%d" % (rng.integers(0, 1000), rng.integers(0, 1000), rng.integers(-20, 20))],
               ["string", "%d.%d" % (neg * rng.integers(0, 10 ** 9), rng.integers(0, 10 ** 6))],
               ["string", "2021-%02d-%02dT%02d:%02d:%02dZ" % (rng.integers(1, 13), rng.integers(1, 29), rng.integers(0, 24), rng.integers(0, 60), rng.integers(0, 60))],
               ["string", "20%02d-%02d-%02d" % (rng.integers(0, 40), rng.integers(1, 13), rng.integers(1, 29))],
               ["string", "%dh%dm%d.%ds" % (rng.integers(0, 100), rng.integers(0, 60), rng.integers(0, 60), rng.integers(0, 1000))],
               ["string", "text %d" % r], ["string", "bytes %d" % r], ["int32", int(r)]]
        if r % 13 == 5:
            row[int(rng.integers(0, 11))] = ["nil", None]
        rows.append(row)
    if bad_at is not None:
        r, c, v = bad_at
        rows[r][c] = ["string", v]
    return rows


@pytest.mark.parametrize("n", [1, 63, 2000])
def test_strings_become_their_columns_types(tf, oracle, n):
    host = abi.batch_from_rows(SCHEMA, NAMES, text_rows(n, 100 + n), "db", "t")
    ref = oracle.strictify(host, SCHEMA)
    assert not ref.errors
    out = tf.strictify(tf.DeviceBatch.upload(host), SCHEMA).download()
    assert_batches_equal(out, ref.batch, "n=%d" % n)
    # the strict Go types, as the serializers then see them
    want = {"i8": abi.R_INT8, "i32": abi.R_INT32, "i64": abi.R_INT64, "u16": abi.R_UINT16, "u64": abi.R_UINT64, "b": abi.R_BOOL, "f": abi.R_FLOAT32,
            "d": abi.R_JSONNUM, "ts": abi.R_TIME, "dt": abi.R_TIME, "iv": abi.R_DURATION, "s": abi.R_STRING, "y": abi.R_BYTES, "k": abi.R_INT64}
    assert {c.name: c.repr for c in out.cols} == want
    # strictify of a strict batch shares it
    again = tf.strictify(tf.DeviceBatch.upload(out), SCHEMA).download()
    assert_batches_equal(again, out, "idempotent")


@pytest.mark.parametrize("bad", [(7, 0, "128"), (7, 3, "-1"), (7, 1, "12x"), (3, 5, "yes"), (40, 10, "5 parsecs"), (40, 8, "not a date"), (0, 7, "1,5"), (59, 6, "1e39"),
                                 (7, 4, "18446744073709551616")])
def test_first_failing_value_fails_the_call(tf, oracle, bad):
    rows = text_rows(60, 9, bad_at=bad)
    if bad[0] < 50:
        rows[50][2] = ["string", "later and also bad"]   # a later row: not the one reported
    host = abi.batch_from_rows(SCHEMA, NAMES, rows, "db", "t")
    ref = oracle.strictify(host, SCHEMA)
    assert ref.errors and ref.errors[0][0] == bad[0]
    with pytest.raises(tf.TfgpuError) as ei:
        tf.strictify(tf.DeviceBatch.upload(host), SCHEMA)
    assert ei.value.code == tf.ERR_INVALID and ei.value.bad == (bad[0], bad[1]), (ei.value.bad, str(ei.value))
    assert "failed to strictify the value of column [%d] \"%s\"" % (bad[1], NAMES[bad[1]]) in str(ei.value)


def test_integer_kinds_convert_with_gos_rules(tf, oracle):
    schema = abi.Schema.of([["a", "int8"], ["b", "uint32"], ["c", "int64"], ["d", "boolean"], ["e", "float"], ["f", "timestamp"], ["g", "interval"], ["h", "uint8"]])
    names = [c.name for c in schema.cols]
    rng = np.random.default_rng(5)
    rows = [[["int64", int(rng.integers(-128, 128))], ["int32", int(rng.integers(0, 1 << 31))], ["uint64", int(rng.integers(0, 1 << 63))], ["int16", int(rng.integers(-2, 3))],
             ["int64", int(rng.integers(-(1 << 40), 1 << 40))], ["uint32", int(rng.integers(0, 1 << 32))], ["int32", int(rng.integers(-1000, 1000))], ["bool", bool(r & 1)]]
            for r in range(500)]
    rows[17][5] = ["nil", None]
    host = abi.batch_from_rows(schema, names, rows, "db", "t")
    ref = oracle.strictify(host, schema)
    assert not ref.errors
    out = tf.strictify(tf.DeviceBatch.upload(host), schema).download()
    assert_batches_equal(out, ref.batch, "ints")
    for bad_col, v, why in [(0, ["int64", 128], "range"), (1, ["int32", -1], "negative"), (7, ["bool", True], None)]:
        rows2 = [list(r) for r in rows]
        rows2[9][bad_col] = v
        h2 = abi.batch_from_rows(schema, names, rows2, "db", "t")
        r2 = oracle.strictify(h2, schema)
        if why is None:
            assert not r2.errors
            continue
        assert r2.errors[0][0] == 9
        with pytest.raises(tf.TfgpuError) as ei:
            tf.strictify(tf.DeviceBatch.upload(h2), schema)
        assert ei.value.code == tf.ERR_INVALID and ei.value.bad == (9, bad_col)


def test_numbers_bools_and_times_under_text_types(tf, oracle):
    """castx.ToStringE of an integer kind / bool / time.Time / time.Duration (caste.go:58-106): strconv's decimal text, FormatBool,
    Time.String(), Duration.String(); castx.ToByteSliceE (:16-28) takes []byte and string only — anything else fails the call at its
    first value."""
    schema = abi.Schema.of([["a", "utf8"], ["b", "utf8"], ["c", "utf8"], ["d", "utf8"], ["e", "utf8"], ["f", "utf8"], ["k", "int32"]])
    names = [c.name for c in schema.cols]
    rng = np.random.default_rng(11)
    rows = [[["int64", int(rng.integers(-(1 << 62), 1 << 62))], ["uint64", int(rng.integers(0, 1 << 63)) * 2 + 1], ["int8", int(rng.integers(-128, 128))], ["bool", bool(r & 1)],
             ["time", "20%02d-%02d-%02dT%02d:%02d:%02d.%09dZ" % (rng.integers(0, 99), rng.integers(1, 13), rng.integers(1, 29), rng.integers(0, 24), rng.integers(0, 60), rng.integers(0, 60), rng.integers(0, 10 ** 9) if r % 3 else 0)],
             ["string", "already text %d" % r], ["int32", r]] for r in range(700)]
    rows[5][0] = ["nil", None]; rows[9][3] = ["nil", None]; rows[11][4] = ["nil", None]
    host = abi.batch_from_rows(schema, names, rows, "db", "t")
    ref = oracle.strictify(host, schema)
    assert not ref.errors
    out = tf.strictify(tf.DeviceBatch.upload(host), schema).download()
    assert_batches_equal(out, ref.batch, "to text")
    assert out.col("a").repr == abi.R_STRING and out.col("a").get_bytes(0) == str(rows[0][0][1]).encode() and out.col("d").get_bytes(1) == b"true"
    # under "string" ([]byte): the first value of the integer column fails the call; a column of nils does not
    bschema = abi.Schema.of([["a", "string"], ["k", "int32"]])
    hb = abi.batch_from_rows(bschema, ["a", "k"], [[["nil", None], ["int32", 0]], [["int64", 7], ["int32", 1]], [["int64", 8], ["int32", 2]]], "db", "t")
    rb = oracle.strictify(hb, bschema)
    assert rb.errors and rb.errors[0][0] == 1
    with pytest.raises(tf.TfgpuError) as ei:
        tf.strictify(tf.DeviceBatch.upload(hb), bschema)
    assert ei.value.code == tf.ERR_INVALID and ei.value.bad == (1, 0)
    nils = abi.batch_from_rows(bschema, ["a", "k"], [[["nil", None], ["int32", 0]], [["nil", None], ["int32", 1]]], "db", "t")
    nils.cols[0] = abi.Column("a", "string", abi.R_INT64, values=np.zeros(2, np.int64), validity=np.zeros(2, bool))
    o2 = tf.strictify(tf.DeviceBatch.upload(nils), bschema).download()
    assert o2.col("a").repr == abi.R_BYTES and [o2.col("a").pyvalue(i) for i in range(2)] == [["nil", None]] * 2


def test_go_floats_under_every_numeric_and_text_type(tf, oracle):
    """A Go float64 / float32 value: FormatFloat(f, 'f', -1, bits) under "utf8" and (castx.ToJSONNumberE) as the json.Number of "double";
    int64(f) / uint64(f) with toSignedInt / toUnsignedInt's limits, f != 0, float32(f) under the numeric types."""
    import struct
    schema = abi.Schema.of([["s", "utf8"], ["d", "double"], ["i", "int32"], ["u", "uint16"], ["b", "boolean"], ["f", "float"], ["t", "utf8"], ["k", "int32"]])
    names = [c.name for c in schema.cols]
    rng = np.random.default_rng(17)
    specials = [0.0, -0.0, 1.0, 0.1, 1e21, 1e-7, 123456789.125, 5e-324, 1.7976931348623157e308, float("inf"), float("-inf"), float("nan"), 4.35, 1e15, 0.000001]
    rows = []
    for r in range(600):
        x = specials[r] if r < len(specials) else struct.unpack("<d", struct.pack("<Q", int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))))[0] if r % 3 == 0 else float(rng.uniform(-1e6, 1e6))
        y = float(np.float32(rng.uniform(-3e4, 3e4)))
        rows.append([["float64", x], ["float64", x], ["float64", float(rng.uniform(-2e9, 2e9))], ["float32", float(np.float32(rng.uniform(0, 65000)))], ["float64", float(r % 3)],
                     ["float64", x if x == x and abs(x) < 3e38 else 1.5], ["float32", y], ["int32", r]])
    rows[20][2] = ["nil", None]; rows[21][0] = ["nil", None]; rows[22][1] = ["nil", None]
    host = abi.batch_from_rows(schema, names, rows, "db", "t")
    ref = oracle.strictify(host, schema)
    assert not ref.errors
    out = tf.strictify(tf.DeviceBatch.upload(host), schema).download()
    assert_batches_equal(out, ref.batch, "floats")
    assert out.col("s").get_bytes(4) == b"1000000000000000000000" and out.col("s").get_bytes(9) == b"+Inf" and out.col("d").repr == abi.R_JSONNUM
    for bad_col, v, code in [(2, ["float64", 3e9], tf.ERR_INVALID), (3, ["float32", -1.0], tf.ERR_INVALID), (3, ["float32", 70000.0], tf.ERR_INVALID), (2, ["float64", float("nan")], tf.ERR_UNSUPPORTED)]:
        rows2 = [list(x) for x in rows]
        rows2[33][bad_col] = v
        h2 = abi.batch_from_rows(schema, names, rows2, "db", "t")
        with pytest.raises(tf.TfgpuError) as ei:
            tf.strictify(tf.DeviceBatch.upload(h2), schema)
        assert ei.value.code == code and ei.value.bad == (33, bad_col), (v, str(ei.value))
        if code == tf.ERR_INVALID:
            assert oracle.strictify(h2, schema).errors[0][0] == 33


def test_pairs_left_to_the_host_are_refused_by_name(tf):
    schema = abi.Schema.of([["a", "interval"]])
    host = abi.batch_from_rows(schema, ["a"], [[["float64", 5.5]], [["float64", 6.25]]], "db", "t")
    with pytest.raises(tf.TfgpuError) as ei:   # cast.ToDurationE(float64): not restated
        tf.strictify(tf.DeviceBatch.upload(host), schema)
    assert ei.value.code == tf.ERR_UNSUPPORTED and "column a" in str(ei.value)
    # a column the schema does not name is left alone
    host = abi.batch_from_rows(abi.Schema.of([["a", "int32"], ["zz", "utf8"]]), ["a", "zz"], [[["string", "7"], ["int64", 1]]], "db", "t")
    out = tf.strictify(tf.DeviceBatch.upload(host), abi.Schema.of([["a", "int32"]])).download()
    assert out.col("a").repr == abi.R_INT32 and int(out.col("a").values[0]) == 7 and out.col("zz").repr == abi.R_INT64
