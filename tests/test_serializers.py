"""Sink-side marshalling: oracle pinned to the reference's vectors (CPU), HIP
kernels against the oracle and the same vectors (GPU)."""
import json

import numpy as np
import pytest

from transferia_amd import abi
from util import golden, item_to_batch
import os as _os

SEED0 = int(_os.environ.get("TFGPU_TEST_SEED", "0"))  # 0 = the committed seeds; other values: soak runs (TFGPU_TEST_SEED=n bash tools/gpu_visit.sh tests TAG)
# the largest random batch: 20 011 rows on the GPU; the CPU pre-flight (a lockstep emulator, tools/hipemu) takes a size that still spans several workgroups
_BIG = 4099 if __import__("os").environ.get("TFGPU_TEST_EMU_LIB") else 20011

FMT = {"ch": abi.FMT_CH_JSON_EACH_ROW, "json": abi.FMT_JSON, "csv": abi.FMT_CSV, "raw": abi.FMT_RAW}


def _case_batch(case):
    """A case holds one item, or (Batch/Stream serializer canon) several items of one schema: those become
    the rows of one batch -- what serializeBatch's loop (batch.go:40-57) sees, table names apart."""
    if "items" in case:
        first = case["items"][0]
        item = {k: v for k, v in first.items() if k != "values"}
        item["rows"] = [it["values"] for it in case["items"]]
        return item_to_batch(item)
    return item_to_batch(case["item"])


def _opts(o):
    return abi.serialize_options(add_closing_newline=o.get("add_closing_newline", False), any_as_string=o.get("any_as_string", False),
                                 ch_types=o.get("ch_types"))


def _check_case(case, text: bytes):
    s = text.decode("utf-8", "surrogateescape")
    if "expect" in case:
        assert s == case["expect"], case["name"]
    for frag in case.get("contains", []):
        assert frag in s, (case["name"], frag, s)
    if "json_roundtrip" in case:
        assert json.loads(s) == case["json_roundtrip"], case["name"]


def test_oracle_reference_vectors(oracle):
    for case in golden("serializers.json")["cases"]:
        b, schema = _case_batch(case)
        out = oracle.serialize(FMT[case["format"]], b, schema, _opts(case["opts"]))
        assert out is not None, case["name"]
        _check_case(case, out)


def _canon_tables():
    """TestBatchSerializer / TestStreamSerializer csv + json over the first 10 items of every all-databases canon case
    (reference_test.go:80-259): per table the rows and the lines of the three canon `result` files that belong to them."""
    out = []
    for t in golden("serializers_canon.json")["tables"]:
        item = dict(t["common"], rows=t["rows"])
        out.append((t, item))
    return out


CANON_VARIANTS = [("json", "json", {}), ("json_newline", "json", {"add_closing_newline": True}), ("csv", "csv", {})]


def _check_canon(serialize):
    done = skipped = 0
    for t, item in _canon_tables():
        if "skip" in t:  # time.Time with a zone offset: the column form holds UTC instants only
            skipped += len(t["rows"])
            continue
        b, schema = item_to_batch(item)
        for key, fmt, o in CANON_VARIANTS:
            out = serialize(FMT[fmt], b, schema, _opts(o))
            assert out is not None, (t["name"], key)
            assert out.decode("utf-8") == t["expect"][key], (t["name"], key)
        done += len(t["rows"])
    assert (done, skipped) == (116, 2)  # of the 118 items of ReadChangeItems(10)


def test_oracle_batch_serializer_canon(oracle):
    _check_canon(lambda f, b, schema, o: oracle.serialize(f, b, schema, o))


def test_oracle_strictify_first_rules(oracle):
    """What the canon above pins, as small cases: the serializers strictify first, so a Go string under "string" leaves as
    base64, a Go float under "double" as FormatFloat(v, 'f', -1, bits); the CSV serializer's `any` is json.Marshal (HTML escaped)."""
    schema = abi.Schema.of([["b", "string", False], ["d", "double", False], ["f", "float", False], ["a", "any", False]])
    rows = [[["string", "hi\x00"], ["float64", 1e21], ["float32", 1e21], ["json", '{"k":"<&>"}']],
            [["string", ""], ["float64", 5e-324], ["float32", 1.5], ["json", '"<"']]]
    b = abi.batch_from_rows(schema, ["b", "d", "f", "a"], rows, "", "t")
    js = oracle.serialize(abi.FMT_JSON, b, schema).decode()
    assert js.split("\n")[0] == '{"a":{"k":"<&>"},"b":"aGkA","d":1000000000000000000000,"f":1e+21}'
    assert js.split("\n")[1] == '{"a":"<","b":"","d":0.' + "0" * 323 + '5,"f":1.5}'
    cs = oracle.serialize(abi.FMT_CSV, b, schema).decode()
    assert cs.split("\n")[0] == 'aGkA,1000000000000000000000,1000000000000000000000,"{""k"":""\\u003c\\u0026\\u003e""}"'
    assert cs.split("\n")[1] == ',0.' + "0" * 323 + '5,1.5,"""\\u003c"""'


def test_oracle_escaping_rules(oracle):
    """encoding/json vs writeQuoted differ exactly where SURVEY §7 says: <>& stay (escapeHTML off),
    U+2028/9 and invalid UTF-8 are rewritten by encoding/json only, writeQuoted keeps raw bytes."""
    schema = abi.Schema.of([["s", "utf8", False]])
    raw = "a<b>& \"\\\n\t\x01\x7f".encode("utf-8") + b"\xfe\xc3"
    b = abi.batch_from_rows(schema, ["s"], [[["string", raw.decode("latin-1")]]], "", "t")
    b.cols[0].data = np.frombuffer(raw, np.uint8).copy(); b.cols[0].offsets = np.array([0, len(raw)], np.uint32)
    js = oracle.serialize(abi.FMT_JSON, b, schema)
    assert js == b'{"s":"a<b>&\\u2028\\"\\\\\\n\\t\\u0001\x7f\\ufffd\\ufffd"}'
    ch = oracle.serialize(abi.FMT_CH_JSON_EACH_ROW, b, schema, _opts({"ch_types": [[1, 0]]}))
    assert ch == b'{"s":"a<b>&\xe2\x80\xa8\\"\\\\\\n\\t\\u0001\x7f\xfe\xc3"}\n'
    cs = oracle.serialize(abi.FMT_CSV, b, schema)
    assert cs == b'"a<b>&\xe2\x80\xa8""\\\n\t\x01\x7f\xfe\xc3"\n'


def _random_batch(rng, n):
    from test_gpu_transformers import _random_batch as rb
    b, schema = rb(rng, n)
    keep = [c for c in b.cols if c.repr not in (abi.R_FLOAT32, abi.R_FLOAT64)]
    b = abi.Batch(keep, n, "db", "tbl")
    # json.Number and pre-marshalled any values
    nums = [b"0", b"-1.5", b"1e5", b"1E+3000", b"-2e400", b"1.7976931348623157e308", b"1.7976931348623159e308", b"inf", b"", b"123456789012345678901234567890",
            b"0.000001e315", b"179769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207096330286416692887910946555547851940402630657488671505820681908902000708383676273854845817711531764475730270069855571366959622842914819860834936475292719074168444365510704342711559699508093042880177904174497792"]
    pick = [nums[i] for i in rng.integers(0, len(nums), n)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in pick])]).astype(np.uint32)
    b.cols.append(abi.Column("num", "double", abi.R_JSONNUM, offsets=off, data=np.frombuffer(b"".join(pick), np.uint8).copy()))
    docs = [b"null", b"{\"a\":[1,2,{\"b\":\"<x>\"}]}", b"[]", b"\"s,\\\"q\\\"\"", b"12", b"true"]
    pick = [docs[i] for i in rng.integers(0, len(docs), n)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in pick])]).astype(np.uint32)
    b.cols.append(abi.Column("doc", "any", abi.R_JSON, offsets=off, data=np.frombuffer(b"".join(pick), np.uint8).copy(), validity=rng.random(n) > 0.2))
    specials = ["", " lead", "a,b", "q\"q", "\\.", "line\nbreak", " nbsp", "plain", " sep", "ctl\x01\x1f", "<&>", "\tx"]
    pick = [specials[i].encode("utf-8") for i in rng.integers(0, len(specials), n)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in pick])]).astype(np.uint32)
    b.cols.append(abi.Column("sp", "utf8", abi.R_STRING, offsets=off, data=np.frombuffer(b"".join(pick), np.uint8).copy()))
    b.cols.append(abi.Column("anys", "any", abi.R_STRING, offsets=off.copy(), data=np.frombuffer(b"".join(pick), np.uint8).copy()))
    schema = abi.Schema.of([[c.name, c.dtype, False] for c in b.cols])
    return b, schema


@pytest.mark.parametrize("n", [1, 7, 300])
def test_oracle_json_rows_parse_back(oracle, n):
    """Every JSON row the oracle emits is valid JSON holding every column."""
    rng = np.random.default_rng(SEED0 + (n))
    b, schema = _random_batch(rng, n)
    b.cols = [c for c in b.cols if c.name != "num"]  # 1E+3000 is not a Python float
    schema = abi.Schema.of([[c.name, c.dtype, False] for c in b.cols])
    out = oracle.serialize(abi.FMT_JSON, b, schema)
    lines = out.decode("utf-8").split("\n")
    assert len(lines) == n
    for ln in lines:
        assert sorted(json.loads(ln).keys()) == sorted(c.name for c in b.cols)


# ---------------------------------------------------------------- GPU -------------
@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


@pytest.mark.gpu
def test_gpu_reference_vectors(tf):
    for case in golden("serializers.json")["cases"]:
        b, _ = _case_batch(case)
        out = tf.serialize(FMT[case["format"]], tf.DeviceBatch.upload(b), _opts(case["opts"])).download()
        _check_case(case, out)


@pytest.mark.gpu
def test_gpu_batch_serializer_canon(tf):
    _check_canon(lambda f, b, schema, o: tf.serialize(f, tf.DeviceBatch.upload(b), o).download())


@pytest.mark.gpu
def test_gpu_canon_corpus_jsoneachrow(tf, oracle):
    """The same corpus as input to MarshalCItoJSON (the reference only benchmarks it there, httpuploader/bench/bench_test.go):
    device against the oracle on every provider's value forms; what the oracle does not restate the device must refuse."""
    same = refused = 0
    for t, item in _canon_tables():
        b, schema = item_to_batch(item)
        for o in ({}, {"any_as_string": True}):
            ref = oracle.serialize(abi.FMT_CH_JSON_EACH_ROW, b, schema, _opts(o))
            if ref is None:
                with pytest.raises(tf.TfgpuError):
                    tf.serialize(abi.FMT_CH_JSON_EACH_ROW, tf.DeviceBatch.upload(b), _opts(o))
                refused += 1
                continue
            got = tf.serialize(abi.FMT_CH_JSON_EACH_ROW, tf.DeviceBatch.upload(b), _opts(o)).download()
            assert got == ref, (t["name"], o)
            same += 1
    assert same >= 60, (same, refused)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 63, 1000, _BIG])
def test_gpu_serializers_match_oracle(tf, oracle, n):
    rng = np.random.default_rng(SEED0 + (100 + n))
    b, schema = _random_batch(rng, n)
    db = tf.DeviceBatch.upload(b)
    variants = [
        (abi.FMT_JSON, {}), (abi.FMT_JSON, {"add_closing_newline": True}), (abi.FMT_JSON, {"any_as_string": True}),
        (abi.FMT_CSV, {}), (abi.FMT_CH_JSON_EACH_ROW, {}), (abi.FMT_CH_JSON_EACH_ROW, {"any_as_string": True}),
        (abi.FMT_CH_JSON_EACH_ROW, {"ch_types": [[(1 << (i % 5)) if i % 3 else 0, i % 10] for i in range(len(b.cols))]}),
    ]
    for fmt, o in variants:
        ref = oracle.serialize(fmt, b, schema, _opts(o))
        assert ref is not None, (fmt, o)
        got = tf.serialize(fmt, db, _opts(o)).download()
        if got != ref:
            k = next(i for i in range(min(len(got), len(ref))) if got[i] != ref[i]) if got[: len(ref)] != ref[: len(got)] else min(len(got), len(ref))
            raise AssertionError(f"fmt={fmt} opts={o}: first difference at byte {k}: gpu={got[max(0,k-60):k+60]!r} ref={ref[max(0,k-60):k+60]!r}")


@pytest.mark.gpu
def test_gpu_serialize_unsupported_and_empty(tf, oracle):
    schema = abi.Schema.of([["f", "double", False]])
    b = abi.batch_from_rows(schema, ["f"], [[["float64", 1.5]]], "", "t")
    assert oracle.serialize(abi.FMT_JSON, b, schema) == tf.serialize(abi.FMT_JSON, tf.DeviceBatch.upload(b)).download() == b'{"f":1.5}'
    bn = abi.batch_from_rows(schema, ["f"], [[["float64", 1.5]], [["float64", float("nan")]]], "", "t")
    assert oracle.serialize(abi.FMT_JSON, bn, schema) is None  # json: unsupported value: NaN
    with pytest.raises(tf.TfgpuError) as ei:
        tf.serialize(abi.FMT_JSON, tf.DeviceBatch.upload(bn))
    assert ei.value.code == tf.ERR_UNSUPPORTED
    # strconv.FormatFloat(f, 'f', -1): ClickHouse numeric columns and the CSV serializer render Go floats on device
    import random
    import struct
    rng = random.Random(SEED0 + (9))
    vals = [1.5, -0.0, 0.1, 1e21, 1e-7, 123456789.125, 5e-324, 1.7976931348623157e308] + [struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0] for _ in range(3000)]
    vals = [v for v in vals if v == v and abs(v) != float("inf")]
    s2 = abi.Schema.of([["f", "double", False], ["g", "float", False], ["i", "int64", True]])
    rows = [[["float64", v], ["float32", float(np.float32(rng.uniform(-1e6, 1e6)))], ["int64", k]] for k, v in enumerate(vals)]
    b2 = abi.batch_from_rows(s2, ["f", "g", "i"], rows, "", "t")
    for fmt in (abi.FMT_CSV, abi.FMT_CH_JSON_EACH_ROW, abi.FMT_JSON):
        ref = oracle.serialize(fmt, b2, s2)
        assert ref is not None
        assert tf.serialize(fmt, tf.DeviceBatch.upload(b2)).download() == ref
    e = abi.Batch([abi.Column("a", "int32", abi.R_INT32, values=np.zeros(0, np.int32))], 0, "", "t")
    for fmt in (abi.FMT_JSON, abi.FMT_CSV, abi.FMT_CH_JSON_EACH_ROW):
        assert tf.serialize(fmt, tf.DeviceBatch.upload(e)).download() == b""


@pytest.mark.gpu
def test_gpu_hits_chain_to_jsoneachrow(tf, oracle):
    """configs[2] shape: parse -> filter -> ClickHouse JSONEachRow, bytes identical to the oracle."""
    from transferia_amd import workload
    schema = workload.hits_schema()
    data = workload.hits_csv(2000)
    opts = workload.hits_csv_options()
    db, _, errs = tf.csv_parse(opts, schema, data)
    res = tf.apply_chain([tf.Transformer("filter_rows", {"filter": "eventdate >= 2013-07-15"})], db)
    got = tf.serialize(abi.FMT_CH_JSON_EACH_ROW, res.transformed).download()
    ref = oracle.csv_parse(opts, schema, data, "", "")
    ref2 = oracle.apply_chain([oracle.Transformer("filter_rows", {"filter": "eventdate >= 2013-07-15"})], ref.batch, ref.schema)
    exp = oracle.serialize(abi.FMT_CH_JSON_EACH_ROW, ref2.batch, ref2.schema)
    assert got == exp and got.count(b"\n") == res.transformed.nrows


def _mirror_batch(datas, keys=None, repr_bytes=False):
    from transferia_amd import queue
    rows = []
    for i, d in enumerate(datas):
        rows.append([["string", "t"], ["int", 0], ["uint64", i], ["time", "2024-01-01T00:00:00Z"],
                     ["nil", None] if d is None else (["bytes", d.encode("utf-8").decode("latin-1")] if repr_bytes else ["string", d]), ["nil", None], ["bytes", "k%d" % i]])
    b = abi.batch_from_rows(queue.RAW_DATA_SCHEMA, list(queue.RAW_DATA_COLUMNS), rows, "", "t")
    b.schema = queue.RAW_DATA_SCHEMA
    return b


def test_oracle_raw_rules(oracle):
    """rawSerializer (raw.go:24-63): the `data` value, '\\n' between items, a closing '\\n' on request; anything that is
    not a mirror item, or a nil `data`, is an error."""
    from transferia_amd import queue
    b = _mirror_batch(["a", "", "b\nc"])
    assert oracle.serialize(abi.FMT_RAW, b, queue.RAW_DATA_SCHEMA) == b"a\n\nb\nc"
    assert oracle.serialize(abi.FMT_RAW, b, queue.RAW_DATA_SCHEMA, _opts({"add_closing_newline": True})) == b"a\n\nb\nc\n"
    assert oracle.serialize(abi.FMT_RAW, _mirror_batch(["a", None]), queue.RAW_DATA_SCHEMA) is None
    s = abi.Schema.of([["a", "int32", True]])
    assert oracle.serialize(abi.FMT_RAW, abi.batch_from_rows(s, ["a"], [[["int32", 1]]], "", "t"), s) is None


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 64, 1000, 50021])
def test_gpu_raw_matches_oracle(tf, oracle, n):
    import random
    from transferia_amd import queue
    rng = random.Random(SEED0 + 77 + n)
    alphabet = "ab\n\"\\é中 {}:,"
    datas = ["".join(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 3, 17, 64, 200]))) for _ in range(n)]
    for rb in (False, True):
        b = _mirror_batch(datas, repr_bytes=rb)
        for o in ({}, {"add_closing_newline": True}):
            ref = oracle.serialize(abi.FMT_RAW, b, queue.RAW_DATA_SCHEMA, _opts(o))
            assert ref is not None
            assert tf.serialize(abi.FMT_RAW, tf.DeviceBatch.upload(b), _opts(o)).download() == ref


def _batch_serializer_go(serialize_rows, n, sep, concurrency, threshold, disable, for_writer):
    """pkg/serializer/batch.go restated over a per-slice serializer (rows [a, b) → the items joined by `sep`, none behind the last):
    Serialize (:73-117) or the Write calls of SerializeAndWrite (:119-209)."""
    if disable:
        concurrency, threshold = 1, 0
    else:
        threshold = threshold or 25000
    if concurrency < 2 or n <= threshold:
        whole = serialize_rows(0, n)
        return [whole] if for_writer else whole
    parts = [serialize_rows(a, min(a + threshold, n)) for a in range(0, n, threshold)]
    if for_writer:
        return [p_ + (sep if i != len(parts) - 1 else b"") for i, p_ in enumerate(parts)]
    joined = sep.join(parts)
    return joined[: len(joined) - len(sep)] if sep and joined.endswith(sep) else joined


@pytest.mark.gpu
def test_gpu_batch_serializer_parts(tf, oracle):
    """batchSerializer as a component (tfgpu_serialize_batch): parts of `threshold` items, the one trimmed separator of Serialize,
    the Write calls of SerializeAndWrite — against batch.go restated over the oracle's serializers."""
    from transferia_amd import queue
    rng = np.random.default_rng(SEED0 + 4321)
    b, schema = _random_batch(rng, 211)
    datas = ["m%d" % i + ("\n" if i % 7 == 0 else "") for i in range(210)] + ["ends with the separator\n"]
    mirror = _mirror_batch(datas, repr_bytes=True)
    cases = [(abi.FMT_JSON, b, schema, {}), (abi.FMT_JSON, b, schema, {"add_closing_newline": True}), (abi.FMT_CSV, b, schema, {}),
             (abi.FMT_RAW, mirror, queue.RAW_DATA_SCHEMA, {}), (abi.FMT_RAW, mirror, queue.RAW_DATA_SCHEMA, {"add_closing_newline": True})]
    for fmt, bt, sch, o in cases:
        db = tf.DeviceBatch.upload(bt)
        sep = b"" if fmt == abi.FMT_CSV or o.get("add_closing_newline") else b"\n"

        def rows(a, e, bt=bt, sch=sch, fmt=fmt, o=o):
            if a >= e:
                return b""
            r = oracle.serialize(fmt, abi.slice_batch(bt, a, e) if hasattr(abi, "slice_batch") else _slice(bt, a, e), sch, _opts(o))
            assert r is not None
            return r
        for conc, thr, dis in [(4, 16, False), (4, 211, False), (4, 210, False), (1, 16, False), (0, 50, False), (4, 16, True), (2, 1, False), (3, 100, False)]:
            want = _batch_serializer_go(rows, bt.nrows, sep, conc or 8, thr, dis, False)
            got = tf.serialize_batch(fmt, db, _opts(o), concurrency=conc, threshold=thr, disable_concurrency=dis, gomaxprocs=8).download()
            assert got == want, (fmt, o, conc, thr, dis, len(got), len(want))
            writes = _batch_serializer_go(rows, bt.nrows, sep, conc or 8, thr, dis, True)
            buf, ends = tf.serialize_batch(fmt, db, _opts(o), concurrency=conc, threshold=thr, disable_concurrency=dis, gomaxprocs=8, for_writer=True)
            text = buf.download()
            assert text == b"".join(writes), (fmt, o, conc, thr, dis)
            assert ends == list(np.cumsum([len(w) for w in writes])), (fmt, o, conc, thr, dis, ends[:4])


def _slice(b, a, e):
    import copy
    out = copy.copy(b)
    out.nrows = e - a
    out.cols = []
    for c in b.cols:
        d = copy.copy(c)
        if c.repr in abi.VAR_REPRS:
            o0, o1 = int(c.offsets[a]), int(c.offsets[e])
            d.offsets = (c.offsets[a:e + 1] - c.offsets[a]).astype(np.uint32)
            d.data = c.data[o0:o1].copy() if o1 > o0 else np.zeros(1, np.uint8)
        else:
            d.values = c.values[a:e].copy()
            if getattr(c, "nanos", None) is not None:
                d.nanos = c.nanos[a:e].copy()
        if getattr(c, "validity", None) is not None:
            d.validity = c.validity[a:e].copy()
        out.cols.append(d)
    for k in ("kind", "src_row", "part_id"):
        v = getattr(b, k, None)
        if v is not None:
            setattr(out, k, v[a:e].copy())
    return out


@pytest.mark.gpu
def test_gpu_raw_refusals(tf):
    with pytest.raises(tf.TfgpuError):  # GetRawMessageData: unexpected data type <nil>
        tf.serialize(abi.FMT_RAW, tf.DeviceBatch.upload(_mirror_batch(["a", None, "b"])))
    s = abi.Schema.of([["a", "int32", True]])
    with pytest.raises(tf.TfgpuError):  # raw.go:29-31: not a mirror item
        tf.serialize(abi.FMT_RAW, tf.DeviceBatch.upload(abi.batch_from_rows(s, ["a"], [[["int32", 1]]], "", "t")))
    e = _mirror_batch([])
    assert tf.serialize(abi.FMT_RAW, tf.DeviceBatch.upload(e)).download() == b""
    assert tf.serialize(abi.FMT_RAW, tf.DeviceBatch.upload(e), _opts({"add_closing_newline": True})).download() == b""


@pytest.mark.gpu
def test_gpu_rows_larger_than_the_lds_image(tf, oracle):
    """ser_chunk_write assembles 64 rows of a column chunk in the wave's 9 KiB LDS image; tiles that outgrow it (one 70 KB cell)
    are written by the same wave straight to HBM.  Mixed batch: both paths, every format."""
    import random
    rng = random.Random(SEED0 + 31)
    schema = abi.Schema.of([["k", "int64", True], ["s", "utf8", False], ["t", "utf8", False], ["d", "double", False]])
    rows = []
    for k in range(300):
        big = "x\"é\n" * rng.choice([0, 1, 3, 17500, 12000]) if k % 37 == 5 else "ab" * (k % 50)
        rows.append([["int64", k * 1_000_003], ["string", big], ["nil", None] if k % 7 == 0 else ["string", "t%d" % k], ["float64", k / 3]])
    b = abi.batch_from_rows(schema, ["k", "s", "t", "d"], rows, "", "t")
    for fmt in (abi.FMT_JSON, abi.FMT_CSV, abi.FMT_CH_JSON_EACH_ROW):
        for o in ({}, {"add_closing_newline": True}):
            ref = oracle.serialize(fmt, b, schema, _opts(o))
            assert ref is not None
            assert tf.serialize(fmt, tf.DeviceBatch.upload(b), _opts(o)).download() == ref, (fmt, o)


@pytest.mark.gpu
@pytest.mark.parametrize("knobs", [{}, {"TFGPU_SLAB_RSHIFT": "2"}, {"TFGPU_SLAB_RSHIFT": "5", "TFGPU_SLAB_THREADS": "512"}, {"TFGPU_SLAB_IMAGE_BYTES": "1100"}, {"TFGPU_SLAB_IMAGE_BYTES": "3000", "TFGPU_SLAB_XCD": "0"}])
def test_gpu_slab_form_of_the_text_serializers(tf, oracle, monkeypatch, knobs):
    """The opt-in slab form (tf_serslab.inc: whole rows rendered into an LDS image, one contiguous run a workgroup) byte for byte against the
    oracle, every format: random batches (nils, `any`, floats, escapes), the big-cell batch (rows longer than the image go straight to HBM),
    and with the image clamped so that ordinary rows take the sub-run and the straight-to-HBM paths."""
    import random
    monkeypatch.setenv("TFGPU_SER_SLAB", "1")
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for n in (1, 63, 1000):
        rng = np.random.default_rng(SEED0 + (700 + n))
        b, schema = _random_batch(rng, n)
        db = tf.DeviceBatch.upload(b)
        for fmt, o in ((abi.FMT_JSON, {}), (abi.FMT_JSON, {"add_closing_newline": True, "any_as_string": True}), (abi.FMT_CSV, {}), (abi.FMT_CH_JSON_EACH_ROW, {}),
                       (abi.FMT_CH_JSON_EACH_ROW, {"ch_types": [[(1 << (i % 5)) if i % 3 else 0, i % 10] for i in range(len(b.cols))]})):
            ref = oracle.serialize(fmt, b, schema, _opts(o))
            assert ref is not None and tf.serialize(fmt, db, _opts(o)).download() == ref, (n, fmt, o)
    prng = random.Random(SEED0 + 32)
    schema = abi.Schema.of([["k", "int64", True], ["s", "utf8", False], ["t", "utf8", False], ["d", "double", False]])
    rows = []
    for k in range(200):
        big = "x\"é\n" * prng.choice([0, 1, 3, 17500]) if k % 37 == 5 else "ab" * (k % 50)
        rows.append([["int64", k * 1_000_003], ["string", big], ["nil", None] if k % 7 == 0 else ["string", "t%d" % k], ["float64", k / 3]])
    b = abi.batch_from_rows(schema, ["k", "s", "t", "d"], rows, "", "t")
    for fmt in (abi.FMT_JSON, abi.FMT_CSV, abi.FMT_CH_JSON_EACH_ROW):
        assert tf.serialize(fmt, tf.DeviceBatch.upload(b), _opts({})).download() == oracle.serialize(fmt, b, schema, _opts({})), fmt


@pytest.mark.gpu
def test_gpu_lean_length_pass_of_jsoneachrow(tf, oracle, monkeypatch):
    """ser_chunk_len_fast (round 6): JSONEachRow lengths without the write pass's walk — digit counts, text as 2 + bytes unless ser_text_flags found a byte
    that escapes in the COLUMN, dates as a constant unless the year leaves 0000-9999, DateTime64 / epoch seconds as digit counts.  Batches made ONLY of the
    kinds it takes (so that it runs: the kernel list says so), every ClickHouse target kind the flags can name, nils, extremes, text with and without
    escapes — byte for byte the oracle's MarshalCItoJSON and the same call with TFGPU_SER_LEN_FAST=0."""
    rng = np.random.default_rng(SEED0 + 4242)
    for n in (1, 65, 1500):
        ext = np.array([0, 1, -1, 9, 10, 99, 100, -128, 127, 255, 32767, -32768, 65535, 99999999, 100000000, -99999999, -100000000, 2**31 - 1, -2**31, 2**32 - 1, 2**63 - 1, -2**63], dtype=object)
        pick = lambda lo, hi: np.array([int(max(lo, min(hi, int(ext[i])))) if rng.random() < 0.5 else int(rng.integers(lo, hi)) for i in rng.integers(0, len(ext), n)], dtype=object)  # noqa: E731
        cols = []
        for name, dt, rp, npdt, lo, hi in (("i8", "int8", abi.R_INT8, np.int8, -128, 127), ("i16", "int16", abi.R_INT16, np.int16, -32768, 32767), ("i32", "int32", abi.R_INT32, np.int32, -2**31, 2**31 - 1),
                                           ("i64", "int64", abi.R_INT64, np.int64, -2**63, 2**63 - 1), ("u8", "uint8", abi.R_UINT8, np.uint8, 0, 255), ("u16", "uint16", abi.R_UINT16, np.uint16, 0, 65535),
                                           ("u32", "uint32", abi.R_UINT32, np.uint32, 0, 2**32 - 1), ("u64", "uint64", abi.R_UINT64, np.uint64, 0, 2**63 - 1), ("ianys", "any", abi.R_INT64, np.int64, -2**63, 2**63 - 1),
                                           ("istr", "utf8", abi.R_INT32, np.int32, -2**31, 2**31 - 1)):
            cols.append(abi.Column(name, dt, rp, values=np.array([int(x) for x in pick(lo, hi)], dtype=npdt), validity=rng.random(n) > 0.15))

        def text(name, dt, rp, words):
            p_ = [words[i] for i in rng.integers(0, len(words), n)]
            off = np.concatenate([[0], np.cumsum([len(x) for x in p_])]).astype(np.uint32)
            return abi.Column(name, dt, rp, offsets=off, data=np.frombuffer(b"".join(p_) or b"\0", np.uint8).copy(), validity=rng.random(n) > 0.1)
        cols.append(text("plain", "utf8", abi.R_STRING, [b"", b"a", b"http://example.org/x?y=1", "café €".encode(), b"x" * 70]))
        cols.append(text("esc", "utf8", abi.R_STRING, [b"", b"q\"q", b"back\\slash", b"line\nbreak\ttab", b"ctl\x01\x1f", b"plain words only"]))
        cols.append(text("blob", "string", abi.R_BYTES, [b"", b"\x00\x01\xff", b"bytes \"quoted\"", b"raw"]))
        cols.append(text("anytext", "any", abi.R_STRING, [b"v", b"\\", b"long " * 9]))
        cols.append(text("dec", "utf8", abi.R_STRING, [b"1.50", b"-0.001", b"12345678901234567890.12"]))
        cols.append(abi.Column("flag", "boolean", abi.R_BOOL, values=rng.integers(0, 2, n).astype(np.uint8), validity=rng.random(n) > 0.2))
        cols.append(abi.Column("flagany", "any", abi.R_BOOL, values=rng.integers(0, 2, n).astype(np.uint8)))
        secs = np.array([int(x) for x in rng.integers(-3 * 10**9, 5 * 10**9, n)], dtype=np.int64)
        edge = [-62167219200, 253402300799, -62167219201, 253402300800, 0, -1, 86399]
        secs[: min(n, len(edge))] = edge[: min(n, len(edge))]
        for name, dt in (("day", "date"), ("at", "datetime"), ("ts", "timestamp"), ("ts3", "timestamp"), ("ts0", "timestamp"), ("tssec", "timestamp")):
            cols.append(abi.Column(name, dt, abi.R_TIME, values=secs.copy(), nanos=rng.integers(0, 10**9, n).astype(np.int32), validity=rng.random(n) > 0.1))
        b = abi.Batch(cols, n, "db", "tbl")
        schema = abi.Schema.of([[c.name, c.dtype, False] for c in cols])
        names = [c.name for c in cols]
        base = [[0, 0] for _ in cols]
        for nm, fl, prec in (("plain", 1, 0), ("esc", 1, 0), ("blob", 1, 0), ("anytext", 1, 0), ("dec", 8, 0), ("istr", 1, 0), ("day", 2, 0), ("at", 0, 0), ("ts", 4, 9), ("ts3", 4, 3), ("ts0", 4, 0), ("tssec", 0, 0)):
            base[names.index(nm)] = [fl, prec]
        quoted = [list(x) for x in base]
        for nm in ("i8", "i64", "u64", "u16"):
            quoted[names.index(nm)] = [1, 0]      # a numeric column under a String target: quoted digits
        db = tf.DeviceBatch.upload(b)
        for o in ({}, {"ch_types": base}, {"ch_types": quoted}, {"ch_types": base, "any_as_string": True}):
            want = oracle.serialize(abi.FMT_CH_JSON_EACH_ROW, b, schema, _opts(o))
            assert want is not None
            tf.prof_reset(); tf.prof_enable(True)
            got = tf.serialize(abi.FMT_CH_JSON_EACH_ROW, db, _opts(o)).download()
            kernels = {k for k, l, ms in tf.prof_get() if l}
            tf.prof_enable(False)
            assert "ser_text_flags" in kernels, kernels     # the lean pass ran
            assert bytes(got) == bytes(want), (n, o.keys())
            monkeypatch.setenv("TFGPU_SER_LEN_FAST", "0")
            slow = tf.serialize(abi.FMT_CH_JSON_EACH_ROW, db, _opts(o)).download()
            monkeypatch.delenv("TFGPU_SER_LEN_FAST")
            assert bytes(slow) == bytes(want)
        if n >= 65:   # a slice: its text columns' offsets start above zero (ser_text_flags takes the byte range from the offsets themselves)
            part = db.slice(16, n - 30)
            whole = tf.serialize(abi.FMT_CH_JSON_EACH_ROW, db, _opts({"ch_types": base})).download()
            lines = bytes(whole).split(b"\n")
            got = tf.serialize(abi.FMT_CH_JSON_EACH_ROW, part, _opts({"ch_types": base})).download()
            assert bytes(got) == b"\n".join(lines[16:n - 14]) + b"\n"
            part.free()
