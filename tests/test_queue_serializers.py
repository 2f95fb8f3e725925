"""Queue serializers (pkg/serializer/queue native / json + their batchers): the oracle against the
reference's canon (CPU), the HIP path against the oracle and the same canon (GPU)."""
import json

import numpy as np
import pytest

from transferia_amd import abi
from util import golden, item_to_batch
import os as _os

SEED0 = int(_os.environ.get("TFGPU_TEST_SEED", "0"))  # 0 = the committed seeds; other values: soak runs (TFGPU_TEST_SEED=n bash tools/gpu_visit.sh tests TAG)
# the largest random batch: 20 011 rows on the GPU; the CPU pre-flight (a lockstep emulator, tools/hipemu) takes a size that still spans several workgroups
_BIG = 4099 if __import__("os").environ.get("TFGPU_TEST_EMU_LIB") else 20011

G = golden("queue_serializers.json")


def _with_old_keys(b, item):
    ok = item.get("old_keys")
    if ok:
        b.old_keys = [abi.column_from_values(nm, "utf8", [r[i] for r in ok["rows"]]) for i, nm in enumerate(ok["names"])]
        if "present" in ok:
            b.old_present = np.array(ok["present"], bool)
    return b


def _meta(m, n):
    return abi.row_meta(n, ids=m.get("ids"), lsns=m.get("lsns"), commit_times=m.get("commit_times"), counters=m.get("counters"),
                        tx_ids=m.get("tx_ids"), queries=m.get("queries"), names_form=m.get("names_form"))


def _native_topic_name():
    c = G["native_topic_name"]
    b, schema = item_to_batch(c["item"])
    return c, b, schema, _meta(c["meta"], b.nrows), abi.queue_options(abi.QFMT_NATIVE, table_schema=schema)


def _marshal_json():
    c = G["marshal_json"]
    b, schema = item_to_batch(c["item"])
    _with_old_keys(b, c["item"])
    o = abi.queue_options(abi.QFMT_NATIVE, table_schema_json=c["table_schema_json"], old_key_types=c["item"]["old_keys"]["types"],
                          group_part_ids=[c["meta"]["part"]], group_rows=[b.nrows])
    return c, b, schema, _meta(c["meta"], b.nrows), o


def _batching_cases(fmt):
    """queue/test.go commonTest: message-size limits are expressed through the element size, as the Go test does."""
    c = G["batching"]
    b, schema = item_to_batch(c["item"])
    for enabled, max_items, size, expected in c["table"]:
        yield b, schema, enabled, max_items, size, expected


def _size(fmt, elem, k, delta):
    if k == 0:
        return delta
    return (2 if fmt == abi.QFMT_NATIVE else 0) + (k - 1) + elem * k + delta   # batchSizer of native_/json_batcher_test.go


def _check_batching(serialize, fmt):
    b, schema = item_to_batch(G["batching"]["item"])
    one = serialize(abi.queue_options(fmt, table_schema=schema), b, schema)
    elem = len(one[0]) - (2 if fmt == abi.QFMT_NATIVE else 0)
    assert len(one) == 5 and all(m == one[0] for m in one)
    body = one[0][1:-1] if fmt == abi.QFMT_NATIVE else one[0]
    sep = b"," if fmt == abi.QFMT_NATIVE else b"\n"
    for enabled, max_items, size, expected in G["batching"]["table"]:
        o = abi.queue_options(fmt, enabled=enabled, max_change_items=max_items, max_message_size=_size(fmt, elem, *size) if size else 0, table_schema=schema)
        msgs = serialize(o, b, schema)
        assert len(msgs) == expected, (fmt, enabled, max_items, size)
        rows = 0
        for m in msgs:  # every message is k copies of the element, joined and wrapped
            inner = m[1:-1] if fmt == abi.QFMT_NATIVE else m
            k = inner.count(body)
            assert inner == sep.join([body] * k) and k >= 1
            if fmt == abi.QFMT_NATIVE:
                assert m[:1] == b"[" and m[-1:] == b"]"
            rows += k
        assert rows == 5


# ---------------------------------------------------------------- oracle (CPU) ----
def test_oracle_native_canon(oracle):
    c, b, schema, meta, o = _native_topic_name()
    msgs = oracle.queue_serialize(o, b, schema, meta)
    assert [m.decode() for m in msgs] == [c["value"]] * 2
    assert c["key"] == b.table_ns + "_" + b.table_name  # Fqtn(), change_item.go:139-141


def test_oracle_marshal_json_canon(oracle):
    c, b, schema, meta, o = _marshal_json()
    msgs = oracle.queue_serialize(o, b, schema, meta)
    assert len(msgs) == 1 and msgs[0].decode() == "[" + c["value"] + "]"


def test_oracle_json_canon(oracle):
    c = G["json_all_types"]
    b, schema = item_to_batch(c["item"])
    msgs = oracle.queue_serialize(abi.queue_options(abi.QFMT_JSON), b, schema)
    assert [m.decode() for m in msgs] == [c["value"]]
    assert c["key"] == b.table_ns + "_" + b.table_name


@pytest.mark.parametrize("fmt", [abi.QFMT_NATIVE, abi.QFMT_JSON])
def test_oracle_batching_table(oracle, fmt):
    _check_batching(lambda o, b, s: oracle.queue_serialize(o, b, s), fmt)


def test_oracle_native_forms(oracle):
    """nil / empty ColumnNames, OldKeys on some rows only, HTML escaping, non-row kinds, update/delete in the JSON format."""
    schema = abi.Schema.of([["k", "int32", True], ["s", "utf8", False]])
    rows = [[["int32", 1], ["string", "<a&b> "]], [["int32", 2], ["nil", None]], [["int32", 3], ["string", "x"]]]
    b = abi.batch_from_rows(schema, ["k", "s"], rows, "ns", "t", ["insert", "update", "delete"])
    b.old_keys = [abi.column_from_values("k", "int32", [["int32", 0], ["int32", 20], ["int32", 30]])]
    b.old_present = np.array([False, True, True])
    meta = abi.row_meta(3, ids=[1, 2, 3], names_form=[0, 0, 1], tx_ids=["", "t<1>", ""])
    msgs = oracle.queue_serialize(abi.queue_options(abi.QFMT_NATIVE, omit_table_schema=True), b, schema, meta)
    assert msgs[0] == (b'[{"id":1,"nextlsn":0,"commitTime":0,"txPosition":0,"kind":"insert","schema":"ns","table":"t","part":"",'
                       b'"columnnames":["k","s"],"columnvalues":[1,"\\u003ca\\u0026b\\u003e\\u2028"],"oldkeys":{},"tx_id":"","query":""}]')
    assert msgs[1] == (b'[{"id":2,"nextlsn":0,"commitTime":0,"txPosition":0,"kind":"update","schema":"ns","table":"t","part":"",'
                       b'"columnnames":["k","s"],"columnvalues":[2,null],"oldkeys":{"keynames":["k"],"keyvalues":[20]},"tx_id":"t\\u003c1\\u003e","query":""}]')
    assert msgs[2] == (b'[{"id":3,"nextlsn":0,"commitTime":0,"txPosition":0,"kind":"delete","schema":"ns","table":"t","part":"",'
                       b'"columnnames":null,"oldkeys":{"keynames":["k"],"keyvalues":[30]},"tx_id":"","query":""}]')
    for m in msgs:
        json.loads(m)
    assert oracle.queue_serialize(abi.queue_options(abi.QFMT_JSON), b, schema) is None  # "JsonSerializer: unsupported kind"
    b.kind = np.array([abi.K_INSERT, abi.K_OTHER, abi.K_INSERT], np.uint8)
    js = oracle.queue_serialize(abi.queue_options(abi.QFMT_JSON), b, schema)
    assert js == [b'{"k":1,"s":"<a&b>\\u2028"}', b"", b'{"k":3,"s":"x"}']  # !IsRowEvent: empty value; SetEscapeHTML(false)
    assert oracle.queue_serialize(abi.queue_options(abi.QFMT_NATIVE), b, schema) is None  # non-row kinds stay with the stock path


def _random_case(rng, n):
    from test_serializers import _random_batch
    b, schema = _random_batch(rng, n)
    b.kind = rng.integers(0, 3, n).astype(np.uint8)
    b.old_keys = [abi.Column("i64", "int64", abi.R_INT64, values=rng.integers(-9, 9, n).astype(np.int64)),
                  abi.column_from_values("sk", "utf8", [["string", "k<%d>" % i] if i % 3 else ["nil", None] for i in range(n)])]
    b.old_present = rng.random(n) < 0.6
    b.part_id = rng.integers(0, 5, n).astype(np.uint32)  # sharder_transformer ran: PartID = itoa
    perm = rng.permutation(n).astype(np.int32)  # the batch went through a row-moving step: meta is read through src_row
    b.src_row = perm
    meta = abi.row_meta(n, ids=rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32), lsns=rng.integers(0, 2**63, n, dtype=np.uint64) * 2 + 1,
                        commit_times=rng.integers(0, 2**62, n, dtype=np.uint64), counters=rng.integers(-5, 1000, n),
                        tx_ids=[("tx&%d" % i) if i % 4 == 0 else "" for i in range(n)], queries=[("select '%d' < 2" % i) if i % 7 == 0 else "" for i in range(n)],
                        names_form=(rng.random(n) < 0.15).astype(np.uint8) * rng.integers(1, 3, n).astype(np.uint8))
    return b, schema, meta


# ---------------------------------------------------------------- GPU -------------
@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


def _gpu(tf, o, b, meta=None):
    r = tf.queue_serialize(o, tf.DeviceBatch.upload(b), meta)
    return r.messages()


@pytest.mark.gpu
def test_gpu_queue_canon(tf):
    c, b, schema, meta, o = _native_topic_name()
    assert [m.decode() for m in _gpu(tf, o, b, meta)] == [c["value"]] * 2
    c, b, schema, meta, o = _marshal_json()
    assert [m.decode() for m in _gpu(tf, o, b, meta)] == ["[" + c["value"] + "]"]
    c = G["json_all_types"]
    b, schema = item_to_batch(c["item"])
    assert [m.decode() for m in _gpu(tf, abi.queue_options(abi.QFMT_JSON), b)] == [c["value"]]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [abi.QFMT_NATIVE, abi.QFMT_JSON])
def test_gpu_queue_batching_table(tf, fmt):
    _check_batching(lambda o, b, s: _gpu(tf, o, b), fmt)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 65, 1000, _BIG])
def test_gpu_queue_matches_oracle(tf, oracle, n):
    rng = np.random.default_rng(SEED0 + (900 + n))
    b, schema, meta = _random_case(rng, n)
    db = tf.DeviceBatch.upload(b)
    groups = [n] if n < 3 else [n // 3, 0, n - n // 3]
    variants = [
        dict(fmt=abi.QFMT_NATIVE, table_schema=schema),
        dict(fmt=abi.QFMT_NATIVE, omit_table_schema=True, old_key_types=["int64", "utf8"], enabled=True, max_change_items=7),
        dict(fmt=abi.QFMT_NATIVE, table_schema_json='[{"x":1}]', enabled=True, max_message_size=3000, group_rows=groups, group_part_ids=["p<%d>" % i for i in range(len(groups))]),
        dict(fmt=abi.QFMT_NATIVE, omit_table_schema=True, enabled=True, max_message_size=1500, max_change_items=3, group_rows=groups),
    ]
    for v in variants:
        o = abi.queue_options(v.pop("fmt"), **v)
        ref = oracle.queue_serialize(o, b, schema, meta)
        assert ref is not None
        res = tf.queue_serialize(o, db, meta)
        got = res.messages()
        assert len(got) == len(ref), (v, len(got), len(ref))
        for i, (x, y) in enumerate(zip(got, ref)):
            if x != y:
                k = next((j for j in range(min(len(x), len(y))) if x[j] != y[j]), min(len(x), len(y)))
                raise AssertionError(f"{v}: message {i} differs at byte {k}: gpu={x[max(0,k-80):k+80]!r} ref={y[max(0,k-80):k+80]!r}")
        assert list(res.msg_row) == oracle.queue_serialize.rows
    # JSON format: inserts only (update / delete fail the call like the reference), with a non-row item in the middle
    b.kind = np.where(rng.random(n) < 0.1, abi.K_OTHER, abi.K_INSERT).astype(np.uint8)
    db = tf.DeviceBatch.upload(b)
    for v in (dict(), dict(enabled=True, max_change_items=5), dict(enabled=True, max_message_size=700, group_rows=groups)):
        o = abi.queue_options(abi.QFMT_JSON, **v)
        ref = oracle.queue_serialize(o, b, schema)
        assert ref is not None and tf.queue_serialize(o, db, None).messages() == ref, v


@pytest.mark.gpu
def test_gpu_queue_cut_plan_sweep(tf, oracle, monkeypatch):
    """The cut plan finds a message's end by a galloping search over the element lengths' prefix sums instead of the batchers' running sum
    (native_batcher.go:10-63, json_batcher.go:11-66): every limit from 'no element fits' to 'everything fits', with and without an item count,
    over groups with an empty one in the middle — against the oracle's batchers, and (TFGPU_CUT_PLAN_CHECK=1) against the literal loop inside
    the library."""
    monkeypatch.setenv("TFGPU_CUT_PLAN_CHECK", "1")
    n = 331
    rng = np.random.default_rng(SEED0 + 4242)
    b, schema, meta = _random_case(rng, n)
    db = tf.DeviceBatch.upload(b)
    groups = [n // 3, 0, 1, n - n // 3 - 1]
    for fmt in (abi.QFMT_NATIVE, abi.QFMT_JSON):
        if fmt == abi.QFMT_JSON:
            b.kind = np.full(n, abi.K_INSERT, np.uint8)
            db = tf.DeviceBatch.upload(b)
        for size in (1, 2, 40, 150, 151, 152, 300, 301, 700, 1500, 4096, 20000, 1 << 22, 0):
            for items in (0, 1, 2, 3, 17, 1000):
                for gr in (None, groups):
                    kw = dict(enabled=True, max_message_size=size, max_change_items=items)
                    if gr:
                        kw["group_rows"] = gr
                    if fmt == abi.QFMT_NATIVE:
                        kw["omit_table_schema"] = True
                    o = abi.queue_options(fmt, **kw)
                    ref = oracle.queue_serialize(o, b, schema, meta if fmt == abi.QFMT_NATIVE else None)
                    assert ref is not None
                    res = tf.queue_serialize(o, db, meta if fmt == abi.QFMT_NATIVE else None)
                    assert res.messages() == ref, (fmt, size, items, bool(gr))
                    assert list(res.msg_row) == oracle.queue_serialize.rows, (fmt, size, items, bool(gr))


@pytest.mark.gpu
def test_gpu_queue_canon_corpus(tf, oracle):
    """The all-databases corpus (tests/golden/serializers_canon.json: arrays, hstore, decimals as json.Number, FixedString, YT
    composite values) as input to the native and JSON queue serializers: device against the oracle."""
    from util import golden
    checked = 0
    for t in golden("serializers_canon.json")["tables"]:
        b, schema = item_to_batch(dict(t["common"], rows=t["rows"]))
        db = tf.DeviceBatch.upload(b)
        for o in (abi.queue_options(abi.QFMT_NATIVE, table_schema=schema), abi.queue_options(abi.QFMT_NATIVE, omit_table_schema=True, enabled=True, max_change_items=2),
                  abi.queue_options(abi.QFMT_JSON), abi.queue_options(abi.QFMT_JSON, enabled=True, max_message_size=600)):
            ref = oracle.queue_serialize(o, b, schema)
            if ref is None:
                with pytest.raises(tf.TfgpuError):
                    tf.queue_serialize(o, db, None)
                continue
            assert tf.queue_serialize(o, db, None).messages() == ref, t["name"]
            checked += 1
    assert checked >= 120, checked


@pytest.mark.gpu
def test_gpu_queue_errors_and_empty(tf, oracle):
    schema = abi.Schema.of([["k", "int32", True]])
    b = abi.batch_from_rows(schema, ["k"], [[["int32", 1]], [["int32", 2]]], "", "t", ["insert", "update"])
    with pytest.raises(tf.TfgpuError) as ei:
        tf.queue_serialize(abi.queue_options(abi.QFMT_JSON), tf.DeviceBatch.upload(b), None)
    assert ei.value.code == tf.ERR_UNSUPPORTED
    b.kind = np.array([abi.K_INSERT, abi.K_OTHER], np.uint8)
    with pytest.raises(tf.TfgpuError) as ei:
        tf.queue_serialize(abi.queue_options(abi.QFMT_NATIVE), tf.DeviceBatch.upload(b), None)
    assert ei.value.code == tf.ERR_UNSUPPORTED
    e = abi.Batch([abi.Column("a", "int32", abi.R_INT32, values=np.zeros(0, np.int32))], 0, "", "t")
    for fmt in (abi.QFMT_NATIVE, abi.QFMT_JSON):
        assert tf.queue_serialize(abi.queue_options(fmt), tf.DeviceBatch.upload(e), None).messages() == []
    fs = abi.Schema.of([["f", "double", False]])
    bn = abi.batch_from_rows(fs, ["f"], [[["float64", 1.5]], [["float64", float("inf")]]], "", "t")
    assert oracle.queue_serialize(abi.queue_options(abi.QFMT_NATIVE), bn, fs) is None
    with pytest.raises(tf.TfgpuError):
        tf.queue_serialize(abi.queue_options(abi.QFMT_NATIVE), tf.DeviceBatch.upload(bn), None)


def test_oracle_native_messages_parse_back(oracle):
    """Every native message of the oracle is valid JSON whose fields, read back by an independent JSON implementation,
    are the inputs: key order of MarshalJSON's putItem sequence, values, OldKeys only on the rows that carry them, row meta
    through src_row."""
    import base64
    from collections import OrderedDict
    n = 300
    rng = np.random.default_rng(SEED0 + 77)
    b, schema, meta = _random_case(rng, n)
    b.cols = [c for c in b.cols if c.name not in ("num",)]  # 1E+3000 is not a Python float
    schema = abi.Schema.of([[c.name, c.dtype, False] for c in b.cols])
    ids, lsns = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32), rng.integers(0, 2**63, n, dtype=np.uint64)
    meta = abi.row_meta(n, ids=ids, lsns=lsns, tx_ids=["tx<%d>" % i for i in range(n)])
    msgs = oracle.queue_serialize(abi.queue_options(abi.QFMT_NATIVE, table_schema=schema, old_key_types=["int64", "utf8"]), b, schema, meta)
    assert len(msgs) == n
    for r, m in enumerate(msgs):
        (it,) = json.loads(m.decode("utf-8"), object_pairs_hook=OrderedDict)
        k = int(b.src_row[r])
        assert list(it)[:9] == ["id", "nextlsn", "commitTime", "txPosition", "kind", "schema", "table", "part", "columnnames"]
        assert list(it)[-3:] == ["oldkeys", "tx_id", "query"]
        assert (it["id"], it["nextlsn"], it["tx_id"], it["kind"], it["part"]) == (int(ids[k]), int(lsns[k]), "tx<%d>" % k, ["insert", "update", "delete"][int(b.kind[r])], str(int(b.part_id[r])))
        assert it["columnnames"] == [c.name for c in b.cols] and len(it["table_schema"]) == len(b.cols)
        for c, v in zip(b.cols, it["columnvalues"]):
            g, x = c.pyvalue(r)
            if g == "nil":
                assert v is None
            elif g == "string":
                assert v == x.decode("utf-8", "replace")
            elif g == "bytes":
                assert base64.b64decode(v) == x
            elif g in ("json", "jsonnum"):
                assert v == json.loads(x)
            elif g == "time":
                assert v.endswith("Z") and abi.parse_rfc3339(v) == (x[0], x[1])
            elif g == "bool":
                assert v is x
            else:
                assert v == x
        if b.old_present[r]:
            assert it["oldkeys"]["keynames"] == ["i64", "sk"] and it["oldkeys"]["keytypes"] == ["int64", "utf8"]
            assert it["oldkeys"]["keyvalues"][0] == int(b.old_keys[0].values[r])
        else:
            assert it["oldkeys"] == {}


@pytest.mark.gpu
def test_sr_to_native_carries_the_colschema_wire_fields(tf, oracle):
    """Confluent-SR → NativeSerializer (the configs[4] sink fed by an SR source): every ColSchema of an SR item carries the
    message's TableSchema / TableName (jsonPropertyToJSONSchemaRow, utils_json.go:71-95) and ChangeItem.MarshalJSON marshals
    them (col_schema.go:14-29) — so must the device, with fake_key / expression / properties besides.  Expected text: the
    struct tags of col_schema.go written out by hand for one column, then device == oracle on the whole messages."""
    import json as pyjson
    from transferia_amd import confluent_sr
    text = pyjson.dumps({"type": "object", "title": "shop.orders", "properties": {"id": {"type": "integer"}, "note": {"type": "string"}}, "required": ["id"]})
    title, rows = confluent_sr.json_schema_rows(text)
    ns, table = confluent_sr.build_json_table_id(title, confluent_sr.POLICY_DEBEZIUM_STYLE, "")
    schema = confluent_sr.table_schema(rows, ns, table)
    schema.cols[1].expression, schema.cols[1].fake_key, schema.cols[1].properties_json = "lower(note)", True, '{"default":"x"}'
    frames = [b"\0" + (7).to_bytes(4, "big") + pyjson.dumps({"id": i, "note": "n%d" % i}).encode() for i in range(50)]
    data, msgs = abi.messages(frames, list(range(50)), [0] * 50)
    opts = confluent_sr.sr_json_options(7, text)
    qo = abi.queue_options(abi.QFMT_NATIVE, table_schema=schema)
    ref = oracle.sr_json_parse(opts, data, msgs)
    exp = oracle.queue_serialize(qo, ref.batch, ref.schema)
    got = tf.queue_serialize(qo, tf.sr_json_parse(opts, data, msgs).device_batch).messages()
    assert got == exp and len(got) == 50
    item = pyjson.loads(got[0])[0]
    assert item["table_schema"][0] == {"table_schema": ns, "table_name": table, "path": "", "name": "id", "type": "int64", "key": False, "fake_key": False,
                                       "required": True, "expression": "", "original_type": ""}
    assert item["table_schema"][1] == {"table_schema": ns, "table_name": table, "path": "", "name": "note", "type": "utf8", "key": False, "fake_key": True,
                                       "required": False, "expression": "lower(note)", "original_type": "", "properties": {"default": "x"}}
    assert b'"fake_key":true,"required":false,"expression":"lower(note)","original_type":"","properties":{"default":"x"}}' in got[0]
