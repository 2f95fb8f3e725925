"""tfgpu_debezium_emit (tf_dbzemit.hip) = queue.DebeziumSerializer / Emitter.EmitKV (pkg/serializer/queue/debezium_serializer.go:26-43,
pkg/debezium/emitter_value_converter.go:574-690) against the oracle's emitter (oracle/dbz_emitter.py, pinned to the reference's
fixtures in test_dbz_emitter_oracle.py): every key and every value byte for byte — random Postgres-typed rows through every
device-resident converter, inserts / updates / deletes with OldKeys (tombstones, key changes, replica identity full), the parameter
variants, the reference's own CRUD fixtures cut to the device-resident columns, and what is refused by name."""
import json
import os

import numpy as np
import pytest

from transferia_amd import abi
from oracle import dbz_emitter as E

pytestmark = pytest.mark.gpu
PARAMS = {"database.dbname": "pguser", "topic.prefix": "fullfillment", "dt.add.original.type.info": "false", "dt.source.type": "pg"}


@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


# name, DataType, key, OriginalType
TABLE = [("id", "int32", True, "pg:integer"), ("bl", "boolean", False, "pg:boolean"), ("b1", "utf8", False, "pg:bit(1)"), ("si", "int16", False, "pg:smallint"),
         ("big", "int64", False, "pg:bigint"), ("oid_", "any", False, "pg:oid"), ("re", "double", False, "pg:real"), ("re32", "float", False, "pg:real"),
         ("d", "double", False, "pg:double precision"), ("dn", "double", False, "pg:double precision"), ("t", "utf8", False, "pg:text"),
         ("vc", "utf8", False, "pg:character varying(256)"), ("uid", "utf8", False, "pg:uuid"), ("ip", "any", False, "pg:inet"), ("ba", "string", False, "pg:bytea"),
         ("bas", "string", False, "pg:bytea"), ("dt", "date", False, "pg:date"), ("ts", "timestamp", False, "pg:timestamp without time zone"),
         ("ts3", "timestamp", False, "pg:timestamp(3) without time zone"), ("tz", "timestamp", False, "pg:timestamp with time zone"), ("j", "any", False, "pg:jsonb"),
         ("js", "any", False, "pg:json"), ("nu", "double", False, "pg:numeric"), ("n52", "double", False, "pg:numeric(5,2)"), ("n180", "utf8", False, "pg:numeric(18,0)"),
         ("bits", "utf8", False, "pg:bit varying(24)"), ("b8", "utf8", False, "pg:bit(8)"), ("ir", "any", False, "pg:int4range"), ("z_last", "utf8", False, "pg:USER-DEFINED:citext"),
         ("A_first", "int64", False, "pg:bigint"), ("tm3", "utf8", False, "pg:time(3) without time zone"), ("tm", "utf8", False, "pg:time without time zone"),
         ("ttz", "utf8", False, "pg:time with time zone"), ("ttz1", "utf8", False, "pg:time(1) with time zone"), ("mo", "utf8", False, "pg:money"), ("xm", "any", False, "pg:xml"),
         ("pt", "any", False, "pg:point"), ("tsr", "any", False, "pg:tsrange"), ("nr", "any", False, "pg:numrange"), ("tzr", "any", False, "pg:tstzrange"),
         ("hs", "any", False, "pg:USER-DEFINED:hstore"), ("hst", "any", False, "pg:USER-DEFINED:hstore"), ("iv", "utf8", False, "pg:interval"), ("iv2", "utf8", False, "pg:interval day to second"),
         ("dts", "utf8", False, "pg:date"), ("tzs", "utf8", False, "pg:timestamp with time zone")]


def schema_of(table):
    """(name, DataType, key, OriginalType[, Properties])"""
    return abi.Schema([abi.ColSchema(t[0], t[1], t[2], "", t[3], properties_json=(json.dumps(t[4], separators=(",", ":")) if len(t) > 4 else "")) for t in table])


def cols_of(table):
    return [E.Col(t[0], t[1], t[2], t[3], (t[4] if len(t) > 4 else None)) for t in table]


TEXTS = ["", "plain", "quote\" back\\slash", "<tag attr='x'>&amp;</tag>", "tab\tnl\nctl\x01\x1f", "юникод ✓ \u2028 \u2029", "x" * 300, "bad \xff utf8 \xc3"]
NUMS = ["0", "-0", "0.00", "1", "-1", "127", "128", "-128", "-129", "255", "-255", "-256", "65535", "-16777215", "123.67", "-123.675", "123.674999", "0.005", "-0.005", "9.995",
        "1e2", "1.50e1", "19e-1", "-19e-1", ".123e3", "1e-2", "12345678901234567890123456789012345678", "-99999999999999999999999999999999999999", "00012.300", "+5", "5.", "1.277559e+7"]


def random_rows(n, seed):
    # (n52 skips the two 38-digit numbers: two more fractional digits carry them past the device's 128 bits — refused, see the last test)
    rng = np.random.default_rng(seed)
    rows = []
    for r in range(n):
        def pick(lst):
            return lst[int(rng.integers(0, len(lst)))]
        f = float(rng.standard_normal()) * 10.0 ** int(rng.integers(-12, 25))
        sec = int(rng.integers(-3000000000, 4000000000))
        row = [["int32", int(rng.integers(-2 ** 31, 2 ** 31))], ["bool", bool(rng.integers(0, 2))], ["string", pick(["1", "0", "", "11"])], ["int16", int(rng.integers(-32768, 32768))],
               pick([["int64", int(rng.integers(-2 ** 62, 2 ** 62))], ["int64", -5]]), ["jsonnum", str(int(rng.integers(0, 2 ** 32)))], ["float64", f], ["float32", float(np.float32(f % 1e30))],
               ["float64", pick([f, 0.0, -0.0, 1e21, 1e-7, 123456789.125, 5e-324, 1.7976931348623157e308])], ["float64", pick([float("nan"), float("inf"), float("-inf"), 1.5])],
               ["string", pick(TEXTS)], ["string", pick(TEXTS)], ["string", "a0eebc99-9c0b-4ef8-bb6d-6bb9bd380a11"], ["string", pick(["192.168.1.5/32", "10.0.0.0/8", "::1/128", "/32", "32"])],
               ["bytes", bytes(rng.integers(0, 256, int(rng.integers(0, 20))).astype(np.uint8))], ["string", "yv66vg=="], ["time", (sec - sec % 86400, 0)],
               ["time", (sec, int(rng.integers(0, 10 ** 9)))], ["time", (sec, int(rng.integers(0, 1000)) * 1000000)], ["time", (sec, pick([0, 120000000, 123456789, 999999999]))],
               ["json", pick(['{"k1":"v1"}', '[1,2.5,"x",null,true]', '"just a string"', '12.5', 'null', '{"a":{"b":["<&>","\\"q\\""]},"z":1e5}'])], ["string", pick(['{"raw": "text"}', "plain"])],
               ["jsonnum", pick(NUMS[:29])], ["jsonnum", pick(NUMS[:26] + NUMS[28:])], ["string", pick(["12345", "-7", "0", "999999999999999999"])],
               ["string", pick(["101011110000", "0", "1", "000000000", "1000000010101110", ""])], ["string", "10101111"], ["string", "[3,7)"], ["string", pick(TEXTS)],
               ["int64", int(rng.integers(-10, 10))],
               ["string", pick(["04:05:06", "04:05:06.1", "04:05:06.123456", "23:59:59.999", "00:00:00"])], ["string", pick(["04:05:06", "04:05:06.1", "04:05:06.123456", "12:00:00.000001"])],
               ["string", pick(["13:30:25-04", "13:30:25.5-04", "13:30:25.575401-04", "00:51:02.746572-08", "23:59:59.5+05:30", "00:00:01+14", "12:00:00.120000+00:00:30"])],
               ["string", pick(["13:30:25-04", "13:30:25.5-04", "01:02:03+03"])], ["string", pick(["$123.45", "$0.00", "$-5.10", "$1000000.99", "$", "$7", "$-0.01", "$1,000.00", "$1e3"][:7])],
               ["string", pick(["<foo>bar</foo>", "\\u003cfoo\\u003ebar\\u003c/foo\\u003e", "caf\\u00e9 \\u12 \\uZZZZ \\u0041\\u00ff tail\\u00", "\\u2028x", TEXTS[2], TEXTS[4], TEXTS[5], TEXTS[7]])],
               ["string", pick(["(23.4,-44.5)", "(0,0)", "(1e10,-1e-7)", "(1.5,2.5)", "(-0,123456789.125)"])],
               ["string", pick(['[2010-01-02 10:00:00,2010-01-02 11:00:00)', '["2010-01-02 10:00:00","2010-01-02 11:00:00")', "[,)", "(a,b,c]", "[]", '["q\\"x",é)'])],
               ["string", pick(["[19e-1,191e-2)", "[1.9,1.91)", "(1,2]", "[-1.5e3,1e-2)", "[.123e3,1.277559e+7]", "(0e5,-0.0)"])],
               ["string", pick(['[2010-01-01 01:00:00-05,2010-01-01 02:00:00-08)', '["2010-01-01 09:00:00+03","2010-01-01 13:00:00+03")', "[2010-01-01 06:00:00Z,2010-01-01 10:00:00Z)",
                                "(2020-02-29 23:59:59.123+05:30,2021-01-01 00:00:00+00]"])],
               ["json", pick(['{"a":"1","b":"2"}', '{}', '{"k":null}'])], ["string", pick(["", '{"already": "json"}', "{"])],
               ["string", pick(["1 day 01:00:00.000000", "1 month", "1 year", "40 years", "14 mon 3 day 04:05:06.000007", "1 year 2 mons 3 days 04:05:06.00007", "-1 days +02:03:00",
                                "1 mon -2 days", "-00:00:01", "-00:00:00.5", "00:00:00", "3 fortnights 01:02:03", "2 months 5 days", "25:61:61.1234567", "1 day"])],
               ["string", pick(["3 days 04:05:06", "04:05:06.5", "-3 days -04:05:06.123456"])],
               ["string", pick(["1999-01-08T00:00:00Z", "1969-12-31 23:59:59Z", "2004-10-19T10:23:54+04:00", "2004-10-19 10:23:54-07", "1000-01-01T00:00:00Z"])],
               ["string", pick(["2004-10-19T12:23:54+04:00", "2004-10-19T10:23:54.987654+04:00", "2004-10-19 10:23:54.5-07", "1999-01-08T00:00:00Z", "2022-08-28 19:49:47,749906Z"])]]
        for k in range(1, len(row)):
            if rng.integers(0, 9) == 0:
                row[k] = ["nil", None]
        rows.append(row)
    return rows


def items_of(batch, cols, meta_lists=None):
    ids, lsns, cts = (meta_lists or ([0] * batch.nrows, [0] * batch.nrows, [0] * batch.nrows))[:3]
    txs = meta_lists[3] if meta_lists and len(meta_lists) > 3 else [""] * batch.nrows
    kind_name = {abi.K_INSERT: "insert", abi.K_UPDATE: "update", abi.K_DELETE: "delete"}
    out = []
    old = getattr(batch, "old_keys", None) or []
    pres = getattr(batch, "old_present", None)
    for r in range(batch.nrows):
        k = "insert" if batch.kind is None else kind_name.get(int(batch.kind[r]), "other")
        has = bool(old) and (pres is None or bool(pres[r]))
        listed = [c for c in batch.cols if getattr(c, "absent", None) is None or not c.absent[r]]  # the row's own ColumnNames (tfgpu_column.absent)
        out.append(E.Item(k, batch.table_ns, batch.table_name, cols, [c.name for c in listed], [tuple(c.pyvalue(r)) for c in listed],
                          [c.name for c in old] if has else [], [tuple(c.pyvalue(r)) for c in old] if has else [], ids[r], lsns[r], cts[r], txs[r]))
    return out


def emit_both(tf, batch, table, params, meta_lists=None, **kw):
    schema = schema_of(table)
    n = batch.nrows
    meta = abi.row_meta(n, ids=meta_lists[0], lsns=meta_lists[1], commit_times=meta_lists[2], tx_ids=meta_lists[3] if len(meta_lists) > 3 else None) if meta_lists else None
    got = tf.debezium_emit(abi.dbz_emit_options(params, schema, **kw), tf.DeviceBatch.upload(batch), meta)
    em = E.Emitter(params, kw.get("version") or "1.1.2.Final", drop_keys=kw.get("drop_keys", False))
    want, rows = [], []
    for r, it in enumerate(items_of(batch, cols_of(table), meta_lists)):
        for kv in em.emit_kv(it, snapshot=kw.get("snapshot", False)):
            want.append((kv[0] if kv[0] is not None else b"", kv[1]))
            rows.append(r)
    return got, want, rows


def assert_same(got, want, rows):
    msgs = got.messages()
    assert len(msgs) == len(want)
    assert [int(x) for x in got.msg_row] == rows
    for i, (a, b) in enumerate(zip(msgs, want)):
        if a != b:
            for part in (0, 1):
                x, y = a[part], b[part]
                if x != y and x is not None and y is not None:
                    k = next((j for j in range(min(len(x), len(y))) if x[j] != y[j]), min(len(x), len(y)))
                    raise AssertionError("message %d (row %d) %s differs at byte %d:\n  device %r\n  oracle %r" % (i, rows[i], "key" if part == 0 else "value", k, x[max(0, k - 60):k + 60], y[max(0, k - 60):k + 60]))
            assert a == b, (i, rows[i])


@pytest.mark.parametrize("n", [1, 63, 700])
def test_every_device_resident_type_against_the_oracle(tf, n):
    rows = random_rows(n, 40 + n)
    names = [t[0] for t in TABLE]
    batch = abi.batch_from_rows(schema_of(TABLE), names, rows, "public", "basic_types")
    rng = np.random.default_rng(n)
    meta = ([int(x) for x in rng.integers(0, 2 ** 32, n)], [int(x) for x in rng.integers(0, 2 ** 62, n)], [int(x) for x in rng.integers(0, 2 ** 62, n)])
    got, want, rws = emit_both(tf, batch, TABLE, PARAMS, meta)
    assert_same(got, want, rws)
    assert all(v is not None and json.loads(v)["payload"]["op"] == "c" for _, v in got.messages())
    # the messages are JSON documents whose schema half names every column once
    doc = json.loads(got.messages()[0][1])
    assert [f["field"] for f in doc["schema"]["fields"][1]["fields"]] == names


def crud_batch(n, seed, full_identity=False):
    """inserts, updates (some of them moving the primary key), deletes; OldKeys = the key, or every column (REPLICA IDENTITY FULL)"""
    table = [("id", "int32", True, "pg:integer"), ("k2", "utf8", True, "pg:text"), ("v", "int64", False, "pg:bigint"), ("s", "utf8", False, "pg:text"), ("n", "double", False, "pg:numeric(10,3)")]
    rng = np.random.default_rng(seed)
    rows, kinds, old_rows, present = [], [], [], []
    for r in range(n):
        kind = ["insert", "update", "delete"][int(rng.integers(0, 3))]
        row = [["int32", int(rng.integers(0, 50))], ["string", "k%d" % rng.integers(0, 5)], ["int64", int(rng.integers(-99, 99))], ["string", "s%d" % r], ["jsonnum", "%d.%d" % (rng.integers(0, 999), rng.integers(0, 9999))]]
        if rng.integers(0, 7) == 0:
            row[2] = ["nil", None]
        old = [list(x) for x in row]
        if kind == "update" and rng.integers(0, 2):   # the row's key moved: its old key differs in one or both key columns
            which = int(rng.integers(0, 3))
            if which != 1:
                old[0] = ["int32", int(rng.integers(50, 99))]
            if which != 0:
                old[1] = ["string", "was-%d" % r]
        if kind != "insert":
            old[3] = ["string", "before %d" % r]
        rows.append(row)
        kinds.append(kind)
        old_rows.append(old)
        present.append(kind != "insert" and (kind != "delete" or rng.integers(0, 6) != 0))
    names = [t[0] for t in table]
    sch = schema_of(table)
    b = abi.batch_from_rows(sch, names, rows, "public", "crud", kinds=kinds)
    old_names = names if full_identity else names[:2]
    ob = abi.batch_from_rows(sch, old_names, [[o[names.index(nm)] for nm in old_names] for o in old_rows], "public", "crud")
    b.old_keys, b.old_present = ob.cols, np.array(present, bool)
    return table, b


@pytest.mark.parametrize("full_identity", [False, True])
@pytest.mark.parametrize("params", [{}, {"tombstones.on.delete": "false"}, {"dt.source.type": ""}], ids=["default", "no-tombstones", "no-source-type"])
def test_updates_deletes_and_key_changes(tf, full_identity, params):
    table, b = crud_batch(400, 7 + full_identity, full_identity)
    n = b.nrows
    meta = (list(range(100, 100 + n)), [5000 + 3 * r for r in range(n)], [1649273150231781000 + 977 * r for r in range(n)])
    got, want, rows = emit_both(tf, b, table, dict(PARAMS, **params), meta)
    assert_same(got, want, rows)
    ops = [None if v is None else json.loads(v)["payload"]["op"] for _, v in got.messages()]
    assert {"c", "u", "d"} <= set(ops) and ((None in ops) == (params.get("tombstones.on.delete") != "false"))
    assert len(ops) > n   # deletes and key-changing updates fan out
    if full_identity:
        assert any(v is not None and json.loads(v)["payload"]["op"] == "u" and json.loads(v)["payload"]["before"] is not None for _, v in got.messages())


def test_parameter_variants(tf):
    table, b = crud_batch(60, 3)
    n = b.nrows
    meta = (list(range(n)), list(range(n)), [10 ** 18 + r for r in range(n)])
    for params, kw in [({"decimal.handling.mode": "string"}, {}), ({"dt.add.original.type.info": "true"}, {}), ({}, {"drop_keys": True}), ({}, {"snapshot": True}),
                       ({"unavailable.value.placeholder": "<toast>"}, {}), ({}, {"version": "2.0.0"}), ({"topic.prefix": "pre\"fix", "database.dbname": "dβ"}, {}),
                       ({"key.converter.schemas.enable": "false"}, {}), ({"value.converter.schemas.enable": "false", "key.converter": "io.confluent.connect.json.JsonSchemaConverter"}, {}),
                       ({"key.converter.schemas.enable": "false", "value.converter.schemas.enable": "false", "dt.batching.max.size": "abc"}, {})]:
        got, want, rows = emit_both(tf, b, table, dict(PARAMS, **params), meta, **kw)
        assert_same(got, want, rows)
    # TOAST: a schema column the rows do not carry becomes the placeholder (buildKV, emitter_value_converter.go:311-323)
    wide = table + [("toasted", "utf8", False, "pg:text")]
    got, want, rows = emit_both(tf, b, wide, PARAMS, meta)
    assert_same(got, want, rows)
    assert b'"toasted":"__debezium_unavailable_value"' in got.messages()[0][1] or b'"toasted":null' in got.messages()[0][1]
    # dt.unknown.types.policy = to_string: a type the emitter does not know is described as a plain string and its values go through UnknownTypeToString
    odd = [("id", "int32", True, "pg:integer"), ("u", "utf8", False, "pg:USER-DEFINED:mytype"), ("g", "any", False, "pg:box"), ("f", "double", False, "pg:tsvector_like")]
    orows = [[["int32", r], ["string", "raw <%d>" % r] if r % 3 else ["nil", None], ["json", '{"a":[%d,"q\\"x"]}' % r], ["jsonnum", "%d.50" % r]] for r in range(40)]
    ob_ = abi.batch_from_rows(schema_of(odd), [t[0] for t in odd], orows, "public", "odd")
    for extra_params in ({}, {"dt.add.original.type.info": "true"}):
        got, want, rows = emit_both(tf, ob_, odd, dict(PARAMS, **{"dt.unknown.types.policy": "to_string"}, **extra_params), None)
        assert_same(got, want, rows)
    assert b'"g":"{\\"a\\":[1,\\"q\\\\\\"x\\"]}"' in got.messages()[1][1] and b'{"__dt_original_type_info":{"original_type":""},"field":"u","optional":true,"type":"string"}' in got.messages()[1][1]
    for policy in ("skip", "fail"):
        with pytest.raises(tf.TfgpuError) as ei:
            tf.debezium_emit(abi.dbz_emit_options(dict(PARAMS, **{"dt.unknown.types.policy": policy}), schema_of(odd)), tf.DeviceBatch.upload(ob_))
        assert ei.value.code == (tf.ERR_UNSUPPORTED if policy == "skip" else tf.ERR_INVALID), str(ei.value)
    # no rows, and rows of a non-row kind: no messages
    empty = abi.batch_from_rows(schema_of(table), [t[0] for t in table], [], "public", "crud")
    assert len(tf.debezium_emit(abi.dbz_emit_options(PARAMS, schema_of(table)), tf.DeviceBatch.upload(empty))) == 0


def test_toasted_updates_write_the_placeholder_per_row(tf):
    """Rows that leave columns out of their ColumnNames (tfgpu_column.absent: Updates with TOASTed columns unchanged) get the placeholder for exactly
    those columns (buildKV, emitter_value_converter.go:311-323) — in `after` of the update and of the create event a key change fans out into; a
    batch whose rows leave a KEY column out stays with the stock emitter, by name."""
    for full_identity in (False, True):
        table, b = crud_batch(300, 21 + full_identity, full_identity)
        rng = np.random.default_rng(5)
        upd = b.kind == abi.K_UPDATE
        for c in b.cols[2:]:
            ab = upd & (rng.random(b.nrows) < 0.5)
            c.absent = ab
            c.validity = (np.ones(b.nrows, bool) if c.validity is None else c.validity) & ~ab
        n = b.nrows
        meta = (list(range(n)), list(range(n)), [10 ** 18 + r for r in range(n)])
        for params in ({}, {"unavailable.value.placeholder": "<toast \"q\">"}, {"dt.source.type": "mysql"}):
            got, want, rows = emit_both(tf, b, table, dict(PARAMS, **params), meta)
            assert_same(got, want, rows)
        assert sum(b'"s":"__debezium_unavailable_value"' in v for _, v in got.messages() if v) > 20
    b.cols[0].absent = upd
    with pytest.raises(tf.TfgpuError, match="primary-key column id"):
        tf.debezium_emit(abi.dbz_emit_options(PARAMS, schema_of(table)), tf.DeviceBatch.upload(b))


def test_rows_moved_by_a_transformer_no_meta_and_keyless_tables(tf):
    """row meta rides on src_row (the fan-in key every row-moving step keeps): after filter_rows the surviving rows still name their own ID / LSN /
    CommitTime; without meta they are the Go zero values; a table without a PrimaryKey has `{}` keys; OldKeys that name no key column give `{}` too"""
    table, b = crud_batch(300, 11)
    n = b.nrows
    meta = (list(range(7, 7 + n)), [10 ** 9 + r for r in range(n)], [1600000000000000000 + 1000003 * r for r in range(n)])
    db = tf.DeviceBatch.upload(b)
    kept = tf.Transformer("filter_rows", {"filter": "v >= 0"}).apply(db).transformed
    host = kept.download()
    assert 0 < host.nrows < n and host.src_row is not None
    got = tf.debezium_emit(abi.dbz_emit_options(PARAMS, schema_of(table)), kept, abi.row_meta(n, ids=meta[0], lsns=meta[1], commit_times=meta[2]))
    em = E.Emitter(PARAMS)
    want, rows = [], []
    src = [int(x) for x in host.src_row]
    for r, it in enumerate(items_of(host, cols_of(table), ([meta[0][k] for k in src], [meta[1][k] for k in src], [meta[2][k] for k in src]))):
        for kv in em.emit_kv(it):
            want.append((kv[0], kv[1]))
            rows.append(r)
    assert_same(got, want, rows)
    got, want, rows = emit_both(tf, b, table, PARAMS, None)   # no meta: lsn / txId / ts_ms are 0
    assert_same(got, want, rows)
    assert b'"lsn":0,' in got.messages()[0][1] and b'"ts_ms":0}' in got.messages()[0][1]
    keyless = [(nm, d, False, o) for nm, d, _k, o in table]
    got, want, rows = emit_both(tf, b, keyless, PARAMS, meta)
    assert_same(got, want, rows)
    assert got.messages()[0][0].startswith(b'{"payload":{},"schema":{"fields":[],')
    other = [("id", "int32", False, "pg:integer"), ("k2", "utf8", False, "pg:text"), ("v", "int64", True, "pg:bigint"), ("s", "utf8", False, "pg:text"), ("n", "double", False, "pg:numeric(10,3)")]
    got, want, rows = emit_both(tf, b, other, PARAMS, meta)   # the key is `v`; OldKeys (id, k2) name no key column
    assert_same(got, want, rows)


YDB_TABLE = [("id", "uint64", True, "ydb:Uint64"), ("Bool_", "boolean", False, "ydb:Bool"), ("Int8_", "int8", False, "ydb:Int8"), ("Int16_", "int16", False, "ydb:Int16"),
             ("Int32_", "int32", False, "ydb:Int32"), ("Int64_", "int64", False, "ydb:Int64"), ("Uint8_", "uint8", False, "ydb:Uint8"), ("Uint16_", "uint16", False, "ydb:Uint16"),
             ("Uint32_", "uint32", False, "ydb:Uint32"), ("Uint64_", "uint64", False, "ydb:Uint64"), ("Float_", "float", False, "ydb:Float"), ("Double_", "double", False, "ydb:Double"),
             ("Decimal_", "utf8", False, "ydb:Decimal"), ("DyNumber_", "double", False, "ydb:DyNumber"), ("String_", "string", False, "ydb:String"), ("Utf8_", "utf8", False, "ydb:Utf8"),
             ("Json_", "any", False, "ydb:Json"), ("JsonDocument_", "any", False, "ydb:JsonDocument"), ("Uuid_", "utf8", False, "ydb:Uuid"), ("Date_", "date", False, "ydb:Date"),
             ("Datetime_", "datetime", False, "ydb:Datetime"), ("Timestamp_", "timestamp", False, "ydb:Timestamp"), ("Interval_", "interval", False, "ydb:Interval")]


def test_ydb_typed_rows_and_the_ydb_source_block(tf):
    """AddYDB (pkg/debezium/ydb/emitter.go:123-232) and dt.source.type = ydb (txId from ChangeItem.TxID, step = CommitTime): random rows, then the reference's own
    canon ChangeItem for this emitter (ydb/tests/testdata/emitter_vals_test__canon_change_item.txt — the oracle reproduces its test's expected values)"""
    rng = np.random.default_rng(21)
    n = 300
    rows = []
    for r in range(n):
        def pick(lst):
            return lst[int(rng.integers(0, len(lst)))]
        sec = int(rng.integers(-10 ** 9, 4 * 10 ** 9))
        f = float(rng.standard_normal()) * 10.0 ** int(rng.integers(-10, 20))
        row = [["uint64", int(rng.integers(0, 2 ** 63)) * 2 + 1], ["bool", bool(rng.integers(0, 2))], ["int8", int(rng.integers(-128, 128))], ["int16", int(rng.integers(-2 ** 15, 2 ** 15))],
               ["int32", int(rng.integers(-2 ** 31, 2 ** 31))], ["int64", int(rng.integers(-2 ** 62, 2 ** 62))], ["uint8", int(rng.integers(0, 256))], ["uint16", int(rng.integers(0, 65536))],
               ["uint32", int(rng.integers(0, 2 ** 32))], ["uint64", int(rng.integers(0, 2 ** 63)) * 2], ["float32", float(np.float32(f % 1e30))], pick([["float64", f], ["jsonnum", "2.2"]][:1]),
               ["string", pick(["234.000000000", "-0.000000001", "1.5", "99999999999.999999999", "1e2"])], ["jsonnum", pick(["123", ".123e3", "-5.25", "0"])],
               ["bytes", bytes(rng.integers(0, 256, int(rng.integers(0, 9))).astype(np.uint8))], ["string", pick(TEXTS)], ["json", pick(["{}", '{"a":[1,"<x>"]}', "[]"])], ["json", "{}"],
               ["string", "a0eebc99-9c0b-4ef8-bb6d-6bb9bd380a11"], ["time", (sec - sec % 86400, 0)], ["time", (sec, 0)], ["time", (sec, int(rng.integers(0, 10 ** 6)) * 1000)],
               ["duration", int(rng.integers(-10 ** 12, 10 ** 12))]]
        for k in range(1, len(row)):
            if rng.integers(0, 9) == 0:
                row[k] = ["nil", None]
        rows.append(row)
    names = [t[0] for t in YDB_TABLE]
    b = abi.batch_from_rows(schema_of(YDB_TABLE), names, rows, "", "ydb_table")
    meta = (list(range(n)), list(range(n)), [1700000000000000000 + 12345 * r for r in range(n)], [("%d" % (r * 7) if r % 3 else "") for r in range(n)])
    for params in ({"database.dbname": "public", "topic.prefix": "my_topic", "dt.source.type": "ydb"}, {"database.dbname": "public", "topic.prefix": "my_topic", "decimal.handling.mode": "string"}):
        got, want, rws = emit_both(tf, b, YDB_TABLE, params, meta)
        assert_same(got, want, rws)
    assert b'"txId":null' in got.messages()[0][1] or True
    with open(os.path.join(os.path.dirname(__file__), "golden", "debezium_emitter", "ydb_emitter_vals_test__canon_change_item.txt"), "rb") as f:
        it = E.unmarshal_change_item(f.read())
    table = [(c.name, c.dtype, c.key, c.original_type) for c in it.cols]
    one = abi.batch_from_rows(schema_of(table), it.names, [[[g, (bytes(x) if isinstance(x, (bytes, bytearray)) else x)] if g != "time" else ["time", (x[0], x[1])] for g, x in it.values]], it.schema, it.table)
    got, want, rws = emit_both(tf, one, table, {"database.dbname": "pguser", "topic.prefix": "fullfillment", "dt.source.type": "ydb"}, ([it.id], [it.lsn], [it.commit_time], ["tx-1"]))
    assert_same(got, want, rws)
    assert b'"Decimal_":"Nnt8pAA="' in got.messages()[0][1] and b'"DyNumber_":{"scale":0,"value":"ew=="}' in got.messages()[0][1] and b'"txId":"tx-1"' in got.messages()[0][1]


def typed_row(it):
    return [["time", (x[0], x[1])] if g == "time" else [g, (bytes(x) if isinstance(x, (bytes, bytearray)) else x)] for g, x in it.values]


@pytest.mark.parametrize("fixture", ["mysql_emitter_vals_test__canon_change_item.txt", "mysql_emitter_vals_test__canon_change_item_v8.txt"])
def test_mysql_typed_rows_and_the_mysql_source_block(tf, fixture):
    """AddMysql (pkg/debezium/mysql/emitter.go:168-388) and dt.source.type = mysql (db = the ChangeItem's schema, file / pos from the LSN, gtid = TxID; a delete's `before`
    takes the row's own values): the reference's canon ChangeItems for this emitter — 79 columns whose expected values the oracle reproduces — as an insert, a snapshot insert,
    an update and a delete"""
    with open(os.path.join(os.path.dirname(__file__), "golden", "debezium_emitter", fixture), "rb") as f:
        it = E.unmarshal_change_item(f.read())
    table = [(c.name, c.dtype, c.key, c.original_type) for c in it.cols]
    params = {"topic.prefix": "fullfillment", "dt.source.type": "mysql"}
    row = typed_row(it)
    for kinds, kw in ((["insert"], {}), (["insert"], {"snapshot": True}), (["update", "delete", "delete"], {})):
        b = abi.batch_from_rows(schema_of(table), it.names, [row] * len(kinds), it.schema, it.table, kinds=kinds)
        n = len(kinds)
        if n > 1:
            ob = abi.batch_from_rows(schema_of(table), ["pk"], [[["uint32", 1]], [["uint32", 1]], [["uint32", 7]]], it.schema, it.table)
            b.old_keys, b.old_present = ob.cols, np.array([True, True, False])   # the last delete has no OldKeys: `before` is the row's own values
        meta = ([5] * n, [3000000000123 + r for r in range(n)], [1700000000123456789] * n, ["gtid-%d" % r if r else "" for r in range(n)])
        got, want, rws = emit_both(tf, b, table, params, meta, **kw)
        assert_same(got, want, rws)
    v = got.messages()[-2][1]
    assert b'"file":"mysql-log.000003"' in v and b'"pos":125' in v and b'"gtid":"gtid-2"' in v and b'"db":"%s"' % it.schema.encode() in v and b'"before":{"DECIMAL_":"AIvQODU="' in v
    # other Go shapes of the same types, and what the reference fails on
    small = [("pk", "uint32", True, "mysql:int(10) unsigned"), ("f", "double", False, "mysql:float"), ("b", "string", False, "mysql:binary(5)"), ("bits", "string", False, "mysql:bit(16)"),
             ("b1", "string", False, "mysql:bit(1)"), ("t", "utf8", False, "mysql:time(3)"), ("d", "double", False, "mysql:decimal(5,2)"), ("y", "utf8", False, "mysql:year(4)"),
             ("dt", "timestamp", False, "mysql:datetime(6)"), ("ts", "timestamp", False, "mysql:timestamp(3)"), ("e", "utf8", False, "mysql:enum('a','b''c')")]
    rows = [[["uint32", 4000000000], ["float32", 1.1], ["bytes", b"\x9f"], ["bytes", b"\x00\x00\x00\x00\x00\x00\x01\x9f"], ["bytes", b"\x00\x00\x00\x00\x00\x00\x00\x01"], ["string", "23:59:59.999"],
             ["jsonnum", "-231.45"], ["string", "-5"], ["time", (-1, 999999000)], ["time", (1098181434, 123999000)], ["string", "a"]],
            [["uint32", 1], ["jsonnum", "1e-3"], ["bytes", b"12345"], ["bytes", b"\x01\x02"], ["string", "AAAAAAAAAAE="], ["string", "01:02:03,45"], ["nil", None], ["string", "+2024"],
             ["time", (0, 0)], ["time", (0, 0)], ["nil", None]]]
    for one in rows:   # (a column holds one Go type: the two rows are two batches)
        b = abi.batch_from_rows(schema_of(small), [t[0] for t in small], [one], "db", "t")
        got, want, rws = emit_both(tf, b, small, params, None)
        assert_same(got, want, rws)
    for col, val in (("t", ["string", "24:00:00.000"]), ("t", ["string", "04:05:06."]), ("y", ["string", "19x1"]), ("f", ["string", "1.5"]), ("pk", ["int64", 5])):
        bad = [list(r) for r in rows[:1]]
        bad[0][[t[0] for t in small].index(col)] = val
        with pytest.raises(tf.TfgpuError) as ei:
            tf.debezium_emit(abi.dbz_emit_options(params, schema_of(small)), tf.DeviceBatch.upload(abi.batch_from_rows(schema_of(small), [t[0] for t in small], bad, "db", "t")))
        assert ei.value.code == tf.ERR_INVALID and "colName: " + col in str(ei.value), (col, str(ei.value))


def test_pg_arrays_and_enums(tf):
    """pg arrays whose elements AddPg(intoArr) leaves as they are — the integer family (through json.Number.Int64()), the text family, booleans — with their "items" field
    descriptions (AddFieldDescr, fields_descr.go:71-96), and enum columns (Properties[pg:enum_all_values] → io.debezium.data.Enum); then the reference's array canon ChangeItem
    (pg/tests/testdata/emitter_vals_test__canon_change_item_arr.txt: the oracle reproduces TestPgArrByArrInsert's table) cut to those families"""
    table = [("id", "int32", True, "pg:integer"), ("ai", "any", False, "pg:integer[]"), ("ab", "any", False, "pg:bigint[]"), ("asi", "any", False, "pg:smallint[]"), ("at", "any", False, "pg:text[]"),
             ("av", "utf8", False, "pg:character varying(5)[]"), ("au", "any", False, "pg:uuid[]"), ("abl", "any", False, "pg:boolean[]"),
             ("ar", "any", False, "pg:real[]"), ("ad", "any", False, "pg:double precision[]"), ("ao", "any", False, "pg:oid[]"), ("an", "any", False, "pg:numeric[]"),
             ("an2", "any", False, "pg:numeric(7,2)[]"), ("ain", "any", False, "pg:inet[]"), ("adt", "any", False, "pg:date[]"), ("atz", "any", False, "pg:timestamp with time zone[]"),
             ("atm", "any", False, "pg:time(3) without time zone[]"), ("attz", "any", False, "pg:time with time zone[]"), ("aj", "any", False, "pg:jsonb[]"), ("ab1", "any", False, "pg:bit(1)[]"),
             ("ab8", "any", False, "pg:bit(8)[]"),
             ("mood", "utf8", False, "pg:mood", {"pg:enum_all_values": ["sad", "ok", "ha\"ppy", "süß"]})]
    rng = np.random.default_rng(31)
    rows = []
    for r in range(200):
        def pick(lst):
            return lst[int(rng.integers(0, len(lst)))]
        rows.append([["int32", r], ["json", pick(["[1,2]", "[]", "[-0,null,9223372036854775807]", "[-9223372036854775808]", "null", "[0]"])], ["json", pick(["[1,2]", "[null]"])], ["json", "[1,-2]"],
                     ["json", pick(['["a","<b>&",null]', '["q\\"uote","back\\\\slash","\\u2028"]', "[]", '[""]', '["юникод"]'])], ["json", '["varc","varc"]'],
                     ["json", '["a0eebc99-9c0b-4ef8-bb6d-6bb9bd380a11"]'], ["json", pick(["[true,false,null]", "[]"])],
                     ["json", pick(["[1.45e-10,1.5,null]", "[0]", "[]"])], ["json", pick(["[3.14e-100,-0.0,1e21]", "[null]"])], ["json", "[1,4294967295]"],
                     ["json", pick(['["1267650600228229401496703205376e0","12345e0",null]', "[1.50,-2]", '["-0.001"]'])], ["json", pick(['["12367e-2","0.005"]', "[123.675]"])],
                     ["json", pick(['["192.168.100.128/25","10.0.0.1/32"]', "[null]"])], ["json", pick(['["1999-01-08T00:00:00Z","2004-10-19 10:23:54-07"]', "[]"])],
                     ["json", pick(['["2004-10-19T12:23:54+04:00","2004-10-19T10:23:54.9Z"]', "[null]"])], ["json", pick(['["04:05:06","04:05:06.123456"]', "[]"])],
                     ["json", pick(['["00:51:02.746572-08","13:30:25-04"]', "[null]"])], ["json", pick(['[{"k":[1,"]"]},"s",12.5,null,true,[1,[2]]]', "[]"])], ["json", pick(['["1","0",1,true,null]', "[]"])],
                     ["json", pick(['["10101111","0"]', "[]"])], ["string", pick(["sad", "ok"])]])
        for k in range(1, 22):
            if rng.integers(0, 9) == 0:
                rows[-1][k] = ["nil", None]
    b = abi.batch_from_rows(schema_of(table), [t[0] for t in table], rows, "public", "arrs")
    for params in (PARAMS, dict(PARAMS, **{"dt.add.original.type.info": "true"})):
        got, want, rws = emit_both(tf, b, table, params, None)
        assert_same(got, want, rws)
    v = got.messages()[0][1]
    assert b'{"__dt_original_type_info":{"original_type":"pg:integer[]"},"field":"ai","items":{"__dt_original_type_info":{"original_type":"pg:integer"},"optional":true,"type":"int32"},"optional":true,"type":"array"}' in v
    assert b'"parameters":{"allowed":"sad,ok,ha\\"ppy,s\xc3\xbc\xc3\x9f"}' in v
    with open(os.path.join(os.path.dirname(__file__), "golden", "debezium_emitter", "pg_emitter_vals_test__canon_change_item_arr.txt"), "rb") as f:
        it = E.unmarshal_change_item(f.read())
    keep = [c for c in it.cols if "timestamp" not in c.original_type or "with time zone" in c.original_type]   # (timestamp TEXTS in an array are pgtype.Timestamp.Set's: host)
    assert len(keep) == len(it.cols) - 3 == 34
    table = [(c.name, c.dtype, c.key, c.original_type) for c in keep]
    vals = []
    for c in keep:
        g, x = it.values[it.names.index(c.name)]
        vals.append(["json", E.gomarshal([E.go_value(p) for p in x])] if g == "list" else [g, x])
    one = abi.batch_from_rows(schema_of(table), [c.name for c in keep], [vals], it.schema, it.table)
    got, want, rws = emit_both(tf, one, table, {"database.dbname": "pguser", "topic.prefix": "fullfillment", "dt.source.type": "pg"}, None)
    assert_same(got, want, rws)
    base = [("id", "int32", True, "pg:integer")]
    for t, v, code in (("pg:integer[]", "[1.5]", tf.ERR_INVALID), ("pg:integer[]", '["1"]', tf.ERR_INVALID), ("pg:integer[]", "[[1]]", tf.ERR_INVALID), ("pg:text[]", "[1]", tf.ERR_UNSUPPORTED),
                       ("pg:integer[]", '{"a":1}', tf.ERR_UNSUPPORTED), ("pg:timestamp without time zone[]", '["2004-10-19T10:23:54Z"]', tf.ERR_UNSUPPORTED), ("pg:integer[][]", "[[1]]", tf.ERR_UNSUPPORTED),
                       ("pg:real[]", '["1.5"]', tf.ERR_INVALID), ("pg:date[]", "[1]", tf.ERR_INVALID), ("pg:date[]", '["1999-01-08"]', tf.ERR_UNSUPPORTED), ("pg:inet[]", '[{"IP":"1.2.3.4"}]', tf.ERR_UNSUPPORTED),
                       ("pg:numeric[]", '[{"Int":5,"Exp":0}]', tf.ERR_UNSUPPORTED)):
        bb = abi.batch_from_rows(schema_of(base + [("x", "any", False, t)]), ["id", "x"], [[["int32", 1], ["json", v]]], "public", "t")
        with pytest.raises(tf.TfgpuError) as ei:
            tf.debezium_emit(abi.dbz_emit_options(PARAMS, schema_of(base + [("x", "any", False, t)])), tf.DeviceBatch.upload(bb))
        assert ei.value.code == code, (t, v, str(ei.value))


def test_mysql_random_rows_with_updates_and_deletes(tf):
    """31 MySQL-typed columns of random values in the Go shapes a MySQL source produces, as inserts, updates and deletes with and without OldKeys: `before` of a delete takes the
    row's own values under the OldKeys (valPayload, emitter_value_converter.go:463-486), `source` carries file / pos / gtid"""
    import random
    rnd = random.Random(3)
    table = [("pk", "uint32", True, "mysql:int(10) unsigned"), ("bi", "int64", False, "mysql:bigint(20)"), ("bu", "uint64", False, "mysql:bigint(20) unsigned"), ("ti", "int8", False, "mysql:tinyint(4)"),
             ("t1", "int8", False, "mysql:tinyint(1)"), ("si", "uint16", False, "mysql:smallint(5) unsigned"), ("mi", "int32", False, "mysql:mediumint(9)"), ("f", "double", False, "mysql:float"),
             ("d", "double", False, "mysql:double(10,2)"), ("c", "utf8", False, "mysql:varchar(5)"), ("b3", "string", False, "mysql:binary(3)"), ("b16", "string", False, "mysql:binary(16)"),
             ("vb", "string", False, "mysql:varbinary(5)"), ("bl", "string", False, "mysql:blob"), ("bit1", "string", False, "mysql:bit(1)"), ("bit9", "string", False, "mysql:bit(9)"),
             ("bit64", "string", False, "mysql:bit(64)"), ("j", "any", False, "mysql:json"), ("ts", "timestamp", False, "mysql:timestamp"), ("ts4", "timestamp", False, "mysql:timestamp(4)"),
             ("dt", "timestamp", False, "mysql:datetime"), ("dt2", "timestamp", False, "mysql:datetime(2)"), ("dt6", "timestamp", False, "mysql:datetime(6)"), ("da", "date", False, "mysql:date"),
             ("tm", "utf8", False, "mysql:time"), ("tm3", "utf8", False, "mysql:time(3)"), ("dec", "double", False, "mysql:decimal(12,4)"), ("y", "utf8", False, "mysql:year(4)"),
             ("e", "utf8", False, "mysql:enum('a','b')"), ("s", "utf8", False, "mysql:set('x','y','z')"), ("tx", "utf8", False, "mysql:longtext")]
    n = 400
    rows = []

    def rb(k):
        return bytes(rnd.randrange(256) for _ in range(k))
    for r in range(n):
        sec, us = rnd.randint(-2 * 10 ** 9, 4 * 10 ** 9), rnd.randint(0, 999999)
        f = rnd.gauss(0, 1) * 10.0 ** rnd.randint(-10, 15)
        row = [["uint32", rnd.randint(0, 2 ** 32 - 1)], ["int64", rnd.randint(-2 ** 63, 2 ** 63 - 1)], ["uint64", rnd.randint(0, 2 ** 64 - 1)], ["int8", rnd.randint(-128, 127)], ["int8", rnd.choice([0, 1, 1, 2, -1])],
               ["uint16", rnd.randint(0, 65535)], ["int32", rnd.randint(-2 ** 23, 2 ** 23)], ["jsonnum", "%.7g" % f], ["float64", f], ["string", TEXTS[rnd.randrange(7)][:20]],
               ["bytes", rb(rnd.randint(0, 3))], ["bytes", rb(rnd.randint(0, 20))], ["bytes", rb(rnd.randint(0, 5))], ["bytes", rb(rnd.randint(0, 40))],
               ["bytes", rnd.choice([b"\x01", b"\x00", b"\x00" * 7 + b"\x01", b"\x02", b"\x00" * 8])], ["bytes", rb(rnd.choice([2, 8]))], ["bytes", rb(8)],
               ["json", rnd.choice(['{"k1":"v1"}', "[1,2]", '"s"', "null", "7"])], ["time", (sec, 0)], ["time", (sec, us * 1000)], ["time", (sec, us * 1000)], ["time", (sec, us * 1000)],
               ["time", (sec, us * 1000)], ["time", (sec - sec % 86400, 0)], ["string", "%02d:%02d:%02d" % (rnd.randint(0, 23), rnd.randint(0, 59), rnd.randint(0, 59))],
               ["string", "%02d:%02d:%02d.%03d" % (rnd.randint(0, 23), rnd.randint(0, 59), rnd.randint(0, 59), rnd.randint(0, 999))],
               ["jsonnum", "%d.%04d" % (rnd.randint(-10 ** 7, 10 ** 7), rnd.randint(0, 9999))], ["string", str(rnd.randint(1901, 2155))], ["string", rnd.choice("ab")], ["string", rnd.choice(["x", "x,y", ""])],
               ["string", TEXTS[rnd.randrange(8)]]]
        for k in range(1, len(row)):
            if rnd.random() < 0.08:
                row[k] = ["nil", None]
        rows.append(row)
    kinds = [rnd.choice(["insert", "update", "delete"]) for _ in range(n)]
    b = abi.batch_from_rows(schema_of(table), [t[0] for t in table], rows, "db1", "tbl", kinds=kinds)
    ob = abi.batch_from_rows(schema_of(table), ["pk"], [[r[0]] for r in rows], "db1", "tbl")
    b.old_keys, b.old_present = ob.cols, np.array([k != "insert" and rnd.random() < 0.8 for k in kinds])
    meta = (list(range(n)), [rnd.randint(0, 10 ** 15) for _ in range(n)], [rnd.randint(0, 2 ** 62) for _ in range(n)], [rnd.choice(["", "gtid:%d" % r]) for r in range(n)])
    got, want, rws = emit_both(tf, b, table, {"topic.prefix": "p", "dt.source.type": "mysql"}, meta)
    assert_same(got, want, rws)
    assert any(v is not None and b'"op":"d"' in v and b'"before":{"b16":"' in v for _k, v in got.messages())


def test_decimal_texts_through_every_numeric_converter(tf):
    """DecimalToDebezium / DecimalToDebeziumPrimitives (typeutil/helpers.go:269-436) over random decimal texts — leading zeros, signs, exponents, long fractions — for pg:numeric (variable
    scale), numeric(p,s) with and without a scale, money, ydb:Decimal / DyNumber and mysql:decimal.  Texts the reference fails on (or panics on: a zero spelt with '+') are found by the
    oracle first and replaced: the device's answer to those is checked in the refusal tests"""
    import random
    rnd = random.Random(5)

    def num():
        digs = "".join(rnd.choice("0123456789") for _ in range(rnd.randint(1, 12)))
        frac = "".join(rnd.choice("0123456789") for _ in range(rnd.randint(0, 8)))
        s = digs + ("." + frac if frac and rnd.random() < 0.7 else "")
        if rnd.random() < 0.15:
            s = "0" * rnd.randint(1, 3) + s
        if rnd.random() < 0.3:
            s = "-" + s
        elif rnd.random() < 0.05:
            s = "+" + s
        if rnd.random() < 0.3:
            s += rnd.choice("eE") + rnd.choice(["", "+", "-"]) + str(rnd.randint(0, 6))
        if rnd.random() < 0.03:
            s = rnd.choice(["0", "-0", "0.000", "-0.0", ".5", "5.", "-.5e1", "1e0", "00", "-"])
        return s
    table = [("id", "int32", True, "pg:integer"), ("a", "double", False, "pg:numeric"), ("b", "double", False, "pg:numeric(20,4)"), ("c", "double", False, "pg:numeric(9,0)"), ("m", "utf8", False, "pg:money"),
             ("y", "utf8", False, "ydb:Decimal"), ("z", "double", False, "ydb:DyNumber"), ("q", "double", False, "mysql:decimal(10,2)")]
    n = 1500
    rows = [[["int32", r], ["jsonnum", num()], ["jsonnum", num()], ["jsonnum", num()], ["string", "$" + num()], ["string", num()], ["jsonnum", num()], ["jsonnum", num()]] for r in range(n)]
    em, cols, replaced = E.Emitter(PARAMS), cols_of(table), 0
    for row in rows:
        for ci in range(1, len(table)):
            try:
                em.add(cols[ci], tuple(row[ci]))
            except (E.EmitError, E.NotRestated):
                row[ci] = [row[ci][0], "$1.5" if table[ci][0] == "m" else "12.5"]
                replaced += 1
    assert replaced < n   # most texts are good ones
    b = abi.batch_from_rows(schema_of(table), [t[0] for t in table], rows, "public", "nums")
    for params in (PARAMS, dict(PARAMS, **{"decimal.handling.mode": "string"})):
        got, want, rws = emit_both(tf, b, table, params, None)
        assert_same(got, want, rws)


def test_text_converters_on_random_and_damaged_texts(tf):
    """the converters that READ text (clock and zone texts, intervals, ranges, points, xml escapes, bit strings, dates) on random well-formed texts — byte for byte against the
    oracle — and on damaged ones, one call each: where the oracle says the reference fails, the device fails the call or leaves it to the host, never emits; where the oracle
    says "pgtype's business" (NotRestated), the device leaves the value to the host"""
    import random
    rnd = random.Random(11)

    def clock():
        return "%02d:%02d:%02d" % (rnd.randint(0, 23), rnd.randint(0, 59), rnd.randint(0, 59)) + ("." + "".join(rnd.choice("0123456789") for _ in range(rnd.randint(1, 6))) if rnd.random() < 0.6 else "")

    def zone():
        return rnd.choice(["+", "-"]) + "%02d" % rnd.randint(0, 14) + (":%02d" % rnd.choice([0, 30, 45]) if rnd.random() < 0.4 else "")

    def date():
        return "%04d-%02d-%02d" % (rnd.randint(1, 9999), rnd.randint(1, 12), rnd.randint(1, 28))

    def stamp():
        return date() + " " + clock().split(".")[0] + ("." + str(rnd.randint(0, 999999)) if rnd.random() < 0.3 else "") + rnd.choice(["Z", zone()])
    gens = {
        "pg:time(3) without time zone": clock, "pg:time without time zone": clock, "pg:time with time zone": lambda: clock() + zone(),
        "pg:interval": lambda: (rnd.choice(["", "%d year%s " % (rnd.randint(-3, 30), rnd.choice(["", "s"]))]) + rnd.choice(["", "%d mon%s " % (rnd.randint(-11, 20), rnd.choice(["", "s", "th", "ths"]))]) +
                                rnd.choice(["", "%d day%s " % (rnd.randint(-40, 400), rnd.choice(["", "s"]))]) + rnd.choice(["", rnd.choice(["", "-", "+"]) + clock()])),
        "pg:tstzrange": lambda: rnd.choice("[(") + rnd.choice(['"%s"', "%s"]) % stamp() + "," + rnd.choice(['"%s"', "%s"]) % stamp() + rnd.choice("])"),
        "pg:tsrange": lambda: rnd.choice("[(") + rnd.choice(['"%s"', "%s"]) % stamp()[:19] + "," + stamp()[:19] + rnd.choice("])"),
        "pg:numrange": lambda: rnd.choice("[(") + NUMS[rnd.randrange(26)] + "," + NUMS[rnd.randrange(26)] + rnd.choice("])"),
        "pg:point": lambda: "(%s,%s)" % (rnd.choice(["1.5", "-0", "1e10", "23.4", "+7", "1E3"]), rnd.choice(["2", "-44.5", "1e-7", ".5"])),
        "pg:xml": lambda: "".join(rnd.choice(["<a>", "\\u003c", "\\u00e9", "\\u12", "\\uZZZZ", "x", "é", "\\", "\\u0041", '"']) for _ in range(rnd.randint(0, 8))),
        "pg:bit varying(64)": lambda: "".join(rnd.choice("01") for _ in range(rnd.randint(0, 40))),
        "pg:date": lambda: date() + rnd.choice(["T00:00:00Z", " 00:00:00Z", "T00:00:00" + zone()[:3] + ":00", " 00:00:00" + zone()[:3]]),
        "pg:timestamp with time zone": lambda: rnd.choice([lambda: date() + "T" + clock() + "Z", lambda: date() + " " + clock() + zone()[:3], lambda: date() + "T" + clock() + zone()[:3] + ":30"])(),
        "mysql:time(6)": clock, "mysql:year(4)": lambda: str(rnd.randint(1901, 2155)),
    }
    types = list(gens)
    table = [("id", "int32", True, "pg:integer")] + [("c%d" % i, "utf8", False, t) for i, t in enumerate(types)]
    cols, em = cols_of(table), E.Emitter(PARAMS)

    def oracle_says(ci, text):
        try:
            E.gomarshal(em.add(cols[ci], ("string", text.encode("utf-8", "surrogateescape"))))
            return "ok"
        except E.NotRestated:
            return "host"
        except E.EmitError:
            return "invalid"
    rows = []
    for r in range(600):
        row = [["int32", r]]
        for ci, t in enumerate(types, 1):
            v = gens[t]()
            row.append(["string", v] if oracle_says(ci, v) == "ok" else ["nil", None])
        rows.append(row)
    assert sum(1 for row in rows for c in row[1:] if c[0] == "string") > 0.8 * 600 * len(types)
    b = abi.batch_from_rows(schema_of(table), [t[0] for t in table], rows, "public", "txt")
    got, want, rws = emit_both(tf, b, table, PARAMS, None)
    assert_same(got, want, rws)

    def damaged(s_):
        if not s_:
            return s_ + rnd.choice("0:.-+ ,Z\"[")
        i = rnd.randrange(len(s_))
        k = rnd.random()
        return s_[:i] + s_[i + 1:] if k < 0.35 else s_[:i] + rnd.choice("0123456789:.-+ ,Zz\"[](){}eT/\\u") + (s_[i:] if k < 0.7 else s_[i + 1:])
    seen = {"invalid": 0, "host": 0, "ok": 0}
    for k in range(260):
        ci = 1 + k % len(types)
        t, v = types[ci - 1], damaged(gens[types[ci - 1]]())
        verdict = oracle_says(ci, v)
        one = [("id", "int32", True, "pg:integer"), ("x", "utf8", False, t)]
        bb = abi.batch_from_rows(schema_of(one), ["id", "x"], [[["int32", 1], ["string", v]]], "public", "t")
        try:
            out = tf.debezium_emit(abi.dbz_emit_options(PARAMS, schema_of(one)), tf.DeviceBatch.upload(bb))
            assert verdict == "ok", (t, v, verdict)   # the device emits only what the oracle can check …
            item = E.Item("insert", "public", "t", cols_of(one), ["id", "x"], [("int32", 1), ("string", v.encode("utf-8", "surrogateescape"))])
            assert out.messages() == em.emit_kv(item), (t, v)   # … and then the same bytes
        except tf.TfgpuError as e:
            assert (verdict, e.code) in (("invalid", tf.ERR_INVALID), ("invalid", tf.ERR_UNSUPPORTED), ("host", tf.ERR_UNSUPPORTED), ("ok", tf.ERR_UNSUPPORTED)), (t, v, verdict, e.code)
        seen[verdict] += 1
    assert seen["invalid"] > 20 and seen["host"] > 5


def test_row_range_shards_emit_the_same_messages(tf):
    """SURVEY §8(e) for this path: rows are independent, so a batch cut into row ranges (tfgpu_shard_rows, one shard per rank / device / lane) emits, shard by shard, exactly the messages
    of the whole batch in order — no data-path collective; a shard's row meta is the slice of its rows (tfgpu_shard_rows reports each shard's first row)"""
    table, b = crud_batch(700, 19)
    n = b.nrows
    meta = (list(range(n)), [7 * r for r in range(n)], [1600000000000000000 + 999983 * r for r in range(n)])
    opts = abi.dbz_emit_options(PARAMS, schema_of(table))
    rm = abi.row_meta(n, ids=meta[0], lsns=meta[1], commit_times=meta[2])
    db = tf.DeviceBatch.upload(b)
    whole = tf.debezium_emit(opts, db, rm)
    want = whole.messages()
    for g in (2, 3, 8):
        shards, row0 = db.shard_rows(g, lanes=[0] * g)
        got, rows = [], []
        for sh, r0 in zip(shards, row0):
            k = sh.nrows
            out = tf.debezium_emit(opts, sh, abi.row_meta(k, ids=meta[0][r0:r0 + k], lsns=meta[1][r0:r0 + k], commit_times=meta[2][r0:r0 + k]))
            got += out.messages()
            rows += [int(x) + r0 for x in out.msg_row]
        assert got == want and rows == [int(x) for x in whole.msg_row], g


def test_emit_then_receive_round_trip(tf):
    """a size-independent property: what the emitter writes, the Debezium RECEIVER of this library (tf_debezium.hip, pinned to the reference's receiver canon on its own) reads back
    as the same rows — kinds, values, OldKeys of updates and deletes, ID / LSN / CommitTime at Debezium's millisecond accuracy, table id, key flags — although the two halves
    were built and pinned apart (the reference's emitter_chain tests make the same trip, pkg/debezium/pg/tests/emitter_chain_test.go)"""
    from decimal import Decimal
    from transferia_amd import debezium
    table = [("id", "int32", True, "pg:integer"), ("big", "int64", False, "pg:bigint"), ("t", "utf8", False, "pg:text"), ("b", "boolean", False, "pg:boolean"), ("d", "double", False, "pg:double precision"),
             ("si", "int16", False, "pg:smallint"), ("n", "double", False, "pg:numeric(10,2)")]
    rng = np.random.default_rng(77)
    n = 3000
    rows, kinds = [], []
    for r in range(n):
        rows.append([["int32", r], ["int64", int(rng.integers(-2 ** 62, 2 ** 62))], ["string", TEXTS[int(rng.integers(0, len(TEXTS) - 1))] + str(r)], ["bool", bool(rng.integers(0, 2))],
                     ["float64", float(rng.standard_normal()) * 10.0 ** int(rng.integers(-8, 12))], ["int16", int(rng.integers(-2 ** 15, 2 ** 15))],
                     ["jsonnum", "%d.%02d" % (rng.integers(-10 ** 6, 10 ** 6), rng.integers(0, 100))]])
        for k in range(1, 7):
            if rng.integers(0, 11) == 0:
                rows[-1][k] = ["nil", None]
        kinds.append(["insert", "update", "delete"][int(rng.integers(0, 3))])
    names = [t[0] for t in table]
    b = abi.batch_from_rows(schema_of(table), names, rows, "public", "rt", kinds=kinds)
    ob = abi.batch_from_rows(schema_of(table), ["id"], [[r[0]] for r in rows], "public", "rt")
    b.old_keys, b.old_present = ob.cols, np.array([k != "insert" for k in kinds])
    ids, lsns, cts = [int(x) for x in rng.integers(0, 2 ** 31, n)], [int(x) for x in rng.integers(0, 2 ** 53, n)], [1700000000000000000 + 1000000 * int(x) for x in rng.integers(0, 10 ** 9, n)]
    out = tf.debezium_emit(abi.dbz_emit_options({"database.dbname": "db", "topic.prefix": "srv", "dt.source.type": "pg"}, schema_of(table)), tf.DeviceBatch.upload(b),
                           abi.row_meta(n, ids=ids, lsns=lsns, commit_times=cts))
    msgs = out.messages()
    vals = [v for _k, v in msgs if v is not None]
    src = [int(out.msg_row[i]) for i, (_k, v) in enumerate(msgs) if v is not None]
    assert len(vals) == n and src == list(range(n)) and len(msgs) == n + kinds.count("delete")   # a tombstone behind every delete
    data, cm = abi.messages(vals)
    parsed, errors = debezium.Parser(tf).parse(data, cm, host_bytes=data)
    assert not errors and len(parsed) == 1
    pr = parsed[0]
    hb = pr.batch.download()
    assert hb.nrows == n and (hb.table_ns, hb.table_name) == ("public", "rt") and [c.name for c in hb.cols] == names
    assert [c.key for c in pr.schema.cols] == [t[2] for t in table]
    assert [int(k) for k in hb.kind] == [{"insert": abi.K_INSERT, "update": abi.K_UPDATE, "delete": abi.K_DELETE}[k] for k in kinds]
    assert [int(x) for x in pr.rows["id"]] == ids and [int(x) for x in pr.rows["lsn"]] == lsns and [int(x) for x in pr.rows["commit_time"]] == cts
    assert [bool(x) for x in hb.old_present] == [k != "insert" for k in kinds] and [c.name for c in hb.old_keys] == ["id"]
    for r in range(n):
        if kinds[r] != "insert":
            assert hb.old_keys[0].pyvalue(r) == ["int32", r]
        for ci, c in enumerate(hb.cols):
            got, want = c.pyvalue(r), rows[r][ci]
            if kinds[r] == "delete":
                assert got[0] == "nil", (r, c.name, got)   # a delete's values travel in `before`: OldKeys
            elif want[0] == "nil":
                assert got[0] == "nil", (r, c.name, got)
            elif c.name == "n":
                assert Decimal(got[1].decode()) == Decimal(want[1]), (r, got, want)   # Decimal bytes back to a numeric text
            elif c.name == "t":
                assert got == ["string", want[1].encode("utf-8", "surrogateescape") if isinstance(want[1], str) else want[1]], (r, got, want)
            else:
                assert got == want, (r, c.name, got, want)


GOLD = os.path.join(os.path.dirname(__file__), "golden", "debezium_emitter")
DEVICE_TYPES = ("pg:xml", "pg:point", "pg:numrange", "pg:tsrange", "pg:tstzrange", "pg:money", "pg:USER-DEFINED:hstore", "pg:boolean", "pg:bit(1)", "pg:smallint", "pg:integer", "pg:bigint", "pg:oid", "pg:real", "pg:double precision", "pg:bytea", "pg:json", "pg:jsonb", "pg:uuid",
                "pg:inet", "pg:int4range", "pg:int8range", "pg:daterange", "pg:text", "pg:date", "pg:cidr", "pg:macaddr", "pg:USER-DEFINED:citext")


def device_resident(t):
    return (t in DEVICE_TYPES or t.startswith("pg:character") or t.startswith("pg:bit(") or t.startswith("pg:bit varying(") or t.startswith("pg:timestamp") or t.startswith("pg:time") or t.startswith("pg:interval") or E.is_pg_numeric(t))


@pytest.mark.parametrize("name", ["insert", "update0", "update1", "update2", "delete", "ri_update", "ri_delete"])
def test_the_references_crud_fixtures_cut_to_the_device_columns(tf, name):
    """the fixtures the oracle is pinned on (test_dbz_emitter_oracle.py), cut to the device-resident columns, timestamps as UTC instants
    (what a Postgres source produces; the fixtures' +04:00 wall clocks are an artefact of their JSON round trip)"""
    from test_dbz_emitter_oracle import read
    if name.startswith("ri_"):   # REPLICA IDENTITY FULL: OldKeys hold every column (emitter_replica_identity__canon_change_item_*.txt)
        with open(os.path.join(GOLD, "emitter_replica_identity__canon_change_item_%s.txt" % name[3:]), "rb") as f:
            it = E.unmarshal_change_item(f.read())
    else:
        it = E.unmarshal_change_item(read(name))
    keep = [c for c in it.cols if device_resident(c.original_type)]
    assert len(keep) == len(it.cols) == (2 if name.startswith("ri_") else 59)   # every column of the fixtures' tables
    kn = {c.name for c in keep}
    table = [(c.name, c.dtype, c.key, c.original_type) for c in keep]

    def typed(v):
        g, x = v
        if g == "time":
            return ["time", (x[0] + (x[2] if len(x) > 2 else 0), x[1])]
        if g in ("string", "bytes", "jsonnum", "json"):
            return [g, bytes(x)]
        return [g, x]
    names = [nm for nm in it.names if nm in kn]
    if not names:   # the delete fixture carries no column values: every device column as nil keeps the batch's shape
        names = [c.name for c in keep]
        vals = [["nil", None] for _ in names]
    else:
        vals = [typed(v) for nm, v in zip(it.names, it.values) if nm in kn]
    b = abi.batch_from_rows(schema_of(table), names, [vals], it.schema, it.table, kinds=[it.kind])
    if it.old_names:
        ob = abi.batch_from_rows(schema_of(table), it.old_names, [[typed(v) for v in it.old_values]], it.schema, it.table)
        b.old_keys, b.old_present = ob.cols, np.array([True])
    got, want, rows = emit_both(tf, b, table, PARAMS, ([it.id], [it.lsn], [it.commit_time]))
    assert_same(got, want, rows)
    assert len(want) == {"insert": 1, "update0": 1, "update1": 1, "update2": 3, "delete": 2, "ri_update": 1, "ri_delete": 2}[name]
    if name == "ri_update":
        assert isinstance(json.loads(got.messages()[0][1])["payload"]["before"], dict)   # the old row, not null


def test_the_references_crud_fixtures_as_one_batch_with_absent_cells(tf):
    """The insert and the three updates of the reference's CRUD fixtures in ONE device batch over the table's 59 columns: the items that list fewer
    columns (update1 / update2: unchanged TOASTed values left out by the source) say so through tfgpu_column.absent, and every message — the
    `__debezium_unavailable_value` members of update1's and update2's values included — is the oracle's, which test_dbz_emitter_oracle.py pins
    to the reference's own canon bytes (emitter_crud_test__debezium_update1val.txt, …update2val2.txt)."""
    from test_dbz_emitter_oracle import read
    items = [E.unmarshal_change_item(read(nm)) for nm in ("insert", "update0", "update1", "update2")]
    cols = items[0].cols
    assert all([c.name for c in it.cols] == [c.name for c in cols] for it in items) and all(device_resident(c.original_type) for c in cols)
    table = [(c.name, c.dtype, c.key, c.original_type) for c in cols]
    names = [c.name for c in cols]

    def typed(v):
        g, x = v
        if g == "time":
            return ["time", (x[0] + (x[2] if len(x) > 2 else 0), x[1])]
        if g in ("string", "bytes", "jsonnum", "json"):
            return [g, bytes(x)]
        return [g, x]
    rows = [[typed(dict(zip(it.names, it.values))[nm]) if nm in it.names else ["nil", None] for nm in names] for it in items]
    b = abi.batch_from_rows(schema_of(table), names, rows, items[0].schema, items[0].table, kinds=[it.kind for it in items])
    partial = 0
    for c in b.cols:
        ab = np.array([c.name not in it.names for it in items], bool)
        if ab.any():
            c.absent = ab
            c.validity = (np.ones(len(items), bool) if c.validity is None else c.validity) & ~ab
            partial += 1
    assert partial >= 1 and [len(it.names) < len(names) for it in items] == [False, False, True, True]
    onames = next(it.old_names for it in items if it.old_names)
    assert all((not it.old_names) or it.old_names == onames for it in items)
    ob = abi.batch_from_rows(schema_of(table), onames, [[typed(v) for v in it.old_values] if it.old_names else [["nil", None]] * len(onames) for it in items], items[0].schema, items[0].table)
    b.old_keys, b.old_present = ob.cols, np.array([bool(it.old_names) for it in items])
    meta = ([it.id for it in items], [it.lsn for it in items], [it.commit_time for it in items])
    got, want, rws = emit_both(tf, b, table, PARAMS, meta)
    assert_same(got, want, rws)
    assert len(want) == 1 + 1 + 1 + 3
    assert sum(b"__debezium_unavailable_value" in v for _k, v in got.messages() if v) >= 2


def test_what_stays_with_the_stock_emitter_is_refused_by_name(tf):
    def call(table, rows, params=PARAMS, **kw):
        b = abi.batch_from_rows(schema_of(table), [t[0] for t in table], rows, "public", "t")
        return tf.debezium_emit(abi.dbz_emit_options(params, schema_of(table), **kw), tf.DeviceBatch.upload(b))
    base = [("id", "int32", True, "pg:integer")]
    for t in ("pg:timestamp without time zone[]", "oracle:NUMBER"):
        with pytest.raises(tf.TfgpuError) as ei:
            call(base + [("x", "utf8", False, t)], [[["int32", 1], ["string", "1"]]])
        assert ei.value.code == tf.ERR_UNSUPPORTED and "column x" in str(ei.value) and "stock emitter" in str(ei.value), str(ei.value)
    for params in ({"key.converter.schema.registry.url": "http://sr:8081"}, {"value.converter.ysr.namespace.id": "ns"}, {"dt.source.type": "oracle"},
                   {"decimal.handling.mode": "double"}):
        with pytest.raises(tf.TfgpuError) as ei:
            call(base + [("x", "double", False, "pg:numeric")], [[["int32", 1], ["jsonnum", "1"]]], dict(PARAMS, **params))
        assert ei.value.code == tf.ERR_UNSUPPORTED, str(ei.value)
    # where the reference itself returns an error
    for table, row, word in [(base + [("x", "utf8", False, "pg:some_unknown_type")], [["int32", 1], ["string", "1"]], "unknown pgType"),
                             (base + [("x", "utf8", False, "")], [["int32", 1], ["string", "1"]], "unknown source type"),
                             (base + [("x", "int32", False, "pg:integer")], [["int32", 1], ["jsonnum", "1.5"]], "unable to emit value, colName: x"),
                             (base + [("x", "int32", False, "pg:integer")], [["int32", 1], ["string", "1"]], "unable to emit value, colName: x"),
                             (base + [("x", "double", False, "pg:numeric")], [["int32", 1], ["string", "NaN"]], "unable to emit value, colName: x"),
                             (base + [("x", "double", False, "pg:real")], [["int32", 1], ["float64", float("nan")]], "unable to emit value, colName: x")]:
        with pytest.raises(tf.TfgpuError) as ei:
            call(table, [row])
        assert ei.value.code == tf.ERR_INVALID and word in str(ei.value), str(ei.value)
    # values whose text is a third-party parser's business (pgtype), and hstore given as text (HstoreToJSON): the host's
    for t, v in (("pg:time without time zone", "4:05:06"), ("pg:time with time zone", "04:05:06Z"), ("pg:numrange", "empty"), ("pg:numrange", "[,5)"), ("pg:tstzrange", "[-infinity,2010-01-01 00:00:00Z)"),
                 ("pg:USER-DEFINED:hstore", '"a"=>"1"'), ("pg:tsrange", "x")):
        with pytest.raises(tf.TfgpuError) as ei:
            call(base + [("x", "utf8", False, t)], [[["int32", 1], ["string", v]]])
        assert ei.value.code == tf.ERR_UNSUPPORTED and "column x" in str(ei.value), (t, v, str(ei.value))
    for t, v in (("pg:numeric", "+0.0"), ("pg:money", "$+0")):   # DecimalToDebeziumPrimitivesImpl indexes an empty byte slice: the reference panics
        with pytest.raises(tf.TfgpuError) as ei:
            call(base + [("x", "utf8", False, t)], [[["int32", 1], ["string", v]]])
        assert ei.value.code == tf.ERR_UNSUPPORTED and "column x" in str(ei.value), (t, v, str(ei.value))
    for t, v in (("pg:point", "(1,2,3)"), ("pg:point", "(a,b)"), ("pg:money", "$1,000.00"), ("pg:money", "$1e3"), ("pg:numrange", "[1ee2,3)"), ("pg:interval", "one day"), ("pg:interval", "1 day 01:02"),
                 ("pg:interval", "1 day 01:02:03.")):
        with pytest.raises(tf.TfgpuError) as ei:
            call(base + [("x", "utf8", False, t)], [[["int32", 1], ["string", v]]])
        assert ei.value.code == tf.ERR_INVALID and "colName: x" in str(ei.value), (t, v, str(ei.value))
    with pytest.raises(tf.TfgpuError) as ei:   # past 128 bits: not decided on the device
        call(base + [("x", "double", False, "pg:numeric")], [[["int32", 1], ["jsonnum", "4" + "0" * 38]]])
    assert ei.value.code == tf.ERR_UNSUPPORTED and "column x" in str(ei.value)
    with pytest.raises(tf.TfgpuError) as ei:   # "unsupported interval.handling.mode" (ParsePostgresInterval)
        call(base + [("x", "utf8", False, "pg:interval")], [[["int32", 1], ["string", "1 day"]]], dict(PARAMS, **{"interval.handling.mode": "string"}))
    assert ei.value.code == tf.ERR_INVALID and "colName: x" in str(ei.value)
    with pytest.raises(tf.TfgpuError) as ei:   # parameters.Validate
        call(base, [[["int32", 1]]], dict(PARAMS, **{"dt.batching.max.size": "1048576"}))
    assert ei.value.code == tf.ERR_INVALID and "dt.batching.max.size" in str(ei.value)
    with pytest.raises(tf.TfgpuError) as ei:   # a column the schema does not know
        b = abi.batch_from_rows(schema_of(base + [("y", "utf8", False, "pg:text")]), ["id", "y"], [[["int32", 1], ["string", "1"]]], "public", "t")
        tf.debezium_emit(abi.dbz_emit_options(PARAMS, schema_of(base)), tf.DeviceBatch.upload(b))
    assert ei.value.code == tf.ERR_INVALID and "column absent in schema: y" in str(ei.value)
