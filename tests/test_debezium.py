"""Debezium parser with inline schemas (the source of configs[4]): the oracle pinned to the reference's
canon (CPU), the HIP path against the oracle and the same canon (GPU)."""
import base64
import json
from decimal import Decimal

import numpy as np
import pytest

from transferia_amd import abi
from util import golden
import os as _os

SEED0 = int(_os.environ.get("TFGPU_TEST_SEED", "0"))  # 0 = the committed seeds; other values: soak runs (TFGPU_TEST_SEED=n bash tools/gpu_visit.sh tests TAG)


def canon_value(v):
    """One value of an oracle item ([gotype, v]) in the form the reference's canon JSON shows it."""
    g, x = v
    if g == "nil":
        return None
    if g == "bytes":
        return base64.b64encode(x).decode()
    if g in ("string", "json"):
        return x.decode("utf-8")
    if g == "jsonnum":
        return Decimal(x.decode())
    return x


def same(a, b):
    if isinstance(a, Decimal) or isinstance(b, Decimal):
        return Decimal(str(a)) == Decimal(str(b)) if not isinstance(a, Decimal) or not isinstance(b, Decimal) else a == b
    if isinstance(a, float) or isinstance(b, float):
        return a is not None and b is not None and float(a) == float(b)
    return a == b and type(a) is type(b)


def check_item(it, exp, ctx):
    assert it["kind"] == exp["kind"], ctx
    assert (it["ns"], it["table"]) == (exp["schema"], exp["table"]), ctx
    assert (it["id"], it["lsn"], it["commit_time"]) == (exp["id"], exp["lsn"], exp["commit_time"]), ctx
    assert [[c.name, c.dtype, c.key, c.table_schema, c.table_name, c.original_type] for c in it["schema"].cols] == exp["table_schema"], ctx
    if exp["names"] is None:
        assert it["names"] == [] and it["names_form"] == 1, ctx
    else:
        assert it["names"] == exp["names"] and it["names_form"] == 0, ctx
        for name, got, want in zip(it["names"], it["values"], exp["values"]):
            assert same(canon_value(got), want), (ctx, name, got, want)
    ok = exp["oldkeys"] or {}
    assert [o[0] for o in it["old"]] == (ok.get("keynames") or []), ctx
    assert [o[1][1] for o in it["old"]] == (ok.get("keyvalues") or []), ctx


def test_oracle_reference_vectors(oracle):
    for case in golden("debezium.json")["cases"]:
        data, msgs = abi.messages([case["message"].encode("utf-8")])
        items, codes = oracle.debezium_parse(data, msgs)
        if case["expect"] is None:
            assert items == [] and codes[0] != abi.ROW_OK, case["name"]
            continue
        assert codes == [abi.ROW_OK] and len(items) == 1, (case["name"], codes)
        check_item(items[0], case["expect"], case["name"])


# ---- a synthetic Postgres CDC stream in Debezium's envelope (the shape of configs[4]) ---------------------------------
FIELDS = [("id", "int64", False, None), ("ver", "int32", True, None), ("flag", "boolean", True, None), ("score", "double", True, None),
          ("name", "string", True, None), ("blob", "bytes", True, None),
          ("amount", "bytes", True, {"name": "org.apache.kafka.connect.data.Decimal", "parameters": {"scale": "2", "connect.decimal.precision": "10"}}),
          ("at", "int64", True, {"name": "io.debezium.time.MicroTimestamp"}), ("tz", "string", True, {"name": "io.debezium.time.ZonedTimestamp"}),
          ("pt", "struct", True, {"name": "io.debezium.data.geometry.Point", "fields": [{"type": "double", "optional": False, "field": "x"},
                                                                                       {"type": "double", "optional": False, "field": "y"}]}),
          ("num", "struct", True, {"name": "io.debezium.data.VariableScaleDecimal", "fields": [{"type": "int32", "optional": False, "field": "scale"},
                                                                                             {"type": "bytes", "optional": False, "field": "value"}]})]


def envelope_schema(table="events", fields=FIELDS):
    def struct(field):
        fs = []
        for n, t, opt, extra in fields:
            f = {"type": t, "optional": opt, "field": n}
            f.update(extra or {})
            fs.append(f)
        return {"type": "struct", "fields": fs, "optional": True, "name": "srv.public.%s.Value" % table, "field": field}
    src = {"type": "struct", "fields": [{"type": "string", "optional": False, "field": "version"}, {"type": "int64", "optional": False, "field": "ts_ms"}],
           "optional": False, "name": "io.debezium.connector.postgresql.Source", "field": "source"}
    return {"type": "struct", "fields": [struct("before"), struct("after"), src, {"type": "string", "optional": False, "field": "op"},
                                         {"type": "int64", "optional": True, "field": "ts_ms"}], "optional": False, "name": "srv.public.%s.Envelope" % table}


def twos(v: int) -> str:
    n = max(1, (v.bit_length() + 8) // 8)
    return base64.b64encode(v.to_bytes(n, "big", signed=True)).decode()


def cdc_payloads(rng, n, table="events", weird=True):
    """The payloads of n Debezium events over n/4 keys, one at a time (the caller shares `rng`): inserts / updates / deletes, nulls,
    escapes, every receiver of FIELDS; with `weird` also payloads the reference turns into `_unparsed` items or panics on."""
    for k in range(n):
        key = rng.randrange(max(n // 4, 1))
        def row():
            return {"id": key, "ver": None if rng.random() < .1 else k, "flag": rng.choice([True, False, None]),
                    "score": rng.choice([None, 0, -1.5e-7, 3.14e100, k / 7, 1e22]),
                    "name": rng.choice([None, "", "plain", "é中\\\"q\\n", "\\u00e9\\ud83d\\ude00 x", "__debezium_unavailable", 17]),
                    "blob": rng.choice([None, "", "yv66vg==", "AA==", base64.b64encode(bytes(rng.randrange(256) for _ in range(k % 23))).decode()]),
                    "amount": rng.choice([None, twos(12345), twos(-1), twos(0), twos(-12345678901234567890), twos(5), twos(10**40 + k)]),
                    "at": 1_600_000_000_000_000 + k, "tz": "2004-10-19T08:23:54Z",
                    "pt": rng.choice([None, {"x": 23.4, "y": -44.5, "wkb": "AQE=", "srid": None}, {"y": 1, "x": "s"}]),
                    "num": rng.choice([None, {"scale": 0, "value": twos(123456)}, {"scale": 3, "value": twos(-7)}, {"value": twos(99), "scale": 1}])}
        op = rng.choice(["c", "c", "u", "u", "d", "r"])
        payload = {"before": row() if op in ("u", "d") else None, "after": None if op == "d" else row(),
                   "source": {"version": "1.9", "connector": "postgresql", "name": "srv", "ts_ms": 1_700_000_000_000 + k, "snapshot": "false", "db": "db", "schema": "public",
                              "table": table, "txId": 500 + k, "lsn": 10_000 + 8 * k, "xmin": None}, "op": op, "ts_ms": 1_700_000_000_123 + k, "transaction": None}
        ps = json.dumps(payload, separators=(",", ":")).replace("\\\\\\\\u", "\\\\u").replace("\\\\u", "\\u").replace('\\\\\\"', '\\"').replace("\\\\n", "\\n")
        if weird and k % 11 == 5:
            ps = rng.choice([ps.replace('"op":"%s"' % op, '"op":"x"'), ps.replace('"op":"%s"' % op, '"op":5'), ps.replace('"lsn":', '"lsn":-'),
                             ps.replace('"txId":', '"txId":99999999999'), ps.replace('"ver":', '"Ver":'), ps.replace('"at":', '"at":1.5e3,"_":'),
                             ps.replace('"flag":true', '"flag":"true"').replace('"flag":false', '"flag":0'), ps.replace('"blob":"yv66vg=="', '"blob":"yv66vg="'),
                             ps.replace('"name":"plain"', '"name":"__debezium_unavailable_value"'), ps.replace('"score":0,', '"score":1e999,'),
                             ps.replace('"id":%d' % key, '"id":1.0'), ps.replace('"tz":"2004-10-19T08:23:54Z"', '"tz":null'),
                             ps.replace('"id":', '"\\u0069d":'), ps.replace('"ver":', '"v\\u0065r" :\t'), ps.replace('"score":0,', '"score":-0,'),
                             ps.replace('"score":0,', '"score":0.1e-400,'), ps.replace('"id":%d' % key, '"id":0%d' % key), ps.replace('"at":', '"at":-9223372036854775808,"at2":'),
                             ps.replace('"at":', '"at":9223372036854775808,"_":'), ps.replace('"source":{', '"source":{"lsn":"x",'), ps.replace('"source":{', '"source":null,"s2":{'),
                             ps.replace('"op":"%s"' % op, '"op":"\\u0063"'), ps.replace('"flag":null', '"flag":null,"flag":true')])
        yield k, ps


def cdc_messages(n, seed=3, table="events", weird=True):
    """The same events in the inline-schema envelope; with `weird` also the messages IncludeSchema.Unpack rejects."""
    import random
    rng = random.Random(seed)
    schema = json.dumps(envelope_schema(table), separators=(",", ":"))
    out = []
    for k, ps in cdc_payloads(rng, n, table, weird):
        if k % 3 == 0:
            msg = '{"schema":%s,"payload":%s}' % (schema, ps)
        elif k % 3 == 1:
            msg = '{"payload": %s ,\n "schema" : %s }' % (ps, schema)
        else:
            msg = '{"x":[1,{"schema":1}],"schema":0,"schema":%s,"payload":%s}' % (schema, ps)
        if weird and k % 17 == 9:
            msg = rng.choice(["", "{}", "null", "[]", msg[:-1], msg + "x", '{"payload":%s}' % ps, '{"schema":%s}' % schema, '{"schema":%s,"payload":null}' % schema,
                              '{"schema":null,"payload":%s}' % ps, '{"schema":7,"payload":%s}' % ps, "\x00\x00\x00\x00\x01" + msg, '{"Schema":%s,"payload":%s}' % (schema, ps),
                              '{"schema":%s,"payload":[1]}' % schema, '{"schema":%s,"pay\\u006coad":%s}' % (schema, ps), '{"schema":%s,"payload":%s} \n\t' % (schema, ps),
                              '{"schema":%s,"payload":%s,}' % (schema, ps), ' {"schema" :%s , "payload": %s}' % (schema, ps)])
        out.append(msg.encode("utf-8"))
    return out


def test_oracle_synthetic_stream_is_consistent(oracle):
    msgs = cdc_messages(400)
    data, m = abi.messages(msgs)
    items, codes = oracle.debezium_parse(data, m)
    assert len(items) == sum(c == abi.ROW_OK for c in codes)
    seen = set(codes)
    for c in (abi.ROW_OK, abi.ROW_DBZ_UNPACK, abi.ROW_DBZ_PAYLOAD, abi.ROW_DBZ_OP, abi.ROW_DBZ_SCHEMA, abi.ROW_DBZ_FIELD, abi.ROW_HOST_FALLBACK):
        assert c in seen, c
    kinds = {it["kind"] for it in items}
    assert kinds == {"insert", "update", "delete"}
    for it in items:
        if it["kind"] == "delete":
            assert it["names"] == [] and [o[0] for o in it["old"]] == ["id"] and it["names_form"] == 1
        else:
            assert it["names"] == [f[0] for f in FIELDS]
            assert ([o[0] for o in it["old"]] == ["id"]) == (it["kind"] == "update")
        assert [c.dtype for c in it["schema"].cols] == ["int64", "int32", "boolean", "double", "utf8", "string", "utf8", "int64", "utf8", "utf8", "double"]
    # known answers of the receivers
    one = json.dumps({"schema": envelope_schema(), "payload": {"before": None, "after": {
        "id": 1, "ver": 4294967297, "flag": True, "score": 1e-7, "name": 12.50, "blob": "yv66vg==", "amount": twos(-5), "at": 5, "tz": "z",
        "pt": {"x": 23.4, "y": None}, "num": {"scale": 2, "value": twos(10)}}, "source": {"schema": "s", "table": "t", "lsn": 7, "txId": 8, "ts_ms": 9}, "op": "c"}})
    data, m = abi.messages([one.encode()])
    items, codes = oracle.debezium_parse(data, m)
    assert codes == [abi.ROW_OK]
    vals = dict(zip(items[0]["names"], items[0]["values"]))
    assert vals["ver"] == ["int32", 1]                       # int32(4294967297): Go's truncating conversion
    assert vals["name"] == ["string", b"12.5"]               # a json.Number is taken by its text
    assert vals["blob"] == ["bytes", bytes.fromhex("cafebabe")]
    assert vals["amount"] == ["string", b"-0.05"]            # scale > len("5"): zero-padded to scale + 1 digits
    assert vals["pt"] == ["string", b"(23.4,<nil>)"]
    assert vals["num"] == ["jsonnum", b".10"]                # scale == len("10"): helpers.go:990 pads only when scale > len
    assert (items[0]["id"], items[0]["lsn"], items[0]["commit_time"]) == (8, 7, 9_000_000)


# ---- the HIP path -------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


def device_items(tf, parsed, rows=None):
    """Rows of a debezium.Parsed as oracle-style items (message index → item); `rows`: only the first so many (bench.py's parity legs)."""
    b = parsed.batch.download()
    out = {}
    names = [c.name for c in b.cols]
    for r in range(b.nrows if rows is None else min(rows, b.nrows)):
        kind = ["insert", "update", "delete"][int(b.kind[r])]
        row = parsed.rows[r]
        it = {"kind": kind, "ns": b.table_ns, "table": b.table_name, "schema": parsed.schema, "id": int(row["id"]), "lsn": int(row["lsn"]),
              "commit_time": int(row["commit_time"]), "names_form": int(row["names_form"]), "src": int(b.src_row[r])}
        assert int(row["msg"]) == int(b.src_row[r])
        if kind == "delete":
            assert not any(c.is_valid(r) for c in b.cols), "a Delete keeps ColumnValues nil"
            it["names"], it["values"] = [], []
        else:  # a row lists the columns it is not ABSENT from (`__debezium_unavailable_value`, receiver.go:98-105)
            listed = [c for c in b.cols if getattr(c, "absent", None) is None or not c.absent[r]]
            it["names"], it["values"] = [c.name for c in listed], [c.pyvalue(r) for c in listed]
        pres = b.old_present is None or bool(b.old_present[r])
        it["old"] = [[k.name, k.pyvalue(r)] for k in (b.old_keys or [])] if pres and b.old_keys else []
        assert pres == (kind != "insert")
        out[int(b.src_row[r])] = it
    return out


def expected_errors(codes, exp_items, msgs):
    """The oracle's per-message fates, plus the device's one-table-per-batch rule: an item of another TableID than the first good
    item of its schema (here: a `source` that is null, so Schema / Table are empty) is handed to the stock code."""
    out = {k: v for k, v in enumerate(codes) if v != abi.ROW_OK}
    groups = {}
    for it in exp_items:   # the device groups by the schema bytes; in these streams they differ by the table's name only
        groups.setdefault(b".orders." in msgs[it["src"]], []).append(it)
    for items in groups.values():
        ref = (items[0]["ns"], items[0]["table"])
        for i in items:
            if (i["ns"], i["table"]) != ref:
                out[i["src"]] = abi.ROW_HOST_FALLBACK
    return out


def norm_old(old):
    return [[n, (v[1] if not isinstance(v[1], (bytes, bytearray)) else bytes(v[1]))] for n, v in old]


def assert_same_items(got, exp, ctx=""):
    for k in ("kind", "ns", "table", "names", "id", "lsn", "commit_time", "names_form"):
        assert got[k] == exp[k], (ctx, k, got[k], exp[k])
    assert [[c.name, c.dtype, c.key, c.table_schema, c.table_name] for c in got["schema"].cols] == [[c.name, c.dtype, c.key, c.table_schema, c.table_name] for c in exp["schema"].cols], ctx
    for n, a, b in zip(got["names"], got["values"], exp["values"]):
        if a[0] in ("string", "bytes", "jsonnum"):
            a = [a[0], bytes(a[1])]
        assert a == b or (a[0] == b[0] == "float64" and a[1] != a[1] and b[1] != b[1]), (ctx, n, a, b)
    assert [o[0] for o in got["old"]] == [o[0] for o in exp["old"]], ctx
    assert [o[1][1] for o in got["old"]] == [o[1][1] for o in exp["old"]], ctx


@pytest.mark.gpu
def test_gpu_reference_vectors(tf):
    from transferia_amd import debezium
    for case in golden("debezium.json")["cases"]:
        data, msgs = abi.messages([case["message"].encode("utf-8")])
        parsed, errors = debezium.Parser(tf).parse(data, msgs)
        if case["expect"] is None:
            assert parsed == [] and list(errors) == [0] and errors[0] != abi.ROW_HOST_FALLBACK, case["name"]
            continue
        assert not errors and len(parsed) == 1, (case["name"], errors)
        items = device_items(tf, parsed[0])
        check_item(items[0], case["expect"], case["name"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,weird", [(1, False), (64, False), (700, True), (3000, True)])
def test_gpu_stream_matches_oracle(tf, oracle, n, weird):
    from transferia_amd import debezium
    msgs = cdc_messages(n, seed=n + SEED0, weird=weird)
    if n == 3000:  # a second table in the same batch: two schemas, two device batches
        other = cdc_messages(200, seed=7 + SEED0, table="orders", weird=False)
        msgs = msgs[:1500] + other + msgs[1500:]
    data, m = abi.messages(msgs)
    exp_items, codes = oracle.debezium_parse(data, m)
    parser = debezium.Parser(tf)
    parsed, errors = parser.parse(data, m)
    want = expected_errors(codes, exp_items, msgs)
    assert want == errors
    got = {}
    for p in parsed:
        got.update(device_items(tf, p))
    exp = {it["src"]: it for it in exp_items if it["src"] not in want}
    assert sorted(got) == sorted(exp)
    for k in exp:
        assert_same_items(got[k], exp[k], k)
    if n == 3000:
        assert len(parsed) == 2 and {p.batch.table_id() for p in parsed} == {("public", "events"), ("public", "orders")}
        assert len(parser.cache) >= 2
    fb = sum(1 for c in errors.values() if c == abi.ROW_HOST_FALLBACK)
    assert fb <= (0 if not weird else n // 8), fb   # the stock path is for the odd message, not the stream


def quick_stream(n, seed):
    """workload.debezium_cdc_messages (every field a type dbz_parse_quick reads) with every 7th message bent: shapes the quick kernel must
    hand to the walker (blanks, other member orders, escapes, repeated keys, wrong types, numbers out of range, bad base64) and
    shapes it must take (nulls, empty strings, negative numbers)."""
    import random
    from transferia_amd import workload
    rng = random.Random(seed)
    msgs = workload.debezium_cdc_messages(n, seed=seed)
    out = []
    for k, raw in enumerate(msgs):
        t = raw.decode()
        if k % 7 == 3:
            i = t.index('"payload":') + 10
            head, ps = t[:i], t[i:-1]
            ps = rng.choice([
                ps.replace('"op":', '"op" :'), ps.replace(',"after":', ', "after":'), ps.replace('"ver":', '"v\\u0065r":'), ps.replace('"ver":', '"Ver":'),
                ps.replace('"id":', '"id":0'), ps.replace('"id":', '"id":-'), ps.replace('"id":', '"id":1.5e1,"x":'), ps.replace('"lsn":', '"lsn":-'),
                ps.replace('"txId":', '"txId":99999999999'), ps.replace('"txId":', '"txId":null,"_":'), ps.replace('"xmin":null', '"xmin":17'), ps.replace('"xmin":null', '"xmin":"17"'),
                ps.replace('"snapshot":"false"', '"snapshot":false'), ps.replace('"snapshot":"false"', '"snapshot":null'), ps.replace('"snapshot":"false"', '"snapshot":"fa\\nlse"'),
                ps.replace('"snapshot":"false"', '"snapshot":"fa\\xlse"'), ps.replace('"table":"events"', '"table":"ev\\u0065nts"'), ps.replace('"table":"events"', '"table":null'),
                ps.replace('"schema":"public"', '"schema":"p\u00fcblic"'), ps.replace('"transaction":null', '"transaction":{"id":"5:7","total_order":1,"data_collection_order":2}'),
                ps.replace('"transaction":null', '"transaction":[1,2]'), ps.replace('"payload":"', '"payload":"__debezium_unavailable_value'), ps.replace('"payload":"', '"payload":"\\"q\\" '),
                ps.replace('"payload":"', '"payload":"\u00e9\u4e2d'), ps.replace('"payload":"', '"payload":"a,b}{"'), ps.replace('"payload":"', '"payload":null,"p2":"'),
                ps.replace('"amount":"', '"amount":"=='), ps.replace('"amount":"', '"amount":"AAAA'), ps.replace('"amount":"', '"amount":null,"a2":"'), ps.replace('"amount":', '"amount":""' + ',"a2":'),
                ps.replace('"ts":', '"ts":null,"t2":'), ps.replace('"ts":', '"ts":9223372036854775808,"t2":'), ps.replace('"ts":', '"ts":-9223372036854775808,"t2":'),
                ps.replace('"op":"c"', '"op":"r"').replace('"op":"u"', '"op":"x"').replace('"op":"d"', '"op":"\\u0064"'), ps.replace('"op":', '"op":5,"op2":'),
                ps.replace('"before":null', '"before":{}'), ps.replace('"after":null', '"after":7'), ps.replace(',"source":{', ',"source":null,"s2":{'),
                ps.replace('"ts_ms":', '"ts_ms":null,"x":', 1), ps.replace('"source":{', '"source":{"lsn":1,'), ps.replace('"db":"db"', '"db":"d\tb"'),
                ps[:-1] + ',"extra":1}', ps.replace('{"before":', '{"after":null,"before":'), ps.replace('"ver":', '"ver":1,"ver":'), '{"op":"c"}', 'null', '{}',
            ])
            t = head + ps + "}"
        out.append(t.encode("utf-8"))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 5000])
def test_gpu_quick_path_matches_oracle(tf, oracle, n, capfd):
    """dbz_parse_quick (the tile kernel for payloads that spell the first payload's members) against the oracle, and that it does
    take the stream: only the bent messages may reach the walker."""
    import os
    from transferia_amd import debezium
    msgs = quick_stream(n, 41 + SEED0)
    data, m = abi.messages(msgs)
    exp_items, codes = oracle.debezium_parse(data, m)
    os.environ["TFGPU_DBZ_DEBUG"] = "1"
    try:
        parser = debezium.Parser(tf)
        parsed, errors = parser.parse(data, m)
    finally:
        del os.environ["TFGPU_DBZ_DEBUG"]
    want = expected_errors(codes, exp_items, msgs)
    assert want == errors
    got = {}
    for p in parsed:
        got.update(device_items(tf, p))
    exp = {it["src"]: it for it in exp_items if it["src"] not in want}
    assert sorted(got) == sorted(exp)
    for k in exp:
        assert_same_items(got[k], exp[k], k)
    err = capfd.readouterr().err
    if os.environ.get("TFGPU_DBZ_QUICK") != "0":
        import re
        mm = re.search(r"tfgpu dbz quick: (\d+) messages, (\d+) tiles, (\d+) to the walker", err)
        assert mm, err[-500:]
        assert int(mm.group(3)) <= n // 7 + 2, mm.group(0)


@pytest.mark.gpu
def test_gpu_claimed_payload_spans(tf, oracle, capfd):
    """From its second batch on (a cached prefix, the tile parser in use) the receiver's unpack claims a payload span from the
    message's end instead of walking it; dbz_parse_quick proves the claim or the walker runs IncludeSchema.Unpack's walk.  Envelopes
    whose tails lie about the span — more members, garbage, a missing brace — must end exactly as the oracle says."""
    import os
    import random
    from transferia_amd import debezium
    rng = random.Random(77 + SEED0)
    parser = debezium.Parser(tf)
    first = quick_stream(300, 5 + SEED0)
    data, m = abi.messages(first)
    parser.parse(data, m)   # the opening batch: every span walked; leaves the prefix and "the tile parser was used"
    msgs = quick_stream(1500, 6 + SEED0)
    for k in range(0, len(msgs), 5):
        t = msgs[k].decode()
        if not t.endswith("}}"):
            continue
        body = t[:-1]   # up to and including the payload's '}'
        t = rng.choice([body + " }", body + "}\n\t ", body + ',"x":1}', body + ',"x":{"y":[1,2]}}', body + "} garbage", body + "}}", body, body + "]", body + ',}',
                        body + ',"payload":null}', body[:-1] + "}", body[:-1] + ',"z":{}}}', body + ' ,"schema":0}', body[:-40] + "}", body + '} {"a":1}', body + ",\"x\":\"}\"}"])
        msgs[k] = t.encode()
    data, m = abi.messages(msgs)
    exp_items, codes = oracle.debezium_parse(data, m)
    os.environ["TFGPU_DBZ_HOSTTIME"] = "1"   # (prints which unpack ran; the marks cost syncs only)
    try:
        parsed, errors = parser.parse(data, m)
    finally:
        del os.environ["TFGPU_DBZ_HOSTTIME"]
    want = expected_errors(codes, exp_items, msgs)
    assert want == errors
    got = {}
    for p in parsed:
        got.update(device_items(tf, p))
    exp = {it["src"]: it for it in exp_items if it["src"] not in want}
    assert sorted(got) == sorted(exp)
    for k in exp:
        assert_same_items(got[k], exp[k], k)
    if os.environ.get("TFGPU_DBZ_QUICK") != "0" and os.environ.get("TFGPU_DBZ_TENTATIVE") != "0":
        assert "claimed spans" in capfd.readouterr().err


@pytest.mark.gpu
def test_gpu_walker_cross_check():
    """TFGPU_DBZ_QUICK=0: every message through dbz_parse.  Both forms must pass the file."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, TFGPU_DBZ_QUICK="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "not cross_check"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.mark.gpu
def test_gpu_stream_feeds_collapse_and_native_serializer(tf, oracle):
    """configs[4] from Debezium-envelope bytes: parse -> sharder -> partition -> exchange (one rank) -> Collapse -> native
    queue serializer.  The Kafka messages equal the oracle's for the same collapsed rows, and those rows are the oracle's
    Collapse of the oracle's parse."""
    from transferia_amd import debezium, workload
    n = 3000
    msgs = workload.debezium_cdc_messages(n, seed=21 + SEED0)
    data, m = abi.messages(msgs)
    parsed, errors = debezium.Parser(tf).parse(data, m)
    assert len(parsed) == 1 and not errors
    p = parsed[0]
    exp_items, codes = oracle.debezium_parse(data, m)
    got = device_items(tf, p)
    for it in exp_items:
        assert_same_items(got[it["src"]], it, it["src"])
    one = tf.Transformer("sharder_transformer", {"shardsCount": "1", "columns": {"includeColumns": ["^id$"]}}).apply(p.batch).transformed
    grouped, counts = tf.partition(one, 1)
    comm = tf.Comm.create(tf.Comm.unique_id(), 0, 1)
    try:
        back, _ = comm.exchange(grouped, counts)
    finally:
        comm.close()
    col = tf.collapse(back)
    assert 0 < col.nrows < p.batch.nrows
    a = col.download()
    host = p.batch.download()
    host.schema = p.schema
    ref = oracle.collapse(host, p.schema).batch
    key = lambda b: sorted((int(b.kind[i]), int(b.src_row[i])) for i in range(b.nrows))
    assert key(a) == key(ref)
    ids, lsns, cts, nf = np.zeros(n, np.uint32), np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint8)
    mm = p.rows["msg"]
    ids[mm], lsns[mm], cts[mm], nf[mm] = p.rows["id"], p.rows["lsn"], p.rows["commit_time"], p.rows["names_form"]
    meta = abi.row_meta(n, ids=ids, lsns=lsns, commit_times=cts, names_form=nf)
    o = abi.queue_options(abi.QFMT_NATIVE, enabled=True, max_message_size=1 << 16, table_schema=p.schema, old_key_types=["int64"])
    out = tf.queue_serialize(o, col, meta).messages()
    a.schema = p.schema
    exp = oracle.queue_serialize(o, a, p.schema, meta)
    assert out == exp and len(out) > 3
    first = json.loads(out[0])[0]
    assert first["schema"] == "public" and first["table"] == "events" and first["table_schema"][0]["table_name"] == "events"
    kinds = [it for x in out[:50] for it in json.loads(x)]
    dels = [it for it in kinds if it["kind"] == "delete"]
    assert dels and all(it["columnnames"] is None and it["oldkeys"]["keynames"] == ["id"] for it in dels)   # a Delete keeps ColumnNames nil


@pytest.mark.gpu
def test_gpu_toasted_stream_stays_on_the_device(tf, oracle):
    """A Postgres topic whose Updates carry `__debezium_unavailable_value` for unchanged TOASTed columns: the receiver leaves those columns out of
    the row's ColumnNames (receiver.go:98-105) — ABSENT cells on the device, parsed (not handed to the host), equal to the oracle's items; the
    rows then go through Collapse (compareColumns merges: the kept value of a left-out column is the chain's), the native queue format (every row
    its own columnnames) and the Debezium emitter (the placeholder again), each against the oracle."""
    import random
    from transferia_amd import debezium
    rng = random.Random(77 + SEED0)
    schema = json.dumps(envelope_schema("events"), separators=(",", ":"))
    msgs, marked = [], 0
    for k, ps in cdc_payloads(rng, 900, "events", weird=False):
        if '"op":"u"' in ps and k % 2:  # the update did not touch the TOASTed columns
            after = ps.index('"after":')
            for field, pat in (("name", '"name":'), ("blob", '"blob":'), ("amount", '"amount":')):
                if rng.random() < 0.6:
                    at = ps.index(pat, after)
                    end = at + len(pat)
                    stop = end
                    if ps[end] == '"':
                        stop = end + 1
                        while ps[stop] != '"' or ps[stop - 1] == "\\":
                            stop += 1
                        stop += 1
                    else:
                        while ps[stop] not in ",}":
                            stop += 1
                    ps = ps[:end] + '"__debezium_unavailable_value"' + ps[stop:]
                    marked += 1
        msgs.append(('{"schema":%s,"payload":%s}' % (schema, ps)).encode("utf-8"))
    assert marked > 100
    data, m = abi.messages(msgs)
    exp_items, codes = oracle.debezium_parse(data, m)
    assert all(c == abi.ROW_OK for c in codes)
    parsed, errors = debezium.Parser(tf).parse(data, m)
    assert not errors and len(parsed) == 1
    p = parsed[0]
    got = device_items(tf, p)
    ragged = 0
    for it in exp_items:
        assert_same_items(got[it["src"]], it, it["src"])
        ragged += it["kind"] == "update" and len(it["names"]) < len(FIELDS)
    assert ragged > 50
    # Collapse: the device's rows equal the oracle's row-wise loop over the oracle's items
    keys = [c.name for c in p.schema.cols if c.key]
    rows_in = [{"kind": it["kind"], "keys": keys, "names": it["names"], "values": [list(v) if v[0] not in ("string", "bytes", "jsonnum") else [v[0], bytes(v[1]).decode("utf-8", "surrogateescape")] for v in it["values"]],
                "old_names": [o[0] for o in it["old"]], "old_values": [list(o[1]) for o in it["old"]]} for it in exp_items]
    col = tf.collapse(p.batch)
    assert 0 < col.nrows < p.batch.nrows
    a = col.download()
    want = oracle.collapse_rows(rows_in)
    from collapse_cases import items_of as rows_as_items
    # a chain that starts with an Update leaving a column out and meets one that lists it has its names out of batch order: the batch then
    # carries every row's own order (the committed seed's stream holds such chains; a soak seed's may not — found by seed 603)
    batch_names = [c.name for c in a.cols]
    out_of_order = any(r["kind"] != "delete" and r["names"] != [x for x in batch_names if x in r["names"]] for r in want)
    if out_of_order:
        assert getattr(a, "col_order", None) is not None
    assert out_of_order or SEED0 != 0
    assert [(r["kind"], r["src"], r["names"]) for r in want] == [(g["kind"], g["src"], [] if g["kind"] == "delete" else g["names"]) for g in rows_as_items(a)]
    # the native queue format and the Debezium emitter over the collapsed rows, against the oracle reading the same (downloaded) rows
    n = len(msgs)
    ids, lsns, cts, nf = np.zeros(n, np.uint32), np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint8)
    mm = p.rows["msg"]
    ids[mm], lsns[mm], cts[mm], nf[mm] = p.rows["id"], p.rows["lsn"], p.rows["commit_time"], p.rows["names_form"]
    meta = abi.row_meta(n, ids=ids, lsns=lsns, commit_times=cts, names_form=nf)
    o = abi.queue_options(abi.QFMT_NATIVE, enabled=True, max_message_size=1 << 14, table_schema=p.schema, old_key_types=["int64"])
    out = tf.queue_serialize(o, col, meta).messages()
    a.schema = p.schema
    assert out == oracle.queue_serialize(o, a, p.schema, meta) and len(out) > 3
    if SEED0 == 0:  # (a sanity check of the committed stream's shape, not of the product: a soak seed's stream may hold no such message — seed 616)
        assert any(b'"columnnames":["id","ver","flag","score","at"' in x for x in out) or any(b'"blob"' not in x.split(b'"columnvalues"')[0] for x in out)


@pytest.mark.gpu
def test_gpu_parse_refuses_foreign_frames(tf):
    """tfgpu_debezium_parse takes the frames back from the host: spans outside their message are refused before any lane runs."""
    from transferia_amd import debezium
    msgs = cdc_messages(8, weird=False)
    data, m = abi.messages(msgs)
    frames = tf.debezium_unpack(data, m)
    comp = debezium.compile_schema(data[int(frames["schema_start"][0]):int(frames["schema_start"][0]) + int(frames["schema_len"][0])])
    key = (int(frames["schema_hash"][0][0]), int(frames["schema_hash"][0][1]))
    bad = frames.copy()
    bad["payload_start"][3] = len(data) + 1000
    with pytest.raises(tf.TfgpuError) as ei:
        tf.debezium_parse(key, comp, data, bad, m)
    assert ei.value.code == tf.ERR_INVALID
    db, rows, errs = tf.debezium_parse(key, comp, data, frames, m)
    assert db.nrows + len(errs) == 8


@pytest.mark.gpu
def test_gpu_prefix_compare_at_every_alignment_and_edge(tf, oracle):
    """dbz_prefix_same reads a message in its own 16-byte lines and the bytes in front of the first / behind the last line apart: messages that
    start at every alignment, each with ONE byte of the shared head changed — the first, those around the first aligned line, one inside, those
    around the last line, the last — must not inherit the reference's frame: their frames, and everything parsed from them, are the full walk's
    (a changed field name shows in the item's ColumnNames, a changed brace in the message's fate)."""
    from transferia_amd import debezium
    base = cdc_messages(3, seed=SEED0 + 31, weird=False)[0]            # k % 3 == 0: {"schema":…,"payload":…}
    head = base.index(b'"payload":') + len(b'"payload":')
    at_name = base.index(b'"field":"ver"') + len(b'"field":"v')         # inside the schema: changes a column's name
    spots = sorted({0, 1, 7, 8, 15, 16, 17, 31, 32, at_name, head // 2, head - 33, head - 17, head - 16, head - 15, head - 9, head - 8, head - 2, head - 1})
    for shift in range(0, 16):
        msgs = [base]
        filler = b'{"schema":null,"payload":null}' + b" " * shift       # moves every later message's start by `shift` bytes
        for sp in spots:
            bad = bytearray(base)
            bad[sp] = ord("#") if bad[sp] != ord("#") else ord("$")
            msgs += [filler, bytes(bad), base]
        msgs += [base] * 8                                              # 64 messages and more: the opening message becomes the batch's reference (dbz_prefix_same)
        assert len(msgs) >= 64
        data, m = abi.messages(msgs)
        exp_items, codes = oracle.debezium_parse(data, m)
        parsed, errors = debezium.Parser(tf).parse(data, m)
        want = expected_errors(codes, exp_items, msgs)
        # (a `before` struct that no longer spells the `after` struct's fields is the stock code's: the device hands such a message over where the
        #  reference, for an insert, never looks at it)
        # … and where the reference DOES look at it (an update / a delete: a soak seed's base message, found by seeds 612 / 614 / 617) it fails the
        #  message with "field not found": the device's hand-over leaves that verdict to the stock code — the one substitution allowed here)
        mine = {k: (want[k] if v == abi.ROW_HOST_FALLBACK and want.get(k) == abi.ROW_DBZ_FIELD else v) for k, v in errors.items()}
        assert {k: v for k, v in mine.items() if k in want} == want and all(v == abi.ROW_HOST_FALLBACK for k, v in errors.items() if k not in want), shift
        got = {}
        for p in parsed:
            got.update(device_items(tf, p))
        for it in exp_items:
            if it["src"] not in errors:
                assert_same_items(got[it["src"]], it, (shift, it["src"]))
        assert len(got) >= len(spots)   # the untouched copies between them all parse
        assert sum(1 for c in codes if c != abi.ROW_OK) >= len(spots) - 2   # nearly every changed byte breaks the message for the reference too


@pytest.mark.gpu
def test_gpu_unpack_with_a_cached_prefix_equals_the_full_walk(tf, oracle):
    """tfgpu_debezium_unpack_cached: the head of an earlier batch's message stands in for the per-batch reference walk; frames and
    everything downstream are the same, for messages that share the head and for those that do not."""
    from transferia_amd import debezium
    first = cdc_messages(50, seed=SEED0 + 5, weird=False)
    parser = debezium.Parser(tf)
    d0, m0 = abi.messages(first)
    parser.parse(d0, m0)
    assert parser.known is not None and parser.known[0].endswith(b'"payload":')
    msgs = cdc_messages(900, seed=SEED0 + 6, weird=True) + cdc_messages(60, seed=SEED0 + 8, table="orders", weird=False)
    data, m = abi.messages(msgs)
    plain = tf.debezium_unpack(data, m)
    cached = tf.debezium_unpack(data, m, parser.known)
    for f in ("schema_start", "payload_start", "schema_len", "payload_len", "code"):
        assert np.array_equal(plain[f], cached[f]), f
    assert np.array_equal(plain["schema_hash"], cached["schema_hash"])
    parsed, errors = parser.parse(data, m)     # the cache-hit path end to end
    exp_items, codes = oracle.debezium_parse(data, m)
    want = expected_errors(codes, exp_items, msgs)
    assert want == errors
    got = {}
    for p in parsed:
        got.update(device_items(tf, p))
    for it in exp_items:
        if it["src"] not in want:
            assert_same_items(got[it["src"]], it, it["src"])
    # the usual batch of the usual topic — every message OK, the known schema, a payload the tile parser takes: the frames stay on the
    # device (FrameCache::uniform in tf_debezium.hip), frame 0 and three counters come down instead, no host loop over the messages runs
    usual = cdc_messages(700, seed=SEED0 + 9, weird=False)
    du, mu = abi.messages(usual)
    tf.prof_enable(True)
    try:
        tf.prof_reset()
        parsed, errors = parser.parse(du, mu)
        assert "dbz_frame_stats" in {name for name, _n, _ms in tf.prof_get()}
    finally:
        tf.prof_enable(False)
    exp_items, codes = oracle.debezium_parse(du, mu)
    assert errors == expected_errors(codes, exp_items, usual) == {}
    got = {}
    for p in parsed:
        got.update(device_items(tf, p))
    assert len(got) == len(exp_items) == 700
    for it in exp_items:
        assert_same_items(got[it["src"]], it, it["src"])
