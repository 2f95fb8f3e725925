"""Debezium events framed by a schema registry (NewDebeziumImpl with a registry client; the f1 remainder of SURVEY §8):
0x00 | schema id | payload, several events per Kafka message, the registry's ConfluentJSONSchema converted to the Kafka Connect form.
CPU: the oracle's format conversion pinned to the reference's fixtures, the C side's schema compilation against it.
GPU: tfgpu_dbz_receive_registry against the oracle."""
import json
import random

import numpy as np
import pytest

from transferia_amd import abi
from util import golden
from test_debezium import SEED0, assert_same_items, cdc_payloads, device_items, envelope_schema


# ---- pkg/schemaregistry/format, pinned to its own tests' fixtures -------------------------------------------------------------
def test_format_fixture_properties():
    """json_schema_format_test.go: TestMarshalUnmarshal{Kafka,Confluent}[Arrays] (the structs keep every field of the fixtures),
    TestKafkaToConfluentToKafka[Arrays] (identity), TestCanonizeMakeClosedContentModelTrue (canon file).  The fixtures are one
    envelope in both forms, so Confluent → Kafka of one must be the other."""
    from oracle import ora_srformat as F
    g = golden("sr_format.json")
    for sfx in ("", "_arr"):
        k, c = g["kafka" + sfx], g["confluent" + sfx]
        kb, cb = F.bind_kafka(k), F.bind_confluent(c)
        assert kb == k and cb == c
        assert F.to_kafka(F.to_confluent(kb)) == kb
        assert F.to_kafka(cb) == kb
    assert F.to_confluent(F.to_kafka(F.bind_confluent(g["confluent"])), closed=True) == g["closed_canon"]


def test_format_edges():
    from oracle import ora_srformat as F
    assert F.to_kafka(F.bind_confluent({"oneOf": [{"type": "null"}, {"type": "integer", "connect.type": "int16"}]})) == {"type": "int16", "optional": True}
    assert F.to_kafka(F.bind_confluent({"type": "number"})) == {"type": "bytes", "optional": False}
    assert F.to_kafka(F.bind_confluent({"type": "number", "connect.type": "float32"}))["type"] == "float"
    assert F.to_kafka(F.bind_confluent({"type": "string", "connect.type": "bytes", "connect.parameters": {"scale": "2"}, "title": "org.apache.kafka.connect.data.Decimal"})) == \
        {"type": "bytes", "optional": False, "name": "org.apache.kafka.connect.data.Decimal", "parameters": {"scale": "2"}}
    with pytest.raises(F.GoPanic):
        F.to_kafka(F.bind_confluent({"properties": {"a": {}, "b": {"connect.index": 1}}}))
    assert F.to_kafka(F.bind_confluent({"properties": {"a": {}}}))["fields"] == [{"type": "", "optional": False, "field": "a"}]   # one property: sort.Slice compares nothing
    with pytest.raises(F.Unbindable):
        F.bind_confluent({"connect.index": 1.5})
    with pytest.raises(F.Unbindable):
        F.bind_confluent({"oneOf": {}})


# ---- the C side's convertSchemaFormat + receiveSchema (host code: no GPU needed) --------------------------------------------
def registry_text(table="events"):
    from oracle import ora_srformat as F
    return json.dumps(F.to_confluent(F.bind_kafka(envelope_schema(table))), separators=(",", ":")).encode()


def test_compile_registry_schema_equals_the_inline_compile():
    from transferia_amd import debezium
    g = golden("sr_format.json")
    for sfx in ("", "_arr"):
        try:
            want = debezium.compile_schema(json.dumps(g["kafka" + sfx]).encode())
        except debezium.HostOnly:
            with pytest.raises(debezium.HostOnly):
                debezium.compile_registry_schema(json.dumps(g["confluent" + sfx]).encode())
            continue
        assert debezium.compile_registry_schema(json.dumps(g["confluent" + sfx]).encode()) == want and len(want) == (60 if not sfx else 37)
    assert debezium.compile_registry_schema(registry_text()) == debezium.compile_schema(json.dumps(envelope_schema()).encode())


BROKEN = [b'{"title": 5}', b"[1]", b"", b'{"connect.version": 1.5}', b'{"properties": {"before": 7}}', b'{"oneOf": [3]}', b'{"connect.parameters": {"scale": 2}}', b"{"]
HOST = [b'{"properties":{"a":{},"b":{}}}', b'{"properties":{"a":{"connect.index":1},"b":{"connect.index":1}}}', b"null", b'{"oneOf":[{"type":"null"}]}', b'{"Type": "object"}']


def test_compile_registry_schema_failures_follow_the_oracle():
    from oracle import ora_srformat as F
    from transferia_amd import debezium
    for text in BROKEN:
        with pytest.raises(F.Unbindable):
            F.convert_schema_format(text)
        with pytest.raises(debezium.SchemaError):
            debezium.compile_registry_schema(text)
    for text in HOST:
        with pytest.raises(debezium.HostOnly):
            debezium.compile_registry_schema(text)


def _random_struct(rng, depth=0):
    types = ["int8", "int16", "int32", "int64", "boolean", "string", "float", "double", "bytes", "struct", "array", "weird"]
    fields = []
    for i in range(rng.randrange(1, 7)):
        t = rng.choice(types if depth < 2 else types[:9])
        f = {"type": t, "optional": rng.random() < 0.5, "field": "f%d_%d" % (depth, i)}
        if t == "bytes" and rng.random() < 0.5:
            f["name"] = "org.apache.kafka.connect.data.Decimal"
            f["parameters"] = {"scale": rng.choice(["2", "0", "x", "-3", ""]), "connect.decimal.precision": "10"}
        if t == "struct":
            f["name"] = rng.choice(["io.debezium.data.geometry.Point", "io.debezium.data.VariableScaleDecimal", "other.Struct"])
            f["fields"] = _random_struct(rng, depth + 1)["fields"]
        if t == "array":
            f["items"] = {"type": "int32", "optional": True}
        if rng.random() < 0.1:
            f["version"] = rng.randrange(1, 4)
        if rng.random() < 0.1:
            f["default"] = rng.choice([0, "d", False, 1.5])
        if rng.random() < 0.05:
            f["__dt_original_type_info"] = {"original_type": "pg:integer"}
        fields.append(f)
    return {"type": "struct", "fields": fields, "optional": False}


def _mutate(rng, node):
    """one random damage to a ConfluentJSONSchema document (in place); returns a label"""
    objs = []

    def walk(o):
        if isinstance(o, dict):
            objs.append(o)
            for v in list(o.values()):
                walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)
    walk(node)
    o = rng.choice(objs)
    kind = rng.randrange(6)
    if kind == 0 and "connect.index" in o:
        del o["connect.index"]; return "no index"
    if kind == 1 and "connect.index" in o:
        o["connect.index"] = 0; return "index 0"
    if kind == 2:
        k = rng.choice(["title", "type", "connect.type", "description"]); o[k] = rng.choice([5, [], {}, True]); return "bad " + k
    if kind == 3:
        k = rng.choice(["connect.version", "connect.index"]); o[k] = rng.choice([1.5, "1", [], 1e30]); return "bad " + k
    if kind == 4:
        k = rng.choice(["oneOf", "properties", "items", "connect.parameters", "additionalProperties"]); o[k] = rng.choice([7, "s", [3], {"a": 4}]); return "bad " + k
    o["type"] = rng.choice(["null", "object", "string", "integer", "bogus"]); return "type swap"


def test_compile_registry_schema_random():
    """random envelopes in both forms compile alike; random damage to the registry form ends where the oracle's conversion + the inline
    compile of its result end (a schema that does not bind / the reference's nil dereference / a receiver-less type / the fields)"""
    from oracle import ora_srformat as F
    from transferia_amd import debezium
    rng = random.Random(77 + SEED0)
    outcomes = {}
    for it in range(300):
        row = _random_struct(rng)
        row["optional"] = True
        env = {"type": "struct", "optional": False, "name": "srv.public.t.Envelope",
               "fields": [dict(row, field="before"), dict(row, field="after"), {"type": "string", "optional": False, "field": "op"}]}
        conf = F.to_confluent(F.bind_kafka(env))
        label = "intact"
        if it % 2:
            label = _mutate(rng, conf)
        text = json.dumps(conf).encode()

        def inline(kafka_text):
            try:
                return ("fields", debezium.compile_schema(kafka_text))
            except debezium.SchemaError:
                return ("schema error", None)
            except debezium.HostOnly:
                return ("host", None)
        try:
            want = inline(F.convert_schema_format(text))
        except F.Unbindable:
            want = ("schema error", None)
        except F.GoPanic:
            want = ("host", None)
        try:
            got = ("fields", debezium.compile_registry_schema(text))
        except debezium.SchemaError:
            got = ("schema error", None)
        except debezium.HostOnly:
            got = ("host", None)
        if it % 2 == 0:
            assert got == inline(json.dumps(env).encode()), (it, "the two forms of one envelope")
        # the C side may always hand a schema to the host; it must never decide differently
        assert got == want or got[0] == "host", (it, label, got[0], want[0], text[:300])
        outcomes[(label.split()[0], got[0], want[0])] = outcomes.get((label.split()[0], got[0], want[0]), 0) + 1
    kinds = {k[1] for k in outcomes}
    assert {"fields", "schema error", "host"} <= kinds, outcomes
    assert sum(v for k, v in outcomes.items() if k[1] == "host" and k[2] != "host") <= 20, outcomes   # conservative answers stay rare


def test_compile_deeply_nested_schema_is_handed_over_not_followed():
    """a schema text nested thousands of levels deep (a registry's answer, or the `schema` member of a hostile message) must not take the
    host's stack with it"""
    from transferia_amd import debezium
    deep = b'{"properties":' * 5000 + b"{}" + b"}" * 5000
    with pytest.raises(debezium.HostOnly):
        debezium.compile_registry_schema(deep)
    deep2 = b'{"fields":[' * 5000 + b"{}" + b"]}" * 5000
    with pytest.raises(debezium.HostOnly):
        debezium.compile_schema(deep2)


# ---- streams ------------------------------------------------------------------------------------------------------------------
def frame(sid, payload: bytes) -> bytes:
    return b"\x00" + int(sid).to_bytes(4, "big") + payload


def sr_stream(n, seed, weird=True):
    """n events of two tables (registry ids 7 and 300) packed into Kafka messages of one to three events; with `weird` also the
    payloads and framings the reference turns into `_unparsed` items, stops a message at, or panics on."""
    rng = random.Random(seed)
    ev = [(7, ps) for _k, ps in cdc_payloads(rng, n, "events", weird)] + [(300, ps) for _k, ps in cdc_payloads(rng, max(n // 8, 1), "orders", False)]
    rng.shuffle(ev)
    msgs, cur = [], b""
    for i, (sid, ps) in enumerate(ev):
        body = ps.encode("utf-8")
        r = rng.random()
        if r < 0.15:
            body = rng.choice([b" ", b"\n\t ", b""]) + body + rng.choice([b"\n", b" \r\n", b"  "])
        if weird and i % 13 == 6:
            body = rng.choice([b"", b"   ", b"null", b"nullx", b"null x", b"[1]", b"7", b'"s"', body[:-1], body + b" trailing garbage", body + b"}", b"{" + body, b"tru", b"{}"])
        if weird and i % 29 == 11:
            sid = rng.choice([41, 42, 43])   # a schema that does not bind / panics in the reference / converts to a struct without `after`
        cur += frame(sid, body)
        tail = weird and i % 37 == 20
        if tail:   # what the next DoOne call meets at the end of a Kafka message: a stump shorter than the prefix (buf[5:] panics), or bytes that stay in the payload
            cur += rng.choice([b"\x00", b"\x00\x00\x00", b"\x01abc", b"garbage"])
        if tail or rng.random() < 0.6:
            msgs.append(cur)
            cur = b""
    msgs.append(cur)
    if weird:
        msgs += [b"", b"\x00\x00", b"x", frame(7, b'{"op":"c"}')]
    return msgs


REGISTRY = {41: b'{"title": 5}', 42: b'{"properties":{"a":{},"b":{}}}', 43: b'{"type":"object","properties":{"x":{"type":"string","connect.index":0}}}'}


def registry():
    reg = dict(REGISTRY)
    reg[7], reg[300] = registry_text("events"), registry_text("orders")
    return reg


def expected_fates(codes, items, events):
    """the oracle's fates plus the device's one-table-per-group rule (an item whose TableID differs from the first good item of its schema id)"""
    out = {e: c for e, c in enumerate(codes) if c}
    first = {}
    for e in sorted(items):
        sid = events[e][2]
        ref = first.setdefault(sid, (items[e]["ns"], items[e]["table"]))
        if (items[e]["ns"], items[e]["table"]) != ref:
            out[e] = abi.ROW_HOST_FALLBACK
    return out


def apply_message_rule(fates, events):
    """DoBuf's rule once more, after fates were added: a message with a host event goes to the host whole, events behind a failed one drop"""
    out = dict(fates)
    by_msg = {}
    for e, (m, _i, _s) in enumerate(events):
        by_msg.setdefault(m, []).append(e)
    for es in by_msg.values():
        host, dead = False, False
        for e in es:
            if out.get(e) == abi.ROW_HOST_FALLBACK:
                host = True
            if out.get(e):
                break
        for e in es:
            if host:
                out[e] = abi.ROW_HOST_FALLBACK
            elif dead:
                out[e] = abi.ROW_DROPPED
            elif out.get(e):
                dead = True
    return out


def test_oracle_sr_stream_is_consistent(oracle):
    """the restated DoBatch over a stream: every event has a fate, good ones are the payloads' own items"""
    msgs = sr_stream(300, 5)
    data, m = abi.messages(msgs)
    events, items, codes = oracle.debezium_parse_sr(data, m, registry())
    assert len(events) == len(codes) and len(items) > 150
    assert all(codes[e] == abi.ROW_OK for e in items)
    seen = {c for c in codes}
    assert {abi.ROW_OK, abi.ROW_DBZ_PAYLOAD, abi.ROW_DBZ_SCHEMA, abi.ROW_HOST_FALLBACK, abi.ROW_DROPPED, abi.ROW_SR_MAGIC} <= seen, seen
    assert {it["table"] for it in items.values()} == {"events", "orders", ""}   # (a payload whose `source` is null: empty Schema / Table)
    for (mm, idx, _sid), nxt in zip(events, events[1:]):
        assert nxt[0] > mm or nxt[1] == idx + 1


@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("n,weird", [(1, False), (200, False), (900, True)])
def test_gpu_registry_stream_matches_oracle(tf, oracle, n, weird):
    from transferia_amd import debezium
    msgs = sr_stream(n, 100 + n + SEED0, weird)
    data, m = abi.messages(msgs)
    reg = registry()
    exp_events, exp_items, codes = oracle.debezium_parse_sr(data, m, reg)
    parser = debezium.Parser(tf)
    parsed, missing, _ = parser.parse_registry(data, m)
    used = sorted({e[2] for e in exp_events if e[2] is not None})
    assert parsed is None and sorted(missing) == used      # nothing registered yet: the shim learns the ids it has to fetch
    for sid in used:
        parser.add_registry_schema(sid, reg[sid])
    parsed, errors, events = parser.parse_registry(data, m)
    assert [(int(e["msg"]), int(e["index"]), None if e["code"] else int(e["schema_id"])) for e in events] == exp_events
    want = apply_message_rule(expected_fates(codes, exp_items, exp_events), exp_events)
    assert errors == want, ({k: (errors.get(k), want.get(k)) for k in set(errors) | set(want) if errors.get(k) != want.get(k)})
    got = {}
    for p in parsed:
        got.update(device_items(tf, p))
    exp = {e: it for e, it in exp_items.items() if e not in want}
    assert sorted(got) == sorted(exp)
    for e in exp:
        assert_same_items(got[e], exp[e], e)
    if n >= 200:
        assert {p.batch.table_id() for p in parsed} == {("public", "events"), ("public", "orders")}
    fb = sum(1 for c in errors.values() if c == abi.ROW_HOST_FALLBACK)
    assert fb <= (0 if not weird else len(exp_events) // 6), fb
    # the receiver's next batch: payload spans are claimed from the events' ends and proven by the tile parser (or walked after all)
    parsed2, errors2, _ = parser.parse_registry(data, m)
    assert errors2 == want
    got2 = {}
    for p in parsed2:
        got2.update(device_items(tf, p))
    assert sorted(got2) == sorted(exp)
    for e in exp:
        assert_same_items(got2[e], exp[e], ("second batch", e))


@pytest.mark.gpu
def test_gpu_registry_usual_batch_keeps_its_frames_on_the_device(tf, oracle):
    """Every event OK, one schema id, payloads the tile parser takes: tfgpu_debezium_registry_frames answers that with frame 0 and three
    counters (FrameCache::uniform in tf_debezium.hip), the receiver runs none of its per-event loops — and the items are the oracle's."""
    import random
    from transferia_amd import debezium
    rng = random.Random(SEED0 + 77)
    ev = [ps.encode("utf-8") for _k, ps in cdc_payloads(rng, 600, "events", False)]
    msgs, cur = [], b""
    for i, body in enumerate(ev):
        cur += frame(7, body)
        if i % 3 == 2:
            msgs.append(cur)
            cur = b""
    if cur:
        msgs.append(cur)
    data, m = abi.messages(msgs)
    reg = registry()
    exp_events, exp_items, codes = oracle.debezium_parse_sr(data, m, reg)
    parser = debezium.Parser(tf)
    parser.add_registry_schema(7, reg[7])
    for round_ in range(2):   # (the second batch also claims the payload spans from the events' ends)
        tf.prof_enable(True)
        try:
            tf.prof_reset()
            parsed, errors, events = parser.parse_registry(data, m)
            assert "dbz_frame_stats" in {name for name, _n, _ms in tf.prof_get()}
        finally:
            tf.prof_enable(False)
        assert errors == {} and len(events) == 600
        got = {}
        for p in parsed:
            got.update(device_items(tf, p))
        assert sorted(got) == sorted(exp_items) and len(got) == 600
        for e in exp_items:
            assert_same_items(got[e], exp_items[e], (round_, e))


@pytest.mark.gpu
def test_gpu_registry_frames_refuse_foreign_events(tf):
    """events that do not come from tfgpu_sr_frames over the same bytes are refused, not followed"""
    import ctypes as C
    L = tf.load()
    data = frame(7, b'{"op":"c"}')
    ev = (abi.CSrFrame * 1)()
    ev[0].msg, ev[0].start, ev[0].len, ev[0].schema_id = 0, 5, 4000, 7
    starts = np.array([0, len(data)], np.uint64)
    em = abi.CMessages()
    em.nmsg, em.start = 1, starts.ctypes.data
    fr = (abi.CDbzFrame * 1)()
    L.tfgpu_debezium_registry_frames.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(abi.CMessages), C.POINTER(abi.CSrFrame), C.POINTER(abi.CDbzFrame)]
    rc = L.tfgpu_debezium_registry_frames(data, len(data), abi.MEM_HOST, C.byref(em), ev, fr)
    assert rc == tf.ERR_INVALID
