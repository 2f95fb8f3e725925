"""Confluent Schema Registry parser, JSON schemas (SURVEY §8f.1, configs[2]): the oracle against the reference's canon
(TestClient) and its own test vectors (CPU); the HIP path against the oracle and the same canon (GPU)."""
import base64
import json

import numpy as np
import pytest

from transferia_amd import abi, confluent_sr
from util import golden
import os as _os

SEED0 = int(_os.environ.get("TFGPU_TEST_SEED", "0"))  # 0 = the committed seeds; other values: soak runs (TFGPU_TEST_SEED=n bash tools/gpu_visit.sh tests TAG)

G = golden("confluent_sr.json")
MSGS = [base64.b64decode(x) for x in G["messages"]]
JSON_IDS = sorted(int(k) for k, v in G["schemas"].items() if v["schemaType"] == "JSON")


def _batch_items(res, frames):
    """Rows of one parse result as canon-shaped dicts keyed by frame ordinal."""
    out = {}
    b = res.batch if hasattr(res, "batch") else res
    for r in range(b.nrows):
        vals = []
        for c in b.cols:
            g, v = c.pyvalue(r)
            if g == "nil":
                vals.append(None)
            elif g == "json":
                vals.append(json.loads(v))
            elif g == "jsonnum":
                vals.append(float(v))
            elif g == "string":
                vals.append(v.decode("utf-8"))
            else:
                vals.append(v)
        out[int(b.src_row[r])] = {"schema": b.table_ns, "table": b.table_name, "names": [c.name for c in b.cols], "values": vals,
                                  "lsn": frames[int(b.src_row[r])][0], "dtypes": [c.dtype for c in b.cols]}
    return out


def _check_canon(parse, frames_fn):
    """TestClient (parser_test.go:76-103): every message through Do, items in message order."""
    nonempty = [(i, m) for i, m in enumerate(MSGS) if m]  # `if len(data) == 0 { continue }`
    data, cm = abi.messages([m for _, m in nonempty], offsets=[i for i, _ in nonempty])
    frames = frames_fn(data, cm)
    assert len([f for f in frames if f[0] == 0]) == 2  # the first message holds two frames (require.Len(result, 2))
    lsn_of = [(nonempty[f[0]][0],) for f in frames]
    got = {}
    for sid in JSON_IDS:
        o = confluent_sr.sr_json_options(sid, G["schemas"][str(sid)]["schema"])
        res = parse(o, data, cm)
        assert not res.errors, res.errors
        got.update(_batch_items(res, lsn_of))
    items = [got[k] for k in sorted(got)]
    assert len(items) == len(G["items"])
    for it, exp in zip(items, G["items"]):
        assert (it["schema"], it["table"], it["lsn"], it["names"]) == (exp["schema"], exp["table"], exp["lsn"], exp["names"])
        assert it["values"] == exp["values"], (it, exp)
        assert it["dtypes"] == [c[1] for c in exp["table_schema"]]


def test_schema_rows_match_canon():
    for sid in JSON_IDS:
        title, rows = confluent_sr.json_schema_rows(G["schemas"][str(sid)]["schema"])
        ns, tb = confluent_sr.build_json_table_id(title)
        exp = next(it for it in G["items"] if (it["schema"], it["table"]) == (ns, tb))
        assert [[n, confluent_sr.SRT_DTYPE[t], r] for n, t, r in rows] == exp["table_schema"]
    # format_json_test.go:14-65 TestJSONTableNamePolicy
    assert confluent_sr.build_json_table_id("public.person", confluent_sr.POLICY_DEBEZIUM_STYLE) == ("public", "person")
    assert confluent_sr.build_json_table_id("public.person", confluent_sr.POLICY_TITLE) == ("", "public.person")
    assert confluent_sr.build_json_table_id("public.person", manual_table_name="blablabla") == ("", "blablabla")
    # utils_json_test.go:13-38 TestProcessPayload: required through the "required" list, `connect.type` ignored
    sv = '{"title":"public.person","type":"object","properties":{"name":{"type":"string"},"id":{"type":"integer","connect.type":"int32"},"email":{"type":"string"}},"required":["name","id"],"additionalProperties":false}'
    assert confluent_sr.json_schema_rows(sv)[1] == [("email", abi.SRT_STRING, False), ("id", abi.SRT_INTEGER, True), ("name", abi.SRT_STRING, True)]


def test_oracle_canon(oracle):
    _check_canon(oracle.sr_json_parse, oracle.sr_frames)


PERSON = '{"title":"public.person","type":"object","properties":{"name":{"type":"string"},"id":{"type":"integer"},"email":{"type":"string"}},"required":["name","id"]}'


def test_oracle_process_payload(oracle):
    """utils_json_test.go TestProcessPayload: 3 names, 3 values; the absent optional field is nil."""
    o = confluent_sr.sr_json_options(7, PERSON)
    res = oracle.sr_json_parse(o, b"\0\0\0\0\x07" + b'{"id": 2, "name": "Bob"}')
    assert abi.batch_rows(res.batch) == [[("nil", None), ("int64", 2), ("string", b"Bob")]] and not res.errors
    assert (res.batch.table_ns, res.batch.table_name) == ("public", "person")


def _vectors():
    """(payload, expected row or error code) for the PERSON schema + an `any` and a number column."""
    schema = ('{"title":"a.b","type":"object","properties":{"b":{"type":"boolean"},"i":{"oneOf":[{"type":"null"},{"type":"integer"}]},'
              '"n":{"type":"number"},"s":{"oneOf":[{"type":"string"},{"type":"null"}]},"x":{"type":"object"},"r":{"type":"integer"}},"required":["r","n","s"]}')
    E = abi.ROWERR_ID
    nil = ("nil", None)
    V = []

    def ok(payload, b=nil, i=nil, n=nil, r=nil, s=nil, x=nil):
        V.append((payload, [b, i, n, r, s, x]))

    def err(payload, code):
        V.append((payload, E[code]))
    ok(b'{"r":1,"n":1.50,"s":"a"}', n=("jsonnum", b"1.50"), r=("int64", 1), s=("string", b"a"))
    ok(b' \t\r\n{"r":-0,"n":-1e+5,"s":null , "b" : true,"i":null,"x":null}trailing garbage', b=("bool", True), n=("jsonnum", b"-1e+5"), r=("int64", 0))
    ok(b'{"r":1,"r":2,"n":0,"s":"\\u0041\\n\\/\\ud83d\\ude00\\ud800x\\udc00"}', n=("jsonnum", b"0"), r=("int64", 2),
       s=("string", "A\n/\U0001F600�x�".encode()))
    ok(b'{"r":9223372036854775807,"n":1E5,"s":"\xff\xe2\x82"}', n=("jsonnum", b"1E5"), r=("int64", 9223372036854775807), s=("string", "���".encode()))
    ok(b'{"r":1,"n":1,"s":"","x":{"a":[1,2.50,{"b":"<\\u003e&\xe2\x80\xa8"}],"c":null},"\\u0062":false}', b=("bool", False), n=("jsonnum", b"1"), r=("int64", 1),
       s=("string", b""), x=("json", b'{"a":[1,2.50,{"b":"\\u003c\\u003e\\u0026\\u2028"}],"c":null}'))
    ok(b'{"r":1,"n":1,"s":"q","x":[ ]}', n=("jsonnum", b"1"), r=("int64", 1), s=("string", b"q"), x=("json", b"[]"))
    # json.Marshal of a map: keys ascending at every level, the last duplicate's value (the sorting emitter on the device)
    ok(b'{"r":1,"n":1,"s":"q","x":{"b":1,"a":{"z":[{"k":2,"j":1}],"y":"<"},"b":{"d":null,"c":[]},"\\u0061a":0, "A" : [1,{"b":1,"a":2}]}}',
       n=("jsonnum", b"1"), r=("int64", 1), s=("string", b"q"),
       x=("json", b'{"A":[1,{"a":2,"b":1}],"a":{"y":"\\u003c","z":[{"j":1,"k":2}]},"aa":0,"b":{"c":[],"d":null}}'))
    ok(b'{"r":1,"n":1,"s":"q","x":[{"b":1,"a":2,"b":3},{"\\u00e9":1,"e":2,"\xc3\xa9":3,"z":0}]}', n=("jsonnum", b"1"), r=("int64", 1), s=("string", b"q"),
       x=("json", '[{"a":2,"b":3},{"e":2,"z":0,"é":3}]'.encode()))
    ok(b'{"r":1,"n":1,"s":"q","x":"str"}', n=("jsonnum", b"1"), r=("int64", 1), s=("string", b"q"), x=("json", b'"str"'))
    ok(b'{"r":1,"n":1,"s":"q","x":12.0e1,"zzz":{"deep":[[[{}]]]}}', n=("jsonnum", b"1"), r=("int64", 1), s=("string", b"q"), x=("json", b"12.0e1"))
    deep = b"[" * 16 + b'{"b":1,"a":2}' + b"]" * 16   # unsorted keys 17 containers deep: beyond the sorting emitter
    V.append((b'{"r":1,"n":1,"s":"q","x":' + deep + b'}', ("fallback", [("nil", None), ("nil", None), ("jsonnum", b"1"), ("int64", 1), ("string", b"q"),
                                                                       ("json", b"[" * 16 + b'{"a":2,"b":1}' + b"]" * 16)])))
    ok(b'{"r":1,"n":1,"s":"q","x":' + b"[" * 15 + b'{"b":1,"a":2}' + b"]" * 15 + b'}', n=("jsonnum", b"1"), r=("int64", 1), s=("string", b"q"),
       x=("json", b"[" * 15 + b'{"a":2,"b":1}' + b"]" * 15))
    err(b'{"n":1,"s":"a"}', "SR_REQUIRED")
    err(b'null', "SR_REQUIRED")                      # nil map: every lookup misses
    err(b'{"r":1.0,"n":1,"s":"a"}', "SR_TYPE")       # Number.Int64: strconv.ParseInt syntax error
    err(b'{"r":9223372036854775808,"n":1,"s":"a"}', "SR_TYPE")
    err(b'{"r":"1","n":1,"s":"a"}', "SR_TYPE")
    err(b'{"r":1,"n":"1","s":"a"}', "SR_TYPE")
    err(b'{"r":1,"n":null,"s":"a"}', "SR_TYPE")      # required → not nullable → nil is a wrong type
    err(b'{"r":1,"n":1,"s":1}', "SR_TYPE")
    err(b'{"r":1,"n":1,"s":"a","b":"true"}', "SR_TYPE")
    err(b'{"r":1,"n":1,"s":"a","i":1e3}', "SR_TYPE")
    for bad in (b'', b'   ', b'{', b'[1]', b'"s"', b'12', b'nullx', b'{"r":1,"n":01,"s":"a"}', b'{"r":1,"n":1.,"s":"a"}', b'{"r":1,"n":1,"s":"a\x01"}',
                b'{"r":1,"n":1,"s":"\\x"}', b'{"r":1,"n":1,"s":"\\u12g4"}', b'{"r":1,"n":1,"s":"a",}', b'{"r":1 "n":1}', b"{'r':1}", b'{"r":+1,"n":1,"s":"a"}',
                b'{"r":1,"n":1e,"s":"a"}', b'{"r":1,"n":-,"s":"a"}', b'{"r":1,"n":1,"s":"a"', b'{"r":1,"n":1,"s":tru}', b'{"r":1,"n":.5,"s":"a"}', b'{r:1}'):
        err(bad, "JSON_SYNTAX")
    return schema, V


def _check_vectors(parse):
    schema, V = _vectors()
    o = confluent_sr.sr_json_options(3, schema)
    data, cm = abi.messages([b"\0\0\0\0\x03" + p for p, _ in V])
    res = parse(o, data, cm)
    rows = {int(res.batch.src_row[r]): row for r, row in enumerate(abi.batch_rows(res.batch))}
    errs = {int(e[0]): int(e[1]) for e in res.errors}
    FB = abi.ROWERR_ID["HOST_FALLBACK"]
    for k, (payload, exp) in enumerate(V):
        if isinstance(exp, tuple) and exp[0] == "fallback":  # the oracle has the row; the device may hand the frame to the host
            if errs.get(k) == FB:
                assert k not in rows
                continue
            exp = exp[1]
        if isinstance(exp, int):
            assert errs.get(k) == exp and k not in rows, (payload, abi.ROWERR.get(errs.get(k)), rows.get(k))
        else:
            assert k not in errs and rows.get(k) == [abi.norm_value(v) for v in exp], (payload, rows.get(k), abi.ROWERR.get(errs.get(k)))
    assert list(res.batch.part_id) == [k for k in range(len(V)) if k in rows]
    return errs


def test_oracle_vectors(oracle):
    _check_vectors(oracle.sr_json_parse)


def _framing_case():
    """DoBuf: frames inside one message; an error frame or a failing payload ends its message."""
    f = lambda sid, p: b"\0" + int(sid).to_bytes(4, "big") + p  # noqa: E731
    good = b'{"id":1,"name":"n"}'
    msgs = [
        f(7, good) + f(7, good) + f(8, b"{}") + f(7, good),   # 4 frames, one of another schema
        b"",                                                   # no frames
        b"\0\0\0\7",                                           # shorter than the prefix
        b"\1\0\0\0\7" + good,                                  # magic byte
        f(7, good) + f(7, b'{"id":"x"}') + f(7, good),         # the 2nd payload fails: the 3rd frame is never looked at
        f(7, good) + b"\0\0\7",                                # a good frame, then a short tail
        f(7, b""),                                             # empty payload: decode error
        f(7, good + b"  ") + b"\2\0\0\0\7" + good,             # the payload runs to the next 0 byte (trailing bytes after the
                                                               # object are ignored); that 0 is the next frame's magic: id 0x0000077b
    ]
    return msgs


def _check_framing(parse, frames_fn):
    msgs = _framing_case()
    data, cm = abi.messages(msgs)
    frames = frames_fn(data, cm)
    E = abi.ROWERR_ID
    assert [(f[0], f[3], f[4], f[5]) for f in frames] == [
        (0, 7, 0, 0), (0, 7, 0, 1), (0, 8, 0, 2), (0, 7, 0, 3), (2, 0, E["SR_SHORT"], 0), (3, 0, E["SR_MAGIC"], 0), (4, 7, 0, 0), (4, 7, 0, 1), (4, 7, 0, 2),
        (5, 7, 0, 0), (5, 0, E["SR_SHORT"], 1), (6, 7, 0, 0), (7, 7, 0, 0), (7, 0x77B, 0, 1)]
    for f in frames:
        if f[4] == 0:
            assert data[f[1] - 5] == 0 and int.from_bytes(data[f[1] - 4:f[1]], "big") == f[3]
    o = confluent_sr.sr_json_options(7, PERSON)
    res = parse(o, data, cm)
    assert [int(x) for x in res.batch.src_row] == [0, 1, 3, 6, 9, 12]
    assert [int(x) for x in res.batch.part_id] == [0, 0, 0, 4, 5, 7]
    assert sorted((int(e[0]), int(e[1])) for e in res.errors) == [(4, E["SR_SHORT"]), (5, E["SR_MAGIC"]), (7, E["SR_TYPE"]), (10, E["SR_SHORT"]),
                                                                   (11, E["JSON_SYNTAX"])]
    o2 = confluent_sr.sr_json_options(7, PERSON, report_frame_errors=False)
    res2 = parse(o2, data, cm)
    assert sorted((int(e[0]), int(e[1])) for e in res2.errors) == [(7, E["SR_TYPE"]), (11, E["JSON_SYNTAX"])]
    assert [int(x) for x in res2.batch.src_row] == [0, 1, 3, 6, 9, 12]


def test_oracle_framing(oracle):
    _check_framing(oracle.sr_json_parse, oracle.sr_frames)


def test_incorrect_magic_byte(oracle):
    """parser_test.go:105-133 TestIncorrectMagicByte: every first byte 1..254 yields one `_unparsed` item."""
    msgs = [bytes([i]) + b"\0\0\0{}" for i in range(1, 255)]
    data, cm = abi.messages(msgs)
    res = oracle.sr_json_parse(confluent_sr.sr_json_options(0, PERSON), data, cm)
    assert res.batch.nrows == 0 and [e[1] for e in res.errors] == [abi.ROWERR_ID["SR_MAGIC"]] * 254


# ---------------------------------------------------------------- GPU -------------
@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


@pytest.mark.gpu
def test_gpu_canon(tf):
    _check_canon(tf.sr_json_parse, tf.sr_frames)


@pytest.mark.gpu
def test_gpu_vectors(tf):
    errs = _check_vectors(tf.sr_json_parse)
    assert list(errs.values()).count(abi.ROWERR_ID["HOST_FALLBACK"]) == 1  # only the value nested beyond the sorting emitter


@pytest.mark.gpu
def test_gpu_framing(tf):
    _check_framing(tf.sr_json_parse, tf.sr_frames)
    msgs = [bytes([i]) + b"\0\0\0{}" for i in range(1, 255)]
    data, cm = abi.messages(msgs)
    res = tf.sr_json_parse(confluent_sr.sr_json_options(0, PERSON), data, cm)
    assert res.batch.nrows == 0 and [e[1] for e in res.errors] == [abi.ROWERR_ID["SR_MAGIC"]] * 254
    # empty batches
    data, cm = abi.messages([b"", b""])
    assert tf.sr_frames(data, cm) == [] and tf.sr_json_parse(confluent_sr.sr_json_options(0, PERSON), data, cm).batch.nrows == 0


def _rand_any_text(rng, depth=0):
    """A JSON value as text: objects with their keys in random order, repeated keys, escaped and non-ASCII keys."""
    t = rng.random()
    if depth >= 4 or t < 0.35:
        return ['1', '-2.50', '"s"', '"<&>"', 'null', 'true', '"\\u00e9"', '1e5', '""'][rng.integers(0, 9)]
    w = lambda: ["", " ", "\n "][rng.integers(0, 3)]  # noqa: E731
    if t < 0.6:
        return "[" + ",".join(w() + _rand_any_text(rng, depth + 1) + w() for _ in range(rng.integers(0, 4))) + "]"
    keys = ['"a"', '"b"', '"ab"', '"B"', '"\\u0061"', '"\u00e9"', '"\\u00e9"', '"z z"', '"<"', '""', '"a\\"q"']
    members = [keys[rng.integers(0, len(keys))] + w() + ":" + w() + _rand_any_text(rng, depth + 1) for _ in range(rng.integers(0, 6))]
    return "{" + ",".join(w() + m + w() for m in members) + "}"


def _random_payloads(rng, n):
    """Payloads for one schema drawn from a small grammar: every value kind, escapes, invalid UTF-8, white space,
    duplicate keys, unknown keys, unsorted objects under `any` (host fallback on the device), type and syntax errors."""
    schema = ('{"title":"db.t","type":"object","properties":{"an":{"type":"array"},"bo":{"oneOf":[{"type":"null"},{"type":"boolean"}]},'
              '"in":{"type":"integer"},"nu":{"oneOf":[{"type":"number"},{"type":"null"}]},"ob":{"type":"object"},"st":{"type":"string"},"\u00e9\u4e2d":{"type":"string"}},'
              '"required":["in","st"]}')
    strs = ['""', '"a"', '"plain text"', '"\\u00e9\\n\\t\\\\\\/\\""', '"\\ud83d\\ude00"', '"\\ud83dx"', '"\\udc00"', '"<&>\u2028"', '"\u00e9\u4e2d\U0001F600"', '"bad \udcff\udcfe"', '"\\u0000\\u001f"']
    nums = ["0", "-0", "12", "-9223372036854775808", "9223372036854775807", "9223372036854775808", "1.5", "1e3", "-1.25E-7", "123.2970700287221458280", "1E+400"]
    anys_ok = ['[]', '{}', '[1, "x", null, true]', '{"a":1,"b":{"c":[{}],"d":"<>"}}', '{"a":"\\u0062","ab":2,"b":3}', '"s"', '1.50', 'true', '[[[[1]]]]', '{ "k" : [ 1 , 2 ] }']
    anys_fb = ['{"b":1,"a":2}', '{"a":1,"a":2}', '[{"z":1,"y":2}]', '{"a":{"d":1,"c":2}}', '{"\\u0062":1,"a":2}']
    ws = ["", " ", "\n", "\t  "]
    out = []
    for _ in range(n):
        kind = rng.random()
        members = []
        if rng.random() < 0.95:
            members.append(('"in"', nums[rng.integers(0, 5)] if rng.random() < 0.9 else nums[rng.integers(0, len(nums))]))
        if rng.random() < 0.95:
            members.append(('"st"', strs[rng.integers(0, len(strs))] if rng.random() < 0.95 else "12"))
        if rng.random() < 0.7:
            members.append(('"nu"', nums[rng.integers(0, len(nums))] if rng.random() < 0.9 else "null"))
        if rng.random() < 0.5:
            members.append(('"bo"', ["true", "false", "null", '"true"'][rng.integers(0, 4)]))
        if rng.random() < 0.6:
            r = rng.random()
            members.append(('"ob"', anys_ok[rng.integers(0, len(anys_ok))] if r < 0.5 else anys_fb[rng.integers(0, len(anys_fb))] if r < 0.65 else _rand_any_text(rng)))
        if rng.random() < 0.4:
            members.append(('"an"', anys_ok[rng.integers(0, len(anys_ok))]))
        if rng.random() < 0.3:
            members.append(('"\u00e9\u4e2d"' if rng.random() < 0.5 else '"\\u00e9\\u4e2d"', strs[rng.integers(0, len(strs))]))
        if rng.random() < 0.3:
            members.append(('"unknown"', anys_fb[rng.integers(0, len(anys_fb))]))
        if rng.random() < 0.15 and members:
            members.append(members[rng.integers(0, len(members))])  # duplicate key: the last one wins
        order = rng.permutation(len(members))
        w = lambda: ws[rng.integers(0, len(ws))]  # noqa: E731
        body = "{" + ",".join(w() + members[i][0] + w() + ":" + w() + members[i][1] + w() for i in order) + "}"
        if kind < 0.04:
            body = body[:-1]
        elif kind < 0.06:
            body = body.replace(":", " ", 1)
        elif kind < 0.08:
            body = "null"
        elif kind < 0.10:
            body = body + " trailing"
        out.append(w().encode() + body.encode("utf-8", "surrogateescape"))
    return schema, out


def _compare_with_oracle(tf, oracle, rng, schema, payloads, upload, min_rows):
    frames = [b"\0" + (5 if rng.random() < 0.9 else 6).to_bytes(4, "big") + p for p in payloads]
    msgs, i = [], 0
    while i < len(frames):  # 1-3 frames per Kafka message
        k = int(rng.integers(1, 4))
        msgs.append(b"".join(frames[i:i + k]))
        i += k
    data, cm = abi.messages(msgs, offsets=np.arange(len(msgs)))
    assert tf.sr_frames(data, cm) == oracle.sr_frames(data, cm)
    o = confluent_sr.sr_json_options(5, schema)
    ref = oracle.sr_json_parse(o, data, cm)
    got = tf.sr_json_parse(o, tf.DeviceBuffer.upload(data) if upload else data, cm)
    FB = abi.ROWERR_ID["HOST_FALLBACK"]
    gerr = {e[0]: e[1] for e in got.errors}
    rerr = {e[0]: e[1] for e in ref.errors}
    fallback = {k for k, c in gerr.items() if c == FB}
    grows = {int(got.batch.src_row[r]): (row, int(got.batch.part_id[r])) for r, row in enumerate(abi.batch_rows(got.batch))}
    rrows = {int(ref.batch.src_row[r]): (row, int(ref.batch.part_id[r])) for r, row in enumerate(abi.batch_rows(ref.batch))}
    # a fallback frame ends its message on the device (the shim re-parses that message); every other message must agree
    fr = oracle.sr_frames(data, cm)
    dead_msgs = {fr[k][0] for k in fallback}
    live = [k for k in range(len(fr)) if fr[k][0] not in dead_msgs]
    assert {k: gerr.get(k) for k in live} == {k: rerr.get(k) for k in live}
    for k in live:
        assert grows.get(k) == rrows.get(k), (k, grows.get(k), rrows.get(k))
    for k in fallback:  # the oracle parsed it: a row whose `any` value needed Go's key ordering
        assert k in rrows
    assert set(grows) <= set(rrows)
    assert not fallback and len(grows) >= min_rows  # unsorted `any` objects are sorted on the device
    assert [c.dtype for c in got.batch.cols] == ["any", "boolean", "int64", "double", "any", "utf8", "utf8"]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 64, 1000, 20000])
def test_gpu_matches_oracle(tf, oracle, n):
    rng = np.random.default_rng(SEED0 + (4000 + n))
    schema, payloads = _random_payloads(rng, n)
    _compare_with_oracle(tf, oracle, rng, schema, payloads, n == 1000, 0 if n < 1000 else n // 4 + 1)


def _regular_payloads(rng, n):
    """What one producer emits — the same keys in the same order, compact or with blanks — with the values and the
    defects the tile parser has to decide or hand over exactly as parse_frame would: escapes, control bytes, numbers at
    the edge of int64, leading zeros, wrong types, nulls under required properties, missing / extra / repeated / reordered keys."""
    schema = _random_payloads(rng, 0)[0]
    strs = ['""', '"a"', '"plain text"', '"x' + "y" * 70 + '"', '"\\u00e9\\n\\t\\\\\\/\\""', '"\\ud83d\\ude00"', '"<&>\u2028"', '"\u00e9\u4e2d"', '"bad \\q"', '"\\u12"', '"tab\there"', '"\\\\"', '"q\\"q"']
    ints = ["0", "-0", "12", "-9223372036854775808", "9223372036854775807", "9223372036854775808", "-9223372036854775809", "99999999999999999999", "007", "1.5", "1e3", "-", '"5"', "null", "true"]
    nums = ["0", "-0", "12", "1.5", "1e3", "-1.25E-7", "123.2970700287221458280", "1E+400", "1.", ".5", "01.5", "null", '"x"']
    out = []
    for _ in range(n):
        usual = rng.random() < 0.85
        members = [('"in"', str(int(rng.integers(-10**9, 10**9))) if usual or rng.random() < 0.5 else ints[rng.integers(0, len(ints))]),
                   ('"st"', strs[rng.integers(0, 4)] if usual or rng.random() < 0.5 else strs[rng.integers(0, len(strs))]),
                   ('"nu"', str(int(rng.integers(0, 1000))) if usual or rng.random() < 0.5 else nums[rng.integers(0, len(nums))]),
                   ('"bo"', ["true", "false", "null"][rng.integers(0, 3)] if usual or rng.random() < 0.5 else ['"true"', "1", "tru"][rng.integers(0, 3)]),
                   ('"\u00e9\u4e2d"', strs[rng.integers(0, 3)]), ('"unknown"', ["1", '"x"', "null", "[1]"][rng.integers(0, 4 if not usual else 3)])]
        r = rng.random()
        if r < 0.01: members[0], members[2] = members[2], members[0]
        elif r < 0.02: del members[int(rng.integers(0, len(members)))]
        elif r < 0.03: members.append(members[int(rng.integers(0, len(members)))])
        elif r < 0.04: members.insert(2, ('"ob"', ['{"b":1,"a":2}', "[]", '"s"'][rng.integers(0, 3)]))
        sep = (",", ":") if rng.random() < 0.9 else [(", ", ": "), (" ,\t", " : ")][rng.integers(0, 2)]
        body = "{" + sep[0].join(k + sep[1] + v for k, v in members) + "}"
        r = rng.random()
        if r < 0.005: body = body[:-1]
        elif r < 0.01: body = body + " trailing"
        elif r < 0.015: body = " " + body
        out.append(body.encode("utf-8"))
    return schema, out


@pytest.mark.gpu
@pytest.mark.parametrize("regular", [False, True])
def test_gpu_is_generate_updates(tf, oracle, regular):
    """isGenerateUpdates (format_json.go:44-47, utils_json.go:57-63): every item is an Update whose ColumnNames hold the optional fields its
    payload holds — ABSENT cells (tfgpu_column.absent) for the others, where the insert form puts a nil.  Cells, kinds and which row lists
    what against the oracle, over the random payloads (the per-frame walker) and the regular ones (the tile parser)."""
    rng = np.random.default_rng(SEED0 + 4300 + regular)
    schema, payloads = (_regular_payloads if regular else _random_payloads)(rng, 3000)
    frames = [b"\0" + (5).to_bytes(4, "big") + p for p in payloads]
    data, cm = abi.messages(frames, offsets=np.arange(len(frames)))
    o = confluent_sr.sr_json_options(5, schema, is_generate_updates=True)
    ref = oracle.sr_json_parse(o, data, cm)
    got = tf.sr_json_parse(o, data, cm)
    assert {e[0]: e[1] for e in got.errors} == {e[0]: e[1] for e in ref.errors}
    gb, rb = got.batch, ref.batch
    assert gb.nrows == rb.nrows > 1000 and np.array_equal(gb.src_row, rb.src_row)
    assert gb.kind is not None and set(int(k) for k in gb.kind) == {abi.K_UPDATE} == set(int(k) for k in rb.kind)
    assert [c.name for c in gb.cols] == [c.name for c in rb.cols]
    some = 0
    for g, r in zip(gb.cols, rb.cols):
        ga = g.absent if g.absent is not None else np.zeros(gb.nrows, bool)
        ra = r.absent if r.absent is not None else np.zeros(rb.nrows, bool)
        assert np.array_equal(ga, ra), g.name
        some += int(ga.sum())
        assert (g.absent is None) == (not ra.any()), g.name   # a column every row lists carries no bitmap
    assert some > 500
    assert abi.batch_rows(gb) == abi.batch_rows(rb)
    # the same bytes as inserts: every row lists every column, the fields that are not there are nils
    ins = tf.sr_json_parse(confluent_sr.sr_json_options(5, schema), data, cm).batch
    assert all(c.absent is None for c in ins.cols) and abi.batch_rows(ins) == abi.batch_rows(gb)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [40, 3000, 20000])
def test_gpu_tile_path_matches_oracle(tf, oracle, n):
    rng = np.random.default_rng(SEED0 + (4100 + n))
    schema, payloads = _regular_payloads(rng, n)
    _compare_with_oracle(tf, oracle, rng, schema, payloads, n == 3000, n // 4)


@pytest.mark.gpu
def test_gpu_tile_path_chunk_border_sweep(tf, oracle):
    """Quotes, escaped quotes and backslash runs moved byte by byte across the chunk / thread borders of the tile parser's
    classification; the five prefix bytes between payloads carry quote and backslash bytes of their own (schema ids 0x22, 0x5C)."""
    rng = np.random.default_rng(SEED0 + 4200)
    schema = _random_payloads(rng, 0)[0]
    tails = ['', '\\"', '\\\\', '\\\\\\"', '\\' * 6, '\\' * 63 + '\\"', '\\' * 64, '\\' * 65 + 'n', '"', 'é', '\\u00e9', '\\q', ',"in":', '}{']
    payloads = []
    for pad in range(0, 140):
        for i, tail in enumerate(tails):
            payloads.append(('{"in":%d,"st":"%s","nu":%d}' % (pad * 16 + i, "x" * pad + tail, i)).encode("utf-8"))
    _compare_with_oracle(tf, oracle, rng, schema, payloads, False, len(payloads) // 3)
    # the same payloads behind schema ids whose bytes are '"' and '\\': not ours, and their prefix bytes must not leak into the next payload's parity
    o = confluent_sr.sr_json_options(5, schema)
    frames = []
    for k, pl in enumerate(payloads[:900]):
        sid = 5 if k % 3 else 0x2222225C
        frames.append(b"\0" + sid.to_bytes(4, "big") + pl)
    data, cm = abi.messages(frames, offsets=np.arange(len(frames)))
    ref = oracle.sr_json_parse(o, data, cm)
    got = tf.sr_json_parse(o, data, cm)
    assert sorted((e[0], e[1]) for e in got.errors) == sorted((e[0], e[1]) for e in ref.errors)
    assert abi.batch_rows(got.batch) == abi.batch_rows(ref.batch) and list(got.batch.src_row) == list(ref.batch.src_row)


@pytest.mark.gpu
def test_gpu_per_frame_path_cross_check():
    """sr_parse_tiles is the default; TFGPU_SR_TILES=0 sends every payload through parse_frame.  Both must pass the suite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TFGPU_SR_TILES="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_confluent_sr.py"), "-m", "gpu", "-q", "-x", "-k",
                        "(canon or vectors or framing or matches_oracle or parser_object) and not cross_check"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.mark.gpu
def test_gpu_tile_front_cross_check():
    """sr_parse_quick is the default front; TFGPU_SR_QUICK=0 sends every batch through sr_parse_tiles.  Both must pass the suite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TFGPU_SR_QUICK="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_confluent_sr.py"), "-m", "gpu", "-q", "-x", "-k",
                        "(canon or vectors or framing or matches_oracle or parser_object or border) and not cross_check"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]


# ---------------------------------------------------------------- oracle vs an independent JSON implementation ----
def _py_rand(rng, depth=0):
    t = rng.random()
    if depth >= 4 or t < 0.4:
        return [None, True, False, 0, -1, 12345678901234567890, 1.5, -2.25e-7, 1e300, "", "plain", "<b>&amp;", "tab\t\"q\"\\", "é\u4e2d\U0001F600", "\u2028\u2029\x7f\x01"][int(rng.integers(0, 15))]
    if t < 0.65:
        return [_py_rand(rng, depth + 1) for _ in range(int(rng.integers(0, 5)))]
    keys = ["a", "b", "B", "ab", "", "é", "z z", "<", "\u2028", "k\"q", "\U0001F600"]
    return {keys[int(rng.integers(0, len(keys)))]: _py_rand(rng, depth + 1) for _ in range(int(rng.integers(0, 6)))}


def _go_marshal(v):
    """json.Marshal of the decoded value, through Python's encoder: sorted keys, no spaces, Go's extra escapes."""
    t = json.dumps(v, sort_keys=True, separators=(",", ":"), ensure_ascii=False)
    return t.replace("<", "\\u003c").replace(">", "\\u003e").replace("&", "\\u0026").replace("\u2028", "\\u2028").replace("\u2029", "\\u2029").replace("\x7f", "\x7f")


def test_oracle_any_marshal_matches_python_json(oracle):
    """The oracle's Decoder + json.Marshal restatement against Python's json module on random documents (any key order,
    escapes or raw UTF-8 in the source): an independent implementation of the same published format."""
    rng = np.random.default_rng(SEED0 + 424242)
    schema = '{"title":"a.b","type":"object","properties":{"x":{"type":"object"}}}'
    docs, payloads = [], []
    for _ in range(3000):
        v = _py_rand(rng)
        src = json.dumps({"pad": 1, "x": v}, ensure_ascii=bool(rng.random() < 0.5), separators=[(",", ":"), (", ", ": "), (" , ", " : ")][int(rng.integers(0, 3))])
        docs.append(v); payloads.append(src.encode("utf-8"))
    data, cm = abi.messages([b"\0\0\0\0\x02" + p for p in payloads])
    res = oracle.sr_json_parse(confluent_sr.sr_json_options(2, schema), data, cm)
    assert not res.errors and res.batch.nrows == len(docs)
    col = res.batch.cols[0]
    for i, v in enumerate(docs):
        if v is None:
            assert not col.is_valid(i)
        else:
            assert col.get_bytes(i).decode("utf-8") == _go_marshal(v), (i, payloads[i])


# ---------------------------------------------------------------- the parser object (DoBatch over mixed schema ids) ----
def _check_parser_object(engine):
    """TestClient through ConfluentSrParser.do_batch: all five JSON schemas and the two protobuf messages in ONE batch."""
    reg = {int(k): v for k, v in G["schemas"].items()}
    nonempty = [(i, m) for i, m in enumerate(MSGS) if m]
    proto = [b"\0\0\0\0\x05" + b"\x08\x01\x12\x03abc", b"\0\0\0\0\x06\x08\x01"]  # protobuf schema ids: the stock parser's frames
    data, cm = abi.messages([m for _, m in nonempty] + proto, offsets=[i for i, _ in nonempty] + [900, 901])
    items = confluent_sr.ConfluentSrParser(reg).do_batch(engine, data, cm)
    rows = [it for it in items if "kind" in it]
    assert len(rows) == len(G["items"]) and [it["msg"] for it in items if it.get("fallback")] == [len(nonempty), len(nonempty) + 1]
    for it, exp in zip(rows, G["items"]):
        assert (it["schema"], it["table"], it["names"], nonempty[it["msg"]][0]) == (exp["schema"], exp["table"], exp["names"], exp["lsn"])
    assert [it["index"] for it in rows[:3]] == [0, 1, 0]  # the first message holds two frames
    # a failing frame ends its message across schema ids: frame 1 (schema 0) is bad, frame 2 (schema 3) is never looked at
    f = lambda sid, p: b"\0" + int(sid).to_bytes(4, "big") + p  # noqa: E731
    d2, m2 = abi.messages([f(0, b'{"col0_number":1}') + f(0, b'{"col0_number":"x"}') + f(3, b'{"col0_bool":true}'), f(3, b'{"col0_bool":false}'), b"\x07junk"])
    got = confluent_sr.ConfluentSrParser(reg).do_batch(engine, d2, m2)
    assert [(x.get("table"), x.get("unparsed"), x["msg"]) for x in got] == [("table_iCMRA", None, 0), (None, abi.ROWERR_ID["SR_TYPE"], 0), ("table_VlgTe", None, 1),
                                                                         (None, abi.ROWERR_ID["SR_MAGIC"], 2)]


def test_oracle_parser_object(oracle):
    _check_parser_object(oracle)


@pytest.mark.gpu
def test_gpu_parser_object(tf):
    _check_parser_object(tf)
