"""Parity of the HIP NDJSON ingest (tfgpu_json_parse: GenericParser{Format:"json"} —
bufio.ScanLines, fastjson grammar, Unmarshal, ParseVal, makeChangeItem's key rules on
device) with the oracle, through the C ABI; needs an MI355X.

What is compared, per non-empty line (ordinal):
  - device row            ⇒ the oracle has a fully restated row with identical values;
  - device `_unparsed`    ⇒ same code (JSON_SYNTAX / PARSE_VAL / NIL_KEY), column and message;
  - device vanishes       ⇒ the oracle skipped it ({} or not an object);
  - device HOST_FALLBACK  ⇒ the line is handed to the stock Go code (counted, bounded per test).
"""
import json
import math
import random
import struct

import numpy as np
import pytest

from transferia_amd import abi
from util import golden
import os as _os

SEED0 = int(_os.environ.get("TFGPU_TEST_SEED", "0"))  # 0 = the committed seeds; other values: soak runs (TFGPU_TEST_SEED=n bash tools/gpu_visit.sh tests TAG)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


def f64_bits(x):
    return struct.pack("<d", float(x))


def expected_json_text(v):
    """json.Marshal of the Go value an `any` column holds, for the forms the device keeps."""
    g, x = v
    if g == "bool":
        return b"true" if x else b"false"
    if g == "string":
        return b'"' + x + b'"'  # the device only keeps strings encoding/json leaves untouched
    if g in ("jsonnum", "json"):
        return x
    if g == "float64":  # encoding/json floatEncoder: 'f' unless < 1e-6 or >= 1e21, then 'e' with e-0X → e-X
        from oracle import oracle as ora
        a = abs(x)
        if a != 0 and (a < 1e-6 or a >= 1e21):
            t = ora.fmt_float(x, "g", 64)
            return (t[:-2] + t[-1] if t[-4:-1] in ("e-0",) else t).encode()
        return ora.fmt_float(x, "f", 64).encode()
    raise AssertionError("device kept an `any` value of Go type %s" % g)


def compare(tf, oracle, opts, fields, data, msgs=None, ctx="", max_fallback=0):
    ref = oracle.json_parse(opts, fields, data, msgs)
    db, errs = tf.json_parse(opts, fields, data, msgs, max_errors=1 << 18)
    out = db.download()
    names = [c.name for c in out.cols]
    assert names == [c.name for c in ref.schema.cols], ctx
    assert [c.dtype for c in out.cols] == [c.dtype for c in ref.schema.cols], ctx
    assert names == [c.name for c in tf.json_result_schema(opts, fields).cols], ctx
    gerr = {e[0]: e for e in errs}
    assert len(gerr) == len(errs), ctx
    src = out.src_row if out.src_row is not None else np.arange(out.nrows, dtype=np.int32)
    row_of = {int(o): i for i, o in enumerate(src)}
    assert len(row_of) == out.nrows and list(src) == sorted(src), ctx
    nfb = 0
    for ordn, (st, code, col, msg, idx, row) in enumerate(ref.lines):
        where = "%s line %d" % (ctx, ordn)
        if ordn in gerr:
            assert ordn not in row_of, where
            _, gcode, gmsg, gcol = gerr[ordn]
            if gcode == "HOST_FALLBACK":
                nfb += 1
                assert st != oracle.JL_SKIPPED, where
                continue
            assert st == oracle.JL_UNPARSED, (where, gcode, st)
            assert (gcode, gcol, gmsg) == (abi.ROWERR[code], col, msg), where
            continue
        if ordn not in row_of:
            assert st == oracle.JL_SKIPPED, (where, st)
            continue
        assert st == oracle.JL_ROW, (where, st)
        i = row_of[ordn]
        assert int(out.part_id[i]) == msg, where
        exp = ref.rows[row]
        for c, col_ in enumerate(out.cols):
            got, want = col_.pyvalue(i), exp[c]
            if want[0] == "nil" or got[0] == "nil":
                assert got[0] == want[0] == "nil", (where, names[c], got, want)
            elif col_.repr == abi.R_JSON:
                assert got[1] == expected_json_text(want), (where, names[c], got, want)
            elif want[0] == "float64":
                assert got[0] == "float64" and (f64_bits(got[1]) == f64_bits(want[1]) or (math.isnan(got[1]) and math.isnan(want[1]))), (where, names[c], got, want)
            else:
                assert abi.norm_value(got) == abi.norm_value(want), (where, names[c], got, want)
    assert len(gerr) + out.nrows <= len(ref.lines), ctx
    assert set(gerr) | set(row_of) <= set(range(len(ref.lines))), ctx
    assert nfb <= max_fallback, (ctx, "host fallbacks", nfb, sorted(o for o, e in gerr.items() if e[1] == "HOST_FALLBACK")[:80])
    return out, errs, nfb


def case_inputs(case):
    o = case["options"]
    opts = abi.json_options(add_rest=o.get("add_rest", False), add_dedupe_keys=o.get("add_dedupe_keys", False),
                            null_keys_allowed=o.get("null_keys_allowed", False), use_numbers_in_any=o.get("use_numbers_in_any", False),
                            unpack_bytes_base64=o.get("unpack_bytes_base64", False), topic=o.get("topic", ""), partition=o.get("partition", ""),
                            unescape_string_values=o.get("unescape_string_values", False), format=o.get("format", "json"))
    fields = abi.Schema.of(case["fields"])
    vals = [m["value_latin1"].encode("latin-1") for m in case["messages"]]
    wts = [abi.parse_rfc3339(m["write_time"])[0] * 10**9 for m in case["messages"]]
    data, msgs = abi.messages(vals, [m["offset"] for m in case["messages"]], wts)
    return opts, fields, data, msgs


@pytest.mark.parametrize("case", golden("json_parser.json")["cases"], ids=lambda c: c["name"])
def test_reference_parser_canon(tf, oracle, case):
    """tests/canon/parser {json, mdb} and pkg/parsers/generic canon inputs."""
    from transferia_amd import lib
    opts, fields, data, msgs = case_inputs(case)
    if case["options"].get("unpack_bytes_base64"):
        with pytest.raises(lib.TfgpuError) as e:
            tf.json_parse(opts, fields, data, msgs)
        assert e.value.code == lib.ERR_UNSUPPORTED
        return
    # free-form datetime strings, float64 inside `any`, unknown keys under AddRest: host lines
    compare(tf, oracle, opts, fields, data, msgs, case["name"], max_fallback=len(case["rows"]))


def test_lookup_complex_on_the_device(tf, oracle):
    """Nested ColSchema.Path (lookupComplex + parseJSON, parsers/generic/lookup.go:10-59) in the per-line parser: what the device
    decides it decides like the oracle (tests/test_json_oracle.py::test_lookup_complex_rules holds the same lines against the
    reference's rules), the rest — a decoded string that differs from its bytes on the way, a retry that could change the text, a
    number / bool / container at the end — is handed to the host and named so."""
    fields = abi.Schema.of([["k", "int32", True], ["s", "utf8", False, "ev.a.b"], ["t", "utf8", False, "ev/c"], ["n", "int32", False, "ev.num"], ["q", "utf8", False, "ev.a.zz"]])
    lines = [
        b'k=1\tev={"a":{"b":"x"},"c":"y","num":"12"}',
        b'k=2\tev={"a":"{\\"b\\":\\"inner\\"}","c":null}',           # a string that holds escaped JSON: the decoded string is not its bytes — host
        b'k=3\tev={\\\\"a\\\\": {\\\\"b\\\\": \\\\"dbl\\\\"}, \\\\"c\\\\": \\\\"e\\\\"}',   # the retries rewrite backslashes — host
        b'k=4\tev=not json', b'k=5\tev=null', b'k=6\tother=1',
        b'k=7\tev={"a":{"b":5},"num":"zz"}',                        # a number at the end: host
        b'k=8\tev={"a":{"b":"x"}} trailing',
        b'k=9\tev={"a":{"b":"q\\"uote \\\\ \\/ \\n\\t end"},"c":"caf\xc3\xa9 \xe2\x82\xac","num":"-7"}',   # simple escapes, UTF-8
        b'k=10\tev={"a":{"b":"one","b":"two"},"c":"","num":"0x1F"}',           # the last duplicate wins; strconv.ParseInt base prefixes
        b'k=11\tev= {"a" : { "zz" : "far" , "b" : null } , "c" : "sp ace" }  ',
        b'k=12\tev={"a":{"b":"\\u0041"},"c":"x"}',                            # a \\u escape at the end: host
        b'k=13\tev={"a":{"b":"bad \xff utf8"},"c":"x"}',                     # U+FFFD replacement: host
        b'k=14\tev={"a":[1,2],"c":"x"}',                                       # unexpected value type on the way: nil
        b'k=15\tev={"a":{"b":"x"},"c":"y","num":"99999999999"}',                # ParseVal error behind lookupComplex: _unparsed
        b'k=16\tev={"a":"{}","c":"y"}',                                         # a plain string on the way is parsed again: no such field
        b'k=17\tev={"a":"[1]","c":"y"}', b'k=18\tev={"a":{"b":"x"},"c":"y",}', b'k=19\tev=[{"a":1}]', b'k=20\tev="str"', b'k=21\tev=',
        b'k=22\tev={"a":{"b":"x"},"c":"y"}\tev={"a":{"b":"second"},"c":"wins"}',  # the last duplicate of the top-level key wins too
        b'k=23\tev={"\\u0061":{"b":"x"},"c":"y"}',                             # a member name with an escape: host
    ]
    data, msgs = abi.messages([b"\n".join(lines)], [0], [0])
    for nka in (False, True):
        for fl in (fields, abi.Schema.of([["k", "int32", True], ["s", "utf8", False, "ev.a.b", "", True], ["t", "utf8", False, "ev/c"]])):
            opts = abi.json_options(topic="t", format="tskv", null_keys_allowed=nka)
            out, errs, nfb = compare(tf, oracle, opts, fl, data, msgs, "lookup nka=%s" % nka, max_fallback=3)
            assert nfb == 3  # lines 7 (a number at the end), 13 (ill-formed UTF-8 to decode), 23 (an escaped member name)
    # Format json: the top-level value must be a JSON string without escapes (the Go string is then its bytes)
    jl = [b'{"k": 1, "ev": "{}"}', b'{"k": 2, "ev": "not json"}', b'{"k": 3, "ev": null}', b'{"k": 4}', b'{"k": 5, "ev": {"a": {"b": "x"}}}', b'{"k": 6, "ev": "{\\"a\\":{\\"b\\":\\"x\\"}}"}', b'{"k": 7, "ev": 12}',
          b'{"k": 8, "ev": "[1, 2]"}', b'{"k": 9, "ev": "null"}']
    data, msgs = abi.messages([b"\n".join(jl)], [0], [0])
    out, errs, nfb = compare(tf, oracle, abi.json_options(topic="t"), fields, data, msgs, "lookup json", max_fallback=3)
    assert nfb == 3  # lines 5, 6, 7


def test_lookup_complex_random(tf, oracle):
    """Random nested documents under a TSKV key: every line the device keeps agrees with the oracle, cell for cell."""
    import json as pyjson
    import random
    rng = random.Random(SEED0 + 4242) if "SEED0" in globals() else random.Random(4242)
    fields = abi.Schema.of([["k", "int32", True], ["s", "utf8", False, "ev.a.b"], ["t", "utf8", False, "ev/c"], ["n", "int64", False, "ev.a.num"], ["u", "any", False, "ev.d.e.f"]])
    words = ["x", "", "long value with spaces", "caf\u00e9", "q\"uote", "back\\slash", "tab\there", "12", "-5", "0x10", "{}", "null", "a/b", "\u20ac"]

    def leaf():
        r = rng.random()
        return rng.choice(words) if r < 0.8 else None if r < 0.9 else rng.randrange(-5, 5) if r < 0.94 else rng.choice([True, [1], {"z": 1}])  # (non-strings at a path's end are the host's)

    lines = []
    for i in range(400):
        doc = {}
        if rng.random() < 0.9:
            a = {}
            if rng.random() < 0.8: a["b"] = leaf()
            if rng.random() < 0.7: a["num"] = rng.choice(["1", "-2", "999999999999", "zz", "12", "0", 7, None])
            if rng.random() < 0.2: a["b "] = "near miss"
            doc["a"] = a if rng.random() < 0.9 else rng.choice(["str", 5, None, [a]])
        if rng.random() < 0.8: doc["c"] = leaf()
        if rng.random() < 0.5: doc["d"] = {"e": {"f": leaf()}} if rng.random() < 0.8 else {"e": leaf()}
        if isinstance(doc.get("a"), dict) and rng.random() < 0.25:  # a string that holds the inner object's JSON: parsed again on the way
            doc["a"] = pyjson.dumps(doc["a"], ensure_ascii=rng.random() < 0.5)
        text = pyjson.dumps(doc, ensure_ascii=rng.random() < 0.5, separators=rng.choice([(",", ":"), (", ", ": ")]))
        if rng.random() < 0.2:   # the metrika logs' double escaping: not JSON as written, JSON after parseJSON's first retry
            text = text.replace('\\"', '\\\\"')
        if rng.random() < 0.05: text = text.replace('"', '\\"')   # … or only after every backslash is dropped
        if rng.random() < 0.05: text = text[:-1]
        if rng.random() < 0.05: text += " x"
        lines.append(("k=%d\tev=%s" % (i, text)).encode("utf-8"))
    data, msgs = abi.messages([b"\n".join(lines)], [0], [0])
    out, errs, nfb = compare(tf, oracle, abi.json_options(topic="t", format="tskv"), fields, data, msgs, "lookup random", max_fallback=400)
    assert out.nrows > 100 and nfb < 200, (out.nrows, nfb)


EDGE_FIELDS = [["i8", "int8"], ["u16", "uint16"], ["i64", "int64"], ["u64", "uint64"], ["d", "double"], ["b", "boolean"],
               ["s", "utf8"], ["y", "string"], ["k", "int32", True], ["t", "datetime"], ["a", "any"]]

EDGE_LINES = [
    b'{"k": 1, "i8": 300, "u16": -1, "i64": 9223372036854775807, "u64": 18446744073709551615, "d": 1.5, "b": true, "s": "x", "y": [1, {"a" : "b\\n"}]}',
    b'{"k": "0x10", "i8": "12", "u16": "65535", "d": "1e3", "b": "T", "s": 12.50, "y": true}',
    b'{"k": 1.5}',
    b'{"k": null}',
    b'{"k": "zz"}',
    b'{"i8": "999", "k": 2}',
    b'{"k": 3,}',
    b'[1,2]',
    b'{}',
    b'  {"k" : 4 , "s" : "a\\u00e9\\ud83d\\ude00\\q" }  ',
    b'{"k": 5, "k": 6}',
    b'{"k": 7} x',
    b'{"k": 1e2}',
    b'{"k": 8, "d": 123456789012345678901234567890}',
    b'{"k": 9, "d": 0.000001234e-5, "b": 1, "i64": -9223372036854775808}',
    # strings: every escape, broken escapes, surrogates
    b'{"k": 10, "s": "q\\"b\\\\s\\/\\b\\f\\n\\r\\t", "y": "\\u0041\\u00e9\\u20ac\\ud83d\\ude00"}',
    b'{"k": 11, "s": "\\ud83d", "y": "\\ud83dx\\ude00"}',
    b'{"k": 12, "s": "\\ud83d\\u0041", "y": "\\udc00\\ud83d"}',
    b'{"k": 13, "s": "\\u12", "y": "\\uZZZZ"}',
    b'{"k": 14, "s": "\\x41\\", "y": "tail\\\\"}',
    b'{"k": 15, "s": "ends with bs\\\\", "y": ""}',
    b'{"k": 16, "s": "tab\there", "y": "raw \x01 ctrl"}',
    # numbers through fastfloat
    b'{"k": 17, "d": -0}', b'{"k": 18, "d": 1.}', b'{"k": 19, "d": .5}', b'{"k": 20, "d": -.5}', b'{"k": 21, "d": 1e}',
    b'{"k": 22, "d": 1e+}', b'{"k": 23, "d": 1e400}', b'{"k": 24, "d": 1e-400}', b'{"k": 25, "d": inf}', b'{"k": 26, "d": -Infinity}',
    b'{"k": 27, "d": nan}', b'{"k": 28, "d": NaN}', b'{"k": 29, "d": 123456789012345678}', b'{"k": 30, "d": 1234567890123456789}',
    b'{"k": 31, "d": 0.1234567890123456}', b'{"k": 32, "d": 0.12345678901234567}', b'{"k": 33, "d": 123.456e2}', b'{"k": 34, "d": 1e308}',
    b'{"k": 35, "d": 1e-308}', b'{"k": 36, "d": 1-2+3}', b'{"k": 37, "d": -}', b'{"k": 38, "d": +1}', b'{"k": 39, "d": 1e301}',
    b'{"k": 40, "d": "1.5"}', b'{"k": 41, "d": "1e22"}', b'{"k": 42, "d": "  1"}', b'{"k": 43, "d": "inf"}', b'{"k": 44, "d": "+nan"}',
    b'{"k": 45, "d": "0x1p-2"}', b'{"k": 46, "d": "1_0"}', b'{"k": 47, "d": "123456789012345678901"}', b'{"k": 48, "d": ""}',
    b'{"k": 49, "d": "1e"}', b'{"k": 50, "d": ".5"}', b'{"k": 51, "d": "5."}', b'{"k": 52, "d": "1e23"}', b'{"k": 53, "d": "9007199254740993"}',
    b'{"k": 54, "d": "-0"}', b'{"k": 55, "d": "0.000001"}', b'{"k": 56, "d": "1e-22"}', b'{"k": 57, "d": "1e-23"}', b'{"k": 58, "d": "123456789e30"}',
    # integers
    b'{"k": 59, "i8": -129, "u16": 65536, "i64": 12345678901234567890, "u64": 1.0}',
    b'{"k": 60, "i8": 127, "u16": 65535, "i64": -0, "u64": 18446744073709551616}',
    b'{"k": 61, "i8": "-128", "u16": "0b101", "i64": "0o17", "u64": "0xFFFFFFFFFFFFFFFF"}',
    b'{"k": 62, "i8": "1_0", "u16": "0x_f", "i64": "-9223372036854775809", "u64": "-1"}',
    b'{"k": 63, "i8": "+5", "u16": "+5", "i64": "", "u64": "018"}',
    b'{"k": 64, "i8": true, "u16": [1], "i64": {"a": 1}, "u64": false}',
    b'{"k": "2147483647"}', b'{"k": "2147483648"}', b'{"k": 2147483648}', b'{"k": -2147483649}', b'{"k": "-2147483648"}',
    # booleans
    b'{"k": 65, "b": false}', b'{"k": 66, "b": "TRUE"}', b'{"k": 67, "b": "tRUE"}', b'{"k": 68, "b": 0}', b'{"k": 69, "b": [true]}', b'{"k": 70, "b": "0"}',
    # text columns: raw tokens and compacted containers
    b'{"k": 71, "s": {"a" : [1, 2 , {"b" : "c d"}], "e" : null}, "y": [ ]}',
    b'{"k": 72, "s": { }, "y": [ [ ], { } ]}',
    b'{"k": 73, "s": {"a\\"b": 1}, "y": {"plain": {"a\\"b": 1}}}',
    b'{"k": 74, "s": -1.50e+3, "y": false}',
    b'{"k": 75, "s": nan, "y": -inf}',
    # datetime
    b'{"k": 76, "t": 1600000000}', b'{"k": 77, "t": -5.9}', b'{"k": 78, "t": 1e30}', b'{"k": 79, "t": true}', b'{"k": 80, "t": "2020-01-01"}', b'{"k": 81, "t": null}',
    b'{"k": 82, "t": [1]}',
    # any
    b'{"k": 83, "a": true}', b'{"k": 84, "a": "plain text"}', b'{"k": 85, "a": "  {\\"x\\":1}"}', b'{"k": 86, "a": "<b>"}', b'{"k": 87, "a": 12}',
    b'{"k": 88, "a": {"z": 1}}', b'{"k": 89, "a": null}', b'{"k": 90, "a": "null"}', b'{"k": 91, "a": "caf\xc3\xa9"}',
    # grammar
    b'{"k": 92, "x": [1, 2,]}', b'{"k": 93, "x": [1 2]}', b'{"k": 94, "x": {"a" 1}}', b'{"k": 95, "x": {"a": }}', b'{"k": 96 "x": 1}', b'{"k": 97, x: 1}',
    b'{"k": 98, "x": tru}', b'{"k": 99, "x": nul}', b'{"k": 100, "x": falsE}', b'{"k": 101, "x": "unterminated}', b'{"k": 102, "x": [}', b'{"k": 103',
    b'{"k": 104, "x": truex}', b'{"k": 105}}', b'{"k": 106, "x": [[[[[[[[[[1]]]]]]]]]]}', b'"just a string"', b'12', b'null', b'nan', b'   ', b'{"k": 107, "x": ]}',
    b'{"k\\u0041": 108}', b'{"\\u006b": 109}', b'{"k": 110, "K": 1, "k ": 2}', b'{"": 1, "k": 111}', b'{"k": 112, "x": "\\"}', b'{"k": 113, "x": "\\\\"}',
    b'{"k":114,"x":1}\r', b'\t{"k":115}\t', b'{"k": 116, "x": I}', b'{"k": 117, "x": +Inf}', b'{"k": 118, "x": -nanx}', b'{"k": 119, "x": --1}',
]


def test_grammar_and_coercion_edge_cases(tf, oracle):
    fields = abi.Schema.of(EDGE_FIELDS)
    for use_numbers in (False, True):
        opts = abi.json_options(topic="some/topic@x", use_numbers_in_any=use_numbers)
        data = b"\n".join(EDGE_LINES) + b"\n"
        out, errs, nfb = compare(tf, oracle, opts, fields, data, None, "edge un=%s" % use_numbers, max_fallback=30)
        assert out.table_name == "some_topic_x"
        # the device must decide the bulk of these lines itself


def test_float_parsing(tf, oracle):
    """fastfloat.ParseBestEffort (JSON numbers) and strconv.ParseFloat (JSON strings) into a double column:
    exact path, Eisel-Lemire, truncated mantissas; only half-way / subnormal inputs may go to the host."""
    rng = random.Random(SEED0 + (11))
    toks = []
    for n in range(12000):
        nd = rng.choice([1, 2, 5, 9, 15, 16, 17, 18, 19, 20, 25])
        digits = "".join(rng.choice("0123456789") for _ in range(nd)).lstrip("0") or "0"
        form = rng.random()
        if form < 0.3:
            t = digits
        elif form < 0.6:
            k = rng.randrange(0, len(digits) + 1)
            t = (digits[:k] or "0") + "." + (digits[k:] or "0")
        else:
            k = rng.randrange(1, len(digits) + 1)
            t = digits[:k] + ("." + digits[k:] if k < len(digits) else "") + rng.choice(["e", "E"]) + rng.choice(["", "+", "-"]) + str(rng.choice([0, 1, 5, 22, 23, 37, 38, 100, 290, 300, 301, 308, 309, 320, 330, 400]))
        if rng.random() < 0.3:
            t = "-" + t
        toks.append(t)
    toks += [repr(rng.uniform(-1e6, 1e6)) for _ in range(3000)] + [repr(rng.random() * 10 ** rng.randrange(-300, 300)) for _ in range(3000)]
    toks += ["4.9e-324", "2.2250738585072014e-308", "1.7976931348623157e308", "1.7976931348623159e308", "9007199254740993", "9007199254740992.5",
             "0.1", "0.3", "1e23", "8.41e21", "1e-320", "123456789012345678901234567890e-10", "0." + "0" * 400 + "1", "1" + "0" * 400]
    fields = abi.Schema.of([["k", "int32", True], ["d", "double"]])
    opts = abi.json_options(topic="t")
    for quoted in (False, True):
        lines = [b'{"k": %d, "d": %s}' % (i, (('"%s"' % t) if quoted else t).encode()) for i, t in enumerate(toks)]
        out, errs, nfb = compare(tf, oracle, opts, fields, b"\n".join(lines), None, "floats quoted=%s" % quoted, max_fallback=len(lines) // 20)


def test_deep_nesting(tf, oracle):
    fields = abi.Schema.of([["k", "int32", True], ["s", "utf8"]])
    opts = abi.json_options(topic="t")
    lines = []
    for depth in (1, 2, 30, 62, 63, 64, 65, 100, 298, 299, 300, 301, 400):
        lines.append(b'{"k": %d, "s": ' % depth + b"[" * depth + b"1" + b"]" * depth + b"}")
        lines.append(b'{"k": %d, "s": ' % depth + b'{"a":' * depth + b"1" + b"}" * depth + b"}")
        lines.append(b'{"k": %d, "s": ' % depth + b"[" * depth + b"1" + b"]" * (depth - 1) + b"}")
    compare(tf, oracle, opts, fields, b"\n".join(lines), None, "depth", max_fallback=len(lines))


def test_key_and_required_rules(tf, oracle):
    lines = [b'{"a": 1, "b": 2, "c": 3}', b'{"a": "x", "b": 2, "c": 3}', b'{"a": 1, "b": "x", "c": 3}', b'{"a": 1, "b": 2, "c": "x"}', b'{"b": 2, "c": 3}',
             b'{"a": 1, "c": 3}', b'{"a": 1, "b": 2}', b'{"a": null, "b": null, "c": null}', b'{"a": "x", "b": "x", "c": "x"}', b'{"c": "x", "b": null, "a": 1}',
             b'{"z": 1}']
    data = b"\n".join(lines)
    for nka in (False, True):
        for spec in ([["a", "int32", True], ["b", "int32", False, "", "", True], ["c", "int32"]],
                     [["a", "int32"], ["b", "int32", True], ["c", "int32", False, "", "", True]],
                     [["a", "int32"], ["b", "int32"], ["c", "int32"]]):
            opts = abi.json_options(topic="t", null_keys_allowed=nka)
            compare(tf, oracle, opts, abi.Schema.of(spec), data, None, "rules nka=%s %s" % (nka, spec))


def test_paths_duplicates_and_aux_columns(tf, oracle):
    # two columns reading one key, ColPath different from ColumnName, IgnoreColumnPaths, name clashes with aux columns
    spec = [["id", "int64", True, "ID"], ["id2", "int64", False, "ID"], ["name", "utf8", False, "n"], ["_rest", "utf8"], ["_idx", "int32"]]
    lines = [b'{"ID": 1, "n": "a", "_rest": "r", "_idx": 5}', b'{"ID": 2, "id": 3, "name": "x", "n": "y"}', b'{"ID": 4, "extra": 1}', b'{"id": 5}']
    vals = [b"\n".join(lines[:2]) + b"\n", b"", lines[2], b"\n\n" + lines[3] + b"\r\n"]
    data, msgs = abi.messages(vals, [10, 11, 12, 13], [1_600_000_000_123_456_789, 0, -1, 5])
    for kw in ({}, {"add_rest": True}, {"add_dedupe_keys": True}, {"add_rest": True, "add_dedupe_keys": True, "mark_dedupe_keys_as_system": True},
               ):
        opts = abi.json_options(topic="t", partition='{"partition":3,"topic":"t"}', **kw)
        compare(tf, oracle, opts, abi.Schema.of(spec), data, msgs, "aux %s" % kw, max_fallback=4)
    # IgnoreColumnPaths re-keys colTypeMap / known by ColumnName; the lookup stays ColPath (generic_parser.go:351, 1217-1233)
    plain = [["id", "int64", True], ["name", "utf8"], ["extra", "boolean"]]
    opts = abi.json_options(topic="t", ignore_column_paths=True, add_rest=True, add_dedupe_keys=True)
    compare(tf, oracle, opts, abi.Schema.of(plain), data, msgs, "ignore paths", max_fallback=4)
    # a key typed through another column's DataType is refused, not approximated
    from transferia_amd import lib
    for bad, kw in (([["id", "int64", True, "ID"], ["id2", "utf8", False, "ID"]], {}), (spec, {"ignore_column_paths": True})):
        with pytest.raises(lib.TfgpuError) as e:
            tf.json_parse(abi.json_options(topic="t", **kw), abi.Schema.of(bad), data, msgs)
        assert e.value.code == lib.ERR_UNSUPPORTED


def test_messages_and_scanlines(tf, oracle):
    fields = abi.Schema.of([["k", "int32", True], ["s", "utf8"]])
    rng = random.Random(SEED0 + (7))
    vals, k = [], 0
    for m in range(300):
        parts = []
        for _ in range(rng.choice([0, 1, 1, 1, 2, 5])):
            k += 1
            parts.append(b'{"k": %d, "s": "m%d"}' % (k, m))
            parts.append(rng.choice([b"\n", b"\r\n", b"\n\n", b"\r\n\r\n", b"\n\r\n"]))
        if parts and rng.random() < 0.5:
            parts.pop()  # last line without a newline
        if rng.random() < 0.1:
            parts.insert(0, rng.choice([b"\n", b"\r\n", b"\r", b" \n"]))
        vals.append(b"".join(parts))
    data, msgs = abi.messages(vals, list(range(100, 100 + len(vals))), [i * 1_000_000_007 - 5 for i in range(len(vals))])
    opts = abi.json_options(topic="t", add_dedupe_keys=True, partition='{"partition":1,"topic":"t"}')
    out, errs, _ = compare(tf, oracle, opts, fields, data, msgs, "messages")
    assert out.nrows > 300
    # one message, no trailing newline; empty input; newline-only input
    compare(tf, oracle, opts, fields, b'{"k": 1}', None, "single")
    compare(tf, oracle, opts, fields, b"", None, "empty")
    compare(tf, oracle, opts, fields, b"\n\r\n\n", None, "blank")


def rand_json_value(rng, depth=0):
    t = rng.random()
    if t < 0.25:
        return rng.choice([0, 1, -1, 7, 255, 256, 65535, 65536, -32769, 2**31, -2**31 - 1, 2**53 + 1, 2**63 - 1, -2**63, 2**64 - 1, 2**64,
                           1.5, -0.25, 1e10, 1.25e-7, 123456.789, 0.1, 1e22, 1e23, 3.141592653589793])
    if t < 0.45:
        return rng.choice(["", "x", "hello world", "12", "-7", "0x1F", "1e3", "true", "F", "2.50", "né", "tab\t", 'q"q', "back\\slash", "<&>", "€", "\U0001F600",
                           "  7", "1_000", "9223372036854775808", "0.30000000000000004"])
    if t < 0.55:
        return rng.choice([True, False, None])
    if depth < 3 and t < 0.75:
        return [rand_json_value(rng, depth + 1) for _ in range(rng.randrange(0, 4))]
    if depth < 3:
        return {rng.choice(["a", "b", "c d", "é", 'q"']): rand_json_value(rng, depth + 1) for _ in range(rng.randrange(0, 4))}
    return 1


def mutate(rng, line: bytes) -> bytes:
    if not line:
        return line
    i = rng.randrange(len(line))
    op = rng.random()
    if op < 0.4:
        return line[:i] + line[i + 1:]
    if op < 0.8:
        return line[:i] + bytes([rng.choice(b'{}[]",:\\ 0a.-e')]) + line[i:]
    return line[:i] + bytes([rng.choice(b'{}[]",:\\ 0a.-e')]) + line[i + 1:]


def test_random_documents_and_mutations(tf, oracle):
    rng = random.Random(SEED0 + (20260923))
    keys = ["i8", "u16", "i64", "u64", "d", "b", "s", "y", "k", "t", "a", "zz"]
    lines = []
    for n in range(6000):
        doc = {}
        for key in rng.sample(keys, rng.randrange(1, len(keys))):
            doc[key] = rand_json_value(rng)
        if rng.random() < 0.7:
            doc["k"] = rng.choice([n, str(n), n % 100])
        sep = rng.choice([(",", ":"), (", ", ": "), (" , ", " : ")])
        line = json.dumps(doc, ensure_ascii=rng.random() < 0.5, separators=sep).encode("utf-8")
        if rng.random() < 0.25:
            line = mutate(rng, line)
        if b"\n" in line:
            continue
        lines.append(line)
    fields = abi.Schema.of(EDGE_FIELDS)
    for kw in ({}, {"use_numbers_in_any": True, "null_keys_allowed": True}, {"add_rest": True, "add_dedupe_keys": True},
               {"add_rest": True, "use_numbers_in_any": True}):
        opts = abi.json_options(topic="t", **kw)
        out, errs, nfb = compare(tf, oracle, opts, fields, b"\n".join(lines), None, "fuzz %s" % kw, max_fallback=len(lines))
        assert out.nrows > 200 and len(errs) > 200


def test_flat_lines_are_device_resident(tf, oracle):
    """The config-3 shape (flat Kafka JSON, one object per message): no line may go to the host."""
    rng = random.Random(SEED0 + (3))
    fields = abi.Schema.of([["watchid", "int64", True], ["title", "utf8"], ["eventtime", "datetime"], ["clientip", "int32"], ["counterid", "uint32"],
                            ["isrefresh", "boolean"], ["ratio", "double"], ["url", "string"]])
    vals = []
    for n in range(20000):
        doc = {"watchid": rng.randrange(-2**63, 2**63), "title": rng.choice(["", "Заголовок", "plain title", 'with "quotes" and \\ and /', "tab\there"]),
               "eventtime": rng.randrange(1_370_000_000, 1_380_000_000), "clientip": rng.randrange(-2**31, 2**31), "counterid": rng.randrange(0, 2**32),
               "isrefresh": rng.random() < 0.5, "ratio": rng.choice([0.5, 1.25, 100.0, 3.0e-3, 12345.678, 1e15, 7]), "url": "http://example.com/?q=%d" % n}
        vals.append(json.dumps(doc, ensure_ascii=rng.random() < 0.5).encode("utf-8"))
    data, msgs = abi.messages(vals, list(range(len(vals))), [1_700_000_000_000_000_000 + i for i in range(len(vals))])
    opts = abi.json_options(topic="hits", add_dedupe_keys=True, add_rest=True, partition='{"partition":0,"topic":"hits"}')
    out, errs, nfb = compare(tf, oracle, opts, fields, data, msgs, "flat", max_fallback=0)
    assert out.nrows == len(vals) and not errs
    # the same bytes HBM-resident
    buf = tf.DeviceBuffer.upload(data)
    db, errs2 = tf.json_parse(opts, fields, buf, msgs)
    out2 = db.download()
    assert out2.nrows == out.nrows and not errs2
    for a, b in zip(out.cols, out2.cols):
        if a.repr in abi.VAR_REPRS:
            assert np.array_equal(a.offsets, b.offsets) and bytes(a.data) == bytes(b.data)
        else:
            assert bytes(a.values) == bytes(b.values)


TSKV_LINES = [
    b"tskv\tk=1\ti8=-5\tu16=65535\ti64=0x10\tu64=18446744073709551615\td=1.5e3\tb=T\ts=plain text\ty=raw \\n \"q\" <&>\ta=word",
    b"k=2\ts=\ty==\ta=a=b=c",                     # empty value, '=' inside a value (SplitN 2)
    b"k=3\tk=4",                                    # the last duplicate wins
    b"tskv\tnofields",                              # no key=value field at all: skipped
    b"\t\tk=5\t\t",                               # empty fields
    b"=nokey\tk=6",                                 # empty key
    b"k=zz",                                         # ParseInt error on the key column → _unparsed
    b"i8=1",                                         # key column absent → nil key → _unparsed
    b"k=7\ti8=999\tu16=-1\tb=maybe\td=abc",        # errors on non-key columns → nil
    b"k=8\tt=1600000000",                           # datetime from a string: dateparse → host
    b"k=9\ta={\"x\":1}",                            # any: a JSON object in a string → host
    b"k=10\ts=caf\xc3\xa9 \xff\ta=caf\xc3\xa9",      # non-ASCII bytes: text keeps them, any goes to the host
    b"k=11\ts=back\\slash\\\ty=\\=\\t",              # backslashes are content when nothing unescapes
    b"k = 12\tk=12",                                # "k " is another key
    b"k=13\tunknown=1",
    b"k=14\r",                                      # ScanLines drops the trailing \r
]


def test_tskv_edge_cases(tf, oracle):
    fields = abi.Schema.of(EDGE_FIELDS)
    data = b"\n".join(TSKV_LINES) + b"\n"
    for opts in (abi.json_options(topic="t", format="tskv"), abi.json_options(topic="t", format="tskv", add_dedupe_keys=True, add_rest=True),
                 abi.json_options(topic="t", format="tskv", unescape_string_values=True, null_keys_allowed=True)):
        compare(tf, oracle, opts, fields, data, None, "tskv", max_fallback=6)
    vals = [TSKV_LINES[0] + b"\n" + TSKV_LINES[1], b"", TSKV_LINES[2] + b"\r\n\r\n" + TSKV_LINES[4]]
    d2, m2 = abi.messages(vals, [7, 8, 9], [10**18, 0, 5])
    compare(tf, oracle, abi.json_options(topic="a/b@c", format="tskv", add_dedupe_keys=True), fields, d2, m2, "tskv messages", max_fallback=0)


def test_tskv_random_lines(tf, oracle):
    rng = random.Random(SEED0 + (77))
    fields = abi.Schema.of(EDGE_FIELDS)
    keys = ["k", "i8", "u16", "i64", "u64", "d", "b", "s", "y", "a", "t", "zz", ""]
    vals = ["0", "1", "-1", "127", "128", "65535", "0x1F", "1_000", "1e3", "1.25", "-0", "inf", "true", "FALSE", "x", "", "a b", "\\n", "\\", "\\q", "tab\\tend",
            "=", "a=b", "\"", "<>", "caf\u00e9", "9223372036854775807", "18446744073709551615", "-9223372036854775809", " 1", "1 "]
    lines = []
    for _ in range(5000):
        nf = rng.randint(0, 8)
        fs = []
        for _ in range(nf):
            r = rng.random()
            if r < 0.1:
                fs.append(rng.choice(["tskv", "", "novalue"]))
            else:
                fs.append(rng.choice(keys) + "=" + rng.choice(vals))
        if rng.random() < 0.8:
            fs.append("k=%d" % rng.randint(-5, 10**6))
        lines.append("\t".join(fs).encode("utf-8"))
    data = b"\n".join(lines)
    for opts in (abi.json_options(topic="t", format="tskv"), abi.json_options(topic="t", format="tskv", unescape_string_values=True, add_rest=True)):
        compare(tf, oracle, opts, fields, data, None, "tskv random", max_fallback=len(lines))


REST_LINES = [
    b'{"k": 1, "zz": 1, "aa": "x<y", "mm": null, "b2": [1, {"q": 2, "p": 1}], "cc": 1.50, "dd": true}',
    b'{"k": 2, "u": 1, "u": 2, "t2": {"b": 1, "a": {"d": 1, "c": 2}, "b": 3}}',          # duplicates: the last one is the map's value
    b'{"k": 3}',                                                                            # no unknown key: {}
    b'{"k": 4, "s": "known", "extra": "tab\\there \\"q\\""}',                                 # an escaped string value: host
    b'{"k": 5, "caf\\u00e9": 1}', b'{"k": 6, "caf\xc3\xa9": 1}',                              # keys that are not plain ASCII: host
    b'{"k": 7, "n1": 1e400, "n2": 0.1}', b'{"k": 8, "n": nan}',                            # floats json.Marshal refuses (without UseNumbers)
    b'{"k": 9, ' + b", ".join(b'"x%02d": %d' % (i, i) for i in range(16)) + b'}',           # 16 unknown keys: the most the device sorts
    b'{"k": 10, ' + b", ".join(b'"x%02d": %d' % (i, i) for i in range(17)) + b'}',          # 17: host
    b'{"k": 11, "_rest": 1}', b'{"k": 12, "_idx": 5}',                                      # aux names are typed as their columns: host
    b'{"zz": 1}',                                                                           # key column absent: _unparsed, whatever _rest holds
    b'{"k": 13, "deep": ' + b"[" * 17 + b"]" * 17 + b'}',                                   # nested beyond the emitter: host
    b'{"k": 14, "e": "", "o": {}, "a2": [], "neg": -0, "big": 123456789012345678901234567890}',
]


def test_rest_column_on_device(tf, oracle):
    """AddRest: `_rest` = json.Marshal of the members no column knows.  Device-resident for plain keys and values."""
    fields = abi.Schema.of(EDGE_FIELDS)
    data = b"\n".join(REST_LINES)
    for kw in ({"add_rest": True}, {"add_rest": True, "use_numbers_in_any": True, "add_dedupe_keys": True}):
        out, errs, nfb = compare(tf, oracle, abi.json_options(topic="t", **kw), fields, data, None, "rest %s" % kw, max_fallback=9)
        rest = out.col("_rest")
        got = {int(out.col("k").values[i]): rest.get_bytes(i) for i in range(out.nrows)}
        assert got[1] == (b'{"aa":"x\\u003cy","b2":[1,{"p":1,"q":2}],"cc":1.50,"dd":true,"mm":null,"zz":1}' if kw.get("use_numbers_in_any")
                          else b'{"aa":"x\\u003cy","b2":[1,{"p":1,"q":2}],"cc":1.5,"dd":true,"mm":null,"zz":1}')
        assert got[2] == b'{"t2":{"a":{"c":2,"d":1},"b":3},"u":2}' and got[3] == b"{}"
        assert 9 in got and 10 not in got and 4 not in got and 5 not in got


def _tile_lines(rng, n):
    """Lines a producer's serializer would emit — one key order, typed values — with the perturbations the tile parser must
    hand to the per-line parser (or decide exactly as it would): another key order, a missing / extra / repeated key, blanks,
    nested values, escapes, numbers at the edges of their types, malformed text."""
    def val(key):
        r = rng.random()
        if rng.random() < 0.85:  # the usual value of the column
            usual = {"i8": lambda: rng.randrange(-128, 128), "u16": lambda: rng.randrange(0, 65536), "k": lambda: rng.randrange(-10**6, 10**6),
                     "i64": lambda: rng.randrange(-2**63, 2**63), "u64": lambda: rng.randrange(0, 2**63), "d": lambda: rng.choice([rng.randrange(-10**6, 10**6), rng.random() * 1000]),
                     "b": lambda: rng.random() < 0.5, "t": lambda: rng.randrange(1_300_000_000, 1_400_000_000), "a": lambda: rng.choice(["D", "5", "ok"]),
                     "s": lambda: rng.choice(["", "plain", "Заголовок", "x" * rng.randrange(0, 60) + '"q"']), "y": lambda: "http://example.com/?q=%d" % rng.randrange(10**6)}
            return usual[key]()
        if key in ("i8", "u16", "k"):
            return rng.choice([0, 1, -1, 127, 128, 255, 65535, 65536, -32769, 2**31 - 1, 2**31, -2**31 - 1, rng.randrange(-10**6, 10**6)])
        if key in ("i64", "u64"):
            return rng.choice([0, -1, 10**17, 10**18 - 1, 10**18, 2**63 - 1, 2**63, 2**64 - 1, -2**63, -2**63 - 1, 10**19 - 1, 10**19, 10**20, rng.randrange(-2**63, 2**63)])
        if key == "d":
            return rng.choice([0, 7, -7, 10**15 - 1, 10**15, 10**16, 0.5, -1.25, 1e5, 1.5e-7, 3.0e300, 12345.678])
        if key == "b":
            return rng.choice([True, False, 0, 1, "true", None]) if r < 0.3 else rng.random() < 0.5
        if key == "t":
            return rng.choice([0, 1_370_000_000, 10**15, 10**16, -5, 1.5, "2020-01-01", None]) if r < 0.3 else rng.randrange(1_300_000_000, 1_400_000_000)
        if key == "a":
            return rng.choice(["D", "5", "", " {x", "nul", "a<b", "q\"q", "é", 5, 1.5, True, None, [1, 2], {"z": 1}, "  n"])
        pad = "x" * rng.randrange(0, 60)  # moves the following bytes across the 16 / 48 / 64-byte chunk borders
        return rng.choice(["", "plain", pad, pad + '"', pad + "\\", pad + "\\\\" + '"', pad + "\\" * rng.randrange(1, 70), "tab\there", "Заголовок", "[{,:}]", "a,b:c", pad + "é"])
    keys = ["k", "i8", "u16", "i64", "u64", "d", "b", "s", "y", "t", "a"]
    out = []
    for i in range(n):
        doc = [(key, val(key)) for key in keys]
        r = rng.random()
        if r < 0.01: doc[2], doc[5] = doc[5], doc[2]
        elif r < 0.02: del doc[rng.randrange(1, len(doc))]
        elif r < 0.03: doc.insert(rng.randrange(len(doc) + 1), ("extra%d" % rng.randrange(3), rng.choice([1, "x", None, [1], 2.5])))
        elif r < 0.04: doc.append(doc[rng.randrange(len(doc))])
        elif r < 0.045: doc = []
        sep = (",", ":") if rng.random() < 0.9 else rng.choice([(", ", ": "), (" ,\t", " : "), (",", ": ")])
        body = sep[0].join(json.dumps(key) + sep[1] + json.dumps(v, ensure_ascii=rng.random() < 0.5, separators=sep) for key, v in doc)
        line = ("{" + rng.choice(["", "", " "]) + body + rng.choice(["", "", " "]) + "}").encode("utf-8")
        r = rng.random()
        if r < 0.005: line = line[:rng.randrange(1, len(line))]                     # cut short
        elif r < 0.01: line = line.replace(b":", b"::", 1)
        elif r < 0.015: line = line.replace(b'"', b"'", 1)
        elif r < 0.017: line = line[:-1] + b',"pad":"' + b"z" * 26000 + b'"}'         # longer than a tile
        elif r < 0.027: line = line.replace(b"7", b"007", 1)
        elif r < 0.037: line = line.replace(b"1", b"-0", 1)
        if b"\n" not in line:
            out.append(line)
    return out


def test_tile_path_against_the_per_line_grammar(tf, oracle):
    """json_parse_tiles decides only what it is sure of and lists the rest: whatever mix of lines it gets, the batch must
    be the oracle's, line for line."""
    rng = random.Random(SEED0 + 4242)
    fields = abi.Schema.of(EDGE_FIELDS)
    lines = _tile_lines(rng, 8000)
    for kw in ({}, {"use_numbers_in_any": True, "null_keys_allowed": True}, {"add_rest": True, "add_dedupe_keys": True}):
        opts = abi.json_options(topic="t", **kw)
        out, errs, nfb = compare(tf, oracle, opts, fields, b"\n".join(lines), None, "tiles %s" % kw, max_fallback=len(lines))
        assert out.nrows > 1000
    # one object per message (no newline between them), the bench's shape
    vals = lines[:1500]
    data, msgs = abi.messages(vals, list(range(len(vals))), [1_700_000_000_000_000_000 + i for i in range(len(vals))])
    compare(tf, oracle, abi.json_options(topic="t", add_dedupe_keys=True), fields, data, msgs, "tiles as messages", max_fallback=len(vals))


def test_tile_path_chunk_border_sweep(tf, oracle):
    """Quotes, escaped quotes and backslash runs moved byte by byte across the 16-byte chunk, the 64-byte thread and the tile
    borders of the tile parser's classification (the escaped-quote carry and the in-string parity cross lanes and waves there)."""
    fields = abi.Schema.of(EDGE_FIELDS)
    tails = ['', '\\"', '\\\\', '\\\\\\"', '\\' * 6, '\\' * 63 + '\\"', '\\' * 64, '\\' * 65 + 'n', '"', 'é', '\\u00e9', ',"k":', '}{']
    lines = []
    for pad in range(0, 140):
        for i, tail in enumerate(tails):
            body = "x" * pad + tail
            # the string is written by hand (json.dumps would re-escape): only well-formed escape sequences stay valid JSON
            lines.append(('{"k":%d,"s":"%s","y":"%s","i8":%d}' % (pad * 16 + i, body, "z" * (pad % 7), i)).encode("utf-8"))
    opts = abi.json_options(topic="t")
    out, errs, nfb = compare(tf, oracle, opts, fields, b"\n".join(lines), None, "border sweep", max_fallback=len(lines))
    assert out.nrows > len(lines) // 2
    vals = lines[:600]
    data, msgs = abi.messages(vals, list(range(len(vals))), [0] * len(vals))
    compare(tf, oracle, opts, fields, data, msgs, "border sweep as messages", max_fallback=len(vals))


def test_per_line_path_cross_check():
    """The tile parser is the default; with TFGPU_JSON_TILES=0 every line takes the per-line parser.  Both must pass the suite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TFGPU_JSON_TILES="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_json.py"), "-m", "gpu", "-q", "-x", "-k",
                        "(edge or random or flat or canon or rules or aux or messages or tile_path or float or rest) and not cross_check"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]


def test_tile_path_cross_check():
    """json_parse_quick is the default front; with TFGPU_JSON_QUICK=0 every batch goes through json_parse_tiles (the form batches
    with float64 columns or > 128 members still take).  Both must pass the suite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TFGPU_JSON_QUICK="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_json.py"), "-m", "gpu", "-q", "-x", "-k",
                        "(edge or random or flat or canon or rules or aux or messages or tile_path or float or rest) and not cross_check"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]


def test_wave_path_cross_check():
    """The wave-cooperative parser (TFGPU_JSON_WAVEPATH=1: simdjson-style stage 1 on the scalar unit, members on lanes)
    and the per-line parser are two independent implementations of the same grammar: the edge-case, fuzz and flat
    suites must pass under both."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TFGPU_JSON_WAVEPATH="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_json.py"), "-m", "gpu", "-q", "-x", "-k",
                        "(edge or random or flat or canon or rules or aux or messages) and not cross_check"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
