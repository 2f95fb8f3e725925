"""Pins the oracle's restatement of the generic JSON parser (oracle/ora_jsonparse.c,
SURVEY §8 a17) to the reference's canon files: tests/canon/parser {json,mdb} canon and
pkg/parsers/generic canon (TestParserNumberTypes, TestBase64Unpack).  CPU only."""
import base64
import json

import pytest

from transferia_amd import abi
from util import golden


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


def marshal_like_go(v):
    """[gotype, value] → what ChangeItem.MarshalJSON + json.load give for it"""
    g, x = v
    if g == "nil":
        return None
    if g == "string":
        return x.decode("utf-8")
    if g == "bytes":
        return base64.b64encode(x).decode()
    if g in ("json", "jsonnum"):
        return json.loads(x)
    return x


@pytest.mark.parametrize("case", golden("json_parser.json")["cases"], ids=lambda c: c["name"])
def test_json_parser_canon(oracle, case):
    opts, fields, data, msgs = case_inputs(case)
    r = oracle.json_parse(opts, fields, data, msgs)
    assert r.nrows == len(case["rows"])
    names = [c.name for c in r.schema.cols]
    for i, exp in enumerate(case["rows"]):
        assert names == exp["names"]
        got = r.rows[i]
        if "values" in exp:
            for c, (g, e) in enumerate(zip(got, exp["values"])):
                if fields.cols[c].dtype == "datetime" if c < len(fields.cols) else False:
                    # free-form datetime strings go through github.com/araddon/dateparse (unpinned): not restated
                    assert g == ["nil", None] and r.lines[i][0] == oracle.JL_UNRESTATED
                    continue
                assert abi.norm_value(g) == abi.norm_value(e), (case["name"], i, names[c])
        elif "marshalled_text" in exp:  # "string" (bytes) columns hold Go strings here: the canon shows their text
            assert [x.decode("utf-8") if g in ("string", "bytes") else x for g, x in got] == exp["marshalled_text"], (case["name"], i)
        else:
            assert [marshal_like_go(v) for v in got] == exp["marshalled"], (case["name"], i)
    sts = {ln[0] for ln in r.lines}
    assert sts <= {oracle.JL_ROW, oracle.JL_UNRESTATED}


def test_fastjson_edge_cases(oracle):
    """Behaviour fixed by the call sites in generic_parser.go:672-731 and fastjson's grammar."""
    fields = abi.Schema.of([["i8", "int8"], ["u16", "uint16"], ["i64", "int64"], ["u64", "uint64"], ["d", "double"], ["b", "boolean"],
                            ["s", "utf8"], ["y", "string"], ["k", "int32", True]])
    opts = abi.json_options(topic="t")
    lines = [
        b'{"k": 1, "i8": 300, "u16": -1, "i64": 9223372036854775807, "u64": 18446744073709551615, "d": 1.5, "b": true, "s": "x", "y": [1, {"a" : "b\\n"}]}',
        b'{"k": "0x10", "i8": "12", "u16": "65535", "d": "1e3", "b": "T", "s": 12.50, "y": true}',
        b'{"k": 1.5}',                 # GetInt of "1.5" → 0
        b'{"k": null}',                # nil key → unparsed
        b'{"k": "zz"}',                # ParseInt error on a key → unparsed
        b'{"i8": "999", "k": 2}',      # range error on a non-key column → nil
        b'{"k": 3,}',                  # syntax
        b'[1,2]',                      # not an object → skipped
        b'{}',                         # empty → skipped
        b'  {"k" : 4 , "s" : "a\\u00e9\\ud83d\\ude00\\q" }  ',
        b'{"k": 5, "k": 6}',           # last duplicate wins
        b'{"k": 7} x',                 # unexpected tail
        b'{"k": 1e2}',                 # GetInt("1e2") → 0
        b'{"k": 8, "d": 123456789012345678901234567890}',
        b'{"k": 9, "d": 0.000001234e-5, "b": 1, "i64": -9223372036854775808}',
    ]
    r = oracle.json_parse(opts, fields, b"\r\n".join(lines) + b"\n\n")
    st = [(ln[0], ln[1], ln[2]) for ln in r.lines]
    J = oracle
    assert st == [(J.JL_ROW, 0, -1), (J.JL_ROW, 0, -1), (J.JL_ROW, 0, -1), (J.JL_UNPARSED, 14, 8), (J.JL_UNPARSED, 13, 8), (J.JL_ROW, 0, -1),
                  (J.JL_UNPARSED, 12, -1), (J.JL_SKIPPED, 0, -1), (J.JL_SKIPPED, 0, -1), (J.JL_ROW, 0, -1), (J.JL_ROW, 0, -1),
                  (J.JL_UNPARSED, 12, -1), (J.JL_ROW, 0, -1), (J.JL_ROW, 0, -1), (J.JL_ROW, 0, -1)]
    rows = r.rows
    assert rows[0] == [["int8", 44], ["uint16", 0], ["int64", 9223372036854775807], ["uint64", 18446744073709551615], ["float64", 1.5], ["bool", True],
                       ["string", b"x"], ["string", b'[1,{"a":"b\\n"}]'], ["int32", 1]]
    assert rows[1] == [["int8", 12], ["uint16", 65535], ["nil", None], ["nil", None], ["float64", 1000.0], ["bool", True], ["string", b"12.50"],
                       ["string", b"true"], ["int32", 16]]
    assert rows[2][8] == ["int32", 0]
    assert rows[3][0] == ["nil", None] and rows[3][8] == ["int32", 2]
    assert rows[4][6] == ["string", "aé\U0001F600\\q".encode("utf-8")] and rows[4][8] == ["int32", 4]
    assert rows[5][8] == ["int32", 6]
    assert rows[6][8] == ["int32", 0]
    assert rows[7][4] == ["float64", 1.2345678901234568e29]
    # fastfloat is not correctly rounded: float64(1234)/1e9 * Pow10(-5), two roundings (1.2340000000000001e-11, not 1.234e-11)
    assert rows[8][4] == ["float64", (1234 / 1e9) * 1e-5] and rows[8][5] == ["bool", False] and rows[8][2] == ["int64", -9223372036854775808]


def test_fastfloat_best_effort(oracle):
    import ctypes as C
    L = oracle.lib()
    for s, want in [("0", 0.0), ("-0", -0.0), ("1.5", 1.5), ("123456789012345678", 1.2345678901234568e17), ("1234567890123456789", 1.2345678901234568e18),
                    ("0.1", 0.1), ("1e5", 1e5), ("1E-5", 1e-5), ("1.", 0.0), (".5", 0.5), ("-.5", -0.5), ("1e", 0.0), ("1e+", 0.0), ("1x", 0.0),
                    ("12345678765432.23456765432", 12345678765432.234), ("1e400", float("inf")), ("inf", float("inf")), ("-Infinity", float("-inf")),
                    ("0.0000001", 1e-7), ("123.456e2", 12345.6)]:
        got = L.ora_fastfloat_parse_best_effort(s.encode(), len(s))
        assert got == want and (str(got) == str(want)), s


def test_any_and_rest_marshal_match_python_json(oracle):
    """go_marshal_any / the `_rest` map of the oracle against Python's json module on random valid documents
    (UseNumbersInAny: numbers keep their text): an independent implementation of the same published format."""
    import numpy as np
    from test_confluent_sr import _go_marshal, _py_rand
    rng = np.random.default_rng(515151)
    fields = abi.Schema.of([["k", "int32", True], ["a", "any"]])
    opts = abi.json_options(topic="t", use_numbers_in_any=True, add_rest=True)
    docs, lines = [], []
    for n in range(2000):
        v, extra = _py_rand(rng, 1), {("u%d" % i): _py_rand(rng, 2) for i in range(int(rng.integers(0, 4)))}
        if not isinstance(v, (dict, list)):
            v = [v]
        doc = dict(extra); doc["k"] = n; doc["a"] = v
        keys = list(doc)
        rng.shuffle(keys)
        lines.append(json.dumps({k: doc[k] for k in keys}, ensure_ascii=bool(rng.random() < 0.5)).encode("utf-8"))
        docs.append((v, extra))
    r = oracle.json_parse(opts, fields, b"\n".join(lines))
    assert r.nrows == len(docs) and all(ln[0] == oracle.JL_ROW for ln in r.lines)
    for (v, extra), row in zip(docs, r.rows):
        assert row[1][1].decode("utf-8") == _go_marshal(v)
        assert row[2][1].decode("utf-8") == _go_marshal(extra)


def test_lookup_complex_rules(oracle):
    """lookupComplex + parseJSON (parsers/generic/lookup.go:10-59) through makeChangeItem (generic_parser.go:323-347): the
    nested-path columns of the metrika canon, as small cases, against the reference's rules (the device runs the same lines against this oracle in tests/test_gpu_json.py::test_lookup_complex_on_the_device)."""
    # [name, type, key, path, original_type, required]
    fields = abi.Schema.of([["k", "int32", True], ["s", "utf8", False, "ev.a.b"], ["t", "utf8", False, "ev/c"], ["n", "int32", False, "ev.num"], ["q", "utf8", False, "ev.a.zz"]])
    opts = abi.json_options(topic="t", format="tskv")
    lines = [
        b'k=1\tev={"a":{"b":"x"},"c":"y","num":"12"}',            # a map inside the decoded map, '/' as the separator, ParseVal of a string
        b'k=2\tev={"a":"{\\"b\\":\\"inner\\"}","c":null}',           # a string that holds JSON again is parsed again; null → nil
        b'k=3\tev={\\\\"a\\\\": {\\\\"b\\\\": \\\\"dbl\\\\"}, \\\\"c\\\\": \\\\"e\\\\"}',   # \\\\" → \\" does not help, dropping every backslash does
        b'k=4\tev=not json',                                         # parseJSON fails three times: the cells stay nil
        b'k=5\tev=null',                                             # a nil map: "unable to get field"
        b'k=6\tother=1',                                             # no top-level value at all
        b'k=7\tev={"a":{"b":5},"num":"zz"}',                        # a number at the end: not restated; ParseVal error on a plain column: _unparsed
        b'k=8\tev={"a":{"b":"x"}} trailing',                        # invalid character after top-level value → retries → error
    ]
    data, msgs = abi.messages([b"\n".join(lines)], [0], [0])
    r = oracle.json_parse(opts, fields, data, msgs)
    rows = {int(r.rows[i][0][1]): r.rows[i] for i in range(r.nrows)}
    assert rows[1][1:5] == [["string", b"x"], ["string", b"y"], ["int32", 12], ["nil", None]]
    assert rows[2][1:3] == [["string", b"inner"], ["nil", None]]
    assert rows[3][1:3] == [["string", b"dbl"], ["string", b"e"]]
    for k in (4, 5, 6, 8):
        assert rows[k][1:5] == [["nil", None]] * 4, k
    assert 7 not in rows  # "ParseVal error n raw(zz)": an _unparsed item whatever the column's flags
    # a required nested column that cannot be found makes the line an _unparsed item
    req = abi.Schema.of([["k", "int32", True], ["s", "utf8", False, "ev.a.b", "", True]])
    r2 = oracle.json_parse(opts, req, data, msgs)
    got = sorted(int(r2.rows[i][0][1]) for i in range(r2.nrows))
    assert got == [1, 2, 3, 7], got   # 4, 5, 6, 8: lookup errors → _unparsed
    assert oracle.JL_UNRESTATED in {ln[0] for ln in r2.lines}  # line 7: a number at the end of the path (ParseVal of a float64) is not restated
