#!/usr/bin/env python3
"""Build tests/golden/*.json from the reference's own golden vectors.

Run in the build container (where /root/reference exists):

    python tools/extract_golden.py

It reads the reference's canon files (expected outputs, byte-exact) and pairs
them with the inputs of the reference tests that produced them.  Go test
inputs cannot be executed here (no Go toolchain), so they are transcribed as
DATA below, each block citing the reference test file:line it mirrors.  The
expected values are never typed by hand when a canon file holds them.

Value encoding: [gotype, value]
  gotype in int8..int64, uint8..uint64, int (Go int = int64), float32, float64,
  bool, string, bytes (latin-1 text or list of ints), time (RFC3339Nano, any
  zone), duration (ns), nil, json (pre-marshalled any).
"""
import json
import os
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
REG = REF + "/pkg/transformer/registry"

YT = {"String": "utf8", "Bytes": "string", "Int8": "int8", "Int16": "int16", "Int32": "int32", "Int64": "int64",
      "Uint8": "uint8", "Uint16": "uint16", "Uint32": "uint32", "Uint64": "uint64", "Float32": "float",
      "Float64": "double", "Boolean": "boolean", "Date": "date", "Datetime": "datetime", "Timestamp": "timestamp",
      "Interval": "interval", "Any": "any"}


def canon(path, key):
    with open(path) as f:
        return json.load(f)[key]


def write(name, obj, compact=False):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name), "w") as f:
        if compact:
            json.dump(obj, f, ensure_ascii=False, sort_keys=True, separators=(",", ":"))
        else:
            json.dump(obj, f, indent=1, ensure_ascii=False, sort_keys=True)
        f.write("\n")
    print("wrote", name)


# ---------------------------------------------------------------------------
# Shared items of mask/sharder/to_string tests
#   mask/hmac_hasher_test.go:31-92, sharder/sharder_test.go:57-113,
#   to_string/to_string_test.go:57-113
# ---------------------------------------------------------------------------
T1_SCHEMA = [["column1", "utf8", True], ["column2", "int64", False], ["column3", "int32", False], ["column4", "boolean", False]]
T2_SCHEMA = [["column1", "string", False], ["column2", "date", False], ["column3", "double", False], ["column4", "float", False]]
T3_SCHEMA = [["column2", "int8", False], ["column3", "uint32", False], ["column4", "date", False]]


def item1():
    return {"ns": "db", "table": "table1", "schema": T1_SCHEMA, "names": ["column1", "column2", "column3", "column4"],
            "values": [["string", "value1"], ["int64", 123], ["int32", 1234], ["bool", True]]}


def item2(first_name):
    return {"ns": "db", "table": "table2", "schema": T2_SCHEMA, "names": [first_name, "column2", "column3", "column4"],
            "values": [["string", "value1"], ["time", "1703-01-02T00:00:00Z"], ["float64", 123.123], ["float32", 312.321]]}


def item3():
    return {"ns": "db", "table": "a_table3", "schema": T3_SCHEMA, "names": ["column2", "column3", "column4"],
            "values": [["int8", -3], ["uint32", 12345], ["duration", 60_000_000_000]]}


def mask():
    # mask/hmac_hasher_test.go:17-95; canon mask/gotest/canondata/result.json
    res = canon(REG + "/mask/gotest/canondata/result.json", "gotest.gotest.TestHmacHasherTransformer")
    t2 = [list(c) for c in T2_SCHEMA]
    items = [item1(), item2("column1"), item3(),
             {"ns": "db", "table": "a_table3", "schema": T3_SCHEMA, "names": [], "values": []}]
    items[1]["schema"] = t2
    items[1]["original_types"] = {"column1": "mysql:blob"}  # hmac_hasher_test.go:45
    cases = []
    for it, r in zip(items, res):
        tr = r["Transformed"][0]
        cases.append({"item": it, "expect_values": tr.get("columnvalues", []),
                      "expect_schema": [[c["name"], c["type"], c["key"], c["original_type"]] for c in tr["table_schema"]],
                      "expect_errors": len(r["Errors"])})
    write("mask.json", {
        "source": "pkg/transformer/registry/mask/hmac_hasher_test.go:17-95 + mask/gotest/canondata/result.json",
        "config": {"maskFunctionHash": {"userDefinedSalt": "the-best-tasty-saint-petersburg-salt"},
                   "columns": ["column1", "column2", "column3", "column4"], "tables": {"excludeTables": []}},
        "cases": cases})


def sharder():
    # sharder/sharder_test.go:15-128; canon sharder/gotest/canondata/result.json
    res = canon(REG + "/sharder/gotest/canondata/result.json", "gotest.gotest.TestSharderTransformer")
    cfgs = [
        {"tables": {"excludeTables": []}, "columns": {"includeColumns": []}, "shardsCount": "2"},
        {"tables": {"includeTables": ["db.table"]}, "columns": {"includeColumns": [], "excludeColumns": ["column2"]}, "shardsCount": "4"},
        {"tables": {"includeTables": ["db.a_table3"]}, "columns": {"includeColumns": ["column1", "column3"]}, "shardsCount": "8"},
    ]
    # Suitable() matrix asserted at sharder_test.go:78-88
    suitable = [[True, True, True], [True, True, False], [False, False, True]]
    items = [item1(), item2("colunm1"), item3()]  # NB the reference test really spells "colunm1"
    cases = []
    k = 0
    for ci, cfg in enumerate(cfgs):
        for ii, it in enumerate(items):
            c = {"config": cfg, "item": it, "suitable": suitable[ci][ii]}
            if suitable[ci][ii]:
                c["expect_part"] = res[k]["Transformed"][0]["part"]
                k += 1
            cases.append(c)
    assert k == len(res)
    write("sharder.json", {"source": "pkg/transformer/registry/sharder/sharder_test.go:15-128 + canon", "cases": cases})


def to_string():
    res = canon(REG + "/to_string/gotest/canondata/result.json", "gotest.gotest.TestToStringTransformer")
    cfgs = [
        {"tables": {"excludeTables": []}, "columns": {"includeColumns": []}},
        {"tables": {"includeTables": ["db.table"]}, "columns": {"includeColumns": [], "excludeColumns": ["column2"]}},
        {"tables": {"includeTables": ["db.a_table3"]}, "columns": {"includeColumns": ["column1", "column3"]}},
    ]
    suitable = [[True, True, True], [True, True, False], [False, False, True]]  # to_string_test.go:78-88
    items = [item1(), item2("colunm1"), item3()]
    cases = []
    k = 0
    for ci, cfg in enumerate(cfgs):
        for ii, it in enumerate(items):
            c = {"config": cfg, "item": it, "suitable": suitable[ci][ii]}
            if suitable[ci][ii]:
                tr = res[k]["Transformed"][0]
                c["expect_values"] = tr["columnvalues"]
                c["expect_types"] = [x["type"] for x in tr["table_schema"]]
                k += 1
            cases.append(c)
    assert k == len(res)
    # to_string_test.go:130-168 TestAllTypesToStringTransformer (expected strings are in the test itself)
    kats = [
        [["json", '[1,"string",3,4.123,6,true]'], "any", '[1,"string",3,4.123,6,true]'],
        [["json", '{"someName":"someValue","someName2":1234}'], "any", '{"someName":"someValue","someName2":1234}'],
        [["int64", 981274987], "int64", "981274987"], [["int32", -12049182], "int32", "-12049182"],
        [["int16", 12313], "int16", "12313"], [["int8", -14], "int8", "-14"],
        [["uint64", 1142423562], "uint64", "1142423562"], [["uint32", 0], "uint32", "0"],
        [["uint16", 65212], "uint16", "65212"], [["uint8", 213], "uint8", "213"],
        [["float32", 123.123], "float", "123.123"], [["float64", -12344.12334341], "double", "-12344.12334341"],
        [["bytes", "bytes"], "string", "bytes"], [["string", "string"], "utf8", "string"], [["bool", True], "boolean", "true"],
        [["time", "-1232-02-23T00:00:00Z"], "date", "-1232-02-23"], [["time", "14124-01-12T00:00:00Z"], "date", "14124-01-12"],
        [["time", "2311-12-01T01:02:04.000000005Z"], "datetime", "2311-12-01T01:02:04.000000005Z"],
        [["time", "1231-05-23T09:08:07.000000006Z"], "timestamp", "1231-05-23T09:08:07.000000006Z"],
        [["duration", (12 * 3600 + 53 * 60 + 21) * 10**9 + 87_182_124], "interval", "12h53m21.087182124s"],
        [["nil", None], "date", "<nil>"], [["nil", None], "datetime", "<nil>"], [["nil", None], "boolean", "<nil>"],
        [["nil", None], "utf8", "<nil>"], [["nil", None], "int64", "<nil>"],
    ]
    write("to_string.json", {"source": "pkg/transformer/registry/to_string/to_string_test.go:15-168 + canon",
                             "cases": cases, "serialize_kats": kats})


def to_datetime():
    res = canon(REG + "/to_datetime/gotest/canondata/result.json", "gotest.gotest.TestToDateTimeTransformer")
    # to_datetime_test.go:15-110
    s1 = [["column1", "utf8", True], ["column2", "int32", False], ["column3", "int16", False]]
    s2 = [["column1", "uint32", False], ["column2", "datetime", False], ["column3", "float", False]]
    s3 = [["column2", "int8", False], ["column3", "uint32", False], ["column4", "datetime", False]]
    items = [
        {"ns": "db", "table": "table1", "schema": s1, "names": ["column1", "column2", "column3"],
         "values": [["string", "value1"], ["int32", 1759143061], ["int16", 1234]]},
        {"ns": "db", "table": "table2", "schema": s2, "names": ["colunm1", "column2", "column3"],
         "values": [["uint32", 1759143071], ["time", "2025-09-29T11:22:00Z"], ["float64", 123.123]]},
        {"ns": "db", "table": "a_table3", "schema": s3, "names": ["column2", "column3", "column4"],
         "values": [["int8", -3], ["uint32", 1759143081], ["time", "2025-09-29T10:00:00Z"]]},
    ]
    cfgs = [{"tables": {}, "columns": {"includeColumns": [], "excludeColumns": []}},
            {"tables": {}, "columns": {"includeColumns": ["column2", "column3"]}}]
    suitable = [[False, False, False], [True, False, True]]  # to_datetime_test.go:70-76
    cases = []
    k = 0
    for ci, cfg in enumerate(cfgs):
        for ii, it in enumerate(items):
            c = {"config": cfg, "item": it, "suitable": suitable[ci][ii]}
            if suitable[ci][ii]:
                tr = res[k]["Transformed"][0]
                c["expect_values"] = tr["columnvalues"]
                c["expect_types"] = [x["type"] for x in tr["table_schema"]]
                k += 1
            cases.append(c)
    assert k == len(res)
    write("to_datetime.json", {"source": "pkg/transformer/registry/to_datetime/to_datetime_test.go:15-124 + canon", "cases": cases})


# ---------------------------------------------------------------------------
# filter_rows: filter_rows/filter_rows_test.go:47-760 (expected kept sets are
# written in the test itself; values there are untyped Go constants, i.e. Go
# `int` / `float64` regardless of the column's schema type).
# ---------------------------------------------------------------------------
I8, I16, I32, I64 = (-128, 127), (-32768, 32767), (-2**31, 2**31 - 1), (-2**63, 2**63 - 1)


def one_col(dtype, gotype, vals, keep, key=True):
    return {"schema": [["column", dtype, key]], "names": ["column"], "rows": [[[gotype, v]] for v in vals],
            "expect_rows": [[[gotype, v]] for v in keep]}


def filter_rows():
    cases = []

    def add(name, cfg, body, nerr=0, err_code=None, suitable=True, ns="db", table="table", kinds=None):
        c = {"name": name, "config": cfg, "ns": ns, "table": table, "expect_errors": nerr, "suitable": suitable}
        c.update(body)
        if err_code:
            c["expect_error_code"] = err_code
        if kinds:
            c["kinds"] = kinds
        cases.append(c)

    # TestIntFiltering :47-144
    f = {"filter": "column > 10 AND column <= 15 AND column IN (11, 15)"}
    for t, lo, hi in [("int8", I8[0], I8[1]), ("int16", I16[0], I16[1]), ("int32", I32[0], I32[1]), ("int64", I64[0], I64[1]),
                      ("uint8", 0, 255), ("uint16", 0, 65535), ("uint32", 0, 2**32 - 1), ("uint64", 0, I64[1])]:
        add("int/" + t, f, one_col(t, "int", [lo, 10, 11, 14, 15, 16, hi], [11, 15]))
    # the reference mixes Go int and uint64 in this column; a columnar batch is homogeneous,
    # so the whole column is carried as uint64 (toInt64E treats both alike below MaxInt64)
    b = one_col("uint64", "uint64", [0, 10, 11, 14, 15, 16], [11, 15])
    b["rows"].append([["uint64", 2**64 - 1]])
    add("int/uint64-int-overflow", f, b, nerr=1, err_code="INT_OVERFLOW")
    # TestFloatFiltering :146-172
    f = {"filter": "column >= 10.1 AND column < 15.3 AND column NOT IN (15.2, 11.0)"}
    MAXF32 = 3.40282346638528859811704183484516925440e+38
    for t in ("float", "double"):
        add("float/" + t, f, one_col(t, "float64", [-1.0, 10.09, 10.1, 11.0, 14.0, 15.0, 15.2, 15.29, 15.3, 16.0, MAXF32],
                                     [10.1, 14.0, 15.0, 15.29]))
    # TestIntFilteringByFloat :174-271
    f = {"filter": "column > 10.1 AND column <= 15.1 AND column IN (11.0, 15.0)"}
    for t, lo, hi in [("int8", I8[0], I8[1]), ("int16", I16[0], I16[1]), ("int32", I32[0], I32[1]), ("int64", I64[0], I64[1]),
                      ("uint8", 0, 255), ("uint16", 0, 65535), ("uint32", 0, 2**32 - 1), ("uint64", 0, I64[1])]:
        add("int-by-float/" + t, f, one_col(t, "int", [lo, 10, 11, 14, 15, 16, hi], [11, 15]))
    # the reference mixes Go int and uint64 in this column; a columnar batch is homogeneous,
    # so the whole column is carried as uint64 (toInt64E treats both alike below MaxInt64)
    b = one_col("uint64", "uint64", [0, 10, 11, 14, 15, 16], [11, 15])
    b["rows"].append([["uint64", 2**64 - 1]])
    add("int-by-float/uint64-int-overflow", f, b, nerr=1, err_code="INT_OVERFLOW")
    # TestFloatFilteringByInt :273-299
    f = {"filter": "column >= 10 AND column < 15 AND column IN (10.0, 11.0, 14.9)"}
    add("float-by-int/float", f, one_col("float", "float64", [-1.0, 10.0, 10.1, 11.0, 14.0, 14.9, 15.0, MAXF32], [10.0, 11.0, 14.9]))
    add("float-by-int/double", f, one_col("double", "float64", [-1.0, 10.0, 10.1, 11.0, 14.0, 14.9, 15.0, 1.79769313486231570814527423731704356798070e+308], [10.0, 11.0, 14.9]))
    # TestBoolFiltering :301-331
    add("bool1", {"filter": "column = true AND column != false AND column > false AND column >= false AND column <= true"},
        one_col("boolean", "bool", [True, False], [True]))
    add("bool2", {"filter": "column = false AND column != true AND column < true AND column >= false AND column <= false"},
        one_col("boolean", "bool", [True, False], [False]))
    # TestNullFiltering :333-363
    add("null", {"filter": "column1 != NULL AND column2 = NULL"}, {
        "schema": [["id", "int32", True], ["column1", "utf8", False], ["column2", "int64", False]], "names": ["id", "column1", "column2"],
        "rows": [[["int", 1], ["string", "abc"], ["int", 128]], [["int", 2], ["string", "str"], ["nil", None]],
                 [["int", 3], ["nil", None], ["int", 32]], [["int", 4], ["nil", None], ["nil", None]]],
        "expect_rows": [[["int", 2], ["string", "str"], ["nil", None]]]})
    # TestTimeFiltering :365-386
    t1, t2, t3 = "1986-04-26T01:23:47+03:00", "1990-07-22T00:00:00+04:00", "1990-07-22T00:00:00.001+04:00"
    t4, t5, t6 = "1991-12-26T00:00:00+03:00", "2003-04-17T10:19:00+03:00", "2003-04-17T10:19:00.001+03:00"
    add("time", {"filter": "column >= %s AND column < %s AND column NOT IN (%s)" % (t2, t6, t3)},
        one_col("timestamp", "time", [t1, t2, t3, t4, t5, t6], [t2, t4, t5]))
    # TestStringFiltering :388-505
    S7 = ["str", "st", "strr", "tr", "", '"', '""']
    S9 = ["ab", "abc", "abca", "abcz", "abd", "ac", "", '"', '""']
    S9b = ["ab", "bcc", "bccz", "bcd", "bcda", "bce", "", '"', '""']
    S10 = ["str", "st", "sstr", "sttr", "strr", "astrb", "rts", "", '"', '""']
    for name, flt, vals, keep in [
        ("=", 'column = "str"', S7, ["str"]),
        ("!=", 'column != "str"', S7, ["st", "strr", "tr", "", '"', '""']),
        (">", 'column > "abc"', S9, ["abca", "abcz", "abd", "ac"]),
        (">=", 'column >= "abc"', S9, ["abc", "abca", "abcz", "abd", "ac"]),
        ("<", 'column < "bcd"', S9b, ["ab", "bcc", "bccz", "", '"', '""']),
        ("<=", 'column <= "bcd"', S9b, ["ab", "bcc", "bccz", "bcd", "", '"', '""']),
        ("~", 'column ~ "str"', S10, ["str", "sstr", "strr", "astrb"]),
        ("!~", 'column !~ "str"', S10, ["st", "sttr", "rts", "", '"', '""']),
        ("in", "column IN ('str', '\"')", S7, ["str", '"']),
    ]:
        add("string/" + name, {"filter": flt}, one_col("utf8", "string", vals, keep))
    # TestBytesFiltering :507-575
    add("bytes/=", {"filter": 'column = "str"'}, one_col("string", "bytes", S7, ["str"]))
    add("bytes/!=", {"filter": 'column != "\x11\x16\x0f"'},
        one_col("string", "bytes", [[12, 7, 31, 52], [7, 14, 27, 43], [17, 22, 15], [15], [43, 21, 15, 2]],
                [[12, 7, 31, 52], [7, 14, 27, 43], [15], [43, 21, 15, 2]]))
    add("bytes/>", {"filter": 'column > "abc"'}, one_col("string", "bytes", S9, ["abca", "abcz", "abd", "ac"]))
    add("bytes/~", {"filter": 'column ~ "☺"'},
        one_col("utf8", "bytes-utf8", ["☺str", "st", "ss☺tr", "sttr", "strr☺", "ast☺rb", "rts", "", '"', '""'],
                ["☺str", "ss☺tr", "strr☺", "ast☺rb"]))
    add("bytes/in", {"filter": "column IN ('str', '\"', '☺')"},
        one_col("utf8", "bytes-utf8", ["str", "st", "strr", "tr", "", '"', '""', "☺"], ["str", '"', "☺"]))
    # TestErrors :577-670
    two = [["colstr", "utf8", False], ["colint", "int32", False]]
    add("errors/different-types", {"filter": 'colstr > 10 AND colint = "str"'},
        {"schema": two, "names": ["colstr", "colint"], "rows": [[["string", "str"], ["int", 1]], [["string", "abc"], ["int", 2]]],
         "expect_rows": []}, nerr=2, suitable=False)
    add("errors/impossible-kinds", {"filter": 'colstr != ""'},
        {"schema": two, "names": ["colstr", "colint"],
         "rows": [[["string", "str"], ["int", 1]], [["string", "abc"], ["int", 2]]] * 3,
         "expect_rows": [[["string", "str"], ["int", 1]], [["string", "abc"], ["int", 2]]]},
        nerr=4, kinds=["insert", "insert", "update", "update", "delete", "delete"])
    # "Table doesn't contain one of columns" :603-642 — three batches with different schemas
    f = {"filter": 'colstr = "str" AND colint > 4'}
    add("errors/missing-column/ok", f, {"schema": two, "names": ["colstr", "colint"],
                                         "rows": [[["string", "str"], ["int", 1]], [["string", "abc"], ["int", 10]], [["string", "str"], ["int", 10]]],
                                         "expect_rows": [[["string", "str"], ["int", 10]]]})
    add("errors/missing-column/bad1", f, {"schema": [["colstr", "utf8", False]], "names": ["colstr"], "rows": [[["string", "str"]]], "expect_rows": []},
        nerr=1, err_code="COLUMN_NOT_FOUND", suitable=False)
    add("errors/missing-column/bad2", f, {"schema": [["colint", "int32", False]], "names": ["colint"], "rows": [[["int", 10]]], "expect_rows": []},
        nerr=1, err_code="COLUMN_NOT_FOUND", suitable=False)
    # TestTablesAndColumnsFiltering :672-720
    inc = [["column", "int32", True], ["excluded-column", "utf8", False]]
    cfg = {"tables": {"excludeTables": ["table-excluded"]}, "filter": "column = 3"}
    add("tables/included", cfg, {"schema": inc, "names": ["column", "excluded-column"],
                                  "rows": [[["int", i], ["string", s]] for i, s in zip(range(1, 6), "abcde")],
                                  "expect_rows": [[["int", 3], ["string", "c"]]]})
    add("tables/excluded-pass-through", cfg,
        {"schema": [["column", "utf8", True], ["excluded-column", "int32", False]], "names": ["column", "excluded-column"],
         "rows": [[["string", "abc"], ["int", 5]]], "expect_rows": [[["string", "abc"], ["int", 5]]]},
        suitable=False, table="table-excluded")
    # TestCoupleInFilters :722-735
    add("couple-in", {"filter": "column IN (10, 11, 14, 15) AND column NOT IN (10) AND column IN (11, 15) AND column NOT IN (10, 11)"},
        one_col("int8", "int", [I8[0], 10, 11, 14, 15, 16, I8[1]], [15]))
    # TestManyFilters :737-768
    add("many-filters", {"filters": ["column1 > 10", 'column2 = "str"']},
        {"schema": [["column1", "int8", True], ["column2", "utf8", True]], "names": ["column1", "column2"],
         "rows": [[["int", 15], ["string", "str"]], [["int", 15], ["string", "aaa"]], [["int", 5], ["string", "str"]], [["int", 5], ["string", "aaa"]]],
         "expect_rows": [[["int", 15], ["string", "str"]], [["int", 15], ["string", "aaa"]], [["int", 5], ["string", "str"]]]},
        ns="", table="")
    write("filter_rows.json", {"source": "pkg/transformer/registry/filter_rows/filter_rows_test.go:47-768",
                               "unparseable": ['column = str"'], "cases": cases})


def csv_reader():
    # pkg/csv/reader_test.go:13-240 (expected field strings are in the test itself)
    D = {}
    cases = [
        {"name": "simple", "opts": D, "input": "1, 2, 3\n\t\ta,b,c\n\t\t7,8,9\n", "expect_nlines": 3, "expect_len_line1": 3},
        {"name": "newline-in-value", "opts": {"newlines_in_value": 1}, "input": "1, 2,   3,\" 4\n\t\t4    \", 5\n",
         "expect_nlines": 1, "expect_field": [0, 3, " 4\n\t\t4    "]},
        {"name": "quoted-newline-disabled", "opts": {"newlines_in_value": 0}, "input": "123123,\"2000-01-01\",\"\nmysuperdata\"",
         "expect_error": "QUOTE"},
        {"name": "quoting-disallowed", "opts": {"quote_char": 0}, "input": "a, b, \"c\", d\n", "expect_error": "QUOTING_DISABLED"},
        {"name": "escape-outside-quotes", "opts": D, "input": "a, \\, \"c \\\" e , f\"\n", "expect": [["a", "\\", "c \\\" e , f"]]},
        {"name": "no-escape-char", "opts": {"escape_char": 0}, "input": "a, \\, \"c \\\" e , f\"\n", "expect": [["a", "\\", "\"c \\\" e", "f\""]]},
        {"name": "double-quote-disallowed", "opts": {"double_quote": 0}, "input": "a, b, \"the main \"\"test\"\" is this\", d\n",
         "expect_error": "DOUBLE_QUOTE"},
        {"name": "double-quote-allowed", "opts": {"double_quote": 1}, "input": "a, b, \"the main \"\"test\"\" is this\", d\n",
         "expect": [["a", "b", "the main \"test\" is this", "d"]]},
        {"name": "delimiter-semicolon", "opts": {"delimiter": ";"}, "input": "a; b; \"c\"; d\n", "expect": [["a", "b", "c", "d"]]},
        {"name": "quote-paren", "opts": {"quote_char": "("}, "input": "a, (b(, (c(, d\n", "expect": [["a", "b", "c", "d"]]},
        {"name": "delimiter-in-quotes", "opts": D, "input": "1, \"2\", \"3 , 3\", 4\n", "expect": [["1", "2", "3 , 3", "4"]]},
        {"name": "double-to-single", "opts": D, "input": "1, \"\"2\"\", 3\n", "expect": [["1", "\"2\"", "3"]]},
        {"name": "latin1-bytes-untouched", "opts": D, "input_latin1": "1, \"\"\xe4\xe4\xe4\xe4\xe4\"\", 3\n",
         "expect_latin1": [["1", "\"\xe4\xe4\xe4\xe4\xe4\"", "3"]]},
        {"name": "quoted-no-quotes", "opts": D, "input": "1, \"check check\", 3\n", "expect": [["1", "check check", "3"]]},
    ]
    # reader_test.go:214-222 — the 51-field pipe-delimited access-log line
    with open(REF + "/pkg/csv/reader_test.go", encoding="utf-8") as f:
        src = f.read()
    a = src.index('content := bytes.NewBufferString(`"5.155.155.155"') + len("content := bytes.NewBufferString(`")
    b = src.index('` + "\\n")', a)
    line = src[a:b]
    ea = src.index('[][]string{{"5.155.155.155"', b) + len("[][]string{{")
    eb = src.index("}}, result)", ea)
    exp = json.loads("[" + src[ea:eb] + "]")
    cases.append({"name": "pipe-access-log", "opts": {"delimiter": "|"}, "input": line + "\n", "expect": [exp]})
    write("csv_reader.json", {"source": "pkg/csv/reader_test.go:13-240", "cases": cases})


def csv_splitter():
    """pkg/csv/splitter_test.go:14-100: the byte streams of the four ConsumeRow tests and the entries they expect (literals of
    the test; a guard checks they are still in the file)."""
    with open(REF + "/pkg/csv/splitter_test.go", encoding="utf-8") as f:
        src = f.read()
    for lit in ['[]byte("12345678901234567890"),', '[]byte(`"2345"` + "\\n"),', '[]byte(`"23456789012345""89`),', '[]byte(`123456"`),',
                'require.EqualValues(t, `"23456789012345""89`+"\\n"+`123456"`, builder.String())']:
        assert lit in src, lit
    cases = [
        {"name": "TestScannerBasic", "input": "a\nb", "rows": ["a\n"], "eof_rest": "b"},
        {"name": "TestScannerBiggerLines", "input": "12345678901234567890\n12345\n", "rows": ["12345678901234567890\n", "12345\n"], "eof_rest": ""},
        {"name": "TestScannerQuotes", "input": '"234567890123456789"\n"2345"\n', "rows": ['"234567890123456789"\n', '"2345"\n'], "eof_rest": ""},
        {"name": "TestScannerLineBreaksInsideQuotes", "input": '"23456789012345""89\n123456"', "rows": [], "eof_rest": '"23456789012345""89\n123456"'},
    ]
    write("csv_splitter.json", {"source": "pkg/csv/splitter_test.go:14-100", "cases": cases})


def csv_typed():
    """The typed half of the CSV ingest: constructCI / getCorrespondingValue / Strictify.

    pkg/providers/s3/reader/registry/csv/reader_csv_test.go:168-430 (the expected values are literals of the tests,
    transcribed with their line) and tests/canon/s3/csv/canondata/*/extracted (Go type + value of every cell the
    validator sink saw; read from the canon files here, never typed)."""
    T = REF + "/pkg/providers/s3/reader/registry/csv/reader_csv_test.go"
    with open(T, encoding="utf-8") as f:
        src = f.read()
    # guard: the literals transcribed below are still in the test file
    for lit in ['require.Equal(t, []any{true, ""}, changeItem.ColumnValues)',
                '"missing row element for column: test-missing-row-column, row elements: 1, columns: 2"',
                'require.Equal(t, []any{"test_file", uint64(1), true, "this is a test string"}, changeItem.ColumnValues)',
                'require.Equal(t, []any{nil, nil, true, "this is a test string"}, changeItem.ColumnValues)',
                'originalValue := "123,456"', 'expected := "123.456"', '"02-Jan-2006",     // dd-Mon-yyyy',
                'TrueValues:       []string{"true", "yes", "1"}', 'NullValues:             []string{"NULL", "NA"}']:
        assert lit in src, lit
    base = [["test-first-column", "boolean", False, "0"], ["test-missing-row-column", "utf8", False, "1"]]
    sys_cols = [["__file_name", "utf8", True, ""], ["__row_index", "uint64", True, ""]]  # AppendSystemColsTableSchema(cols, true), util.go:210-215
    construct = [
        # TestConstructCI, reader_csv_test.go:168-275: row []string given to constructCI directly (one CSV line here)
        {"name": "missing cols are included", "line": 204, "opts": {"include_missing_columns": 1}, "schema": base, "row": ["true"],
         "expect": [["bool", True], ["string", ""]]},
        {"name": "missing cols flag is disabled", "line": 213, "opts": {}, "schema": base, "row": ["true"], "expect_error": "MISSING_CELL",
         "expect_error_col": "test-missing-row-column"},
        {"name": "missing cols flag is disabled but all elements present", "line": 221, "opts": {}, "schema": base,
         "row": ["true", "this is a test string"], "expect": [["bool", True], ["string", "this is a test string"]]},
        {"name": "schema contains sys cols", "line": 230, "opts": {"file_name": "test_file", "row_number_base": 1}, "schema": sys_cols + base,
         "row": ["true", "this is a test string"],
         "expect": [["string", "test_file"], ["uint64", 1], ["bool", True], ["string", "this is a test string"]]},
        {"name": "hide sys cols", "line": 241, "opts": {"file_name": "test_file", "row_number_base": 1, "hide_system_cols": 1},
         "schema": sys_cols + base, "row": ["true", "this is a test string"],
         "expect": [["nil", None], ["nil", None], ["bool", True], ["string", "this is a test string"]]},
    ]
    corresponding = [
        # TestParseFloatValue :273-303 — parseFloatValue returns a string either way
        {"fn": "parseFloatValue", "line": 278, "opts": {"decimal_point": ","}, "dtype": "double", "in": "123,456", "expect": ["string", "123.456"]},
        {"fn": "parseFloatValue", "line": 285, "opts": {"decimal_point": "."}, "dtype": "double", "in": "123.456", "expect": ["string", "123.456"]},
        {"fn": "parseFloatValue", "line": 291, "opts": {"decimal_point": "."}, "dtype": "double", "in": "abc", "expect": ["string", "abc"]},
        {"fn": "parseFloatValue", "line": 298, "opts": {"decimal_point": ""}, "dtype": "double", "in": "123.456", "expect": ["string", "123.456"]},
        # TestParseNullValues :305-356 — col is the empty ColSchema: DefaultValue(&col) is nil (change_item_builders.go:88-109, no type matches)
        {"fn": "parseNullValues", "line": 316, "opts": {"strings_can_be_null": 1, "quoted_strings_can_be_null": 1, "null_values": ["NULL", "NA"]},
         "dtype": "invalid", "in": "\"NULL\"", "expect": ["nil", None]},
        {"fn": "parseNullValues", "line": 323, "opts": {"strings_can_be_null": 1, "quoted_strings_can_be_null": 1, "null_values": ["NULL", "NA"]},
         "dtype": "invalid", "in": "\"notnull\"", "expect": ["string", "\"notnull\""]},
        {"fn": "parseNullValues", "line": 329, "opts": {"strings_can_be_null": 1, "quoted_strings_can_be_null": 1, "null_values": ["NULL", "NA"]},
         "dtype": "invalid", "in": "NULL", "expect": ["nil", None]},
        {"fn": "parseNullValues", "line": 335, "opts": {"strings_can_be_null": 1, "quoted_strings_can_be_null": 1, "null_values": ["NULL", "NA"]},
         "dtype": "invalid", "in": "notnull", "expect": ["string", "notnull"]},
        {"fn": "parseNullValues", "line": 343, "opts": {"null_values": ["NULL", "NA"]}, "dtype": "invalid", "in": "\"NULL\"",
         "expect": ["string", "\"NULL\""]},
        {"fn": "parseNullValues", "line": 351, "opts": {"strings_can_be_null": 1, "quoted_strings_can_be_null": 1, "null_values": ["NULL", "NA"]},
         "dtype": "invalid", "in": "notnull", "expect": ["string", "notnull"]},
        # TestParseDateValue :358-393 — expected = time.Parse(layout, value), written here as RFC3339
        {"fn": "parseDateValue", "line": 371, "opts": {"timestamp_parsers": ["2006-01-02", "02-Jan-2006", "January 2, 2006"]}, "dtype": "date",
         "in": "2024-03-22", "expect": ["time", "2024-03-22T00:00:00Z"]},
        {"fn": "parseDateValue", "line": 377, "opts": {"timestamp_parsers": ["2006-01-02", "02-Jan-2006", "January 2, 2006"]}, "dtype": "date",
         "in": "22-Mar-2024", "expect": ["time", "2024-03-22T00:00:00Z"]},
        {"fn": "parseDateValue", "line": 383, "opts": {"timestamp_parsers": ["2006-01-02", "02-Jan-2006", "January 2, 2006"]}, "dtype": "date",
         "in": "March 22, 2024", "expect": ["time", "2024-03-22T00:00:00Z"]},
        {"fn": "parseDateValue", "line": 389, "opts": {"timestamp_parsers": ["2006-01-02", "02-Jan-2006", "January 2, 2006"]}, "dtype": "date",
         "in": "2024/03/22", "expect": ["string", "2024/03/22"]},
    ]
    B = {"strings_can_be_null": 1, "null_values": ["NULL", "NA"], "true_values": ["true", "yes", "1"], "false_values": ["false", "no", "0"]}
    for line, v, e in [(407, "NULL", ["bool", False]), (412, "true", ["bool", True]), (417, "false", ["bool", False]), (422, "TRUE", ["bool", True]),
                       (427, "random", ["string", "random"])]:  # TestParseBooleanValue :395-431
        corresponding.append({"fn": "parseBooleanValue", "line": line, "opts": B, "dtype": "boolean", "in": v, "expect": e})

    # S3 canon: what the validator sink received (Go type + value per cell).  The input object (test_csv_all_types/all_types.csv)
    # lives in the test bucket, not in the repository; the canon pins the OUTPUT — DefaultValue fill of missing columns with
    # IncludeMissingColumns, the system columns, and the three leading typed cells of every row.
    canon_cases = []
    CD = REF + "/tests/canon/s3/csv/canondata/"
    for d, opts in [("csv.csv.TestNativeS3MissingColumnsAreFilled_canon_0#01", {"include_missing_columns": 1, "hide_system_cols": 1}),
                    ("csv.csv.TestNativeS3WithProvidedSchemaAndSystemCols_canon_0#01", {})]:
        with open(CD + d + "/extracted") as f:
            items = json.load(f)
        rows, schema = [], None
        for it in items:
            names = it["ColumnNames"]["value"]
            vals = [[v["type"], v["value"]] for v in it["ColumnValues"]["value"]]
            ts = it["TableSchema"]["value"]
            sch = [[c["name"], c["type"], bool(c["key"]), c["path"]] for c in ts]
            assert [c[0] for c in sch] == names
            schema = schema or sch
            assert sch == schema
            rows.append(vals)
            assert it["Kind"]["value"] == "insert"
        o = dict(opts)
        if any(n == "__file_name" for n in names):
            o["file_name"] = rows[0][names.index("__file_name")][1]
            o["row_number_base"] = rows[0][names.index("__row_index")][1]
        canon_cases.append({"name": d, "opts": o, "schema": schema, "expect_rows": rows})
    write("csv_typed.json", {
        "source": "pkg/providers/s3/reader/registry/csv/reader_csv_test.go:168-440 + tests/canon/s3/csv/canondata/*/extracted (canon_test.go:188-330)",
        "construct_ci": construct, "corresponding_value": corresponding, "s3_canon": canon_cases})


# ---------------------------------------------------------------------------
# serializers: pkg/serializer/reference/reference_test.go:19-78 (one raw-message item, canon files
# hold the exact bytes), json_test.go:52-110, httpuploader/marshal_test.go:18-233
# ---------------------------------------------------------------------------
def serializers():
    ref = REF + "/pkg/serializer/reference/canondata/reference.reference.TestSerialize_%s/result"

    def canon_bytes(name):
        with open(ref % name, "rb") as f:
            return f.read().decode("utf-8")
    # abstract.MakeRawMessage([]byte("stub"), "table", time.Time{}, "topic", 42, 42, []byte("data"))
    # (pkg/abstract/changeitem/mirror.go:23-67): partition is a Go int, meta a nil map (json: null)
    raw_schema = [["topic", "utf8", True], ["partition", "uint32", True], ["seq_no", "uint64", True], ["write_time", "datetime", True],
                  ["data", "utf8", False], ["meta", "any", False], ["sequence_key", "string", False]]
    raw_item = {"ns": "", "table": "table", "schema": raw_schema, "names": [c[0] for c in raw_schema],
                "values": [["string", "topic"], ["int", 42], ["uint64", 42], ["time", "0001-01-01T00:00:00Z"], ["string", "data"],
                           ["json", "null"], ["bytes", "stub"]]}
    cases = [
        {"name": "TestSerialize/csv:default", "format": "csv", "opts": {}, "item": raw_item, "expect": canon_bytes("csv_default")},
        {"name": "TestSerialize/json:default", "format": "json", "opts": {}, "item": raw_item, "expect": canon_bytes("json_default")},
        {"name": "TestSerialize/json:newline", "format": "json", "opts": {"add_closing_newline": True}, "item": raw_item,
         "expect": canon_bytes("json_newline")},
    ]
    # raw serializer: TestSerialize raw:* (one item), TestBatchSerializer / TestStreamSerializer raw:* over MakeChangeItems(10)
    # (reference_test.go:261-278: data = "data<i>", table<i>, topic<i>, shard i, offset i; the stream serializer always closes lines)
    cases.append({"name": "TestSerialize/raw:default", "format": "raw", "opts": {}, "item": raw_item, "expect": canon_bytes("raw_default")})
    cases.append({"name": "TestSerialize/raw:newline", "format": "raw", "opts": {"add_closing_newline": True}, "item": raw_item, "expect": canon_bytes("raw_newline")})

    def raw_items(n):
        return [{"ns": "", "table": "table%d" % i, "schema": raw_schema, "names": [c[0] for c in raw_schema],
                 "values": [["string", "topic%d" % i], ["int", i], ["uint64", i], ["time", "0001-01-01T00:00:00Z"], ["string", "data%d" % i],
                            ["json", "null"], ["bytes", "stub"]]} for i in range(n)]
    cbase = REF + "/pkg/serializer/reference/canondata/reference.reference.Test%s/result"
    for test, name, opts in [("BatchSerializer_raw_default", "TestBatchSerializer/raw:default", {}),
                             ("BatchSerializer_raw_newline", "TestBatchSerializer/raw:newline", {"add_closing_newline": True}),
                             ("StreamSerializer_raw_default", "TestStreamSerializer/raw:default", {"add_closing_newline": True}),
                             ("StreamSerializer_raw_newline", "TestStreamSerializer/raw:newline", {"add_closing_newline": True})]:
        with open(cbase % test, "rb") as f:
            cases.append({"name": name, "format": "raw", "opts": opts, "items": raw_items(10), "expect": f.read().decode("utf-8")})
    # json_test.go:52-110 TestJSONSerializerComplexAsStr (require.Contains fragments)
    complex_item = {"ns": "", "table": "", "schema": [["id", "int16", False], ["jsonObject", "any", False], ["jsonArray", "any", False], ["nil", "any", False]],
                    "names": ["id", "jsonObject", "jsonArray", "nil"],
                    "values": [["int", 1], ["json", "{\"key\":\"value\"}"], ["json", "[1,2,3]"], ["nil", None]]}
    cases.append({"name": "TestJSONSerializerComplexAsStr/as-string", "format": "json", "opts": {"any_as_string": True}, "item": complex_item,
                  "contains": ["\"id\":1", "\"jsonObject\":\"{\\\"key\\\":\\\"value\\\"}\"", "\"jsonArray\":\"[1,2,3]\"", "\"nil\":null"]})
    cases.append({"name": "TestJSONSerializerComplexAsStr/as-is", "format": "json", "opts": {}, "item": complex_item,
                  "contains": ["\"id\":1", "\"jsonObject\":{\"key\":\"value\"}", "\"jsonArray\":[1,2,3]", "\"nil\":null"]})
    # httpuploader/marshal_test.go:18-39 TestDatetime64Marshal: time.Date(2020,2,2,10,2,22,123456789,UTC)
    dt = ["time", "2020-02-02T10:02:22.123456789Z"]
    for prec, exp in enumerate(["15806377421", "158063774212", "1580637742123", "15806377421234", "158063774212345", "1580637742123456",
                                "15806377421234567", "158063774212345678", "1580637742123456789"], start=1):
        cases.append({"name": "TestDatetime64Marshal/DateTime64(%d)" % prec, "format": "ch", "opts": {"ch_types": [[4, prec]]},
                      "item": {"ns": "", "table": "t", "schema": [["ts", "timestamp", False]], "names": ["ts"], "values": [dt]},
                      "expect": "{\"ts\":%s}\n" % exp})
    # marshal_test.go:41-80 TestValidJSON: []byte values under "utf8" round-trip through JSON
    for i, val in enumerate(["\"[{\\\"foo\\\":0}]\"", "[{\\\"foo\\\":0}]", "\"[{\\\\\\\"foo\\\":0}]"]):
        cases.append({"name": "TestValidJSON/tc_%d" % i, "format": "ch", "opts": {"ch_types": [[0, 0]]},
                      "item": {"ns": "", "table": "t", "schema": [["bytes_with_jsons", "utf8", False]], "names": ["bytes_with_jsons"],
                               "values": [["bytes", val]]}, "json_roundtrip": {"bytes_with_jsons": val}})
    # marshal_test.go:116-233 TestNullValueMarshal
    one = [["recipient", "utf8", False]]
    cases.append({"name": "TestNullValueMarshal/single_null_column", "format": "ch", "opts": {"ch_types": [[0, 0]]},
                  "item": {"ns": "", "table": "t", "schema": one, "names": ["recipient"], "values": [["nil", None]]}, "expect": "{}\n"})
    three = [["col1", "utf8", False], ["col2", "utf8", False], ["col3", "utf8", False]]
    cases.append({"name": "TestNullValueMarshal/multiple_columns_all_null", "format": "ch", "opts": {"ch_types": [[0, 0], [0, 0]]},
                  "item": {"ns": "", "table": "t", "schema": three[:2], "names": ["col1", "col2"], "values": [["nil", None], ["nil", None]]},
                  "expect": "{}\n"})
    cases.append({"name": "TestNullValueMarshal/mixed_null_and_values", "format": "ch", "opts": {"ch_types": [[0, 0]] * 3},
                  "item": {"ns": "", "table": "t", "schema": three, "names": ["col1", "col2", "col3"],
                           "values": [["nil", None], ["string", "value2"], ["nil", None]]}, "expect": "{\"col2\":\"value2\"}\n"})
    # marshal_test.go:235-264 TestJSON: "a\nb" must come out as valid JSON
    cases.append({"name": "TestJSON", "format": "ch", "opts": {"ch_types": [[0, 0]]},
                  "item": {"ns": "", "table": "t", "schema": [["data", "utf8", False]], "names": ["data"], "values": [["string", "a\nb"]]},
                  "expect": "{\"data\":\"a\\nb\"}\n"})
    write("serializers.json", {"cases": cases})


# ---------------------------------------------------------------------------
# Batch / Stream serializer canon over the all-databases fixture corpus:
#   pkg/serializer/reference/reference_test.go:80-148 (TestBatchSerializer csv / json), :150-232 (TestStreamSerializer),
#   ReadChangeItems :234-259 = the first 10 items of every tests/canon/<provider>/canondata/*/extracted, cases ordered by
#   name descending (tests/canon/all_databases.go:68-124).  One output line per item, so the canon `result` files split
#   into per-case expectations.  `any` values are written the way the cgo side hands them over: the JSON text
#   encoding/json produces for the Go value with HTML escaping off (sorted map keys, floatEncoder digits, base64 []byte).
# ---------------------------------------------------------------------------
class _Raw(str):
    """a JSON number literal as written (json.Number values must not pass through a Python float)"""


def _go_float(x, bits):
    """encoding/json floatEncoder (encode.go): shortest digits, %e below 1e-6 and from 1e21 on, exponent without padding"""
    import math
    import numpy as np
    if x == 0:
        return "-0" if math.copysign(1, x) < 0 else "0"
    t = np.format_float_scientific(np.float32(x) if bits == 32 else np.float64(x), unique=True, trim="-")
    mant, exp = t.split("e")
    e = int(exp)
    neg = mant.startswith("-")
    digits = mant.lstrip("-").replace(".", "")
    if e < -6 or e >= 21:
        out = digits[0] + ("." + digits[1:] if len(digits) > 1 else "") + "e" + ("-" if e < 0 else "+") + ("%02d" % abs(e))
        if out[-4:-1] in ("e-0", "e+0"):
            out = out[:-2] + out[-1]
    elif e >= 0:
        out = digits + "0" * (e + 1 - len(digits)) if len(digits) <= e + 1 else digits[:e + 1] + "." + digits[e + 1:]
    else:
        out = "0." + "0" * (-e - 1) + digits
    return ("-" if neg else "") + out


def _go_string(s):
    out = ['"']
    for ch in s:
        o = ord(ch)
        if ch in '"\\':
            out.append("\\" + ch)
        elif ch in "\n\r\t\b\f":
            out.append({"\n": "\\n", "\r": "\\r", "\t": "\\t", "\b": "\\b", "\f": "\\f"}[ch])
        elif o < 0x20 or o in (0x2028, 0x2029):
            out.append("\\u%04x" % o)
        else:
            out.append(ch)
    out.append('"')
    return "".join(out)


_GO_INTS = ("int", "int8", "int16", "int32", "int64", "uint", "uint8", "uint16", "uint32", "uint64")


def _go_marshal(v):
    """json.Marshal (escapeHTML off) of a canonized typed value"""
    t, x = v["type"], v["value"]
    if t == "nil" or x is None:
        return "null"
    if t in _GO_INTS or t == "time.Duration":
        return str(int(x))
    if t in ("float64", "float32"):
        return _go_float(float(x), 64 if t == "float64" else 32)
    if t == "json.Number":
        return str(x)
    if t == "bool":
        return "true" if x else "false"
    if t in ("string", "time.Time", "[]uint8"):  # canon already holds []byte as base64
        return _go_string(x)
    if t == "[]interface {}":
        return "[" + ",".join(_go_marshal(e) for e in x) + "]"
    if t == "map[string]interface {}":
        return "{" + ",".join(_go_string(k) + ":" + _go_marshal(x[k]) for k in sorted(x, key=lambda k: k.encode())) + "}"
    if t.startswith("[]"):
        et = t[3:] if t.startswith("[]*") else t[2:]
        return "[" + ",".join(_go_marshal({"type": et, "value": e}) for e in x) + "]"
    raise ValueError(t)


def _typed_cell(v, dtype):
    import base64
    t, x = v["type"], v["value"]
    if t == "nil":
        return ["nil", None]
    if dtype == "any":
        return ["json", _go_marshal(v)]
    if t == "time.Time":
        return ["time", x]
    if t == "time.Duration":
        return ["duration", int(x)]
    if t == "[]uint8":
        return ["bytes", base64.b64decode(x).decode("latin-1")]
    if t == "json.Number":
        return ["jsonnum", str(x)]
    if t == "string":
        return ["string", x]
    if t in _GO_INTS:
        return [t, int(x)]
    if t in ("float32", "float64"):
        return [t, float(x)]
    if t == "bool":
        return [t, bool(x)]
    raise ValueError((t, dtype))


def serializer_canon():
    import glob
    roots = [("clickhouse", "tests/canon/clickhouse/canondata/*/extracted"), ("mysql", "tests/canon/mysql/canondata/*/extracted"),
             ("postgres", "tests/canon/postgres/gotest/canondata/*/extracted"), ("ydb", "tests/canon/ydb/canondata/*/extracted"),
             ("yt", "tests/canon/yt/canondata/*/extracted")]
    tables = []
    for prov, g in roots:
        for f in sorted(glob.glob(REF + "/" + g)):
            with open(f) as fh:
                items = json.load(fh, parse_float=_Raw, parse_int=_Raw)[:10]
            tables.append((f.split("/")[-2].split(".")[-1], prov, items, f[len(REF) + 1:]))
    tables.sort(key=lambda c: c[0], reverse=True)

    def result(name):
        with open(REF + "/pkg/serializer/reference/canondata/reference.reference.Test%s/result" % name, "rb") as f:
            return f.read().decode("utf-8")
    js, jn, cs = result("BatchSerializer_json_default"), result("BatchSerializer_json_newline"), result("BatchSerializer_csv_default")
    assert jn == js + "\n" and result("StreamSerializer_json_default") == jn and result("StreamSerializer_json_newline") == jn
    assert result("StreamSerializer_csv_default") == cs and cs.endswith("\n")
    jl, cl = js.split("\n"), cs[:-1].split("\n")
    assert len(jl) == len(cl) == sum(len(t[2]) for t in tables)
    cases, pos = [], 0
    for name, prov, items, path in tables:
        sch = items[0]["TableSchema"]["value"]
        k = len(items)
        common = {"ns": items[0]["Schema"]["value"], "table": items[0]["Table"]["value"], "schema": [[c["name"], c["type"], bool(c["key"])] for c in sch],
                  "original_types": {c["name"]: c["original_type"] for c in sch}, "names": items[0]["ColumnNames"]["value"]}
        its = []
        for it in items:
            assert it["ColumnNames"]["value"] == common["names"] and it["Kind"]["value"] == "insert"
            its.append({"values": [_typed_cell(v, c["type"]) for v, c in zip(it["ColumnValues"]["value"], sch)]})
        zoned = any(g == "time" and not x.endswith("Z") for it in its for g, x in it["values"])
        # json: TestBatchSerializer/json:default; json_newline: TestBatchSerializer/json:newline = TestStreamSerializer/json:default
        # = TestStreamSerializer/json:newline; csv: TestBatchSerializer/csv:default = TestStreamSerializer/csv:default
        case = {"name": "%s:%s" % (prov, name), "ref": path, "common": common, "rows": [it["values"] for it in its],
                "expect": {"json": "\n".join(jl[pos:pos + k]), "json_newline": "".join(x + "\n" for x in jl[pos:pos + k]), "csv": "".join(x + "\n" for x in cl[pos:pos + k])}}
        if zoned:  # the column form holds UTC instants only: such items stay with the stock code
            case["skip"] = "time.Time values with a non-UTC Location"
        cases.append(case)
        pos += k
    write("serializers_canon.json", {"tables": cases}, compact=True)


# ---------------------------------------------------------------------------
# generic JSON parser (a17)
#   tests/canon/parser/canon_static_generic_test.go:20-44 + samples/static/generic/{json,mdb}.*
#   pkg/parsers/generic/parser_test.go:129-231 (TestParserNumberTypes), :233-276 (TestBase64Unpack)
# ---------------------------------------------------------------------------
def _canon_value(v):
    """tests/canon/validator canonized value {type, value} -> [gotype, value]"""
    t, x = v["type"], v["value"]
    if t == "nil":
        return ["nil", None]
    if t == "time.Time":
        return ["time", x]
    if t == "string":
        return ["string", x]
    if t == "map[string]interface {}":
        return ["json", json.dumps(x, separators=(",", ":"), sort_keys=True)]
    if t in ("uint8", "uint16", "uint32", "uint64", "int8", "int16", "int32", "int64", "float64", "bool"):
        return [t, x]
    raise ValueError(t)


def json_parser():
    cases = []
    base = REF + "/tests/canon/parser"
    for name in ("json", "mdb"):
        cfg = json.load(open(f"{base}/samples/static/generic/{name}.config.json"))
        pc = cfg["ParserConfig"]["json.lb"]
        sample = open(f"{base}/samples/static/generic/{name}.sample", "rb").read()
        ext = json.load(open(f"{base}/gotest/canondata/gotest.gotest.TestGenericParsers_{name}_canon_0/extracted"))
        # ParserConfigJSONLb → GenericParserConfig (pkg/parsers/registry/json/parser_json.go:62-86): AddDedupeKeys = !SkipSystemKeys… for
        # the Lb flavour dedupe keys are always added (lb.go), AddRest from the config
        fields = [[f["name"], f["type"], bool(f["key"]), f.get("path", ""), "", bool(f.get("required"))] for f in pc["Fields"]]
        topic = cfg["GroupTopics"][0]
        rows = []
        for it in ext:
            assert it["ColumnNames"]["value"][:len(fields)] == [f[0] for f in fields]
            rows.append({"table": it["Table"]["value"], "names": it["ColumnNames"]["value"], "values": [_canon_value(v) for v in it["ColumnValues"]["value"]],
                         "part": it["PartID"]["value"]})
        cases.append({"name": "canon_" + name, "ref": f"tests/canon/parser/samples/static/generic/{name}.sample",
                      "options": {"add_rest": bool(pc["AddRest"]), "add_dedupe_keys": True, "null_keys_allowed": bool(pc["NullKeysAllowed"]),
                                  "topic": topic, "partition": json.dumps({"partition": 0, "topic": topic}, separators=(",", ":"))},
                      "fields": fields,
                      # testcase.MakeDefaultPersqueueReadMessage (tests/canon/parser/testcase/test_case.go:66-76)
                      "messages": [{"offset": 123, "write_time": "2020-02-02T10:02:20Z", "value_latin1": sample.decode("latin-1")}],
                      "rows": rows})
    # the tskv.lb flavours of the same canon test that need no TimeField (taxi parses its time column through
    # github.com/araddon/dateparse, unpinned: not extracted).  metrika / metrika_complex read nested ColSchema.Paths
    # (EventValue.LogInfo …: lookupComplex) — oracle only, the device refuses such a schema.
    for name in ("tskv", "tm-5249", "metrika", "metrika_complex"):
        cfg = json.load(open(f"{base}/samples/static/generic/{name}.config.json"))
        pc = cfg["ParserConfig"]["tskv.lb"]
        assert not pc["TimeField"] and not pc["TableSplitter"]
        nested = any("." in (f.get("path") or "") or "/" in (f.get("path") or "") for f in pc["Fields"])
        sample = open(f"{base}/samples/static/generic/{name}.sample", "rb").read()
        ext = json.load(open(f"{base}/gotest/canondata/gotest.gotest.TestGenericParsers_{name}_canon_0/extracted"))
        # (metrika_complex spells one type "UInt32": to Go it is just another string that ParseVal's switch does not know; its
        # cells are nil in the canon — lookups that fail — so the spelling never meets a value here)
        fields = [[f["name"], f["type"].lower(), bool(f["key"]), f.get("path", ""), "", bool(f.get("required"))] for f in pc["Fields"]]
        topic = cfg["GroupTopics"][0]
        rows = [{"table": it["Table"]["value"], "names": it["ColumnNames"]["value"], "values": [_canon_value(v) for v in it["ColumnValues"]["value"]],
                 "part": it["PartID"]["value"]} for it in ext]
        cases.append({"name": "canon_" + name, "ref": f"tests/canon/parser/samples/static/generic/{name}.sample", "nested_paths": nested,
                      # ParserConfigTSKVLb → GenericParserConfig (pkg/parsers/registry/tskv/parser_tskv.go): Format "tskv"
                      "options": {"format": "tskv", "add_rest": bool(pc["AddRest"]), "add_dedupe_keys": True, "null_keys_allowed": bool(pc["NullKeysAllowed"]),
                                  "topic": topic, "partition": json.dumps({"partition": 0, "topic": topic}, separators=(",", ":"))},
                      "fields": fields,
                      "messages": [{"offset": 123, "write_time": "2020-02-02T10:02:20Z", "value_latin1": sample.decode("latin-1")}],
                      "rows": rows})
    # TestParserNumberTypes: one line per message, Offset = line index, canon = MarshalJSON of the ChangeItems
    gdir = REF + "/pkg/parsers/generic"
    res = json.load(open(gdir + "/gotest/canondata/result.json"))
    nums = open(gdir + "/test_data/parser_numbers_test.jsonl", "rb").read()
    nfields = [["id", "int8", False], ["number_field", "int64", False], ["float_field", "double", False], ["obj_field", "any", False], ["array_field", "any", False]]
    for key, use_numbers in (("UseNumbersFalse", False), ("UseNumbersTrue", True)):
        items = res["gotest.gotest.TestParserNumberTypes"][key]
        lines = [ln for ln in nums.split(b"\n")]
        msgs, rows = [], []
        for i, ln in enumerate(lines):
            if ln == b"":
                continue
            msgs.append({"offset": i, "write_time": "1970-01-01T00:00:00Z", "value_latin1": ln.decode("latin-1")})
        for it in items:
            # canon values went through ChangeItem.MarshalJSON: ints/floats as JSON numbers, any as JSON
            rows.append({"table": it["table"], "names": it["columnnames"], "marshalled": it["columnvalues"], "part": it["part"]})
        cases.append({"name": "numbers_" + key, "ref": "pkg/parsers/generic/parser_test.go:129-231",
                      "options": {"use_numbers_in_any": use_numbers, "topic": "my_topic_name", "partition": '{"partition":0,"topic":""}'},
                      "fields": nfields, "messages": msgs, "rows": rows})
    b64 = open(gdir + "/test_data/parse_base64_packed.jsonl", "rb").read()
    items = res["gotest.gotest.TestBase64Unpack"]
    msgs = [{"offset": i, "write_time": "1970-01-01T00:00:00Z", "value_latin1": ln.decode("latin-1")} for i, ln in enumerate(b64.split(b"\n")) if ln]
    cases.append({"name": "base64_unpack", "ref": "pkg/parsers/generic/parser_test.go:233-276",
                  "options": {"unpack_bytes_base64": True, "topic": "my_topic_name", "partition": '{"partition":0,"topic":""}'},
                  "fields": [["id", "int8", False], ["stringVal", "utf8", False], ["bytesVal", "string", False]], "messages": msgs,
                  "rows": [{"table": it["table"], "names": it["columnnames"], "marshalled": it["columnvalues"], "part": it["part"]} for it in items]})
    # TestUnescapeTSKV (parser_test.go:323-366): Format "tskv", UnescapeStringValues, one line per message
    tsk = open(gdir + "/test_data/parser_unescape_test.tskv", "rb").read()
    items = res["gotest.gotest.TestUnescapeTSKV"]
    msgs = [{"offset": i, "write_time": "1970-01-01T00:00:00Z", "value_latin1": ln.decode("latin-1")} for i, ln in enumerate(tsk.split(b"\n")) if ln]
    cases.append({"name": "tskv_unescape", "ref": "pkg/parsers/generic/parser_test.go:323-366",
                  "options": {"format": "tskv", "unescape_string_values": True, "topic": "my_topic_name", "partition": '{"partition":0,"topic":""}'},
                  "fields": [["id", "int8", False], ["message", "string", False]], "messages": msgs,
                  "rows": [{"table": it["table"], "names": it["columnnames"], "marshalled_text": it["columnvalues"], "part": it["part"]} for it in items]})
    write("json_parser.json", {"cases": cases})

def collapse():
    """abstract.Collapse cases of TestCollapse (pkg/abstract/changeitem/change_item_test.go:205-964), transcribed as
    data; expectations are the test's own asserts.  Items: kind, keys (PrimaryKey names of the item's TableSchema),
    names / values (ColumnNames / ColumnValues), old_names / old_values (OldKeys.KeyNames / KeyValues)."""
    src = REF + "/pkg/abstract/changeitem/change_item_test.go"
    lines = open(src, encoding="utf-8").read().split("\n")

    def it(kind, keys, names, values, old=None):
        d = {"kind": kind, "keys": keys, "names": names, "values": values}
        if old:
            d["old_names"], d["old_values"] = old
        return d

    I = lambda v: ["int", v]
    S = lambda v: ["string", v]
    mw = '["бесплатно","борода","игры","порно"]'
    fio = 'ООО "РостовПромПокрытия"'
    cid = 51615524
    cases = []

    def case(line, name, items, expect):
        assert ('t.Run("%s"' % name) in lines[line - 1], (line, name, lines[line - 1])
        cases.append({"name": name, "ref": "change_item_test.go:%d" % line, "items": items, "expect": expect})

    case(206, "insert and update primary key",
         [it("insert", ["id"], ["id"], [I(1)]),
          it("update", ["id"], ["id"], [I(2)], (["id"], [I(1)])),
          it("update", ["id"], ["id"], [I(3)], (["id"], [I(2)]))],
         {"len": 1, "kinds": {"0": "insert"}, "values_equal": {"0": [I(3)]}})
    case(243, "insert and update primary key sequentially",
         [it("insert", ["id"], ["id"], [I(1)])] +
         [it("update", ["id"], ["id"], [I(i + 1)], (["id"], [I(i)])) for i in range(1, 10)],
         {"len": 1, "kinds": {"0": "insert"}, "values_equal": {"0": [I(10)]}})
    case(270, "Update Update, Diff toast",
         [it("update", ["cid"], ["cid", "FIO", "minus_words"], [I(cid), S(fio), S(mw)]),
          it("update", ["cid"], ["cid", "meaningful_goals"], [I(cid), S('[{"goal_id":"114403594","value":500}]')])],
         {"len": 1, "kinds": {"0": "update"}, "values_contain": {"0": [S(mw), S('[{"goal_id":"114403594","value":500}]')]},
          "names_contain": {"0": ["minus_words"]}})
    mw2 = '["бесплатно","борода","игры","порно","без смс"]'
    case(313, "Update Update, Diff full",
         [it("update", ["cid"], ["cid", "FIO", "minus_words"], [I(cid), S(fio), S(mw)],
             (["cid", "FIO", "minus_words"], [S("51615524"), S(""), S("")])),
          it("update", ["cid"], ["cid", "FIO", "minus_words"], [I(cid), S(fio), S(mw2)],
             (["cid", "FIO", "minus_words"], [S("51615524"), S(fio), S(mw)]))],
         {"len": 1, "kinds": {"0": "update"}, "values_contain": {"0": [S(mw2), S(fio), I(cid)]},
          "names_contain": {"0": ["minus_words", "FIO", "cid"]}})
    case(393, "Insert Update, Diff Toast",
         [it("insert", ["cid"], ["cid", "minus_words", "meaningful_goals"], [I(cid), S(mw), S('[{"value":100}]')]),
          it("update", ["cid"], ["cid", "meaningful_goals"], [I(cid), S('[{"value":500}]')])],
         {"len": 1, "kinds": {"0": "insert"}, "values_contain": {"0": [S(mw), S('[{"value":500}]')]},
          "values_not_contain": {"0": [S('[{"value":100}]')]}, "names_contain": {"0": ["minus_words"]}})
    case(438, "Insert Update, multiple PK",
         [it("insert", ["i2", "i1", "i3"], ["i1", "i2", "i3", "t"], [I(11), I(21), I(31), S("test1")]),
          it("update", ["i1", "i3", "i2"], ["i2", "i1", "i3", "t"], [I(21), I(11), I(31), S("test2")])],
         {"len": 1, "kinds": {"0": "insert"}, "values_contain": {"0": [I(11), I(21), I(31), S("test2")]}})
    case(469, "Insert Update, multiple PK, toast",
         [it("insert", ["i2", "i1"], ["i1", "i2", "t"], [I(11), I(21), S("test1")]),
          it("update", ["i1", "i2"], ["t"], [S("test2")], (["i2", "i1"], [I(21), I(11)]))],
         {"len": 1, "kinds": {"0": "insert"}, "values_contain": {"0": [I(11), I(21), S("test2")]}})
    iud = [it("insert", ["cid"], ["cid", "minus_words", "meaningful_goals"], [I(cid), S(mw), S('[{"value":100}]')]),
           it("update", ["cid"], ["cid", "meaningful_goals"], [I(cid), S('[{"value":500}]')]),
           it("delete", ["cid"], ["cid"], [I(cid)])]
    case(503, "Insert Update Delete", iud, {"len": 1, "kinds": {"0": "delete"}})
    case(556, "Update primary key and Delete",
         [it("update", ["cid"], ["cid", "meaningful_goals"], [I(51615525), S('[{"value":500}]')], (["cid"], [I(cid)])),
          it("delete", ["cid"], [], [], (["cid"], [I(51615525)]))],
         {"len": 1, "kinds": {"0": "delete"}, "old_value0": {"0": I(cid)}})
    case(596, "Insert Update Delete Insert",
         iud + [it("insert", ["cid"], ["cid", "minus_words", "meaningful_goals"], [I(cid), S(mw), S('[{"value":200}]')])],
         {"len": 1, "kinds": {"0": "insert"}, "values_contain": {"0": [S('[{"value":200}]')]}, "names_contain": {"0": ["minus_words"]}})
    case(668, "Delete Update (Upsert scenario)",
         [it("delete", ["cid"], ["cid"], [I(cid)]),
          it("update", ["cid"], ["cid", "meaningful_goals"], [I(cid), S('[{"value":500}]')])],
         {"len": 1, "kinds": {"0": "update"}})
    mw3 = '["платно", "качественный пристойный контент"]'
    case(704, "Primary key change",
         [it("insert", ["cid"], ["cid", "minus_words", "meaningful_goals"], [I(cid), S(mw), S('[{"value":100}]')]),
          it("update", ["cid"], ["cid", "meaningful_goals"], [I(51615525), S('[{"value":500}]')], (["cid"], [I(cid)])),
          it("insert", ["cid"], ["cid", "minus_words", "meaningful_goals"], [I(51615526), S(mw3), S('[{"value":200}]')])],
         {"len": 2, "kinds": {"0": "insert", "1": "insert"}, "nvalues": {"0": 3, "1": 3},
          "col_value": {"0": {"cid": I(51615525), "minus_words": S(mw), "meaningful_goals": S('[{"value":500}]')},
                        "1": {"cid": I(51615526), "minus_words": S(mw3), "meaningful_goals": S('[{"value":200}]')}}})
    cu = ["cid", "uniq"]
    # this reference case runs Collapse twice (changes, changes2): two golden cases
    case(760, "Bad collapse with extra contstraint",
         [it("insert", ["cid"], cu, [I(1), S("first_value")]), it("insert", ["cid"], cu, [I(2), S("second_value")])],
         {"len": 2, "kinds": {"0": "insert"}, "equals_input": True})
    case(760, "Bad collapse with extra contstraint",
         [it("update", ["cid"], cu, [I(1), S("temp_value")]), it("update", ["cid"], cu, [I(2), S("first_value")]),
          it("update", ["cid"], cu, [I(1), S("second_value")])],
         {"len": 2, "kinds": {"0": "update"}})
    case(819, "Collapse with no primary keys should be no-op",
         [it("insert", [], cu, [I(1), S("first_value")]), it("insert", [], cu, [I(2), S("second_value")])],
         {"len": 2, "equals_input": True})
    # json.Unmarshal gives []interface{} of map[string]interface{}: json.Marshal renders map keys sorted
    mkey = '[{"Key":"issueId","Value":"60be59389f7e4745883817d6"},{"Key":"linkField","Value":"parentIssueLinkChain"}]'
    doc = '{"chain":[{"height":1,"issue":"608999a8a702ca1877a3ddf2"},{"height":2,"issue":"6024fc6b5dc51554943bed7f"},{"height":3,"issue":"6017da0f3ed4d47bd0582d07"}]}'
    case(842, "Collapse with mongo replication delete->insert records with same pkey",
         [it("delete", ["_id"], [], [], (["_id"], [["json", mkey]])),
          it("insert", ["_id"], ["_id", "document"], [["json", mkey], ["json", doc]])],
         {"len": 1, "equals_item": {"0": 1}})
    case(933, "update insert",
         [it("update", ["id"], ["id"], [I(2)], (["id"], [I(1)])), it("insert", ["id"], ["id"], [I(1)])],
         {"len": 2, "equals_input": True})
    write("collapse.json", {"source": "pkg/abstract/changeitem/change_item_test.go:205-964 (TestCollapse)", "cases": cases})


def keys_changed():
    """ChangeItem.KeysChanged (TestPkeyChange, change_item_test.go:966-1235) and SplitUpdatedPKeys
    (TestSplitUnchangedKeys, :1294-1410), transcribed as data with the tests' own expectations."""
    src = REF + "/pkg/abstract/changeitem/change_item_test.go"
    lines = open(src, encoding="utf-8").read().split("\n")
    I = lambda v: ["int", v]
    S = lambda v: ["string", v]

    def upd(keys, names, values, old=None):
        d = {"kind": "update", "keys": keys, "names": names, "values": values}
        if old:
            d["old_names"], d["old_values"] = old
        return d
    cases = []

    def case(line, needle, item, expect):
        assert needle in lines[line - 1], (line, needle, lines[line - 1])
        cases.append({"ref": "change_item_test.go:%d" % line, "item": item, "changed": expect})
    iv = ["id", "value"]
    for line, vals, exp in [(976, [I(1), S("kek")], False), (987, [I(1), S("lel")], False), (998, [I(2), S("kek")], True), (1009, [I(2), S("lel")], True)]:
        case(line, "require.", upd(["id"], iv, vals, (iv, [I(1), S("kek")])), exp)
    n4 = ["id1", "value1", "id2", "value2"]
    k2 = ["id1", "id2"]
    case(1036, "require.False", upd(k2, n4, [I(1), S("olel"), I(100), S("okek")], (n4, [I(1), S("lel"), I(100), S("kek")])), False)
    case(1047, "require.True", upd(k2, n4, [I(1), S("lel"), I(200), S("lel")], (n4, [I(1), S("kek"), I(100), S("lel")])), True)
    case(1058, "require.True", upd(k2, n4, [I(1), S("lel"), I(200), S("lel")], (k2, [I(1), I(100)])), True)
    case(1071, "one column - PrimaryKey, changed", upd(["a"], ["a"], [I(1)], (["a"], [I(123)])), True)
    case(1093, "one column - not a PrimaryKey, changed", upd([], ["a"], [I(1)]), False)
    ab = ["a", "b"]
    case(1110, "one 1st changed", upd(ab, ab, [I(1), I(2)], (ab, [I(1), I(3)])), True)
    case(1135, "one 2nd changed", upd(ab, ab, [I(1), I(2)], (ab, [I(2), I(2)])), True)
    case(1160, "both changed", upd(ab, ab, [I(1), I(2)], (ab, [I(3), I(4)])), True)
    case(1185, "both changed", upd(ab, ab, [I(1), I(2)], (ab, [I(1), I(2)])), False)
    abc = ["a", "b", "c"]
    case(1210, "not a PrimaryKey changed", upd(ab, abc, [I(1), I(2), I(3)], (abc, [I(1), I(2), I(4)])), False)

    assert "func TestSplitUnchangedKeys" in lines[1293]
    na = ["name", "address"]

    def it(kind, vals, old=None):
        d = {"kind": kind, "keys": ["name"], "names": na, "values": [S(v) for v in vals]}
        if old:
            d["old_names"], d["old_values"] = na, [S(v) for v in old]
        return d
    changes = [it("insert", ["John", "123 Street"]), it("update", ["John", "124 Street"], ["John", "123 Street"]),
               it("update", ["Susan", "124 Street"], ["John", "124 Street"]), it("insert", ["Ben", "100 Street"]),
               it("update", ["Ben", "124 Street"], ["Ben", "100 Street"])]
    expected = [[changes[0], changes[1]],
                [{"kind": "delete", "keys": ["name"], "names": [], "values": [], "old_names": na, "old_values": [S("John"), S("124 Street")]},
                 it("insert", ["Susan", "124 Street"])],
                [changes[3], changes[4]]]
    write("keys_changed.json", {"source": "pkg/abstract/changeitem/change_item_test.go:966-1235 (TestPkeyChange), :1294-1410 (TestSplitUnchangedKeys)",
                                "keys_changed": cases, "split": {"ref": "change_item_test.go:1294", "items": changes, "expected": expected}})


# ---------------------------------------------------------------------------
# queue serializers: pkg/serializer/queue native_serializer_test.go:18-87, json_serializer_test.go:32-141,
# test.go:36-114 (commonTest table, native_batcher_test.go / json_batcher_test.go sizers),
# pkg/abstract/changeitem/change_item_test.go:1445-1455 (TestMarshalJSON)
# ---------------------------------------------------------------------------
def queue_serializers():
    q = REF + "/pkg/serializer/queue/gotest/canondata/"
    res = canon(q + "result.json", "gotest.gotest.TestNativeSerializerTopicName/saveTxOrder-enabled")
    key = res['""'][0]
    with open(q + "gotest.gotest.TestNativeSerializerTopicName_saveTxOrder-enabled/extracted") as f:
        native_value = f.read()
    with open(q + "gotest.gotest.TestNativeSerializerTopicName_saveTxOrder-disabled/extracted") as f:
        assert f.read() == native_value
    # nativeSerializerTestTypicalChangeItem with Table = "table1" (both variables alias one item)
    native_item = {"ns": "public", "table": "table1", "schema": [["id", "int32", True], ["val", "int32", False]], "names": ["id", "val"],
                   "original_types": {"id": "pg:integer", "val": "pg:integer"},
                   "rows": [[["int32", 1], ["int32", -8388605]]] * 2, "kinds": ["insert"] * 2}
    native_meta = {"ids": [601] * 2, "lsns": [25051056] * 2, "commit_times": [1643660670333075000] * 2, "counters": [0] * 2}
    jres = canon(q + "result.json", "gotest.gotest.TestJSONSerializerTopicNameAllTypes")
    with open(q + "gotest.gotest.TestJSONSerializerTopicNameAllTypes/extracted") as f:
        json_value = f.read()
    all_schema = [["val_int64", "int64", False], ["val_int32", "int32", False], ["val_int16", "int16", False], ["val_int8", "int8", False],
                  ["val_uint64", "uint64", False], ["val_uint32", "uint32", False], ["val_uint16", "uint16", False], ["val_uint8", "uint8", False],
                  ["val_float", "float", False], ["val_double", "double", False], ["val_string", "string", False], ["val_utf8", "utf8", False],
                  ["val_boolean", "boolean", False], ["val_any", "any", False], ["val_date", "date", False], ["val_datetime", "datetime", False],
                  ["val_timestamp", "timestamp", False], ["val_interval", "interval", False]]
    all_vals = [["int64", -1234567899123456789], ["int32", -123456789], ["int16", -12345], ["int8", -123], ["uint64", 123456789123456789],
                ["uint32", 123456789], ["uint16", 12345], ["uint8", 123], ["float32", 1.23], ["float64", 1.234], ["string", "bla bla bla"],
                ["string", "utf8 bla bla bla"], ["bool", True], ["json", "{\"123\":123,\"key\":\"val\"}"], ["time", "2021-02-03T00:00:00Z"],
                ["time", "2021-03-04T05:06:07.000000008Z"], ["duration", 1000000123], ["duration", 1000000321]]
    json_item = {"ns": "public", "table": "table0", "schema": all_schema, "names": [c[0] for c in all_schema], "rows": [all_vals], "kinds": ["insert"]}
    # TestMarshalJSON: testChangeItem (change_item_test.go), canon = json.MarshalIndent of the same MarshalJSON bytes
    with open(REF + "/pkg/abstract/changeitem/gotest/canondata/gotest.gotest.TestMarshalJSON/extracted") as f:
        mj = json.load(f)
    marshal_item = {"ns": "schema", "table": "table", "schema": [["a", "utf8", True], ["b", "utf8", True]], "names": ["a", "b"],
                    "rows": [[["string", "av"], ["string", "bv"]]], "kinds": ["insert"],
                    "old_keys": {"names": ["a", "b"], "types": ["at", "bt"], "rows": [[["string", "av"], ["string", "bv"]]]}}
    marshal_meta = {"ids": [100], "lsns": [200], "commit_times": [1961], "counters": [9], "tx_ids": ["tx_id"], "queries": ["query"], "part": "part_id"}
    # commonTest (test.go:52-114): 5 copies of the master item; size = (k elements fit exactly) + delta
    master = {"ns": "public", "table": "timmyb32r_test_tm_2174_pg_src2", "schema": [["id", "int32", True], ["val1", "int32", False], ["val2", "int32", False]],
              "names": ["id", "val1", "val2"], "rows": [[["int32", 4], ["int32", 5], ["int32", 6]]] * 5, "kinds": ["insert"] * 5}
    table = [  # [enabled, MaxChangeItems, [k, delta] or null, expected #messages]
        [False, 0, None, 5], [True, 1, None, 5], [True, 2, None, 3], [True, 0, [0, 1], 5], [True, 0, [1, 0], 5], [True, 0, [2, -1], 5],
        [True, 0, [2, 0], 3], [True, 0, [2, 1], 3], [True, 0, [3, -1], 3], [True, 0, [3, 0], 2], [True, 0, [3, 1], 2],
        [True, 1, [2, 0], 5], [True, 2, [2, 0], 3], [True, 2, [2, 0], 3], [True, 2, [1, 0], 5]]
    write("queue_serializers.json", {
        "native_topic_name": {"item": native_item, "meta": native_meta, "key": key, "value": native_value},
        "json_all_types": {"item": json_item, "key": jres[0]["Keys"][0], "value": json_value},
        "marshal_json": {"item": marshal_item, "meta": marshal_meta, "table_schema_json": json.dumps(mj["table_schema"], separators=(",", ":")),
                         "value": json.dumps(mj, separators=(",", ":"))},
        "batching": {"item": master, "table": table}})


# ---------------------------------------------------------------------------
# Confluent SR parser, JSON schemas: engine/parser_test.go:21-103 (TestClient) over
# testdata/test_schemas.json + testdata/test_raw_json_messages, canon gotest/canondata/result.json.
# The canon went through float64 (json.Number values print with 17 digits), so numbers pin to float64 precision.
# ---------------------------------------------------------------------------
def confluent_sr():
    import base64
    eng = REF + "/pkg/parsers/registry/confluentschemaregistry/engine/"
    with open(eng + "testdata/test_schemas.json") as f:
        schemas = json.load(f)
    # init(): every schema object is re-marshalled; the registry mock serves {"schema": text, "schemaType": ...}
    schemas = {k: {"schema": v["schema"], "schemaType": v.get("schemaType", "")} for k, v in schemas.items()}
    with open(eng + "testdata/test_raw_json_messages", "rb") as f:
        lines = f.read().split(b"\n")
    items = canon(eng + "gotest/canondata/result.json", "gotest.gotest.TestClient")
    json_tables = set()
    for v in schemas.values():
        if v["schemaType"] == "JSON":
            ns, tb = json.loads(v["schema"])["title"].split(".", 1)
            json_tables.add((ns, tb))
    exp = [{"schema": it["schema"], "table": it["table"], "lsn": it["nextlsn"], "names": it["columnnames"], "values": it["columnvalues"],
            "table_schema": [[c["name"], c["type"], c["required"]] for c in it["table_schema"]]}
           for it in items if (it["schema"], it["table"]) in json_tables]
    write("confluent_sr.json", {"schemas": schemas, "messages": [base64.b64encode(x).decode() for x in lines], "items": exp})


def debezium():
    """Debezium parser, inline schemas: TestParser's canon (engine/gotest/canondata/result.json over engine/parser_test.jsonl,
    one 13.6 KB Postgres event with every pg type), receiver_test.go's TestDelete message with the ChangeItem JSON it requires
    (the test rewrites KeyTypes / CommitTime before comparing; the original values are kept here), TestUnparsed's `{}`."""
    import re
    eng = REF + "/pkg/parsers/registry/debezium/engine/"
    with open(eng + "parser_test.jsonl") as f:
        lines = [x for x in f.read().split("\n") if x]
    items = canon(eng + "gotest/canondata/result.json", "gotest.gotest.TestParser")
    assert len(lines) == len(items)
    cases = []
    for line, it in zip(lines, items):
        cases.append({"name": "TestParser", "message": line, "expect": {
            "kind": it["kind"], "schema": it["schema"], "table": it["table"], "id": it["id"], "lsn": it["nextlsn"], "commit_time": it["commitTime"],
            "names": it["columnnames"], "values": it["columnvalues"], "oldkeys": it.get("oldkeys") or {},
            "table_schema": [[c["name"], c["type"], c["key"], c["table_schema"], c["table_name"], c["original_type"]] for c in it["table_schema"]]}})
    with open(REF + "/pkg/debezium/receiver_test.go") as f:
        src = f.read()
    msg = re.search(r"func TestDelete.*?debeziumMsg := `(.*?)`", src, re.S).group(1)
    want = json.loads(re.search(r"func TestDelete.*?require\.Equal\(t, `(.*?)`, changeItem\.ToJSONString", src, re.S).group(1))
    cases.append({"name": "TestDelete", "message": msg, "expect": {
        "kind": want["kind"], "schema": want["schema"], "table": want["table"], "id": want["id"], "lsn": want["nextlsn"],
        "commit_time": 1672943646565 * 1000000,   # source.ts_ms * 1e6; the test overwrites it with a wall-clock constant before comparing
        "names": want["columnnames"], "values": None, "oldkeys": {"keynames": want["oldkeys"]["keynames"], "keyvalues": want["oldkeys"]["keyvalues"]},
        "table_schema": [[c["name"], c["type"], c["key"], c["table_schema"], c["table_name"], c["original_type"]] for c in want["table_schema"]]}})
    cases.append({"name": "TestUnparsed", "message": "{}", "expect": None})
    write("debezium.json", {"cases": cases})


def sr_format():
    """pkg/schemaregistry/format: the two fixture pairs json_schema_format_test.go embeds (the Postgres all-types Debezium envelope in Kafka
    Connect and in Confluent JSON-schema form, without and with array columns) and the canon of TestCanonizeMakeClosedContentModelTrue."""
    d = REF + "/pkg/schemaregistry/format/"
    out = {"source": "pkg/schemaregistry/format/{full_*_test.json, gotest/canondata/result.json}"}
    for name in ("kafka", "confluent", "kafka_arr", "confluent_arr"):
        a, b = name.split("_")[0], "_arr" if name.endswith("_arr") else ""
        with open(d + "full_%s_json_schema%s_test.json" % (a, b)) as f:
            out[name] = json.load(f)
    out["closed_canon"] = canon(d + "gotest/canondata/result.json", "gotest.gotest.TestCanonizeMakeClosedContentModelTrue")
    write("sr_format.json", out, compact=True)


def sr_protobuf():
    """Confluent-SR parser, PROTOBUF schemas: the two protobuf schemas of engine/testdata/test_schemas.json (ids 5, 6), the two wire
    messages (testdata/test_protobuf_{0,1}.bin) and their items in TestClient's canon (engine/gotest/canondata/result.json: the last two
    of its 55 — parser_test.go:49-52 appends the .bin messages after the JSON lines)."""
    import base64
    eng = REF + "/pkg/parsers/registry/confluentschemaregistry/engine/"
    with open(eng + "testdata/test_schemas.json") as f:
        schemas = json.load(f)
    items = canon(eng + "gotest/canondata/result.json", "gotest.gotest.TestClient")
    cases = []
    for k, (sid, it) in enumerate(zip(("5", "6"), items[-2:])):
        assert schemas[sid]["schemaType"] == "PROTOBUF"
        with open(eng + "testdata/test_protobuf_%d.bin" % k, "rb") as f:
            msg = f.read()
        cases.append({"schema_id": int(sid), "schema": schemas[sid]["schema"], "message_b64": base64.b64encode(msg).decode(),
                      "expect": {"kind": it["kind"], "schema": it["schema"], "table": it["table"], "names": it["columnnames"], "values": it["columnvalues"],
                                 "table_schema": [[c["name"], c["type"], c["key"], c["table_schema"], c["table_name"]] for c in it["table_schema"]]}})
    # TestUnpackVal (types_protobuf_test.go:20-97): the message is proto.Marshal'ed by the test from literal values, so only its canon
    # OUTPUT is a file; the literals are transcribed here and the tests re-encode them.  Names / Vals as the canon holds them.
    u = canon(eng + "gotest/canondata/result.json", "gotest.gotest.TestUnpackVal")
    with open(eng + "testdata/types_protobuf_test_data/std_data_types.proto") as f:
        proto = f.read()
    unpack = {"proto": proto, "names": u["Names"], "vals": u["Vals"],
              "literals": {"doubleField": 1.11, "floatField": 2.2, "int32Field": 2, "int64Field": 3, "uint32Field": 4, "uint64Field": 5, "sint32Field": 6, "sint64Field": 7,
                           "fixed32Field": 8, "fixed64Field": 9, "sfixed32Field": 10, "sfixed64Field": 11, "boolField": True, "stringField": "string", "bytesField": "bytes",
                           "repeatedField": ["1", "2", "3"], "msgField": {"stringField": "stringField", "int32Field": 2, "enumField": 1}}}
    write("sr_protobuf.json", {"cases": cases, "unpack_val": unpack})


def parquet_reader():
    """The reference's 30 Parquet reader canon outputs (tests/canon/s3/parquet/canondata/*/extracted, produced by canon_test.go
    TestCanonSource through reader_parquet.go:137-340 and parquet_schema_resolver.go:81-158 over the apache/parquet-testing corpus;
    rowsCutter keeps the first three row items of the first push): per file the ColumnNames, the TableSchema
    (name, type, original_type, key, required) and every kept row as [Go type, value] cells — nested values as the canon's own tree."""
    import glob
    base = REF + "/tests/canon/s3/parquet/canondata/"
    files = {}
    for d in sorted(glob.glob(base + "parquet.parquet.TestCanonSource_*_canon_0")):
        name = os.path.basename(d)[len("parquet.parquet.TestCanonSource_"):-len("_canon_0")]
        with open(d + "/extracted") as f:
            items = json.load(f)
        it0 = items[0]
        rows = []
        for it in items:
            assert it["ColumnNames"]["value"] == it0["ColumnNames"]["value"] and it["Kind"]["value"] == "insert"
            assert it["Schema"]["value"] == "s3_source_parquet" and it["Table"]["value"] == name
            rows.append([[v["type"], v["value"]] for v in it["ColumnValues"]["value"]])
        files[name] = {"names": it0["ColumnNames"]["value"], "counters": [it["Counter"]["value"] for it in items],
                       "table_schema": [[c["name"], c["type"], c["original_type"], c["key"], c["required"]] for c in it0["TableSchema"]["value"]],
                       "rows": rows}
    assert len(files) == 30
    write("parquet_reader.json", {"source": "tests/canon/s3/parquet/canondata/parquet.parquet.TestCanonSource_<file>_canon_0/extracted (canon_test.go:92-141)",
                                  "files": files}, compact=True)



# ---------------------------------------------------------------------------
# The ONE Parquet object the reference holds: TestBatchSerializer/parquet:default (reference_test.go:117-124) — the 118 items of
# ReadChangeItems(10) written by parquetBatchSerializer with the FIRST item's TableSchema (parquet.go:69-84): every item's values are looked
# up by field name in item.AsMap() (parquet_format.go:82-100), so the items of the other 31 tables contribute nils — and a nil in the
# Required `__primary_key` comes out as 0.  The object is parquet-go's (not rebuildable here); what it SAYS — field order, physical /
# logical types, repetition, row count, every value — is read with pyarrow and kept as data.
# ---------------------------------------------------------------------------
def parquet_writer_canon():
    import glob
    import pyarrow.parquet as pq
    path = REF + "/pkg/serializer/reference/canondata/reference.reference.TestBatchSerializer_parquet_default/result"
    f = pq.ParquetFile(path)
    sch, md = f.schema, f.metadata
    fields = []
    for i in range(len(sch.names)):
        c = sch.column(i)
        fields.append({"name": c.name, "physical": c.physical_type, "logical": str(c.logical_type), "converted": str(c.converted_type), "required": c.max_definition_level == 0,
                       "max_repetition_level": c.max_repetition_level})
    rg = md.row_group(0)
    chunks = [{"name": rg.column(i).path_in_schema, "compression": rg.column(i).compression, "num_values": rg.column(i).num_values} for i in range(rg.num_columns)]
    t = f.read()
    rows = [[r[n] for n in sch.names] for r in t.to_pylist()]
    # the TableSchema the reference built its parquet.Schema from: items[0] of the corpus in the order ReadChangeItems gives (serializer_canon's order)
    roots = ["tests/canon/clickhouse/canondata/*/extracted", "tests/canon/mysql/canondata/*/extracted", "tests/canon/postgres/gotest/canondata/*/extracted",
             "tests/canon/ydb/canondata/*/extracted", "tests/canon/yt/canondata/*/extracted"]   # (serializer_canon's corpus)
    first = sorted([p_ for g in roots for p_ in glob.glob(REF + "/" + g)], key=lambda p_: p_.split("/")[-2].split(".")[-1], reverse=True)[0]
    with open(first) as fh:
        it0 = json.load(fh)[0]
    ts = [[c["name"], c["type"], bool(c["key"]), bool(c["required"])] for c in it0["TableSchema"]["value"]]
    write("parquet_writer_canon.json", {"ref": path[len(REF) + 1:], "created_by": md.created_by, "num_rows": md.num_rows, "num_row_groups": md.num_row_groups, "root": "table",
                                         "fields": fields, "chunks": chunks, "rows": rows, "table_schema": ts, "table_schema_from": first[len(REF) + 1:],
                                         "items": "the 118 items of tests/golden/serializers_canon.json, in its order"})


def hits_schema():
    with open(REF + "/pkg/providers/postgres/testdata/hits_data.json") as f:
        d = json.load(f)
    cols = [[c["name"], c["type"], bool(c.get("key"))] for c in d["parse_schema"]]
    import base64
    row = [base64.b64decode(x).decode("utf-8", "replace") for x in d["data"][0]]
    write("hits_schema.json", {"source": "pkg/providers/postgres/testdata/hits_data.json (parse_schema, data[0])",
                               "columns": cols, "sample_row_text": row})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present: run this in the build container")
    mask(); sharder(); to_string(); to_datetime(); filter_rows(); csv_reader(); csv_splitter(); csv_typed(); serializers(); serializer_canon(); json_parser(); hits_schema(); collapse(); keys_changed(); queue_serializers(); confluent_sr(); debezium(); sr_format(); sr_protobuf(); parquet_reader(); parquet_writer_canon()
