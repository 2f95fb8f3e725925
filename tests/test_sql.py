"""The `sql` transformer (pkg/transformer/registry/clickhouse/clickhouse_local.go:97-294), predicate + cast subset.

The reference runs the query in an external clickhouse-local process; its own tests (clickhouse_local_test.go) need that
binary and assert schemas, row counts, kinds and ONE value.  Those assertions are replayed here (device and oracle), the
rest is device-vs-oracle parity over the documented subset (transferia_amd/csrc/tf_sql.cpp header; PARITY UNPINNED beyond
the replayed assertions: oracle/ora_sql.py restates ClickHouse's documented typing, not a ClickHouse run)."""
import numpy as np
import pytest

from transferia_amd import abi, lib

TABLES = {"include_tables": [".*"]}


def sql(query):
    return lib.Transformer("sql", {"tables": TABLES, "query": query})


# ---- host side: the plan, ResultSchema, refusals (no GPU) ------------------------------------------------------------------
def test_reference_schema_assertions():
    s_nokey = abi.Schema.of([["id", "int32"], ["val", "any"]])
    s = abi.Schema.of([["id", "int32", True], ["val", "any"]])
    # "table_schema": select *, (1+1) as res → three columns, a primary key (clickhouse_local_test.go:77-92)
    t = sql("select *, (1+1) as res from table")
    assert t.type() == "sql" and t.description() == "SQL transfer"
    assert t.suitable("", "test", s)
    r = t.result_schema(s)
    assert [[c.name, c.dtype, c.key, c.original_type] for c in r.cols] == [["id", "int32", True, "ch:Int32"], ["val", "string", False, "ch:String"], ["res", "uint16", False, "ch:UInt16"]]
    # "actual_data": toInt8(id+1) as res (:93-124)
    r = sql("\nselect\n    id,\n    val,\n    toInt8(id+1) as res\nfrom table").result_schema(s)
    assert [[c.name, c.dtype, c.key] for c in r.cols] == [["id", "int32", True], ["val", "string", False], ["res", "int8", False]]
    # "exclude tables with no PKey": ResultSchema refuses a result without a key (:26-44)
    with pytest.raises(lib.TfgpuError) as ei:
        sql("select id, val from table").result_schema(s_nokey)
    assert ei.value.code == lib.ERR_CONFIG and "no primary key" in str(ei.value)
    # "invalid query" (:63-75): refused when the transformer is built (the reference finds out at ResultSchema)
    with pytest.raises(lib.TfgpuError) as ei:
        sql("selet *, 1+1 as res from table")
    assert ei.value.code == lib.ERR_CONFIG
    # "valid query" (:45-62) and "exclude tables with no PKey" (:26-44): cityHash64(val) over the `any` column
    vq = "\nselect\n\tid,\n\tval,\n\tcityHash64(val) as hashed_title\nfrom table;\n"
    assert sql(vq).suitable("", "test", s)
    assert [[c.name, c.dtype, c.key, c.original_type] for c in sql(vq).result_schema(s).cols] == [["id", "int32", True, "ch:Int32"], ["val", "string", False, "ch:String"], ["hashed_title", "uint64", False, "ch:UInt64"]]
    with pytest.raises(lib.TfgpuError) as ei:
        sql(vq).result_schema(s_nokey)
    assert ei.value.code == lib.ERR_CONFIG and "no primary key" in str(ei.value)


def test_refusals_name_the_construct():
    for q, what in [("select id from table group by id", "group"), ("select id from table order by id", "order"), ("select id from table limit 5", "limit"),
                    ("select id from other", "FROM"), ("select id, 1.5 as f from table", "floating"), ("select id, id / 2 as h from table", "Float64"),
                    ("select id, id % 2 as h from table", "modulo"), ("select id, concat(val, 'x') as l from table", "concat"), ("select id + 1 from table", "alias"),
                    ("select id, length(val, val) as l from table", "one argument"), ("select id from table where val like 'a%'", "like")]:
        with pytest.raises(lib.TfgpuError) as ei:
            sql(q)
        assert ei.value.code == lib.ERR_UNSUPPORTED and what.lower() in str(ei.value).lower(), (q, str(ei.value))
    for q in ["select from table", "select id table", "select id from table where id =", "select id from table where id in 1", "select 'x from table"]:
        with pytest.raises(lib.TfgpuError) as ei:
            sql(q)
        assert ei.value.code == lib.ERR_CONFIG, q


def test_expression_types_and_schema_refusals(oracle):
    """The general expressions: ClickHouse's result types (device plan against the oracle's restatement) and what a schema makes
    ClickHouse refuse (TFGPU_ERR_CONFIG) or the device subset decline (TFGPU_ERR_UNSUPPORTED)."""
    from oracle import ora_sql
    s = abi.Schema.of([["id", "int32", True], ["k", "uint8"], ["big", "uint64"], ["w", "int16"], ["name", "utf8"], ["d", "double"], ["t", "timestamp"], ["b", "boolean"]])
    q = ("select id, id * k as m, k * k as kk, big * 2 as b2, -k as nk, -w as nw, -(1 + 1) as n2, id - big as diff, toInt8(id * 300) as c8, length(name) as l, cityHash64(name) as h, "
         "lower(name) as lo, upper(lower(name)) as up, toString(id + k) as s1, id > k as gt, not (k in (1, 2)) as ni, (id = 1 or k > 2) and name != 'zz' as f, 'a' < name as lt, "
         "length(lower(name)) + w as lw, b and k as bk from table where (id >= 0 or not b) and (length(name) > 2 or name in ('a', 'b'))")
    got = [[c.name, c.dtype, c.key] for c in sql(q).result_schema(s).cols]
    assert got == [[n, yt, key] for n, _, _, yt, key in ora_sql.resolve(q, [(c.name, c.dtype, c.key) for c in s.cols])]
    assert dict((n, t) for n, t, _ in got) == {"id": "int32", "m": "int64", "kk": "uint16", "b2": "uint64", "nk": "int16", "nw": "int16", "n2": "int32", "diff": "int64", "c8": "int8", "l": "uint64",
                                               "h": "uint64", "lo": "string", "up": "string", "s1": "string", "gt": "uint8", "ni": "uint8", "f": "uint8", "lt": "uint8", "lw": "int64", "bk": "uint8"}
    for q, code in [("select id, name + 1 as x from table", lib.ERR_CONFIG), ("select id, lower(id) as x from table", lib.ERR_CONFIG), ("select id from table where name = 1", lib.ERR_CONFIG),
                    ("select id from table where id in ('a')", lib.ERR_CONFIG), ("select id from table where name", lib.ERR_CONFIG), ("select id, nosuch + 1 as x from table", lib.ERR_CONFIG),
                    ("select id, d + 1 as x from table", lib.ERR_UNSUPPORTED), ("select id, length(id) as x from table", lib.ERR_UNSUPPORTED), ("select id, toInt8(name) as x from table", lib.ERR_UNSUPPORTED),
                    ("select id, toDateTime(id + 1) as x from table", lib.ERR_UNSUPPORTED), ("select id from table where name = lower(name)", lib.ERR_UNSUPPORTED)]:
        with pytest.raises(lib.TfgpuError) as ei:
            sql(q).result_schema(s)
        assert ei.value.code == code, (q, str(ei.value))
        with pytest.raises((ora_sql.QueryError, ora_sql.Unsupported)) as oi:
            ora_sql.resolve(q, [(c.name, c.dtype, c.key) for c in s.cols])
        assert (oi.type is ora_sql.QueryError) == (code == lib.ERR_CONFIG), (q, oi.type)


def test_city_hash64_restatement_agrees_with_itself():
    """CityHash64 v1.0.2 is restated twice (oracle: Python integers; device: tf_transform.hip) from the published algorithm; the
    only value known without a ClickHouse run is the empty string's (k2).  The device side is compared in test_general_expressions."""
    from oracle import ora_sql
    assert ora_sql.city_hash64(b"") == 0x9AE16A3B2F90404F
    seen = {ora_sql.city_hash64(bytes([65 + i % 26 for i in range(n)])) for n in range(0, 300)}
    assert len(seen) == 300


def test_result_types_follow_clickhouse(oracle):
    from oracle import ora_sql
    s = abi.Schema.of([["id", "int32", True], ["k", "uint8"], ["big", "int64"], ["name", "utf8"], ["ts", "uint32"], ["b", "boolean"], ["d", "double"], ["t", "timestamp"]])
    q = ("select id, k + 1 as a, k - 1 as b2, id + 300 as c, big + 1 as d2, toUInt16(id) as e, toInt64(k) + 70000 as f, -5 as g, 70000 as h, 'x' as i, toString(id) as j, "
         "toDateTime(ts) as dt, b, d, t, name from table where id >= -3 and name != 'zz' or k in (1, 2)")
    got = [[c.name, c.dtype, c.key] for c in sql(q).result_schema(s).cols]
    exp = [[n, yt, key] for n, _, _, yt, key in ora_sql.resolve(q, [(c.name, c.dtype, c.key) for c in s.cols])]
    assert got == exp
    assert dict((n, t) for n, t, _ in got) == {"id": "int32", "a": "uint16", "b2": "int16", "c": "int64", "d2": "int64", "e": "uint16", "f": "int64", "g": "int8", "h": "uint32", "i": "string",
                                               "j": "string", "dt": "datetime", "b": "uint8", "d": "double", "t": "timestamp", "name": "string"}


# ---- device ----------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tf():
    lib.init()
    return lib


def _rows_of(batch, schema):
    """The collapsed batch row-wise, as ora_sql.apply takes it."""
    kid = {abi.K_INSERT: "insert", abi.K_UPDATE: "update", abi.K_DELETE: "delete"}
    rows = []
    for i in range(batch.nrows):
        vals = {c.name: c.pyvalue(i) for c in batch.cols}
        rows.append({"kind": kid[int(batch.kind[i])] if batch.kind is not None else "insert", "src": int(batch.src_row[i]) if batch.src_row is not None else i, "values": vals})
    return rows


def _check(tf, oracle, query, batch, schema):
    from oracle import ora_sql
    batch.schema = schema
    res = tf.Transformer("sql", {"tables": TABLES, "query": query}).apply(tf.DeviceBatch.upload(batch))
    assert not res.errors
    out = res.transformed.download()
    col = oracle.collapse(batch, schema).batch
    exp = ora_sql.apply(query, _rows_of(col, schema), [(c.name, c.dtype, c.key) for c in schema.cols])
    assert out.nrows == len(exp), (query, out.nrows, len(exp))
    rs = ora_sql.resolve(query, [(c.name, c.dtype, c.key) for c in schema.cols])
    assert [c.name for c in out.cols] == [r[0] for r in rs] and [c.dtype for c in out.cols] == [r[3] for r in rs]
    kid = {"insert": abi.K_INSERT, "update": abi.K_UPDATE, "delete": abi.K_DELETE}
    src = out.src_row if out.src_row is not None else np.arange(out.nrows)
    for i, e in enumerate(exp):
        assert (int(out.kind[i]) if out.kind is not None else abi.K_INSERT) == kid[e["kind"]], (query, i)
        assert int(src[i]) == e["src"], (query, i)
        got = [c.pyvalue(i) for c in out.cols]
        if e["values"] is None:
            assert all(g == ["nil", None] for g in got), (query, i, got)
        else:
            assert [[g[0], g[1] if not isinstance(g[1], tuple) else tuple(g[1])] for g in got] == [[v[0], v[1]] for v in e["values"]], (query, i)
        old = getattr(out, "old_keys", None)
        has_old = bool(old) and (out.old_present is None or bool(out.old_present[i]))
        assert has_old == (e["old"] is not None), (query, i)
        if has_old:
            assert [c.name for c in old] == [c.name for c in out.cols]
            ov = [c.pyvalue(i) for c in old]
            assert [[g[0], g[1] if not isinstance(g[1], tuple) else tuple(g[1])] for g in ov] == [[v[0], v[1]] for v in e["old"]], (query, i)
    return out


@pytest.mark.gpu
def test_reference_actual_data(tf, oracle):
    """clickhouse_local_test.go:93-124: id 1, val "part" through toInt8(id+1) → [int32(1), "part", int8(2)]."""
    s = abi.Schema.of([["id", "int32", True], ["val", "any"]])
    b = abi.batch_from_rows(s, ["id", "val"], [[["int32", 1], ["string", "part"]]], "", "test_table", ["insert"])
    out = _check(tf, oracle, "\nselect\n    id,\n    val,\n    toInt8(id+1) as res\nfrom table", b, s)
    assert [c.pyvalue(0) for c in out.cols] == [["int32", 1], ["string", b"part"], ["int8", 2]]


@pytest.mark.gpu
def test_reference_sequences(tf, oracle):
    """"collapse, single delete" and "insert update insert" (:125-170): the canonized replication sequences through
    `select * from table where …`: row counts and kinds as the reference asserts them, OldKeys = the result row."""
    S = abi.Schema.of([["i1", "int32", True], ["i2", "int32", True], ["t", "utf8"]])
    names = ["i1", "i2", "t"]

    def seq(items):
        rows, kinds, olds = [], [], []
        for kind, vals, old in items:
            rows.append([["int32", vals[0]], ["int32", vals[1]], ["string", vals[2]]] if vals else [["nil", None]] * 3)
            kinds.append(kind)
            olds.append(old)
        b = abi.batch_from_rows(S, names, rows, "public", "seq", kinds)
        pres = np.array([o is not None for o in olds], bool)
        b.old_keys = [abi.Column("i1", "int32", abi.R_INT32, values=np.array([(o or (0, 0))[0] for o in olds], np.int32), validity=pres.copy()),
                      abi.Column("i2", "int32", abi.R_INT32, values=np.array([(o or (0, 0))[1] for o in olds], np.int32), validity=pres.copy())]
        b.old_present = pres
        return b
    # insert_update_delete: insert (2,2,b), update (2,2,c) with the same keys, delete of (2,2)
    b = seq([("insert", (2, 2, "b"), None), ("update", (2, 2, "c"), (2, 2)), ("delete", None, (2, 2))])
    out = _check(tf, oracle, "select * from table where i1 in (1, 2) and i2 in (2, 3)", b, S)
    assert len(out.cols) == 3
    # insert_update_insert: keys 1, 2, 3 each inserted and updated in place
    b = seq([("insert", (1, 1, "1a"), None), ("update", (1, 1, "1b"), (1, 1)), ("insert", (2, 2, "2a"), None), ("update", (2, 2, "2b"), (2, 2)),
             ("insert", (3, 3, "3a"), None), ("update", (3, 3, "3b"), (3, 3))])
    out = _check(tf, oracle, "select * from table where i2 in (1, 2)", b, S)
    assert out.nrows >= 1


@pytest.mark.gpu
def test_random_parity(tf, oracle):
    rng = np.random.default_rng(20260923)
    S = abi.Schema.of([["id", "int64", True], ["k", "uint8"], ["x", "int32"], ["u", "uint32"], ["name", "utf8"], ["w", "int16"], ["flag", "boolean"]])
    names = [c.name for c in S.cols]
    queries = ["select * from table",
               "select id, x + 1 as x1, toInt8(x + 1) as x8, toUInt16(w) as w16, k - 1 as km, toString(x) as xs, toDateTime(u) as ut, 'c' as cst, 300 as n300, name from table",
               "select id, name, x from table where x >= 0 and k in (0, 1, 2, 250) or name = 'n7'",
               "select id, toInt64(k) + 70000 as big, toUInt8(x - 5) as wrapd, flag from table where w != 3 and x < 100000 or k not in (1, 2, 3)",
               "select id, k from table where name in ('n1', 'n2', 'n-never') and id > -1000"]
    for n in (1, 7, 257, 5003):
        rows = []
        for i in range(n):
            nil = rng.random() < 0.1
            rows.append([["int64", int(rng.integers(-5, 1 << 40)) if i % 11 else i], ["uint8", int(rng.integers(0, 256))], ["nil", None] if nil else ["int32", int(rng.integers(-2 ** 31, 2 ** 31))],
                         ["uint32", int(rng.integers(0, 2 ** 32))], ["nil", None] if rng.random() < 0.1 else ["string", "n%d" % int(rng.integers(0, 12))], ["int16", int(rng.integers(-2 ** 15, 2 ** 15))],
                         ["bool", bool(rng.integers(0, 2))]])
        for r in rows:  # unique keys (Collapse would otherwise fold repeats; covered by test_reference_sequences)
            pass
        seen = set()
        for i, r in enumerate(rows):
            while r[0][1] in seen:
                r[0][1] += 1 << 41
            seen.add(r[0][1])
        b = abi.batch_from_rows(S, names, rows, "db", "t", ["insert"] * n)
        for q in queries:
            _check(tf, oracle, q, b, S)


@pytest.mark.gpu
def test_reference_valid_query_cityhash(tf, oracle):
    """clickhouse_local_test.go:45-62, the query of "valid query": id, val, cityHash64(val) over an `any` column holding strings."""
    s = abi.Schema.of([["id", "int32", True], ["val", "any"]])
    vals = ["", "a", "part", "0123456789abcdef", "0123456789abcdefg", "x" * 32, "y" * 33, "z" * 64, "w" * 65, "v" * 128, "u" * 129, "кириллица и пробелы " * 7]
    b = abi.batch_from_rows(s, ["id", "val"], [[["int32", i], ["string", v]] for i, v in enumerate(vals)], "", "test", ["insert"] * len(vals))
    out = _check(tf, oracle, "\nselect\n\tid,\n\tval,\n\tcityHash64(val) as hashed_title\nfrom table;\n", b, s)
    assert out.cols[2].pyvalue(0) == ["uint64", 0x9AE16A3B2F90404F]  # CityHash64 of the empty string: k2


@pytest.mark.gpu
def test_general_expressions(tf, oracle):
    """Column arithmetic, parentheses, NOT, comparisons as values, text functions — the expression program against the oracle."""
    rng = np.random.default_rng(20260924)
    S = abi.Schema.of([["id", "int64", True], ["k", "uint8"], ["x", "int32"], ["u", "uint32"], ["big", "uint64"], ["name", "utf8"], ["w", "int16"], ["flag", "boolean"]])
    names = [c.name for c in S.cols]
    queries = ["select id, x * k as m, x + w as s, x - u as d, -k as nk, -x as nx, k * k * k as k3, big + 1 as b1, big * 3 as b3, toInt8(x * w) as c8, (x + 1) * (w - 1) as prod from table",
               "select id, length(name) as l, cityHash64(name) as h, lower(name) as lo, upper(name) as up, cityHash64(upper(name)) as hu, toString(x * 2) as s2, toString(length(name) + k) as s3 from table",
               "select id, x > w as gt, big > x as ux, x <= big as xu, big > 9223372036854775807 as top, not flag as nf, (k in (1, 2, 250)) or flag as o, 'M' <= name as ge, name < 'n5' as lt from table",
               "select id, name from table where (x >= 0 or k < 10) and not (name in ('N1', 'n2') or w > 30000)",
               "select id, k from table where x * 2 > w + u or length(name) = 2 and cityHash64(name) > 9223372036854775807",
               "select id, big from table where big > x and not (big in (1, 2)) and lower(name) != 'n3'",
               "select id, x from table where (k = 1 or k = 2 or (k > 100 and k <= 200)) and 5 < x"]
    for n in (1, 64, 1001):
        rows = []
        for i in range(n):
            rows.append([["int64", i * 3 - 5], ["uint8", int(rng.integers(0, 256))], ["nil", None] if rng.random() < 0.1 else ["int32", int(rng.integers(-2 ** 31, 2 ** 31))],
                         ["uint32", int(rng.integers(0, 2 ** 32))], ["uint64", int(rng.integers(0, 2 ** 64, dtype=np.uint64)) if i % 3 else int(rng.integers(0, 5))],
                         ["nil", None] if rng.random() < 0.1 else ["string", ("N%d" if i % 4 == 0 else "n%d") % int(rng.integers(0, 12)) * int(rng.integers(1, 30 if i % 5 else 3))],
                         ["int16", int(rng.integers(-2 ** 15, 2 ** 15))], ["bool", bool(rng.integers(0, 2))]])
        b = abi.batch_from_rows(S, names, rows, "db", "t", ["insert"] * n)
        for q in queries:
            _check(tf, oracle, q, b, S)


@pytest.mark.gpu
def test_random_expression_trees(tf, oracle):
    """Random integer / boolean expression trees over columns of every integer width (and literals at the type edges): ClickHouse's
    result types and wrapped values, device program against the oracle's typed evaluation — select items and WHERE alike."""
    import random
    import os
    rng = random.Random(20260925 + int(os.environ.get("TFGPU_TEST_SEED", "0")))
    S = abi.Schema.of([["id", "int64", True], ["a", "int8"], ["b", "uint8"], ["c", "int16"], ["d", "uint16"], ["e", "int32"], ["f", "uint32"], ["g", "int64"], ["h", "uint64"], ["s", "utf8"]])
    names = [c.name for c in S.cols]
    ints = ["a", "b", "c", "d", "e", "f", "g", "h"]
    lits = ["0", "1", "2", "7", "127", "128", "255", "256", "32767", "65535", "65536", "2147483647", "4294967295", "4294967296", "9223372036854775807", "-1", "-128", "-129", "-32768", "-2147483648", "-9223372036854775807"]

    def iexpr(depth):
        r = rng.random()
        if depth <= 0 or r < 0.25:
            return rng.choice(ints) if rng.random() < 0.7 else rng.choice(lits)
        if r < 0.6:
            return "(%s %s %s)" % (iexpr(depth - 1), rng.choice(["+", "-", "*"]), iexpr(depth - 1))
        if r < 0.7:
            return "-(%s)" % iexpr(depth - 1)
        if r < 0.85:
            return "to%s(%s)" % (rng.choice(["Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16", "UInt32", "UInt64"]), iexpr(depth - 1))
        if r < 0.92:
            return "length(%s)" % rng.choice(["s", "lower(s)", "toString(%s)" % iexpr(depth - 1)])
        return bexpr(depth - 1)

    def bexpr(depth):
        r = rng.random()
        if depth <= 0 or r < 0.45:
            return "(%s %s %s)" % (iexpr(depth), rng.choice(["=", "!=", "<", "<=", ">", ">="]), iexpr(depth))
        if r < 0.6:
            return "(%s %sin (%s))" % (iexpr(depth), rng.choice(["", "not "]), ", ".join(rng.sample(lits, rng.randrange(1, 5))))
        if r < 0.7:
            return "(s %s '%s')" % (rng.choice(["=", "!=", "<", ">="]), rng.choice(["", "n1", "N3", "zz"]))
        if r < 0.8:
            return "(not %s)" % bexpr(depth - 1)
        return "(%s %s %s)" % (bexpr(depth - 1), rng.choice(["and", "or"]), bexpr(depth - 1))

    edges = {"a": [-128, 127, 0, -1], "b": [0, 255, 1], "c": [-32768, 32767], "d": [0, 65535], "e": [-2 ** 31, 2 ** 31 - 1], "f": [0, 2 ** 32 - 1], "g": [-2 ** 63, 2 ** 63 - 1, 0], "h": [0, 2 ** 64 - 1, 2 ** 63]}
    go = {"a": "int8", "b": "uint8", "c": "int16", "d": "uint16", "e": "int32", "f": "uint32", "g": "int64", "h": "uint64"}
    rows = []
    for i in range(300):
        row = [["int64", i]]
        for n in ints:
            lo, hi = min(edges[n]), max(edges[n])
            row.append([go[n], rng.choice(edges[n]) if rng.random() < 0.3 else rng.randint(lo, hi) if rng.random() < 0.5 else rng.randint(max(lo, -50), min(hi, 50))])
        row.append(["string", rng.choice(["", "n1", "N3", "zz", "longer text", "n1"])])
        rows.append(row)
    b = abi.batch_from_rows(S, names, rows, "db", "t", ["insert"] * len(rows))
    done = 0
    for _ in range(60):
        items = ", ".join("%s as x%d" % (iexpr(3) if rng.random() < 0.7 else bexpr(2), k) for k in range(rng.randrange(1, 5)))
        q = "select id, %s from table" % items + (" where %s" % bexpr(2) if rng.random() < 0.6 else "")
        try:
            _check(tf, oracle, q, b, S)
            done += 1
        except lib.TfgpuError as e:  # a tree deeper than the program's sixteen slots is the only refusal a generated query may meet
            assert e.code == lib.ERR_UNSUPPORTED and "sixteen" in str(e), (q, str(e))
    assert done >= 50


@pytest.mark.gpu
def test_key_must_pass_through_and_pkey_moves_stay_on_the_host(tf):
    S = abi.Schema.of([["id", "int32", True], ["v", "int32"]])
    b = abi.batch_from_rows(S, ["id", "v"], [[["int32", 1], ["int32", 2]]], "db", "t", ["insert"])
    b.schema = S
    with pytest.raises(tf.TfgpuError) as ei:
        tf.Transformer("sql", {"tables": TABLES, "query": "select id + 1 as id, v from table"}).apply(tf.DeviceBatch.upload(b))
    assert ei.value.code == tf.ERR_UNSUPPORTED and "pass through" in str(ei.value)
    b = abi.batch_from_rows(S, ["id", "v"], [[["int32", 1], ["int32", 2]], [["int32", 5], ["int32", 3]]], "db", "t", ["insert", "update"])
    b.schema = S
    pres = np.array([False, True])
    b.old_keys = [abi.Column("id", "int32", abi.R_INT32, values=np.array([0, 1], np.int32), validity=pres.copy())]
    b.old_present = pres
    with pytest.raises(tf.TfgpuError) as ei:
        tf.Transformer("sql", {"tables": TABLES, "query": "select * from table"}).apply(tf.DeviceBatch.upload(b))
    assert ei.value.code == tf.ERR_UNSUPPORTED and "SplitUpdatedPKeys" in str(ei.value)


@pytest.mark.gpu
def test_after_replace_primary_key_like_configs2(tf, oracle):
    """bench.py --workload configs2: a keyless table (Confluent-SR JSON: every integer int64) gets its key from
    replace_primary_key, then the sql transformer filters and casts — through the Apply loop, against the oracle."""
    from oracle import ora_sql
    rng = np.random.default_rng(7)
    S0 = abi.Schema.of([["watchid", "int64"], ["userid", "int64"], ["regionid", "int64"], ["eventtime", "int64"], ["url", "utf8"]])
    n = 3001
    rows = [[["int64", i * 7 + 1], ["int64", int(rng.integers(-2 ** 62, 2 ** 62))], ["int64", int(rng.integers(0, 100))], ["int64", int(rng.integers(0, 2 ** 31))],
             ["string", "http://x/%d" % int(rng.integers(0, 50))]] for i in range(n)]
    b = abi.batch_from_rows(S0, [c.name for c in S0.cols], rows, "default", "hits", ["insert"] * n)
    b.schema = S0
    q = "select *, toString(userid) as userid_s, toInt32(regionid) as region32, toDateTime(eventtime) as eventtime_dt from table where regionid >= 40"
    chain = [tf.Transformer("replace_primary_key", {"keys": ["watchid"], "tables": {}}), tf.Transformer("sql", {"tables": TABLES, "query": q})]
    res = tf.apply_chain(chain, tf.DeviceBatch.upload(b))
    assert not res.errors
    out = res.transformed.download()
    S1 = abi.Schema.of([["watchid", "int64", True], ["userid", "int64"], ["regionid", "int64"], ["eventtime", "int64"], ["url", "utf8"]])
    exp = ora_sql.apply(q, _rows_of(b, S1), [(c.name, c.dtype, c.key) for c in S1.cols])
    assert out.nrows == len(exp) and 0 < out.nrows < n
    assert [c.name for c in out.cols] == ["watchid", "userid", "regionid", "eventtime", "url", "userid_s", "region32", "eventtime_dt"]
    assert [c.dtype for c in out.cols] == ["int64", "int64", "int64", "int64", "string", "string", "int32", "datetime"]
    for i, e in enumerate(exp):
        got = [c.pyvalue(i) for c in out.cols]
        assert [[g[0], g[1] if not isinstance(g[1], tuple) else tuple(g[1])] for g in got] == [[v[0], v[1]] for v in e["values"]], i
