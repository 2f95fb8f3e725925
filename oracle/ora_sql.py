"""CPU restatement of the `sql` transformer (pkg/transformer/registry/clickhouse/clickhouse_local.go:97-294) for the
predicate + cast subset the device takes.  TEST INFRASTRUCTURE: only tests/, smoke() and bench.py's cpu_baseline leg may
import this; the product path (transferia_amd/) never does.

The reference evaluates the query in an external `clickhouse-local` process.  That binary is not in /root/reference nor
in this image, so this file restates ClickHouse's *documented* behaviour for the subset:
  * integer literal types: the narrowest of UInt8/16/32/64 (Int8/16/32/64 when negative);
  * a + b, a - b over integers: signed if either side is (always for minus), bits = min(64, 2 * max(bits));
  * toIntN / toUIntN: two's complement truncation;  toString(int) = decimal text;  toDateTime(int) = epoch seconds;
  * comparisons / IN over integers and strings; AND / OR without parentheses (AND binds tighter);
  * JSONEachRow input: a column a row does not carry takes the type's default (0, '').
and the Go code around it: SplitUpdatedPKeys is NOT restated (batches with such Updates are outside the subset), Collapse
is the oracle's own (ora_collapse, pinned to TestCollapse), result types map back to YT types through
pkg/providers/clickhouse/typesystem.go:15-33, a result column is a key iff it carries the name of an input key column
(:393-413), rows are re-attached to their input row by primary key (:248-286): Update / Delete get the result row as
OldKeys, Delete loses its column values.

PARITY UNPINNED except for what clickhouse_local_test.go asserts (tests/test_sql.py replays those assertions)."""
from __future__ import annotations

import re

INT_TYPES = ["Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16", "UInt32", "UInt64"]
YT_TO_CH = {"int8": "Int8", "int16": "Int16", "int32": "Int32", "int64": "Int64", "uint8": "UInt8", "uint16": "UInt16", "uint32": "UInt32", "uint64": "UInt64",
            "float": "Float64", "double": "Float64", "string": "String", "utf8": "String", "any": "String", "boolean": "UInt8", "date": "Date", "datetime": "DateTime",
            "timestamp": "DateTime64(9)"}  # typesystem.go TargetRule
CH_TO_YT = {"Int8": "int8", "Int16": "int16", "Int32": "int32", "Int64": "int64", "UInt8": "uint8", "UInt16": "uint16", "UInt32": "uint32", "UInt64": "uint64",
            "Float64": "double", "String": "string", "Date": "date", "DateTime": "datetime", "DateTime64(9)": "timestamp"}  # typesystem.go SourceRules


class Unsupported(Exception):
    pass


class QueryError(Exception):
    pass


_TOKEN = re.compile(r"\s*(?:(?P<id>[A-Za-z_][A-Za-z_0-9]*)|`(?P<bq>[^`]*)`|\"(?P<dq>[^\"]*)\"|(?P<num>\d+(?:\.\d+|[eE][-+]?\d+)?)|'(?P<str>(?:[^'\\]|\\.|'')*)'|(?P<op><=|>=|!=|<>|==|[(),*+\-=<>;]))")


def _tokens(q):
    out, i = [], 0
    q = q.rstrip()
    while i < len(q):
        m = _TOKEN.match(q, i)
        if not m or m.end() == i:
            if q[i:].strip() == "":
                break
            raise QueryError("unexpected character %r" % q[i:].lstrip()[:1])
        i = m.end()
        if m.group("id") is not None:
            out.append(("id", m.group("id")))
        elif m.group("bq") is not None:
            out.append(("id", m.group("bq")))
        elif m.group("dq") is not None:
            out.append(("id", m.group("dq")))
        elif m.group("num") is not None:
            if not m.group("num").isdigit():
                raise Unsupported("floating point literal")
            out.append(("int", int(m.group("num"))))
        elif m.group("str") is not None:
            s = m.group("str").replace("''", "'")
            s = re.sub(r"\\(.)", lambda g: {"n": "\n", "t": "\t", "0": "\0"}.get(g.group(1), g.group(1)), s)
            out.append(("str", s))
        else:
            out.append(("op", m.group("op")))
    return out


def lit_type(v):
    if v >= 0:
        return "UInt8" if v <= 0xFF else "UInt16" if v <= 0xFFFF else "UInt32" if v <= 0xFFFFFFFF else "UInt64"
    return "Int8" if v >= -128 else "Int16" if v >= -32768 else "Int32" if v >= -2 ** 31 else "Int64"


def bits(t):
    return int(re.sub(r"\D", "", t))


def add_type(a, b, minus):
    n = min(64, 2 * max(bits(a), bits(b)))
    signed = minus or a.startswith("Int") or b.startswith("Int")
    return ("Int" if signed else "UInt") + str(n)


def wrap(v, t):
    n = bits(t)
    v &= (1 << n) - 1
    if t.startswith("Int") and v >= 1 << (n - 1):
        v -= 1 << n
    return v


class _P:
    def __init__(self, q):
        self.t, self.i = _tokens(q), 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else ("end", None)

    def take(self):
        x = self.peek()
        self.i += 1
        return x

    def kw(self, w):
        k, v = self.peek()
        return k == "id" and v.lower() == w

    def op(self, o):
        return self.peek() == ("op", o)

    def need(self, o):
        if not self.op(o):
            raise QueryError("expected %r" % o)
        self.take()

    # expression → ("col", name) | ("int", value, type) | ("str", s) | ("iexpr", col, [steps]) | ("tostr", col) | ("todt", col)
    def primary(self):
        k, v = self.peek()
        if self.op("("):
            self.take(); e = self.expr(); self.need(")"); return e
        if self.op("-"):
            self.take(); k, v = self.take()
            if k != "int":
                raise Unsupported("unary minus")
            return ("int", -v, lit_type(-v))
        if k == "int":
            self.take(); return ("int", v, lit_type(v))
        if k == "str":
            self.take(); return ("str", v)
        if k != "id":
            raise QueryError("expected an expression")
        self.take()
        if not self.op("("):
            return ("col", v)
        self.take()
        if v in ("to" + t for t in INT_TYPES):
            a = self.expr(); self.need(")")
            ty = v[2:]
            if a[0] == "int":
                return ("int", wrap(a[1], ty), ty)
            if a[0] == "col":
                return ("iexpr", a[1], [("cast", ty)])
            if a[0] == "iexpr":
                return ("iexpr", a[1], a[2] + [("cast", ty)])
            raise Unsupported("cast of a non-integer expression")
        if v in ("toString", "toDateTime"):
            k2, c = self.take()
            if k2 != "id":
                raise Unsupported(v + " of an expression")
            self.need(")")
            return ("tostr" if v == "toString" else "todt", c)
        raise Unsupported("function " + v)

    def expr(self):
        a = self.primary()
        while self.op("+") or self.op("-"):
            minus = self.take()[1] == "-"
            b = self.primary()
            if b[0] != "int":
                if a[0] == "int" and not minus and b[0] in ("col", "iexpr"):
                    a, b = b, a
                else:
                    raise Unsupported("arithmetic")
            if a[0] == "int":
                ty = add_type(a[2], b[2], minus)
                a = ("int", wrap(a[1] - b[1] if minus else a[1] + b[1], ty), ty)
            elif a[0] == "col":
                a = ("iexpr", a[1], [("add", -b[1] if minus else b[1], b[2], minus)])
            elif a[0] == "iexpr":
                a = ("iexpr", a[1], a[2] + [("add", -b[1] if minus else b[1], b[2], minus)])
            else:
                raise Unsupported("arithmetic on a non-integer")
        return a

    def term(self):
        k, c = self.take()
        if k != "id":
            raise Unsupported("WHERE term")

        def lit():
            neg = False
            if self.op("-"):
                self.take(); neg = True
            k2, v = self.take()
            if k2 == "int":
                return -v if neg else v
            if k2 == "str" and not neg:
                return v
            raise Unsupported("WHERE literal")
        neg = False
        if self.kw("not"):
            self.take(); neg = True
            if not self.kw("in"):
                raise Unsupported("NOT")
        if self.kw("in"):
            self.take(); self.need("(")
            vals = [lit()]
            while self.op(","):
                self.take(); vals.append(lit())
            self.need(")")
            if len({type(x) for x in vals}) > 1:
                raise Unsupported("mixed list")
            return (c, "not in" if neg else "in", vals)
        k2, o = self.take()
        if k2 != "op" or o not in ("=", "==", "!=", "<>", "<", "<=", ">", ">="):
            raise QueryError("expected a comparison")
        return (c, {"==": "=", "<>": "!="}.get(o, o), lit())


def parse(query):
    """→ (items, where): items = [("*",) | (expr, name)], where = None | [[term, …], …] (OR of ANDs)."""
    p = _P(query)
    if not p.kw("select"):
        raise QueryError("the query must start with SELECT")
    p.take()
    items = []
    while True:
        if p.op("*"):
            p.take(); items.append(("*",))
        else:
            e = p.expr()
            name = e[1] if e[0] == "col" else None
            if p.kw("as"):
                p.take(); name = p.take()[1]
            elif p.peek()[0] == "id" and not p.kw("from"):
                name = p.take()[1]
            if name is None:
                raise Unsupported("an expression without an alias")
            items.append((e, name))
        if p.op(","):
            p.take(); continue
        break
    if not p.kw("from"):
        raise QueryError("expected FROM")
    p.take()
    if not p.kw("table"):
        raise Unsupported("FROM")
    p.take()
    where = None
    if p.kw("where"):
        p.take()
        where, conj = [], [p.term()]
        while True:
            if p.kw("and"):
                p.take(); conj.append(p.term())
            elif p.kw("or"):
                p.take(); where.append(conj); conj = [p.term()]
            else:
                break
        where.append(conj)
    if p.op(";"):
        p.take()
    if p.peek()[0] != "end":
        raise Unsupported("trailing clause %r" % (p.peek()[1],))
    return items, where


def resolve(query, schema_triples):
    """The result columns over an input schema [(name, yt type, is key)]: [(name, kind tuple, ch type, yt type, key)]."""
    items, _ = parse(query)
    names = [n for n, _, _ in schema_triples]
    types = {n: t for n, t, _ in schema_triples}
    keys = {n for n, _, k in schema_triples if k}
    out = []

    def col(c):
        if c not in types:
            raise QueryError("unknown column " + c)
        return YT_TO_CH[types[c]]
    for it in items:
        if it == ("*",):
            for n in names:
                out.append((n, ("col", n), YT_TO_CH[types[n]]))
            continue
        e, name = it
        if e[0] == "col":
            out.append((name, e, col(e[1])))
        elif e[0] == "int":
            out.append((name, e, e[2]))
        elif e[0] == "str":
            out.append((name, e, "String"))
        elif e[0] == "tostr":
            if col(e[1]) not in INT_TYPES + ["String"]:
                raise Unsupported("toString of " + types[e[1]])
            out.append((name, e, "String"))
        elif e[0] == "todt":
            if col(e[1]) not in ("Int32", "UInt32", "Int64", "UInt64", "DateTime"):
                raise Unsupported("toDateTime of " + types[e[1]])
            out.append((name, e, "DateTime"))
        elif e[0] == "iexpr":
            ty = col(e[1])
            if ty not in INT_TYPES:
                raise Unsupported("integer expression on " + types[e[1]])
            for s in e[2]:
                ty = s[1] if s[0] == "cast" else add_type(ty, s[2], s[3])
            out.append((name, e, ty))
    if len({o[0] for o in out}) != len(out):
        raise Unsupported("duplicate result column")
    res = [(n, e, ch, CH_TO_YT[ch], n in keys) for n, e, ch in out]
    if not any(r[4] for r in res):
        raise QueryError("result table has no primary key")
    return res


_GO = {"int8": "int8", "int16": "int16", "int32": "int32", "int64": "int64", "uint8": "uint8", "uint16": "uint16", "uint32": "uint32", "uint64": "uint64"}


def apply(query, rows, schema_triples):
    """rows: the Collapse'd sub-batch as [{"kind", "src", "values": {name: [gotype, value]}}] (a nil / absent value takes
    the ClickHouse default).  → [{"kind", "src", "values": [[gotype, value] …] | None, "old": [[gotype, value] …] | None}]."""
    _, where = parse(query)
    res = resolve(query, schema_triples)
    types = {n: t for n, t, _ in schema_triples}
    for k in (n for n, _, key in schema_triples if key):
        if not any(r[0] == k and r[1] == ("col", k) for r in res):
            raise Unsupported("key column does not pass through")

    def value(row, c):  # the cell as ClickHouse reads it from JSONEachRow
        v = row["values"].get(c)
        ch = YT_TO_CH[types[c]]
        if v is None or v[0] == "nil":
            return 0 if ch in INT_TYPES else (b"" if ch == "String" else (0, 0))
        if ch in INT_TYPES:
            return int(v[1])
        return v[1]

    def cmp(a, o, b):
        if isinstance(b, str):
            b = b.encode()
        if type(a) is not type(b) and not (isinstance(a, int) and isinstance(b, int)):
            raise Unsupported("comparison of %r with %r" % (type(a), type(b)))
        return {"=": a == b, "!=": a != b, "<": a < b, "<=": a <= b, ">": a > b, ">=": a >= b}[o]

    def holds(row):
        if where is None:
            return True
        for conj in where:
            ok = True
            for c, o, lit in conj:
                if c not in types:
                    raise QueryError("unknown column " + c)
                a = value(row, c)
                if o in ("in", "not in"):
                    hit = any(cmp(a, "=", x) for x in lit)
                    ok = ok and (hit if o == "in" else not hit)
                else:
                    ok = ok and cmp(a, o, lit)
            if ok:
                return True
        return False
    out = []
    for row in rows:
        if not holds(row):
            continue
        vals = []
        for name, e, ch, yt, key in res:
            if e[0] == "col":
                v = value(row, e[1])
                if ch in INT_TYPES:
                    vals.append([_GO[yt], int(v)])
                elif ch == "String":
                    vals.append(["string", v])
                elif ch == "Float64":
                    vals.append(["float64", float(v)])
                else:
                    vals.append(["time", tuple(v)])
            elif e[0] == "int":
                vals.append([_GO[yt], e[1]])
            elif e[0] == "str":
                vals.append(["string", e[1].encode()])
            elif e[0] == "tostr":
                v = value(row, e[1])
                vals.append(["string", v if isinstance(v, bytes) else str(v).encode()])
            elif e[0] == "todt":
                v = value(row, e[1])
                vals.append(["time", (int(v), 0) if isinstance(v, int) else tuple(v)])
            else:
                v = value(row, e[1])
                ty = YT_TO_CH[types[e[1]]]
                for s in e[2]:
                    if s[0] == "cast":
                        ty = s[1]; v = wrap(v, ty)
                    else:
                        ty = add_type(ty, s[2], s[3]); v = wrap(v + s[1], ty)
                vals.append([_GO[yt], v])
        kind = row["kind"]
        out.append({"kind": kind, "src": row["src"], "values": None if kind == "delete" else vals, "old": vals if kind in ("update", "delete") else None})
    return out
