"""CPU restatement of the `sql` transformer (pkg/transformer/registry/clickhouse/clickhouse_local.go:97-294) for the
predicate + cast subset the device takes.  TEST INFRASTRUCTURE: only tests/, smoke() and bench.py's cpu_baseline leg may
import this; the product path (transferia_amd/) never does.

The reference evaluates the query in an external `clickhouse-local` process.  That binary is not in /root/reference nor
in this image, so this file restates ClickHouse's *documented* behaviour for the subset:
  * integer literal types: the narrowest of UInt8/16/32/64 (Int8/16/32/64 when negative);
  * a + b, a - b over integers: signed if either side is (always for minus), bits = min(64, 2 * max(bits));
  * toIntN / toUIntN: two's complement truncation;  toString(int) = decimal text;  toDateTime(int) = epoch seconds;
  * comparisons / IN over integers and strings; AND / OR / NOT, parentheses; comparisons and logic yield UInt8;
  * a * b typed like a + b; -a is signed (an unsigned operand takes the next size); integers of different signedness
    compare by value; length(s) = bytes as UInt64; lower / upper touch ASCII letters only; cityHash64(String) = CityHash64
    of CityHash v1.0.2 over the bytes (restated below from the published algorithm: unpinned like the rest);
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


_TOKEN = re.compile(r"\s*(?:(?P<id>[A-Za-z_][A-Za-z_0-9]*)|`(?P<bq>[^`]*)`|\"(?P<dq>[^\"]*)\"|(?P<num>\d+(?:\.\d+|[eE][-+]?\d+)?)|'(?P<str>(?:[^'\\]|\\.|'')*)'|(?P<op><=|>=|!=|<>|==|[(),*+\-=<>;/%]))")


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


M64 = (1 << 64) - 1
K0, K1, K2, K3 = 0xc3a5c85c97cb3127, 0xb492b66fbe98f273, 0x9ae16a3b2f90404f, 0xc949d7c7509e6557


def city_hash64(s: bytes) -> int:
    """CityHash64 of CityHash v1.0.2 (city.cc of that release), 64-bit arithmetic modulo 2^64."""
    def f64(i):
        return int.from_bytes(s[i:i + 8], "little")

    def f32(i):
        return int.from_bytes(s[i:i + 4], "little")

    def rot(v, k):
        return v if k == 0 else ((v >> k) | (v << (64 - k))) & M64

    def mix(v):
        return v ^ (v >> 47)

    def h16(u, v):
        m = 0x9ddfea08eb382d69
        a = ((u ^ v) * m) & M64
        a ^= a >> 47
        b = ((v ^ a) * m) & M64
        b ^= b >> 47
        return (b * m) & M64

    def weak(w, x, y, z, a, b):
        a = (a + w) & M64
        b = rot((b + a + z) & M64, 21)
        c = a
        a = (a + x + y) & M64
        b = (b + rot(a, 44)) & M64
        return (a + z) & M64, (b + c) & M64

    def weak_at(i, a, b):
        return weak(f64(i), f64(i + 8), f64(i + 16), f64(i + 24), a, b)
    n = len(s)
    if n <= 16:
        if n > 8:
            a, b = f64(0), f64(n - 8)
            return h16(a, rot((b + n) & M64, n)) ^ b
        if n >= 4:
            return h16((n + (f32(0) << 3)) & M64, f32(n - 4))
        if n > 0:
            y = (s[0] + (s[n >> 1] << 8)) & 0xFFFFFFFF
            z = (n + (s[n - 1] << 2)) & 0xFFFFFFFF
            return (mix(((y * K2) & M64) ^ ((z * K3) & M64)) * K2) & M64
        return K2
    if n <= 32:
        a, b, c, d = (f64(0) * K1) & M64, f64(8), (f64(n - 8) * K2) & M64, (f64(n - 16) * K0) & M64
        return h16((rot((a - b) & M64, 43) + rot(c, 30) + d) & M64, (a + rot(b ^ K3, 20) - c + n) & M64)
    if n <= 64:
        z = f64(24)
        a = (f64(0) + (n + f64(n - 16)) * K0) & M64
        b = rot((a + z) & M64, 52)
        c = rot(a, 37)
        a = (a + f64(8)) & M64
        c = (c + rot(a, 7)) & M64
        a = (a + f64(16)) & M64
        vf, vs = (a + z) & M64, (b + rot(a, 31) + c) & M64
        a = (f64(16) + f64(n - 32)) & M64
        z = f64(n - 8)
        b = rot((a + z) & M64, 52)
        c = rot(a, 37)
        a = (a + f64(n - 24)) & M64
        c = (c + rot(a, 7)) & M64
        a = (a + f64(n - 16)) & M64
        wf, ws = (a + z) & M64, (b + rot(a, 31) + c) & M64
        r = mix(((vf + ws) * K2 + (wf + vs) * K0) & M64)
        return (mix((r * K0 + vs) & M64) * K2) & M64
    x, y, z = f64(0), f64(n - 16) ^ K1, f64(n - 56) ^ K0
    v = weak_at(n - 64, n, y)
    w = weak_at(n - 32, (n * K1) & M64, K0)
    z = (z + mix(v[1]) * K1) & M64
    x = (rot((z + x) & M64, 39) * K1) & M64
    y = (rot(y, 33) * K1) & M64
    left, i = (n - 1) & ~63, 0
    while True:
        x = (rot((x + y + v[0] + f64(i + 16)) & M64, 37) * K1) & M64
        y = (rot((y + v[1] + f64(i + 48)) & M64, 42) * K1) & M64
        x ^= w[1]
        y ^= v[0]
        z = rot(z ^ w[0], 33)
        v = weak_at(i, (v[1] * K1) & M64, (x + w[0]) & M64)
        w = weak_at(i + 32, (z + w[1]) & M64, y)
        z, x = x, z
        i += 64
        left -= 64
        if left == 0:
            break
    return h16((h16(v[0], w[0]) + mix(y) * K1 + z) & M64, (h16(v[1], w[1]) + x) & M64)


def neg_type(a):
    return "Int" + str(bits(a) if a.startswith("Int") else min(64, 2 * bits(a)))


CMP = ("=", "!=", "<", "<=", ">", ">=")


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

    # nodes: ("col", name) | ("int", value, type) | ("str", s) | ("bin", op, a, b) | ("neg", a) | ("cast", type, a) | ("fn", name, a)
    #        | ("cmp", op, a, b) | ("and" | "or", a, b) | ("not", a) | ("in", negated, a, [literal nodes])
    def primary(self):
        k, v = self.peek()
        if self.op("("):
            self.take(); e = self.bexpr(); self.need(")"); return e
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
            a = self.bexpr(); self.need(")")
            return ("cast", v[2:], a)
        if v in ("toString", "toDateTime", "length", "cityHash64", "lower", "upper", "lcase", "ucase", "not"):
            a = self.bexpr()
            if self.op(","):
                raise Unsupported(v + " with several arguments")
            self.need(")")
            return ("not", a) if v == "not" else ("fn", {"lcase": "lower", "ucase": "upper"}.get(v, v), a)
        raise Unsupported("function " + v)

    def unary(self):
        if self.op("-"):
            self.take()
            k, v = self.peek()
            if k == "int":
                self.take(); return ("int", -v, lit_type(-v))
            return ("neg", self.unary())
        if self.op("+"):
            self.take(); return self.unary()
        return self.primary()

    def mul(self):
        a = self.unary()
        while True:
            if self.op("*"):
                self.take(); a = ("bin", "*", a, self.unary()); continue
            if self.op("/") or self.op("%"):
                raise Unsupported("division / modulo")
            return a

    def addsub(self):
        a = self.mul()
        while self.op("+") or self.op("-"):
            o = self.take()[1]
            a = ("bin", o, a, self.mul())
        return a

    def literal(self):
        neg = False
        if self.op("-"):
            self.take(); neg = True
        k, v = self.take()
        if k == "int":
            v = -v if neg else v
            return ("int", v, lit_type(v))
        if k == "str" and not neg:
            return ("str", v)
        if k == "end":
            raise QueryError("expected a literal")
        raise Unsupported("IN literal")

    def cmp(self):
        a = self.addsub()
        neg = False
        if self.kw("not"):
            self.take(); neg = True
            if not self.kw("in"):
                raise Unsupported("NOT")
        if self.kw("in"):
            self.take(); self.need("(")
            vals = [self.literal()]
            while self.op(","):
                self.take(); vals.append(self.literal())
            self.need(")")
            if len({x[0] for x in vals}) > 1:
                raise Unsupported("mixed list")
            return ("in", neg, a, vals)
        k, o = self.peek()
        if k == "op" and o in ("=", "==", "!=", "<>", "<", "<=", ">", ">="):
            self.take()
            if self.peek()[0] == "end":
                raise QueryError("expected an expression")
            return ("cmp", {"==": "=", "<>": "!="}.get(o, o), a, self.addsub())
        return a

    def bnot(self):
        if self.kw("not"):
            self.take(); return ("not", self.bnot())
        return self.cmp()

    def band(self):
        a = self.bnot()
        while self.kw("and"):
            self.take(); a = ("and", a, self.bnot())
        return a

    def bexpr(self):
        a = self.band()
        while self.kw("or"):
            self.take(); a = ("or", a, self.band())
        return a


def parse(query):
    """→ (items, where): items = [("*",) | (node, name)], where = None | node."""
    p = _P(query)
    if not p.kw("select"):
        raise QueryError("the query must start with SELECT")
    p.take()
    items = []
    while True:
        if p.op("*"):
            p.take(); items.append(("*",))
        else:
            e = p.bexpr()
            name = e[1] if e[0] == "col" else None
            if p.kw("as"):
                p.take()
                k, name = p.take()
                if k != "id":
                    raise QueryError("expected an alias")
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
        if p.peek()[0] == "end":
            raise QueryError("expected a condition")
        where = p.bexpr()
    if p.op(";"):
        p.take()
    if p.peek()[0] != "end":
        raise Unsupported("trailing clause %r" % (p.peek()[1],))
    return items, where


def type_of(e, types, top=False):
    """ClickHouse's type of a node over {column: yt type}.  `top`: the node is a select item (any column type passes through)."""
    k = e[0]
    if k == "col":
        if e[1] not in types:
            raise QueryError("unknown column " + e[1])
        t = YT_TO_CH[types[e[1]]]
        if not top and t not in INT_TYPES and t != "String":
            raise Unsupported("a %s column inside an expression" % types[e[1]])
        return t
    if k == "int":
        return e[2]
    if k == "str":
        return "String"

    def integer(x):
        t = type_of(x, types)
        if t not in INT_TYPES:
            raise (QueryError if t == "String" else Unsupported)("an integer is needed")
        return t
    if k == "bin":
        return add_type(integer(e[2]), integer(e[3]), e[1] == "-")
    if k == "neg":
        return neg_type(integer(e[1]))
    if k == "cast":
        if type_of(e[2], types) == "String":
            raise Unsupported("toIntN of text")
        integer(e[2])
        return e[1]
    if k == "fn":
        f = e[1]
        if f == "toDateTime":
            if not top or e[2][0] != "col":
                raise Unsupported("toDateTime of an expression")
            if type_of(e[2], types, True) not in ("Int32", "UInt32", "Int64", "UInt64", "DateTime"):
                raise Unsupported("toDateTime of " + types[e[2][1]])
            return "DateTime"
        a = type_of(e[2], types, top and f == "toString" and e[2][0] == "col")
        if f in ("length", "cityHash64"):
            if a != "String":
                raise Unsupported(f + " of a non-text value")
            return "UInt64"
        if f in ("lower", "upper"):
            if a != "String":
                raise QueryError(f + " of a non-text value")
            return "String"
        if f == "toString":
            if a not in INT_TYPES and a != "String":
                raise Unsupported("toString of " + a)
            return "String"
    if k == "cmp":
        a, b = type_of(e[2], types), type_of(e[3], types)
        if (a == "String") != (b == "String"):
            raise QueryError("text compared with a number")
        if a == "String" and (e[2][0] == "str") == (e[3][0] == "str"):
            raise Unsupported("text comparison without exactly one literal")
        return "UInt8"
    if k in ("and", "or"):
        integer(e[1]); integer(e[2])
        return "UInt8"
    if k == "not":
        integer(e[1])
        return "UInt8"
    if k == "in":
        a = type_of(e[2], types)
        if (a == "String") != (e[3][0][0] == "str"):
            raise QueryError("IN list of another type")
        if a != "String":
            integer(e[2])
        return "UInt8"
    raise AssertionError(e)


def resolve(query, schema_triples):
    """The result columns over an input schema [(name, yt type, is key)]: [(name, node, ch type, yt type, key)]."""
    items, where = parse(query)
    names = [n for n, _, _ in schema_triples]
    types = {n: t for n, t, _ in schema_triples}
    keys = {n for n, _, k in schema_triples if k}
    out = []
    for it in items:
        if it == ("*",):
            for n in names:
                out.append((n, ("col", n), YT_TO_CH[types[n]]))
            continue
        e, name = it
        out.append((name, e, type_of(e, types, True)))
    if where is not None and type_of(where, types) == "String":
        raise QueryError("the WHERE condition is text")
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

    def ev(e, row):
        """→ the value: a Python int (exact, within its ClickHouse type's range) or bytes"""
        k = e[0]
        if k == "col":
            return value(row, e[1])
        if k == "int":
            return e[1]
        if k == "str":
            return e[1].encode() if isinstance(e[1], str) else e[1]
        if k == "bin":
            a, b = ev(e[2], row), ev(e[3], row)
            return wrap(a + b if e[1] == "+" else a - b if e[1] == "-" else a * b, type_of(e, types))
        if k == "neg":
            return wrap(-ev(e[1], row), type_of(e, types))
        if k == "cast":
            return wrap(ev(e[2], row), e[1])
        if k == "fn":
            a = ev(e[2], row)
            if e[1] == "length":
                return len(a)
            if e[1] == "cityHash64":
                return city_hash64(bytes(a))
            if e[1] == "lower":
                return bytes(c + 32 if 65 <= c <= 90 else c for c in a)
            if e[1] == "upper":
                return bytes(c - 32 if 97 <= c <= 122 else c for c in a)
            if e[1] == "toString":
                return a if isinstance(a, (bytes, bytearray)) else str(a).encode()
        if k == "cmp":
            a, b = ev(e[2], row), ev(e[3], row)
            if isinstance(a, (bytes, bytearray)):
                a, b = bytes(a), bytes(b)
            return int({"=": a == b, "!=": a != b, "<": a < b, "<=": a <= b, ">": a > b, ">=": a >= b}[e[1]])
        if k == "and":
            return int(ev(e[1], row) != 0 and ev(e[2], row) != 0)
        if k == "or":
            return int(ev(e[1], row) != 0 or ev(e[2], row) != 0)
        if k == "not":
            return int(ev(e[1], row) == 0)
        if k == "in":
            a = ev(e[2], row)
            if isinstance(a, (bytes, bytearray)):
                a = bytes(a)
            hit = any(a == ev(x, row) for x in e[3])
            return int(hit != e[1])
        raise AssertionError(e)
    out = []
    for row in rows:
        if where is not None and ev(where, row) == 0:
            continue
        vals = []
        for name, e, ch, yt, key in res:
            if e[0] == "fn" and e[1] == "toDateTime":
                v = value(row, e[2][1])
                vals.append(["time", (int(v), 0) if isinstance(v, int) else tuple(v)])
                continue
            v = ev(e, row)
            if ch in INT_TYPES:
                vals.append([_GO[yt], int(v)])
            elif ch == "String":
                vals.append(["string", bytes(v)])
            elif ch == "Float64":
                vals.append(["float64", float(v)])
            else:
                vals.append(["time", tuple(v)])
        kind = row["kind"]
        out.append({"kind": kind, "src": row["src"], "values": None if kind == "delete" else vals, "old": vals if kind in ("update", "delete") else None})
    return out
