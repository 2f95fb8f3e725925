// tf_sql.cpp — the `sql` transformer's query, for the subset that runs on the device.
//
// The reference hands every batch to an external `clickhouse-local --query Q` process (pkg/transformer/registry/
// clickhouse/clickhouse_local.go:97-294): whatever ClickHouse's SQL engine accepts is legal there.  SURVEY.md §7 and
// BASELINE.json configs[2] ("SQL-transformer predicate+cast") ask for the predicate + cast part of that on the device.
// This file parses exactly that subset and refuses everything else by name (TFGPU_ERR_UNSUPPORTED), so a query the
// device cannot evaluate the way ClickHouse does never runs here:
//
//   SELECT item [, item …] FROM table [WHERE cond] [;]
//   item  :=  *  |  expr [AS alias]                     (an expression that is not a plain column needs an alias)
//   expr  :=  column | integer | 'string' | ( expr ) | expr + integer | expr - integer | integer + integer
//          |  toInt8 … toInt64 ( expr ) | toUInt8 … toUInt64 ( expr ) | toString ( column ) | toDateTime ( column )
//             (toString of an integer / text column; toDateTime of a 32 / 64-bit integer column holding epoch seconds that
//              fit DateTime, 0 … 2^32 - 1: ClickHouse saturates outside, the device does not — values are the caller's)
//   cond  :=  conj { OR conj }        conj := term { AND term }
//   term  :=  column (= | == | != | <> | < | <= | > | >=) literal  |  column [NOT] IN ( literal, … )
//
// Typing follows ClickHouse: an integer literal has the narrowest type that holds it (UInt8 … UInt64, Int8 … Int64 when
// negative), a + b of integers is signed if either is, with twice the wider operand's bits (at most 64), toIntN wraps
// (two's complement truncation).  Result columns map back to YT types through typesystem.go's Source rules
// (pkg/providers/clickhouse/typesystem.go:15-33: String → `string`, DateTime → `datetime`, …), and a result column is a
// primary key when it carries the NAME of an input key column (clickhouse_local.go:393-421).
//
// PARITY: the reference's tests for this transformer need the clickhouse binary and assert row counts, kinds and one
// value (clickhouse_local_test.go:87-124: toInt8(id+1) == int8(2)); tests/test_sql.py replays those.  Everything else
// here follows ClickHouse's documented behaviour and is labelled "parity unpinned" in DESIGN.md §4.
#include <cctype>
#include <cstdlib>

#include "tf_plan.hpp"

namespace tf {
namespace {

struct Tok { enum K { End, Ident, Int, Str, Punct } k = End; std::string s; int64_t v = 0; bool neg_overflow = false; };

struct Lexer {
  const std::string &q; size_t i = 0;
  explicit Lexer(const std::string &s) : q(s) {}
  [[noreturn]] static void bad(const std::string &m) { throw Error(TFGPU_ERR_CONFIG, "sql: " + m); }
  Tok next() {
    while (i < q.size() && std::isspace((unsigned char)q[i])) i++;
    Tok t;
    if (i >= q.size()) return t;
    const char c = q[i];
    if (std::isalpha((unsigned char)c) || c == '_') {
      size_t j = i;
      while (j < q.size() && (std::isalnum((unsigned char)q[j]) || q[j] == '_')) j++;
      t.k = Tok::Ident; t.s = q.substr(i, j - i); i = j; return t;
    }
    if (c == '`' || c == '"') {  // quoted identifier
      size_t j = q.find(c, i + 1);
      if (j == std::string::npos) bad("unterminated quoted identifier");
      t.k = Tok::Ident; t.s = q.substr(i + 1, j - i - 1); i = j + 1; return t;
    }
    if (std::isdigit((unsigned char)c)) {
      size_t j = i; unsigned long long v = 0;
      while (j < q.size() && std::isdigit((unsigned char)q[j])) { if (v > 1844674407370955161ull) bad("integer literal out of range"); v = v * 10 + (unsigned)(q[j] - '0'); j++; }
      if (j < q.size() && (q[j] == '.' || q[j] == 'e' || q[j] == 'E')) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: floating point literals are outside the device subset");
      if (v > 9223372036854775807ull) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: integer literals above Int64 are outside the device subset");
      t.k = Tok::Int; t.v = (int64_t)v; i = j; return t;
    }
    if (c == '\'') {
      std::string s; size_t j = i + 1;
      for (;;) {
        if (j >= q.size()) bad("unterminated string literal");
        if (q[j] == '\\') { if (j + 1 >= q.size()) bad("unterminated string literal"); const char e = q[j + 1]; s += e == 'n' ? '\n' : e == 't' ? '\t' : e == '0' ? '\0' : e; j += 2; continue; }
        if (q[j] == '\'') { if (j + 1 < q.size() && q[j + 1] == '\'') { s += '\''; j += 2; continue; } break; }
        s += q[j++];
      }
      t.k = Tok::Str; t.s = s; i = j + 1; return t;
    }
    static const char *two[] = {"<=", ">=", "!=", "<>", "=="};
    for (const char *p : two) if (q.compare(i, 2, p) == 0) { t.k = Tok::Punct; t.s = p; i += 2; return t; }
    if (std::string("(),*+-=<>;").find(c) != std::string::npos) { t.k = Tok::Punct; t.s = std::string(1, c); i++; return t; }
    bad(std::string("unexpected character '") + c + "'");
  }
};

bool ieq(const std::string &a, const char *b) {
  size_t n = std::char_traits<char>::length(b);
  if (a.size() != n) return false;
  for (size_t i = 0; i < n; i++) if (std::tolower((unsigned char)a[i]) != std::tolower((unsigned char)b[i])) return false;
  return true;
}

struct Parser {
  Lexer lx; Tok cur;
  explicit Parser(const std::string &q) : lx(q) { cur = lx.next(); }
  void adv() { cur = lx.next(); }
  bool punct(const char *p) const { return cur.k == Tok::Punct && cur.s == p; }
  bool kw(const char *w) const { return cur.k == Tok::Ident && ieq(cur.s, w); }
  void expect_punct(const char *p) { if (!punct(p)) Lexer::bad(std::string("expected '") + p + "'" + (cur.k == Tok::End ? " at the end of the query" : " near '" + cur.s + "'")); adv(); }

  // ClickHouse's type of an integer literal
  static int lit_type(int64_t v) {
    if (v >= 0) return v <= 0xFF ? SQL_U8 : v <= 0xFFFF ? SQL_U16 : v <= 0xFFFFFFFFll ? SQL_U32 : SQL_U64;
    return v >= -128 ? SQL_I8 : v >= -32768 ? SQL_I16 : v >= -2147483648ll ? SQL_I32 : SQL_I64;
  }

  // expr: yields an item whose `name` is the canonical text (used when the item is a plain column)
  SqlItem primary() {
    SqlItem it;
    if (punct("(")) { adv(); it = expr(); expect_punct(")"); return it; }
    if (punct("-")) {
      adv();
      if (cur.k != Tok::Int) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: unary minus is only taken in front of an integer literal");
      it.kind = SQL_CONST_INT; it.ival = -cur.v; it.cast = lit_type(it.ival); adv(); return it;
    }
    if (cur.k == Tok::Int) { it.kind = SQL_CONST_INT; it.ival = cur.v; it.cast = lit_type(cur.v); adv(); return it; }
    if (cur.k == Tok::Str) { it.kind = SQL_CONST_STR; it.sval = cur.s; it.cast = SQL_STRING; adv(); return it; }
    if (cur.k != Tok::Ident) Lexer::bad("expected an expression" + (cur.k == Tok::End ? std::string(" at the end of the query") : " near '" + cur.s + "'"));
    const std::string id = cur.s;
    adv();
    if (!punct("(")) { it.kind = SQL_COLUMN; it.src = id; it.name = id; return it; }
    adv();
    static const struct { const char *fn; int ty; } casts[] = {{"toInt8", SQL_I8}, {"toInt16", SQL_I16}, {"toInt32", SQL_I32}, {"toInt64", SQL_I64},
                                                                {"toUInt8", SQL_U8}, {"toUInt16", SQL_U16}, {"toUInt32", SQL_U32}, {"toUInt64", SQL_U64}};
    for (auto &c : casts)
      if (id == c.fn) {  // ClickHouse function names are case-sensitive
        SqlItem a = expr();
        expect_punct(")");
        if (a.kind == SQL_CONST_INT) { a.ival = wrap(a.ival, c.ty); a.cast = c.ty; return a; }
        if (a.kind != SQL_COLUMN && a.kind != SQL_INT_EXPR) throw Error(TFGPU_ERR_UNSUPPORTED, std::string("sql: ") + id + "() of a non-integer expression");
        a.kind = SQL_INT_EXPR; a.name.clear();
        a.steps.push_back(SqlStep{true, c.ty, 0, false});
        return a;
      }
    if (id == "toString" || id == "toDateTime") {
      if (cur.k != Tok::Ident) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: " + id + "() takes a plain column in the device subset");
      it.kind = id == "toString" ? SQL_TO_STRING : SQL_TO_DATETIME; it.src = cur.s; it.cast = id == "toString" ? SQL_STRING : SQL_DATETIME;
      adv();
      expect_punct(")");
      return it;
    }
    throw Error(TFGPU_ERR_UNSUPPORTED, "sql: function " + id + "() is outside the device subset (predicate + cast): keep this transformer on the host");
  }
  static int64_t wrap(int64_t v, int ty) {
    switch (ty) {
      case SQL_I8: return (int8_t)v; case SQL_I16: return (int16_t)v; case SQL_I32: return (int32_t)v;
      case SQL_U8: return (uint8_t)v; case SQL_U16: return (uint16_t)v; case SQL_U32: return (uint32_t)v;
      default: return v;
    }
  }
  static int bits_of(int ty) { return ty == SQL_I8 || ty == SQL_U8 ? 8 : ty == SQL_I16 || ty == SQL_U16 ? 16 : ty == SQL_I32 || ty == SQL_U32 ? 32 : 64; }
  static bool signed_of(int ty) { return ty == SQL_I8 || ty == SQL_I16 || ty == SQL_I32 || ty == SQL_I64; }
  // the type of a + b / a - b over integers (NumberTraits::ResultOfAdditionMultiplication / Subtraction)
  static int add_type(int a, int b, bool minus) {
    const int bits = std::min(64, 2 * std::max(bits_of(a), bits_of(b)));
    const bool sg = minus || signed_of(a) || signed_of(b);
    return sg ? (bits == 16 ? SQL_I16 : bits == 32 ? SQL_I32 : SQL_I64) : (bits == 16 ? SQL_U16 : bits == 32 ? SQL_U32 : SQL_U64);
  }
  SqlItem expr() {
    SqlItem a = primary();
    while (punct("+") || punct("-")) {
      const bool minus = punct("-");
      adv();
      SqlItem b = primary();
      if (b.kind != SQL_CONST_INT && !(a.kind == SQL_CONST_INT && !minus && (b.kind == SQL_COLUMN || b.kind == SQL_INT_EXPR))) {
        throw Error(TFGPU_ERR_UNSUPPORTED, "sql: arithmetic is column ± integer (or integer ± integer) in the device subset");
      }
      if (b.kind != SQL_CONST_INT) std::swap(a, b);  // integer + column
      if (a.kind == SQL_CONST_INT) {  // constant folding, with ClickHouse's result type
        const int ty = add_type(a.cast, b.cast, minus);
        a.ival = wrap(minus ? a.ival - b.ival : a.ival + b.ival, ty); a.cast = ty;
        continue;
      }
      if (a.kind != SQL_COLUMN && a.kind != SQL_INT_EXPR) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: arithmetic on a non-integer expression");
      // column ± literal: the column's ClickHouse type is known when the schema is (the steps are typed in sql_resolve)
      a.steps.push_back(SqlStep{false, b.cast, minus ? -b.ival : b.ival, minus});
      a.kind = SQL_INT_EXPR; a.name.clear();
    }
    return a;
  }

  FTerm term() {
    if (cur.k != Tok::Ident) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: a WHERE term starts with a column in the device subset (parenthesised conditions are not taken)");
    FTerm t; t.attr = cur.s;
    adv();
    auto literal = [&](FTerm &ft) {
      bool neg = false;
      if (punct("-")) { neg = true; adv(); }
      if (cur.k == Tok::End) Lexer::bad("expected a literal at the end of the query");
      if (cur.k == Tok::Int) { if (ft.strs.size()) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: a list mixes strings and integers"); ft.vtype = FV_INT; ft.ints.push_back(neg ? -cur.v : cur.v); }
      else if (cur.k == Tok::Str && !neg) { if (ft.ints.size()) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: a list mixes strings and integers"); ft.vtype = FV_STRING; ft.strs.push_back(cur.s); }
      else throw Error(TFGPU_ERR_UNSUPPORTED, "sql: the right side of a WHERE term is an integer or string literal in the device subset");
      adv();
    };
    bool negate = false;
    if (kw("not")) { negate = true; adv(); if (!kw("in")) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: NOT is taken in front of IN only"); }
    if (kw("in")) {
      adv(); expect_punct("(");
      t.op = negate ? F_NOTIN : F_IN; t.is_list = true;
      literal(t);
      while (punct(",")) { adv(); literal(t); }
      expect_punct(")");
      return t;
    }
    if (cur.k != Tok::Punct) Lexer::bad("expected a comparison after " + t.attr);
    const std::string op = cur.s;
    if (op == "=" || op == "==") t.op = F_EQ; else if (op == "!=" || op == "<>") t.op = F_NE; else if (op == "<") t.op = F_LT; else if (op == "<=") t.op = F_LE;
    else if (op == ">") t.op = F_GT; else if (op == ">=") t.op = F_GE; else Lexer::bad("expected a comparison after " + t.attr);
    adv();
    literal(t);
    return t;
  }
};

}  // namespace

// the query → plan fields (items, WHERE as OR of ANDs)
void sql_parse(const std::string &query, tfgpu_plan &p) {
  Parser ps(query);
  if (!ps.kw("select")) throw Error(TFGPU_ERR_CONFIG, "sql: the query must start with SELECT" + (ps.cur.k == Tok::End ? std::string() : " (got '" + ps.cur.s + "')"));
  ps.adv();
  for (;;) {
    SqlItem it;
    if (ps.punct("*")) { ps.adv(); it.kind = SQL_STAR; }
    else {
      it = ps.expr();
      if (ps.kw("as")) { ps.adv(); if (ps.cur.k != Tok::Ident) Lexer::bad("expected an alias after AS"); it.name = ps.cur.s; ps.adv(); }
      else if (ps.cur.k == Tok::Ident && !ps.kw("from")) { it.name = ps.cur.s; ps.adv(); }  // alias without AS
      if (it.name.empty()) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: an expression in the select list needs an alias in the device subset (ClickHouse would name the column after the expression's text)");
    }
    p.sql_items.push_back(std::move(it));
    if (ps.punct(",")) { ps.adv(); continue; }
    break;
  }
  if (!ps.kw("from")) Lexer::bad("expected FROM" + (ps.cur.k == Tok::End ? std::string(" at the end of the query") : " near '" + ps.cur.s + "'"));
  ps.adv();
  if (!ps.kw("table")) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: FROM takes the input `table` only");
  ps.adv();
  if (ps.kw("where")) {
    ps.adv();
    FExpr conj;
    conj.terms.push_back(ps.term());
    for (;;) {
      if (ps.kw("and")) { ps.adv(); conj.terms.push_back(ps.term()); continue; }
      if (ps.kw("or")) { ps.adv(); p.exprs.push_back(std::move(conj)); conj = FExpr(); conj.terms.push_back(ps.term()); continue; }
      break;
    }
    p.exprs.push_back(std::move(conj));
    p.sql_has_where = true;
  }
  if (ps.punct(";")) ps.adv();
  if (ps.cur.k != Tok::End) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: '" + ps.cur.s + "' (GROUP BY / ORDER BY / LIMIT / JOIN / subqueries …) is outside the device subset: keep this transformer on the host");
}

// YT type of a ClickHouse result type (typesystem.go Source rules), and back for the input side (Target rules)
int sql_yt_of(int ty) {
  switch (ty) {
    case SQL_I8: return TFGPU_T_INT8; case SQL_I16: return TFGPU_T_INT16; case SQL_I32: return TFGPU_T_INT32; case SQL_I64: return TFGPU_T_INT64;
    case SQL_U8: return TFGPU_T_UINT8; case SQL_U16: return TFGPU_T_UINT16; case SQL_U32: return TFGPU_T_UINT32; case SQL_U64: return TFGPU_T_UINT64;
    case SQL_F64: return TFGPU_T_FLOAT64; case SQL_STRING: return TFGPU_T_BYTES; case SQL_DATE: return TFGPU_T_DATE; case SQL_DATETIME: return TFGPU_T_DATETIME;
    case SQL_DATETIME64: return TFGPU_T_TIMESTAMP;
    default: return TFGPU_T_INVALID;
  }
}
int sql_ch_of(int yt) {
  switch (yt) {
    case TFGPU_T_INT8: return SQL_I8; case TFGPU_T_INT16: return SQL_I16; case TFGPU_T_INT32: return SQL_I32; case TFGPU_T_INT64: return SQL_I64;
    case TFGPU_T_UINT8: return SQL_U8; case TFGPU_T_UINT16: return SQL_U16; case TFGPU_T_UINT32: return SQL_U32; case TFGPU_T_UINT64: return SQL_U64;
    case TFGPU_T_BOOLEAN: return SQL_U8; case TFGPU_T_FLOAT64: return SQL_F64; case TFGPU_T_FLOAT32: return SQL_F64;
    case TFGPU_T_BYTES: case TFGPU_T_UTF8: case TFGPU_T_ANY: return SQL_STRING;
    case TFGPU_T_DATE: return SQL_DATE; case TFGPU_T_DATETIME: return SQL_DATETIME; case TFGPU_T_TIMESTAMP: return SQL_DATETIME64;
    default: return SQL_PENDING;
  }
}

// The select list over a concrete input schema: every item with its source column, ClickHouse type and YT type.
// Throws TFGPU_ERR_CONFIG for what ClickHouse would refuse (unknown column) and TFGPU_ERR_UNSUPPORTED for what the device
// subset does not take.
std::vector<SqlOut> sql_resolve(const tfgpu_plan &p, const std::vector<SchemaCol> &in) {
  std::vector<SqlOut> out;
  auto find = [&](const std::string &n) -> int { for (size_t i = 0; i < in.size(); i++) if (in[i].name == n) return (int)i; return -1; };
  auto need = [&](const std::string &n) { const int i = find(n); if (i < 0) throw Error(TFGPU_ERR_CONFIG, "sql: unknown column " + n + " (ClickHouse: Missing columns)"); return i; };
  auto is_int = [](int ty) { return ty >= SQL_I8 && ty <= SQL_U64; };
  for (const SqlItem &it : p.sql_items) {
    switch (it.kind) {
      case SQL_STAR:
        for (size_t i = 0; i < in.size(); i++) { SqlOut o; o.kind = SQL_COLUMN; o.name = in[i].name; o.src = (int)i; o.ch = sql_ch_of(in[i].dtype); out.push_back(o); }
        break;
      case SQL_COLUMN: { SqlOut o; o.kind = SQL_COLUMN; o.name = it.name; o.src = need(it.src); o.ch = sql_ch_of(in[(size_t)o.src].dtype); out.push_back(o); break; }
      case SQL_CONST_INT: { SqlOut o; o.kind = SQL_CONST_INT; o.name = it.name; o.ival = it.ival; o.ch = it.cast; out.push_back(o); break; }
      case SQL_CONST_STR: { SqlOut o; o.kind = SQL_CONST_STR; o.name = it.name; o.sval = it.sval; o.ch = SQL_STRING; out.push_back(o); break; }
      case SQL_TO_STRING: case SQL_TO_DATETIME: {
        SqlOut o; o.kind = it.kind; o.name = it.name; o.src = need(it.src); o.ch = it.cast;
        const int sch = sql_ch_of(in[(size_t)o.src].dtype);
        if (it.kind == SQL_TO_STRING && !(is_int(sch) || sch == SQL_STRING)) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: toString() of a " + type_name(in[(size_t)o.src].dtype) + " column (ClickHouse's text form of dates / floats) is outside the device subset");
        if (it.kind == SQL_TO_DATETIME && !(sch == SQL_I32 || sch == SQL_U32 || sch == SQL_I64 || sch == SQL_U64 || sch == SQL_DATETIME))
          throw Error(TFGPU_ERR_UNSUPPORTED, "sql: toDateTime() takes a 32 / 64-bit integer (epoch seconds) column in the device subset");
        out.push_back(o);
        break;
      }
      case SQL_INT_EXPR: {
        SqlOut o; o.kind = SQL_INT_EXPR; o.name = it.name; o.src = need(it.src);
        int ty = sql_ch_of(in[(size_t)o.src].dtype);
        if (!is_int(ty)) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: integer arithmetic / casts on the " + type_name(in[(size_t)o.src].dtype) + " column " + it.src);
        // replay the steps with ClickHouse's typing: each ± literal widens, each cast wraps; the device evaluates in int64
        // with a wrap to `bits` after every step that narrows
        for (const SqlStep &s : it.steps) {
          ty = s.is_cast ? s.ty : Parser::add_type(ty, s.ty, s.minus);
          o.ops.push_back(SqlOp{s.is_cast ? 0 : s.addend, ty});
        }
        o.ch = ty;
        out.push_back(o);
        break;
      }
      default: throw Error(TFGPU_ERR_INVALID, "sql: internal item kind");
    }
  }
  for (size_t i = 0; i < out.size(); i++) {
    for (size_t j = 0; j < i; j++) if (out[i].name == out[j].name) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: two result columns are named " + out[i].name);
    out[i].yt = sql_yt_of(out[i].ch);
    if (out[i].yt == TFGPU_T_INVALID) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: column " + out[i].name + " has no device type");
    const int k = find(out[i].name);
    out[i].key = k >= 0 && (in[(size_t)k].flags & TFGPU_COL_KEY) != 0;  // keys[col.Name] (clickhouse_local.go:393-413)
  }
  return out;
}

}  // namespace tf
