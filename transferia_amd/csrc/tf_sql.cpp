// tf_sql.cpp — the `sql` transformer's query, for the subset that runs on the device.
//
// The reference hands every batch to an external `clickhouse-local --query Q` process (pkg/transformer/registry/
// clickhouse/clickhouse_local.go:97-294): whatever ClickHouse's SQL engine accepts is legal there.  SURVEY.md §7 and
// BASELINE.json configs[2] ("SQL-transformer predicate+cast") ask for the predicate + cast part of that on the device.
// This file parses row-wise SELECTs over integer and text expressions and refuses everything else by name (TFGPU_ERR_UNSUPPORTED), so a query the
// device cannot evaluate the way ClickHouse does never runs here:
//
//   SELECT item [, item …] FROM table [WHERE cond] [;]
//   item  :=  *  |  expr [AS alias]                     (an expression that is not a plain column needs an alias)
//   expr  :=  or-expr;  OR < AND < NOT < comparison (= == != <> < <= > >=, [NOT] IN (literal, …)) < + - < * < unary -
//   atoms :=  column | integer | 'string' | ( expr ) | toInt8 … toInt64 ( expr ) | toUInt8 … toUInt64 ( expr )
//          |  toString ( expr ) | toDateTime ( column ) | length ( text ) | lower / upper ( text ) | cityHash64 ( text ) | not ( expr )
//             (toDateTime of a 32 / 64-bit integer column holding epoch seconds that fit DateTime, 0 … 2^32 - 1: ClickHouse
//              saturates outside, the device does not — values are the caller's; a / b, a % b, LIKE, BETWEEN, functions of
//              several arguments, floats and dates inside expressions are refused by name)
//   cond  :=  expr of an integer type; rows whose value is not zero stay
//
// Typing follows ClickHouse: an integer literal has the narrowest type that holds it (UInt8 … UInt64, Int8 … Int64 when
// negative), a + b and a * b of integers are signed if either side is, a - b always, with twice the wider operand's bits (at
// most 64); -a is signed (an unsigned operand takes the next size); toIntN wraps (two's complement truncation); comparisons,
// AND / OR / NOT, IN yield UInt8, integers of different signedness compare by value; length() counts bytes (UInt64); lower /
// upper touch ASCII letters; cityHash64(String) is CityHash64 of CityHash v1.0.2 over the bytes; text compares bytewise, against
// a literal.  Constants fold while the query is read.  The forms that had kernels of their own before the general tree (a column,
// a constant, casts and ± literal over one column, toString / toDateTime of a column, a WHERE that is an OR of ANDs of column
// against literal) are recognised in the tree and keep them; everything else runs the expression program (sql_expr_kernel,
// tf_transform.hip).  Result columns map back to YT types through typesystem.go's Source rules
// (pkg/providers/clickhouse/typesystem.go:15-33: String → `string`, DateTime → `datetime`, …), and a result column is a
// primary key when it carries the NAME of an input key column (clickhouse_local.go:393-421).
//
// PARITY: the reference's tests for this transformer need the clickhouse binary and assert schemas, row counts, kinds and one
// value (clickhouse_local_test.go:26-124: cityHash64(val) is Suitable, toInt8(id+1) == int8(2)); tests/test_sql.py replays those.  Everything else
// here follows ClickHouse's documented behaviour and is labelled "parity unpinned" in DESIGN.md §4.
#include <cctype>
#include <cstdlib>
#include <functional>

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
    if (std::string("(),*+-=<>;/%").find(c) != std::string::npos) { t.k = Tok::Punct; t.s = std::string(1, c); i++; return t; }
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
  Lexer lx; Tok cur; std::vector<SqlNode> &nodes;
  Parser(const std::string &q, std::vector<SqlNode> &n) : lx(q), nodes(n) { cur = lx.next(); }
  void adv() { cur = lx.next(); }
  bool punct(const char *p) const { return cur.k == Tok::Punct && cur.s == p; }
  bool kw(const char *w) const { return cur.k == Tok::Ident && ieq(cur.s, w); }
  void expect_punct(const char *p) { if (!punct(p)) Lexer::bad(std::string("expected '") + p + "'" + (cur.k == Tok::End ? " at the end of the query" : " near '" + cur.s + "'")); adv(); }
  int mk(int op, std::vector<int> kids = {}, int ty = SQL_PENDING, int64_t iv = 0, std::string str = std::string()) {
    SqlNode n; n.op = op; n.ty = ty; n.ival = iv; n.s = std::move(str); n.kids = std::move(kids);
    nodes.push_back(std::move(n));
    return (int)nodes.size() - 1;
  }
  bool is_int_lit(int i) const { return nodes[(size_t)i].op == SN_INT; }

  // ClickHouse's type of an integer literal
  static int lit_type(int64_t v) {
    if (v >= 0) return v <= 0xFF ? SQL_U8 : v <= 0xFFFF ? SQL_U16 : v <= 0xFFFFFFFFll ? SQL_U32 : SQL_U64;
    return v >= -128 ? SQL_I8 : v >= -32768 ? SQL_I16 : v >= -2147483648ll ? SQL_I32 : SQL_I64;
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
  static int int_type(bool sg, int bits) { return sg ? (bits == 8 ? SQL_I8 : bits == 16 ? SQL_I16 : bits == 32 ? SQL_I32 : SQL_I64) : (bits == 8 ? SQL_U8 : bits == 16 ? SQL_U16 : bits == 32 ? SQL_U32 : SQL_U64); }
  // the type of a + b, a * b / a - b over integers (NumberTraits::ResultOfAdditionMultiplication / ResultOfSubtraction)
  static int add_type(int a, int b, bool minus) {
    return int_type(minus || signed_of(a) || signed_of(b), std::min(64, 2 * std::max(bits_of(a), bits_of(b))));
  }
  // -a (NumberTraits::ResultOfNegate): signed; a signed operand keeps its size, an unsigned one takes the next
  static int neg_type(int a) { return int_type(true, signed_of(a) ? bits_of(a) : std::min(64, 2 * bits_of(a))); }

  // constants fold while the query is read, with ClickHouse's result types ((1+1) is a UInt16 2)
  int fold(int i) {
    SqlNode &n = nodes[(size_t)i];
    for (int k : n.kids) if (!is_int_lit(k)) return i;
    auto K = [&](int j) -> const SqlNode & { return nodes[(size_t)n.kids[(size_t)j]]; };
    int ty; int64_t v;
    switch (n.op) {
      case SN_ADD: ty = add_type(K(0).ty, K(1).ty, false); v = wrap((int64_t)((uint64_t)K(0).ival + (uint64_t)K(1).ival), ty); break;
      case SN_SUB: ty = add_type(K(0).ty, K(1).ty, true); v = wrap((int64_t)((uint64_t)K(0).ival - (uint64_t)K(1).ival), ty); break;
      case SN_MUL: ty = add_type(K(0).ty, K(1).ty, false); v = wrap((int64_t)((uint64_t)K(0).ival * (uint64_t)K(1).ival), ty); break;
      case SN_NEG: ty = neg_type(K(0).ty); v = wrap((int64_t)(0 - (uint64_t)K(0).ival), ty); break;
      case SN_CAST: ty = n.ty; v = wrap(K(0).ival, ty); break;
      default: return i;
    }
    n.op = SN_INT; n.ty = ty; n.ival = v; n.kids.clear();
    return i;
  }

  int primary() {
    if (punct("(")) { adv(); const int e = bexpr(); expect_punct(")"); return e; }
    if (cur.k == Tok::Int) { const int e = mk(SN_INT, {}, lit_type(cur.v), cur.v); adv(); return e; }
    if (cur.k == Tok::Str) { const int e = mk(SN_STR, {}, SQL_STRING, 0, cur.s); adv(); return e; }
    if (cur.k != Tok::Ident) Lexer::bad("expected an expression" + (cur.k == Tok::End ? std::string(" at the end of the query") : " near '" + cur.s + "'"));
    const std::string id = cur.s;
    adv();
    if (!punct("(")) return mk(SN_COL, {}, SQL_PENDING, 0, id);
    adv();
    static const struct { const char *fn; int ty; } casts[] = {{"toInt8", SQL_I8}, {"toInt16", SQL_I16}, {"toInt32", SQL_I32}, {"toInt64", SQL_I64},
                                                                {"toUInt8", SQL_U8}, {"toUInt16", SQL_U16}, {"toUInt32", SQL_U32}, {"toUInt64", SQL_U64}};
    for (auto &c : casts)
      if (id == c.fn) {  // ClickHouse function names are case-sensitive
        const int a = bexpr();
        expect_punct(")");
        return fold(mk(SN_CAST, {a}, c.ty));
      }
    static const struct { const char *fn; int op; } fns[] = {{"toString", SN_TOSTR}, {"toDateTime", SN_TODT}, {"length", SN_LEN}, {"cityHash64", SN_CITY64}, {"lower", SN_LOWER}, {"upper", SN_UPPER},
                                                             {"lcase", SN_LOWER}, {"ucase", SN_UPPER}, {"not", SN_NOT}};
    for (auto &f : fns)
      if (id == f.fn) {
        const int a = bexpr();
        if (punct(",")) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: " + id + "() takes one argument in the device subset");
        expect_punct(")");
        return mk(f.op, {a});
      }
    throw Error(TFGPU_ERR_UNSUPPORTED, "sql: function " + id + "() is outside the device subset (integer arithmetic, casts, comparisons, length / lower / upper / cityHash64 of text): keep this transformer on the host");
  }
  int unary() {
    if (punct("-")) {
      adv();
      if (cur.k == Tok::Int) { const int64_t v = -cur.v; adv(); return mk(SN_INT, {}, lit_type(v), v); }  // a negative literal, as ClickHouse's parser reads it
      return fold(mk(SN_NEG, {unary()}));
    }
    if (punct("+")) { adv(); return unary(); }
    return primary();
  }
  int mul() {
    int a = unary();
    for (;;) {
      if (punct("*")) { adv(); const int b = unary(); a = fold(mk(SN_MUL, {a, b})); continue; }
      if (punct("/")) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: a / b is a Float64 in ClickHouse: outside the device subset");
      if (punct("%")) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: the modulo operator is outside the device subset");
      return a;
    }
  }
  int addsub() {
    int a = mul();
    while (punct("+") || punct("-")) {
      const bool minus = punct("-");
      adv();
      const int b = mul();
      a = fold(mk(minus ? SN_SUB : SN_ADD, {a, b}));
    }
    return a;
  }
  int literal() {  // an element of an IN list
    bool neg = false;
    if (punct("-")) { neg = true; adv(); }
    if (cur.k == Tok::End) Lexer::bad("expected a literal at the end of the query");
    int e;
    if (cur.k == Tok::Int) { const int64_t v = neg ? -cur.v : cur.v; e = mk(SN_INT, {}, lit_type(v), v); }
    else if (cur.k == Tok::Str && !neg) e = mk(SN_STR, {}, SQL_STRING, 0, cur.s);
    else throw Error(TFGPU_ERR_UNSUPPORTED, "sql: an IN list holds integer or string literals in the device subset");
    adv();
    return e;
  }
  int cmp() {
    const int a = addsub();
    bool negate = false;
    if (kw("not")) {
      negate = true; adv();
      if (!kw("in")) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: NOT behind an expression is taken in front of IN only (LIKE / BETWEEN are outside the device subset)");
    }
    if (kw("in")) {
      adv(); expect_punct("(");
      std::vector<int> kids{a, literal()};
      while (punct(",")) { adv(); kids.push_back(literal()); }
      expect_punct(")");
      for (size_t i = 2; i < kids.size(); i++)
        if (nodes[(size_t)kids[i]].op != nodes[(size_t)kids[1]].op) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: a list mixes strings and integers");
      return mk(negate ? SN_NOTIN : SN_IN, kids);
    }
    if (cur.k == Tok::Punct) {
      const std::string op = cur.s;
      const int k = (op == "=" || op == "==") ? SN_EQ : (op == "!=" || op == "<>") ? SN_NE : op == "<" ? SN_LT : op == "<=" ? SN_LE : op == ">" ? SN_GT : op == ">=" ? SN_GE : -1;
      if (k >= 0) {
        adv();
        if (cur.k == Tok::End) Lexer::bad("expected an expression at the end of the query");
        const int b = addsub();
        return mk(k, {a, b});
      }
    }
    return a;
  }
  int bnot() {
    if (kw("not")) { adv(); return mk(SN_NOT, {bnot()}); }
    return cmp();
  }
  int band() {
    int a = bnot();
    while (kw("and")) { adv(); const int b = bnot(); a = mk(SN_AND, {a, b}); }
    return a;
  }
  int bexpr() {
    int a = band();
    while (kw("or")) { adv(); const int b = band(); a = mk(SN_OR, {a, b}); }
    return a;
  }
};

// a select-list tree in one of the forms that have a kernel of their own (a column, a constant, casts and ± literal over ONE
// column, toString / toDateTime of a column); everything else is SQL_EXPR, evaluated by the expression program
SqlItem lower_item(const std::vector<SqlNode> &nodes, int root) {
  SqlItem it;
  const SqlNode &n = nodes[(size_t)root];
  if (n.op == SN_COL) { it.kind = SQL_COLUMN; it.src = n.s; it.name = n.s; return it; }
  if (n.op == SN_INT) { it.kind = SQL_CONST_INT; it.ival = n.ival; it.cast = n.ty; return it; }
  if (n.op == SN_STR) { it.kind = SQL_CONST_STR; it.sval = n.s; it.cast = SQL_STRING; return it; }
  if ((n.op == SN_TOSTR || n.op == SN_TODT) && nodes[(size_t)n.kids[0]].op == SN_COL) {
    it.kind = n.op == SN_TOSTR ? SQL_TO_STRING : SQL_TO_DATETIME; it.src = nodes[(size_t)n.kids[0]].s; it.cast = n.op == SN_TOSTR ? SQL_STRING : SQL_DATETIME;
    return it;
  }
  std::vector<SqlStep> steps;  // outermost first
  int i = root;
  for (;;) {
    const SqlNode &m = nodes[(size_t)i];
    if (m.op == SN_COL) {
      it.kind = SQL_INT_EXPR; it.src = m.s;
      it.steps.assign(steps.rbegin(), steps.rend());
      return it;
    }
    if (m.op == SN_CAST) { steps.push_back(SqlStep{true, m.ty, 0, false}); i = m.kids[0]; continue; }
    if ((m.op == SN_ADD || m.op == SN_SUB) && nodes[(size_t)m.kids[1]].op == SN_INT) {
      const SqlNode &l = nodes[(size_t)m.kids[1]];
      steps.push_back(SqlStep{false, l.ty, m.op == SN_SUB ? -l.ival : l.ival, m.op == SN_SUB});
      i = m.kids[0]; continue;
    }
    if (m.op == SN_ADD && nodes[(size_t)m.kids[0]].op == SN_INT) {  // integer + column
      const SqlNode &l = nodes[(size_t)m.kids[0]];
      steps.push_back(SqlStep{false, l.ty, l.ival, false});
      i = m.kids[1]; continue;
    }
    break;
  }
  it.kind = SQL_EXPR; it.root = root;
  return it;
}

// WHERE as filter_rows' OR of ANDs of column-against-literal terms, when it has that shape (parentheses are gone in the tree)
bool lower_where(const std::vector<SqlNode> &nodes, int root, std::vector<FExpr> &out) {
  std::vector<int> ors, stack{root};
  while (!stack.empty()) { const int i = stack.back(); stack.pop_back(); if (nodes[(size_t)i].op == SN_OR) { stack.push_back(nodes[(size_t)i].kids[1]); stack.push_back(nodes[(size_t)i].kids[0]); } else ors.push_back(i); }
  std::vector<FExpr> res;
  for (int d : ors) {
    std::vector<int> ands; stack = {d};
    while (!stack.empty()) { const int i = stack.back(); stack.pop_back(); if (nodes[(size_t)i].op == SN_AND) { stack.push_back(nodes[(size_t)i].kids[1]); stack.push_back(nodes[(size_t)i].kids[0]); } else ands.push_back(i); }
    FExpr conj;
    for (int ti : ands) {
      const SqlNode &t = nodes[(size_t)ti];
      FTerm ft;
      auto lit = [&](const SqlNode &l) -> bool {
        if (l.op == SN_INT) { if (!ft.strs.empty()) return false; ft.vtype = FV_INT; ft.ints.push_back(l.ival); return true; }
        if (l.op == SN_STR) { if (!ft.ints.empty()) return false; ft.vtype = FV_STRING; ft.strs.push_back(l.s); return true; }
        return false;
      };
      if (t.op >= SN_EQ && t.op <= SN_GE) {
        if (nodes[(size_t)t.kids[0]].op != SN_COL || !lit(nodes[(size_t)t.kids[1]])) return false;
        ft.attr = nodes[(size_t)t.kids[0]].s;
        ft.op = t.op == SN_EQ ? F_EQ : t.op == SN_NE ? F_NE : t.op == SN_LT ? F_LT : t.op == SN_LE ? F_LE : t.op == SN_GT ? F_GT : F_GE;
      } else if (t.op == SN_IN || t.op == SN_NOTIN) {
        if (nodes[(size_t)t.kids[0]].op != SN_COL) return false;
        ft.attr = nodes[(size_t)t.kids[0]].s; ft.op = t.op == SN_IN ? F_IN : F_NOTIN; ft.is_list = true;
        for (size_t k = 1; k < t.kids.size(); k++) if (!lit(nodes[(size_t)t.kids[k]])) return false;
      } else return false;
      conj.terms.push_back(std::move(ft));
    }
    res.push_back(std::move(conj));
  }
  out = std::move(res);
  return true;
}

}  // namespace

// the query → plan fields (items; WHERE as OR of ANDs, or as a tree)
void sql_parse(const std::string &query, tfgpu_plan &p) {
  Parser ps(query, p.sql_nodes);
  if (!ps.kw("select")) throw Error(TFGPU_ERR_CONFIG, "sql: the query must start with SELECT" + (ps.cur.k == Tok::End ? std::string() : " (got '" + ps.cur.s + "')"));
  ps.adv();
  for (;;) {
    SqlItem it;
    if (ps.punct("*")) { ps.adv(); it.kind = SQL_STAR; }
    else {
      it = lower_item(p.sql_nodes, ps.bexpr());
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
    if (ps.cur.k == Tok::End) Lexer::bad("expected a condition at the end of the query");
    const int root = ps.bexpr();
    p.sql_where_tree = root;
    if (!lower_where(p.sql_nodes, root, p.exprs)) p.sql_where_root = root;
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

// filter_rows' predicate program compares integers the way Go's matchValue does — a uint64 above MaxInt64 is an error row there, a
// value like any other in ClickHouse — so a WHERE in filter_rows' shape that names a UInt64 column runs as the expression program too
bool sql_where_as_tree(const tfgpu_plan &p, const std::vector<SchemaCol> &in) {
  if (p.sql_where_root >= 0) return true;
  for (const FExpr &e : p.exprs) for (const FTerm &t : e.terms)
    for (auto &c : in) if (c.name == t.attr && c.dtype == TFGPU_T_UINT64) return p.sql_where_tree >= 0;
  return false;
}

// ClickHouse's type of every node of the expression trees over a concrete schema.  TFGPU_ERR_CONFIG: what ClickHouse refuses
// (unknown column, text against a number); TFGPU_ERR_UNSUPPORTED: what the device subset does not take.
std::vector<int> sql_node_types(const tfgpu_plan &p, const std::vector<SchemaCol> &in) {
  std::vector<int> ty(p.sql_nodes.size(), SQL_PENDING);
  auto is_int = [](int t) { return t >= SQL_I8 && t <= SQL_U64; };
  static const char *const OPN[] = {"column", "literal", "literal", "+", "-", "*", "negate", "cast", "length", "cityHash64", "lower", "upper", "toString", "toDateTime",
                                    "=", "!=", "<", "<=", ">", ">=", "AND", "OR", "NOT", "IN", "NOT IN"};
  std::function<int(int)> go = [&](int i) -> int {
    const SqlNode &n = p.sql_nodes[(size_t)i];
    auto kid = [&](int k) { return go(n.kids[(size_t)k]); };
    auto need_int = [&](int t) { if (!is_int(t)) throw Error(t == SQL_STRING ? TFGPU_ERR_CONFIG : TFGPU_ERR_UNSUPPORTED, std::string("sql: ") + OPN[n.op] + " of a value that is not an integer (ClickHouse: illegal type of argument)"); };
    int t = SQL_PENDING;
    switch (n.op) {
      case SN_COL: {
        int ci = -1;
        for (size_t c = 0; c < in.size(); c++) if (in[c].name == n.s) { ci = (int)c; break; }
        if (ci < 0) throw Error(TFGPU_ERR_CONFIG, "sql: unknown column " + n.s + " (ClickHouse: Missing columns)");
        t = sql_ch_of(in[(size_t)ci].dtype);
        if (!is_int(t) && t != SQL_STRING) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: the " + type_name(in[(size_t)ci].dtype) + " column " + n.s + " inside an expression is outside the device subset (integers and text)");
        break;
      }
      case SN_INT: t = n.ty; break;
      case SN_STR: t = SQL_STRING; break;
      case SN_ADD: case SN_SUB: case SN_MUL: { const int a = kid(0), b = kid(1); need_int(a); need_int(b); t = Parser::add_type(a, b, n.op == SN_SUB); break; }
      case SN_NEG: { const int a = kid(0); need_int(a); t = Parser::neg_type(a); break; }
      case SN_CAST: { const int a = kid(0); if (a == SQL_STRING) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: toIntN() of text (ClickHouse parses the text) is outside the device subset"); need_int(a); t = n.ty; break; }
      case SN_LEN: case SN_CITY64: { const int a = kid(0); if (a != SQL_STRING) throw Error(TFGPU_ERR_UNSUPPORTED, std::string("sql: ") + OPN[n.op] + "() takes text in the device subset"); t = SQL_U64; break; }
      case SN_LOWER: case SN_UPPER: { const int a = kid(0); if (a != SQL_STRING) throw Error(TFGPU_ERR_CONFIG, std::string("sql: ") + OPN[n.op] + "() of a value that is not text (ClickHouse: illegal type of argument)"); t = SQL_STRING; break; }
      case SN_TOSTR: { const int a = kid(0); if (!is_int(a) && a != SQL_STRING) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: toString() of this type is outside the device subset"); t = SQL_STRING; break; }
      case SN_TODT: throw Error(TFGPU_ERR_UNSUPPORTED, "sql: toDateTime() takes a plain column in the device subset");
      case SN_EQ: case SN_NE: case SN_LT: case SN_LE: case SN_GT: case SN_GE: {
        const int a = kid(0), b = kid(1);
        if ((a == SQL_STRING) != (b == SQL_STRING)) throw Error(TFGPU_ERR_CONFIG, "sql: text compared with a number (ClickHouse: there is no supertype for String and an integer)");
        if (a == SQL_STRING) {
          const bool la = p.sql_nodes[(size_t)n.kids[0]].op == SN_STR, lb = p.sql_nodes[(size_t)n.kids[1]].op == SN_STR;
          if (la == lb) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: a text comparison has a string literal on exactly one side in the device subset");
        } else { need_int(a); need_int(b); }
        t = SQL_U8;
        break;
      }
      case SN_AND: case SN_OR: { const int a = kid(0), b = kid(1); need_int(a); need_int(b); t = SQL_U8; break; }
      case SN_NOT: { const int a = kid(0); need_int(a); t = SQL_U8; break; }
      case SN_IN: case SN_NOTIN: {
        const int a = kid(0);
        const bool lit_str = p.sql_nodes[(size_t)n.kids[1]].op == SN_STR;
        if ((a == SQL_STRING) != lit_str) throw Error(TFGPU_ERR_CONFIG, "sql: IN list of another type than its left side (ClickHouse: type mismatch in IN)");
        if (a != SQL_STRING) need_int(a);
        t = SQL_U8;
        break;
      }
      default: throw Error(TFGPU_ERR_INVALID, "sql: internal node");
    }
    ty[(size_t)i] = t;
    return t;
  };
  for (const SqlItem &it : p.sql_items) if (it.kind == SQL_EXPR) go(it.root);
  if (sql_where_as_tree(p, in)) { const int t = go(p.sql_where_tree); if (!is_int(t)) throw Error(TFGPU_ERR_CONFIG, "sql: the WHERE condition is text (ClickHouse: illegal type for filter)"); }
  return ty;
}

// The select list over a concrete input schema: every item with its source column, ClickHouse type and YT type.
// Throws TFGPU_ERR_CONFIG for what ClickHouse would refuse (unknown column) and TFGPU_ERR_UNSUPPORTED for what the device
// subset does not take.
std::vector<SqlOut> sql_resolve(const tfgpu_plan &p, const std::vector<SchemaCol> &in) {
  std::vector<SqlOut> out;
  auto find = [&](const std::string &n) -> int { for (size_t i = 0; i < in.size(); i++) if (in[i].name == n) return (int)i; return -1; };
  auto need = [&](const std::string &n) { const int i = find(n); if (i < 0) throw Error(TFGPU_ERR_CONFIG, "sql: unknown column " + n + " (ClickHouse: Missing columns)"); return i; };
  auto is_int = [](int ty) { return ty >= SQL_I8 && ty <= SQL_U64; };
  std::vector<int> types;
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
        if (ty == SQL_STRING && !it.steps.empty() && !it.steps[0].is_cast) throw Error(TFGPU_ERR_CONFIG, "sql: arithmetic on the text column " + it.src + " (ClickHouse: illegal types of arguments)");
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
      case SQL_EXPR: {
        if (types.empty()) types = sql_node_types(p, in);
        SqlOut o; o.kind = SQL_EXPR; o.name = it.name; o.root = it.root; o.ch = types[(size_t)it.root];
        out.push_back(o);
        break;
      }
      default: throw Error(TFGPU_ERR_INVALID, "sql: internal item kind");
    }
  }
  if (sql_where_as_tree(p, in) && types.empty()) types = sql_node_types(p, in);  // a WHERE ClickHouse would refuse is refused with the schema
  for (const FExpr &e : p.exprs) for (const FTerm &t : e.terms) {  // the WHERE in filter_rows' form: the same refusals
    const int ct = sql_ch_of(in[(size_t)need(t.attr)].dtype);
    if ((ct == SQL_STRING && t.vtype == FV_INT) || (is_int(ct) && t.vtype == FV_STRING))
      throw Error(TFGPU_ERR_CONFIG, "sql: column " + t.attr + (t.is_list ? ": IN list of another type than the column (ClickHouse: type mismatch in IN)" : " compared with a literal of another kind (ClickHouse: there is no supertype for String and an integer)"));
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
