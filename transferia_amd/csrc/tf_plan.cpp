// tf_plan.cpp — transformer factories: JSON config → immutable plan.
// Mirrors the reference constructors (citations inline).  Host-only code.
#include "tf_plan.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>

namespace tf {

[[noreturn]] static void cfg_error(const std::string &m) { throw Error(TFGPU_ERR_CONFIG, m); }

// ===================== JSON =====================
namespace {
struct JP {
  const char *p, *e;
  void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
  [[noreturn]] void bad(const char *m) { cfg_error(std::string("config json: ") + m); }
  static void put_utf8(std::string &o, unsigned cp) {
    if (cp < 0x80) o += (char)cp;
    else if (cp < 0x800) { o += (char)(0xC0 | cp >> 6); o += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { o += (char)(0xE0 | cp >> 12); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    else { o += (char)(0xF0 | cp >> 18); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
  }
  unsigned hex4() {
    if (e - p < 4) bad("short \\u escape");
    unsigned v = 0;
    for (int i = 0; i < 4; i++) { char c = *p++; v = v * 16 + (unsigned)(std::isdigit((unsigned char)c) ? c - '0' : (std::tolower(c) - 'a' + 10)); }
    return v;
  }
  std::string str() {
    if (p >= e || *p != '"') bad("expected string");
    p++;
    std::string o;
    while (p < e && *p != '"') {
      if (*p == '\\') {
        p++;
        if (p >= e) break;
        char c = *p++;
        switch (c) {
          case 'n': o += '\n'; break; case 't': o += '\t'; break; case 'r': o += '\r'; break;
          case 'b': o += '\b'; break; case 'f': o += '\f'; break;
          case 'u': {
            unsigned cp = hex4();
            if (cp >= 0xD800 && cp < 0xDC00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') { p += 2; unsigned lo = hex4(); cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); }
            put_utf8(o, cp); break;
          }
          default: o += c;
        }
      } else o += *p++;
    }
    if (p >= e) bad("unterminated string");
    p++;
    return o;
  }
  int depth = 0;  // (encoding/json stops at 10 000 levels; nothing on this path nests past a few dozen — a text that does is refused by name
                  //  rather than followed down the stack)
  Json val() {
    struct Depth { int &d; JP &jp; Depth(int &x, JP &j) : d(x), jp(j) { if (++d > 512) jp.bad("nesting deeper than 512 levels"); } ~Depth() { d--; } } guard(depth, *this);
    ws();
    if (p >= e) bad("unexpected end");
    Json j;
    char c = *p;
    if (c == '{') {
      j.type = Json::Obj; p++; ws();
      if (p < e && *p == '}') { p++; return j; }
      for (;;) {
        ws(); std::string k = str(); ws();
        if (p >= e || *p != ':') bad("expected ':'");
        p++;
        j.obj.emplace_back(k, val());
        ws();
        if (p < e && *p == ',') { p++; continue; }
        if (p < e && *p == '}') { p++; return j; }
        bad("expected ',' or '}'");
      }
    }
    if (c == '[') {
      j.type = Json::Arr; p++; ws();
      if (p < e && *p == ']') { p++; return j; }
      for (;;) {
        j.arr.push_back(val()); ws();
        if (p < e && *p == ',') { p++; continue; }
        if (p < e && *p == ']') { p++; return j; }
        bad("expected ',' or ']'");
      }
    }
    if (c == '"') { j.type = Json::Str; j.str = str(); return j; }
    if (e - p >= 4 && !std::strncmp(p, "true", 4)) { j.type = Json::Bool; j.b = true; p += 4; return j; }
    if (e - p >= 5 && !std::strncmp(p, "false", 5)) { j.type = Json::Bool; j.b = false; p += 5; return j; }
    if (e - p >= 4 && !std::strncmp(p, "null", 4)) { p += 4; return j; }
    if (c == '-' || std::isdigit((unsigned char)c)) {
      const char *b = p;
      while (p < e && (std::isdigit((unsigned char)*p) || std::strchr("+-.eE", *p))) p++;
      j.type = Json::Num; j.str.assign(b, p); j.num = std::strtod(j.str.c_str(), nullptr);
      return j;
    }
    bad("unexpected character");
  }
};
bool ieq(const std::string &a, const std::string &b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); i++) if (std::tolower((unsigned char)a[i]) != std::tolower((unsigned char)b[i])) return false;
  return true;
}
}  // namespace

Json Json::parse(const std::string &text) {
  JP s{text.data(), text.data() + text.size()};
  Json j = s.val();
  s.ws();
  if (s.p != s.e) s.bad("trailing data");
  return j;
}
const Json *Json::get(const std::string &key) const {
  if (type != Obj) return nullptr;
  for (auto &kv : obj) if (kv.first == key) return &kv.second;
  for (auto &kv : obj) if (ieq(kv.first, key)) return &kv.second;
  return nullptr;
}
std::string Json::s(const std::string &key, const std::string &dflt) const { const Json *j = get(key); return (j && j->type == Str) ? j->str : dflt; }
bool Json::flag(const std::string &key, bool dflt) const { const Json *j = get(key); return (j && j->type == Bool) ? j->b : dflt; }
std::vector<std::string> Json::strings(const std::string &key) const {
  std::vector<std::string> out;
  const Json *j = get(key);
  if (j && j->type == Arr) for (auto &x : j->arr) out.push_back(x.type == Str ? x.str : "");
  return out;
}

// ===================== name filters =====================
void NameFilter::init(const std::vector<std::string> &inc, const std::vector<std::string> &exc) {
  include_src = inc; exclude_src = exc;
  // Go regexp is RE2 syntax; ECMAScript covers the constructs transformer
  // configs use (anchors, classes, alternation, quantifiers, \d \w \s).
  auto compile = [](const std::string &src, const char *what) {
    try { return std::regex(src, std::regex::ECMAScript | std::regex::optimize); }
    catch (const std::regex_error &) { cfg_error(std::string("unable to compile ") + what + " regexp: " + src); }
  };
  for (auto &s : inc) include.push_back(compile(s, "include"));
  for (auto &s : exc) exclude.push_back(compile(s, "exclude"));
}
bool NameFilter::match(const std::string &v) const {  // filter.go:27-44
  for (auto &re : exclude) if (std::regex_search(v, re)) return false;
  if (include_src.empty()) return true;
  for (auto &re : include) if (std::regex_search(v, re)) return true;
  return false;
}
static std::string dq(const std::string &s) {
  std::string o = "\"";
  for (char c : s) { if (c == '"') o += '"'; o += c; }
  return o + "\"";
}
bool NameFilter::match_table(const std::string &ns, const std::string &name) const {
  if (empty()) return true;
  std::string full = ns.empty() ? name : ns + "." + name;
  std::string fqtn;
  if (!ns.empty()) fqtn = dq(ns) + ".";
  fqtn += (name == "*") ? name : dq(name);
  return match(full) || match(fqtn);
}

bool is_system_table(const std::string &name) {
  // names registered through abstract.RegisterSystemTables by provider init()s
  static const char *sys[] = {"__wal", "__table_transfer_progress", "__tm_gtid_keeper", "__tm_keeper", "__consumer_keeper",
                              "__data_transfer_lsn", "__data_transfer_signal_table", "__data_transfer", "__dt_cluster_time"};
  for (auto s : sys) if (name == s) return true;
  return false;
}

// ===================== filter grammar =====================
// library/go/yandex/cloud/filter/grammar/grammar.go:255-313 (lexer alternatives in
// priority order) and filters.go:240-311 (validateTerm / Parse).
namespace {
enum Tk { T_OP, T_STR, T_DT, T_ID, T_FLOAT, T_INT, T_PUNCT, T_WS, T_EOF };
struct Tok { Tk t; std::string s; };
bool dg(char c) { return c >= '0' && c <= '9'; }
bool al(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

std::vector<Tok> lex(const std::string &in) {
  std::vector<Tok> out;
  size_t i = 0, n = in.size();
  const char *s = in.data();
  while (i < n) {
    size_t r = n - i;
    const char *p = s + i;
    size_t len = 0; Tk t = T_EOF;
    if (r >= 2 && ((p[0] == '!' && p[1] == '=') || (p[0] == '<' && p[1] == '=') || (p[0] == '>' && p[1] == '=') || (p[0] == '!' && p[1] == '~'))) { t = T_OP; len = 2; }
    else if (p[0] == '=' || p[0] == '<' || p[0] == '>' || p[0] == '~') { t = T_OP; len = 1; }
    else if (p[0] == '\'' || p[0] == '"') {
      char q = p[0]; size_t k = 1; long last_pair = -1;
      while (k < r) {
        if (p[k] == '\\' && k + 1 < r && p[k + 1] == q) { last_pair = (long)k + 1; k += 2; continue; }
        if (p[k] == q) { len = k + 1; break; }
        k++;
      }
      if (!len && last_pair >= 0) len = (size_t)last_pair + 1;  // regexp backtracking
      if (!len) cfg_error("filter: invalid token at position " + std::to_string(i));
      t = T_STR;
    } else if (r >= 10 && dg(p[0]) && dg(p[1]) && dg(p[2]) && dg(p[3]) && p[4] == '-' && dg(p[5]) && dg(p[6]) && p[7] == '-' && dg(p[8]) && dg(p[9])) {
      size_t k = 10;
      if (r >= k + 6 && p[k] == 'T' && dg(p[k + 1]) && dg(p[k + 2]) && p[k + 3] == ':' && dg(p[k + 4]) && dg(p[k + 5])) {
        k += 6;
        if (r >= k + 3 && p[k] == ':' && dg(p[k + 1]) && dg(p[k + 2])) {
          k += 3;
          if (r >= k + 2 && p[k] == '.' && dg(p[k + 1])) { k += 2; while (k < r && dg(p[k])) k++; }
        }
        if (k < r && p[k] == 'Z') k++;
        else if (k + 1 < r && (p[k] == '+' || p[k] == '-') && dg(p[k + 1])) {
          k += 2; while (k < r && dg(p[k])) k++;
          if (k + 1 < r && p[k] == ':' && dg(p[k + 1])) { k += 2; while (k < r && dg(p[k])) k++; }
        }
      }
      t = T_DT; len = k;
    } else if (al(p[0])) { size_t k = 1; while (k < r && (al(p[k]) || dg(p[k]) || p[k] == '_' || p[k] == '.')) k++; t = T_ID; len = k; }
    else {
      size_t k = (p[0] == '-' || p[0] == '+') ? 1 : 0;
      if (k < r && dg(p[k])) {
        size_t j = k; while (j < r && dg(p[j])) j++;
        if (j + 1 < r && p[j] == '.' && dg(p[j + 1])) { j += 2; while (j < r && dg(p[j])) j++; t = T_FLOAT; }
        else t = T_INT;
        len = j;
      } else if (p[0] == '(' || p[0] == ')' || p[0] == ',') { t = T_PUNCT; len = 1; }
      else if (std::isspace((unsigned char)p[0])) { size_t j = 1; while (j < r && std::isspace((unsigned char)p[j])) j++; t = T_WS; len = j; }
      else cfg_error("filter: invalid token at position " + std::to_string(i));
    }
    out.push_back({t, in.substr(i, len)});
    i += len;
  }
  out.push_back({T_EOF, ""});
  return out;
}

bool kw(const Tok &t, const char *k) { return t.t == T_ID && ieq(t.s, k); }

// participle.Unquote → strconv.UnquoteChar loop
std::string go_unquote(const std::string &tok) {
  char q = tok[0];
  std::string s = tok.substr(1, tok.size() - 2), o;
  for (size_t i = 0; i < s.size();) {
    unsigned char c = (unsigned char)s[i];
    if (c == (unsigned char)q) cfg_error("filter: invalid string literal");
    if (c != '\\') { o += (char)c; i++; continue; }
    if (i + 1 >= s.size()) cfg_error("filter: invalid string literal");
    char e = s[i + 1]; i += 2;
    auto hexv = [&](int nd) {
      if (i + (size_t)nd > s.size()) cfg_error("filter: invalid escape");
      unsigned v = 0;
      for (int k = 0; k < nd; k++) { char h = s[i + (size_t)k]; int d = dg(h) ? h - '0' : (h >= 'a' && h <= 'f') ? h - 'a' + 10 : (h >= 'A' && h <= 'F') ? h - 'A' + 10 : -1; if (d < 0) cfg_error("filter: invalid escape"); v = v * 16 + (unsigned)d; }
      i += (size_t)nd; return v;
    };
    switch (e) {
      case 'a': o += '\a'; break; case 'b': o += '\b'; break; case 'f': o += '\f'; break; case 'n': o += '\n'; break;
      case 'r': o += '\r'; break; case 't': o += '\t'; break; case 'v': o += '\v'; break; case '\\': o += '\\'; break;
      case '\'': case '"': if (e != q) cfg_error("filter: invalid escape"); o += e; break;
      case 'x': o += (char)hexv(2); break;
      case 'u': JP::put_utf8(o, hexv(4)); break;
      case 'U': JP::put_utf8(o, hexv(8)); break;
      default:
        if (e >= '0' && e <= '7') {
          if (i + 2 > s.size()) cfg_error("filter: invalid escape");
          unsigned v = (unsigned)(e - '0');
          for (int k = 0; k < 2; k++) { char h = s[i + (size_t)k]; if (h < '0' || h > '7') cfg_error("filter: invalid escape"); v = v * 8 + (unsigned)(h - '0'); }
          i += 2; if (v > 255) cfg_error("filter: invalid escape"); o += (char)v;
        } else cfg_error("filter: invalid escape");
    }
  }
  return o;
}

int64_t days_from_civil(int64_t y, int m, int d) {
  y -= m <= 2;
  int64_t era = (y >= 0 ? y : y - 399) / 400;
  int64_t yoe = y - era * 400;
  int64_t doy = (153 * (m > 2 ? m - 3 : m + 9) + 2) / 5 + d - 1;
  int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + doe - 719468;
}
int dim(int m, int64_t y) {
  static const int t[] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  return (m == 2 && (y % 4 == 0 && (y % 100 != 0 || y % 400 == 0))) ? 29 : t[m - 1];
}

// grammar.go:114-144 findTimeLayout + time.Parse on the DateTime token → UnixMicro
int64_t parse_filter_time(const std::string &v) {
  auto num = [&](size_t at, int n) { int x = 0; for (int i = 0; i < n; i++) x = x * 10 + (v[at + (size_t)i] - '0'); return x; };
  int64_t y = num(0, 4); int mo = num(5, 2), d = num(8, 2), h = 0, mi = 0, se = 0; int64_t ns = 0; int off = 0;
  size_t k = 10;
  if (k < v.size() && v[k] == 'T') {
    h = num(k + 1, 2); mi = num(k + 4, 2); k += 6;
    if (k < v.size() && v[k] == ':') {
      se = num(k + 1, 2); k += 3;
      if (k < v.size() && v[k] == '.') { k++; int nd = 0; while (k < v.size() && dg(v[k])) { if (nd < 9) { ns = ns * 10 + (v[k] - '0'); nd++; } k++; } while (nd < 9) { ns *= 10; nd++; } }
    }
    if (k < v.size()) {
      if (v[k] == 'Z') k++;
      else {
        // layout Z07:00 if the zone text holds ':', else Z07 — both need exactly two digits per field
        int sign = v[k] == '-' ? -1 : 1; k++;
        size_t b = k; while (k < v.size() && dg(v[k])) k++;
        if (k - b != 2) cfg_error("filter: cannot parse datetime " + v);
        int hh = num(b, 2), mm = 0;
        if (k < v.size() && v[k] == ':') { k++; size_t c = k; while (k < v.size() && dg(v[k])) k++; if (k - c != 2) cfg_error("filter: cannot parse datetime " + v); mm = num(c, 2); }
        off = sign * (hh * 3600 + mm * 60);
      }
    }
  }
  if (k != v.size() || mo < 1 || mo > 12 || d < 1 || d > dim(mo, y) || h > 23 || mi > 59 || se > 59) cfg_error("filter: cannot parse datetime " + v);
  int64_t sec = days_from_civil(y, mo, d) * 86400 + h * 3600 + mi * 60 + se - off;
  return sec * 1000000 + ns / 1000;  // Time.UnixMicro()
}

struct P {
  std::vector<Tok> t; size_t pos = 0;
  Tok &peek() { return t[pos]; }
  void ws() { if (peek().t == T_WS) pos++; }
};

// strconv.ParseInt(s, 10, 64)
bool parse_i64(const std::string &s, int64_t &out) {
  size_t i = 0; bool neg = false;
  if (i < s.size() && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
  if (i >= s.size()) return false;
  unsigned long long v = 0;
  for (; i < s.size(); i++) {
    if (!dg(s[i])) return false;
    unsigned d = (unsigned)(s[i] - '0');
    if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10) return false;
    v = v * 10 + d;
  }
  if (!neg && v > 0x7FFFFFFFFFFFFFFFull) return false;
  if (neg && v > 0x8000000000000000ull) return false;
  out = neg ? (int64_t)(0 - v) : (int64_t)v;
  return true;
}

void scalar_into(P &p, FTerm &term, bool in_list) {
  Tok &t = p.peek();
  int32_t ty;
  switch (t.t) {
    case T_STR: ty = FV_STRING; break;
    case T_DT: ty = FV_TIME; break;
    case T_FLOAT: ty = FV_FLOAT; break;
    case T_INT: ty = FV_INT; break;
    case T_ID:
      if (kw(t, "TRUE") || kw(t, "FALSE")) ty = FV_BOOL;
      else if (kw(t, "NULL") || kw(t, "NIL")) ty = FV_NULL;
      else cfg_error("filter: unexpected token \"" + t.s + "\"");
      break;
    case T_PUNCT:
      if (t.s == "(" && in_list) cfg_error("filter: nested list are not supported");
      /* fallthrough */
    default: cfg_error("filter: unexpected token \"" + t.s + "\"");
  }
  bool first = term.ints.empty() && term.floats.empty() && term.strs.empty();
  if (in_list && !first && ty != term.vtype) cfg_error("filter: list items should have same type");
  term.vtype = ty;
  switch (ty) {
    case FV_STRING: term.strs.push_back(go_unquote(t.s)); break;
    case FV_TIME: term.ints.push_back(parse_filter_time(t.s)); break;
    case FV_FLOAT: term.floats.push_back(std::strtod(t.s.c_str(), nullptr)); break;
    case FV_INT: { int64_t v; if (!parse_i64(t.s, v)) cfg_error("filter: value out of range"); term.ints.push_back(v); break; }
    case FV_BOOL: term.ints.push_back(kw(t, "TRUE") ? 1 : 0); break;
    case FV_NULL: term.ints.push_back(0); break;
  }
  p.pos++;
}
}  // namespace

std::vector<FTerm> parse_filter(const std::string &src) {
  std::vector<FTerm> out;
  if (src.empty()) return out;
  P p; p.t = lex(src);
  p.ws();
  bool first = true;
  while (p.peek().t != T_EOF) {
    if (!first) {
      p.ws();
      if (!kw(p.peek(), "AND")) cfg_error("filter: unexpected token \"" + p.peek().s + "\"");
      p.pos++; p.ws();
    }
    first = false;
    FTerm term;
    if (p.peek().t != T_ID) cfg_error("filter: unexpected token \"" + p.peek().s + "\"");
    term.attr = p.peek().s; p.pos++; p.ws();
    Tok &o = p.peek();
    if (o.t == T_OP) {
      term.op = o.s == "=" ? F_EQ : o.s == "!=" ? F_NE : o.s == "<" ? F_LT : o.s == "<=" ? F_LE : o.s == ">" ? F_GT : o.s == ">=" ? F_GE : o.s == "~" ? F_MATCH : F_NOTMATCH;
      p.pos++;
    } else if (kw(o, "IN")) { term.op = F_IN; p.pos++; }
    else if (kw(o, "NOT")) {
      p.pos++;
      while (p.peek().t == T_WS) p.pos++;
      if (!kw(p.peek(), "IN")) cfg_error("filter: unexpected token \"" + p.peek().s + "\"");
      term.op = F_NOTIN; p.pos++;
    } else cfg_error("filter: unexpected token \"" + o.s + "\"");
    p.ws();
    if (p.peek().t == T_PUNCT && p.peek().s == "(") {
      term.is_list = true;
      p.pos++; p.ws();
      for (;;) {
        scalar_into(p, term, true);
        p.ws(); p.ws();
        if (p.peek().t == T_PUNCT && p.peek().s == ",") { p.pos++; p.ws(); continue; }
        if (p.peek().t == T_PUNCT && p.peek().s == ")") { p.pos++; break; }
        cfg_error("filter: unexpected token \"" + p.peek().s + "\"");
      }
    } else scalar_into(p, term, false);
    p.ws();
    // validateTerm filters.go:240-272
    if (term.is_list) { if (term.op != F_IN && term.op != F_NOTIN) cfg_error("filter: list values require [ NOT ] IN operator"); }
    else if (term.op == F_IN || term.op == F_NOTIN) cfg_error("filter: IN operator expect list value");
    if (!term.is_list && term.vtype == FV_NULL && term.op != F_EQ && term.op != F_NE) cfg_error("filter: NULL expects \"=\" or \"!=\" operator");
    out.push_back(std::move(term));
  }
  return out;
}

// ===================== SHA-256 midstate (host) =====================
void sha256_midstate(const uint8_t block[64], uint32_t h[8]) {
  static const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
      0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
      0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
      0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
      0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
      0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = (uint32_t)block[4 * i] << 24 | (uint32_t)block[4 * i + 1] << 16 | (uint32_t)block[4 * i + 2] << 8 | block[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; i++) {
    uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
    uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

static void sha256_host(const uint8_t *msg, size_t n, uint8_t out[32]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  std::vector<uint8_t> m(msg, msg + n);
  m.push_back(0x80);
  while (m.size() % 64 != 56) m.push_back(0);
  uint64_t bits = (uint64_t)n * 8;
  for (int i = 0; i < 8; i++) m.push_back((uint8_t)(bits >> (56 - 8 * i)));
  for (size_t i = 0; i < m.size(); i += 64) sha256_midstate(m.data() + i, h);
  for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}

// ===================== factories =====================
static NameFilter tables_of(const Json &cfg) {
  NameFilter f;
  const Json *t = cfg.get("tables");
  if (t) f.init(t->strings("includeTables"), t->strings("excludeTables")); else f.init({}, {});
  return f;
}
static NameFilter columns_of(const Json &cfg) {
  NameFilter f;
  const Json *t = cfg.get("columns");
  if (t && t->type == Json::Obj) f.init(t->strings("includeColumns"), t->strings("excludeColumns")); else f.init({}, {});
  return f;
}

static std::string join_keys(const std::vector<std::string> &v) {
  std::string o;
  for (size_t i = 0; i < v.size(); i++) { if (i) o += ", "; o += v[i]; }
  return o;
}
std::unique_ptr<tfgpu_plan> make_plan(const std::string &type_name, const std::string &config_json) {
  Json cfg = Json::parse(config_json.empty() ? "{}" : config_json);
  auto p = std::make_unique<tfgpu_plan>();
  p->type_name = type_name;
  if (type_name == "mask_field") {  // mask/mask.go:21-42, hmac_hasher.go:108-117
    p->kind = PK_MASK;
    p->tables = tables_of(cfg);
    p->columns.init({}, {});
    const Json *h = cfg.get("maskFunctionHash");
    p->salt = h ? h->s("userDefinedSalt") : "";
    p->mask_cols = cfg.strings("columns");
    // hmac key block: keys longer than the block size are hashed first (crypto/hmac)
    uint8_t k0[64] = {0};
    if (p->salt.size() > 64) sha256_host((const uint8_t *)p->salt.data(), p->salt.size(), k0);
    else std::memcpy(k0, p->salt.data(), p->salt.size());
    uint8_t ipad[64], opad[64];
    for (int i = 0; i < 64; i++) { ipad[i] = k0[i] ^ 0x36; opad[i] = k0[i] ^ 0x5c; }
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    std::memcpy(p->ipad_state, iv, sizeof iv); std::memcpy(p->opad_state, iv, sizeof iv);
    sha256_midstate(ipad, p->ipad_state);
    sha256_midstate(opad, p->opad_state);
  } else if (type_name == "rename_tables") {  // rename/rename.go:20-41,85-93
    p->kind = PK_RENAME;
    p->tables.init({}, {}); p->columns.init({}, {});
    const Json *rt = cfg.get("renameTables");
    if (rt && rt->type == Json::Arr)
      for (auto &r : rt->arr) {
        const Json *o = r.get("originalName"), *n = r.get("newName");
        std::array<std::string, 4> e = {o ? o->s("nameSpace") : "", o ? o->s("name") : "", n ? n->s("nameSpace") : "", n ? n->s("name") : ""};
        bool replaced = false;
        for (auto &x : p->renames) if (x[0] == e[0] && x[1] == e[1]) { x = e; replaced = true; }  // map: last entry wins
        if (!replaced) p->renames.push_back(e);
      }
  } else if (type_name == "filter_columns") {  // filter/filter_columns_transformer.go:16-31
    p->kind = PK_FILTER_COLUMNS;
    p->tables = tables_of(cfg); p->columns = columns_of(cfg);
  } else if (type_name == "skip_events") {  // filter/skip_events.go:24-47
    p->kind = PK_SKIP_EVENTS;
    p->tables = tables_of(cfg); p->columns.init({}, {});
    for (auto &e : cfg.strings("events")) {  // kinds compare as exact strings
      if (e == "insert") p->skip[TFGPU_K_INSERT] = true;
      else if (e == "update") p->skip[TFGPU_K_UPDATE] = true;
      else if (e == "delete") p->skip[TFGPU_K_DELETE] = true;
    }
  } else if (type_name == "filter_rows") {  // filter_rows/filter_rows.go:42-96
    p->kind = PK_FILTER_ROWS;
    p->tables = tables_of(cfg); p->columns.init({}, {});
    std::string single = cfg.s("filter");
    std::vector<std::string> many = cfg.strings("filters");
    if (!single.empty() && !many.empty()) cfg_error("Settings 'filters' and 'filter' cannot be enabled at the same time");
    if (many.empty()) many.push_back(single);
    for (auto &f : many) {
      FExpr e;
      try { e.terms = parse_filter(f); }
      catch (const Error &er) { cfg_error("Unable to parse filter '" + f + "': " + er.what()); }
      for (auto &t : e.terms)  // valuesListToSet util.go:84-112
        if (t.is_list && (t.vtype == FV_BOOL || t.vtype == FV_NULL)) cfg_error("Unable to prepare term values: not appropriate type of list values");
      p->exprs.push_back(std::move(e));
    }
  } else if (type_name == "convert_to_string") {  // to_string/to_string.go:19-46
    p->kind = PK_TO_STRING;
    p->tables = tables_of(cfg); p->columns = columns_of(cfg);
    p->to_bytes = cfg.flag("convert_to_bytes"); p->skip_utc = cfg.flag("skip_utc_conversion");
  } else if (type_name == "convert_to_datetime") {  // to_datetime/to_datetime.go:23-47
    p->kind = PK_TO_DATETIME;
    p->tables = tables_of(cfg); p->columns = columns_of(cfg);
  } else if (type_name == "sharder_transformer") {  // sharder/sharder.go:20-68
    p->kind = PK_SHARDER;
    p->tables = tables_of(cfg);
    p->is_random = cfg.flag("is_random");
    if (p->is_random) p->columns.init({}, {}); else p->columns = columns_of(cfg);
    std::string sc = cfg.s("shardsCount");
    if (!parse_i64(sc, p->shards)) cfg_error("cannot parse param as int: " + sc);
    if (p->is_random) throw Error(TFGPU_ERR_UNSUPPORTED, "sharder_transformer is_random=true draws from math/rand on the host; not a device transform");
    if (p->shards <= 0 || p->shards > 0xFFFFFFFFll) cfg_error("shardsCount out of range");
  } else if (type_name == "replace_primary_key") {  // replace_primary_key/replace_primary_key.go:24-27, 134-150
    p->kind = PK_REPLACE_PK;
    p->tables = tables_of(cfg); p->columns.init({}, {});
    p->new_keys = cfg.strings("keys");
    for (size_t i = 0; i < p->new_keys.size(); i++)
      for (size_t j = 0; j < i; j++)
        if (p->new_keys[i] == p->new_keys[j]) cfg_error("Can't use same keys column names twice: " + join_keys(p->new_keys));
  } else if (type_name == "sql") {
    // clickhouse/clickhouse_local.go:60-63, 436-452: {tables, query}.  The reference runs the query in an external
    // clickhouse-local process; the device takes the predicate + cast subset tf_sql.cpp documents and refuses the rest by
    // name, so such a transformer stays on the host.
    p->kind = PK_SQL;
    p->tables = tables_of(cfg); p->columns.init({}, {});
    p->sql_query = cfg.s("query");
    sql_parse(p->sql_query, *p);
  } else if (type_name == "lambda" || type_name == "dbt" || type_name == "logger" || type_name == "yt_dict_transformer" || type_name == "raw_doc_grouper" ||
             type_name == "raw_cdc_doc_grouper" || type_name == "table_splitter_transformer" || type_name == "number_to_float_transformer" ||
             type_name == "problem_item_detector" || type_name == "batch_splitter" || type_name == "filter_strm_access_log" || type_name == "jsonparser" ||
             type_name == "regex_replace_transformer" || type_name == "filter_rows_by_ids" || type_name == "mongo_pk_extender") {
    // registered in the reference (pkg/transformer/registry/*), outside the device subset (out of scope)
    throw Error(TFGPU_ERR_UNSUPPORTED, "transformer type " + type_name + " has no device plan: keep it on the host");
  } else {
    throw Error(TFGPU_ERR_UNKNOWN_TYPE, "transformer type not registered: " + type_name);
  }
  return p;
}

static bool todt_col(const tfgpu_plan &p, const std::string &name, int dtype) {
  return p.columns.match(name) && (dtype == TFGPU_T_INT32 || dtype == TFGPU_T_UINT32);
}

static bool fr_col_suitable(const FTerm &t, int dt) {  // filter_rows.go:453-490
  if (t.is_list) return true;
  switch (t.vtype) {
    case FV_BOOL: return dt == TFGPU_T_BOOLEAN;
    case FV_FLOAT: case FV_INT: return (dt >= TFGPU_T_INT8 && dt <= TFGPU_T_UINT64) || dt == TFGPU_T_FLOAT32 || dt == TFGPU_T_FLOAT64;
    case FV_STRING: return dt == TFGPU_T_UTF8 || dt == TFGPU_T_BYTES || dt == TFGPU_T_ANY;
    case FV_TIME: return dt == TFGPU_T_TIMESTAMP || dt == TFGPU_T_DATE || dt == TFGPU_T_DATETIME;
    case FV_NULL: return true;
  }
  return false;
}

bool plan_suitable(const tfgpu_plan &p, const std::string &ns, const std::string &name, const tfgpu_schema &s) {
  switch (p.kind) {
    case PK_RENAME:
      for (auto &r : p.renames) if (r[0] == ns && r[1] == name) return true;
      return false;
    case PK_MASK:  // hmac_hasher.go:76-89
      if (!p.tables.match_table(ns, name)) return false;
      if (p.mask_cols.empty()) return true;
      for (int i = 0; i < s.ncols; i++) if (p.mask_has(s.cols[i].name)) return true;
      return false;
    case PK_FILTER_COLUMNS:  // filter_columns_transformer.go:215-226
      if (!p.tables.match_table(ns, name)) return false;
      for (int i = 0; i < s.ncols; i++) if (!p.columns.match(s.cols[i].name) && (s.cols[i].flags & TFGPU_COL_KEY)) return false;
      return true;
    case PK_SKIP_EVENTS: return p.tables.match_table(ns, name);
    case PK_FILTER_ROWS:  // filter_rows.go:418-451
      if (!p.tables.match_table(ns, name)) return false;
      for (auto &e : p.exprs)
        for (auto &t : e.terms) {
          bool found = false;
          for (int i = 0; i < s.ncols; i++) if (t.attr == s.cols[i].name) { found = true; if (!fr_col_suitable(t, s.cols[i].dtype)) return false; break; }
          if (!found && s.ncols > 0) return false;
        }
      return true;
    case PK_TO_STRING: case PK_SHARDER:
      if (!p.tables.match_table(ns, name)) return false;
      if (p.columns.empty()) return true;
      for (int i = 0; i < s.ncols; i++) if (p.columns.match(s.cols[i].name)) return true;
      return false;
    case PK_REPLACE_PK: {  // replace_primary_key.go:104-106: containsAllKeys counts the schema's columns that are new keys
      if (!p.tables.match_table(ns, name)) return false;
      size_t hits = 0;
      for (int i = 0; i < s.ncols; i++) if (p.is_new_key(s.cols[i].name ? s.cols[i].name : "")) hits++;
      return hits == p.new_keys.size();
    }
    case PK_SQL:  // clickhouse_local.go:335-349: the table filter decides; a result without columns / key is only warned about
      return p.tables.match_table(ns, name);
    case PK_TO_DATETIME:  // to_datetime.go:63-76
      if (!p.tables.match_table(ns, name)) return false;
      if (p.columns.empty()) return false;
      for (int i = 0; i < s.ncols; i++) if (todt_col(p, s.cols[i].name, s.cols[i].dtype)) return true;
      return false;
  }
  return false;
}

static std::string join(const std::vector<std::string> &v, const char *sep) {
  std::string o;
  for (size_t i = 0; i < v.size(); i++) { if (i) o += sep; o += v[i]; }
  return o;
}
static std::string trim100(std::string s) { if (s.size() > 100) s.resize(100); return s; }

std::string plan_description(const tfgpu_plan &p) {
  switch (p.kind) {
    case PK_MASK:  // hmac_hasher.go:95-103
      return "Hash table columns: columns: " + join(p.mask_cols, ",") + ", includedtables: " + join(p.tables.include_src, ",") + ", excluded tables: " + join(p.tables.exclude_src, ",");
    case PK_RENAME: {
      std::vector<std::string> r;
      for (auto &x : p.renames) r.push_back((x[0].empty() ? "" : dq(x[0]) + ".") + dq(x[1]) + "->" + (x[2].empty() ? "" : dq(x[2]) + ".") + dq(x[3]));
      return "Rename tables: " + join(r, ", ");
    }
    case PK_FILTER_COLUMNS: return "Column filter";
    case PK_SKIP_EVENTS: return "skips the following event types";
    case PK_FILTER_ROWS: return "Transformer for filtering rows by provided filter.";
    case PK_SQL: return "SQL transfer";  // clickhouse_local.go:430-432
    case PK_TO_STRING:
      if (p.columns.empty()) return "Transform to string all column values";
      return "Transform to string column values (include: " + trim100(join(p.columns.include_src, "|")) + ", exclude: " + trim100(join(p.columns.exclude_src, "|")) + ")";
    case PK_TO_DATETIME:
      if (p.columns.empty()) return "Transform to datetime uint32 column values";
      return "Transform to datetime uint32 column values (include: " + trim100(join(p.columns.include_src, "|")) + ", exclude: " + trim100(join(p.columns.exclude_src, "|")) + ")";
    case PK_REPLACE_PK: return "Replace primary keys to: " + join_keys(p.new_keys) + " ";
    case PK_SHARDER:
      if (p.columns.empty()) return "Transform to shard tables by field values";
      return "Transform to shard tables by field values (include: " + trim100(join(p.columns.include_src, "|")) + ", exclude: " + trim100(join(p.columns.exclude_src, "|")) + ", shards_num: " + std::to_string(p.shards) + ")";
  }
  return "";
}

}  // namespace tf
