/*
 * oracle/ora_jsonparse.c — CPU restatement of the generic JSON parser
 * (SURVEY.md §8 a17).  TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 * Follows, line by line:
 *   pkg/parsers/generic/generic_parser.go
 *     doGenericParser :519-555   bufio.ScanLines split, empty lines skipped before idx++
 *     Unmarshal (json) :672-731  fastjson parse, per-key typed extraction
 *     makeChangeItem   :297-404  ParseVal per raw field, key/required rules, _rest, dedupe keys
 *     ParseVal         :888-1123 coercion matrix
 *     extractTimeValue :818-886  datetime sources
 *     addAuxFields     :99-154   aux schema columns
 *
 * Third-party arithmetic that is NOT in /root/reference (go.mod pins github.com/valyala/fastjson
 * v1.6.4), restated here from its published algorithm and pinned through the reference's own
 * canon files (tests/golden/json_parser.json: tests/canon/parser json + mdb canon, pkg/parsers/
 * generic canon TestParserNumberTypes):
 *   fastjson parser.go  Parse/parseValue/parseObject/parseArray/parseRawString/parseRawKey/
 *                       parseRawNumber, skipWS, unescapeStringBestEffort, Value.MarshalTo
 *   fastjson fastfloat  ParseBestEffort, ParseInt64BestEffort, ParseUint64BestEffort
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include "ora.h"

/* ------------------------------------------------------------------ */
/* fastjson value tree                                                 */
/* ------------------------------------------------------------------ */
typedef enum { FJ_NULL, FJ_OBJ, FJ_ARR, FJ_STR, FJ_NUM, FJ_TRUE, FJ_FALSE } fj_type;
typedef struct fj_value fj_value;
struct fj_value {
  fj_type t;
  const char *s; size_t n;   /* FJ_STR: raw (still escaped) contents; FJ_NUM: raw token */
  int nkids;
  fj_value **kids;           /* array items / object values */
  const char **keys; size_t *klen; /* object keys, raw */
};
#define FJ_MAXDEPTH 300

static void fj_free(fj_value *v) {
  if (!v) return;
  for (int i = 0; i < v->nkids; i++) fj_free(v->kids[i]);
  free(v->kids); free(v->keys); free(v->klen); free(v);
}
static fj_value *fj_new(fj_type t) { fj_value *v = (fj_value *)calloc(1, sizeof *v); v->t = t; return v; }
static void fj_push(fj_value *p, fj_value *kid, const char *k, size_t kl) {
  p->kids = (fj_value **)realloc(p->kids, (size_t)(p->nkids + 1) * sizeof(fj_value *));
  p->kids[p->nkids] = kid;
  if (p->t == FJ_OBJ) {
    p->keys = (const char **)realloc(p->keys, (size_t)(p->nkids + 1) * sizeof(char *));
    p->klen = (size_t *)realloc(p->klen, (size_t)(p->nkids + 1) * sizeof(size_t));
    p->keys[p->nkids] = k; p->klen[p->nkids] = kl;
  }
  p->nkids++;
}

typedef struct { const char *p, *e; const char *err; } fjp;

/* skipWS: 0x20, 0x0A, 0x09, 0x0D */
static void fj_ws(fjp *s) { while (s->p < s->e && (*s->p == ' ' || *s->p == '\n' || *s->p == '\t' || *s->p == '\r')) s->p++; }

/* parseRawString (s points after the opening quote): the closing quote is the first '"' preceded by an
 * even number of backslashes */
static int fj_raw_string(fjp *s, const char **out, size_t *n) {
  const char *b = s->p;
  const char *q = b;
  for (;;) {
    q = (const char *)memchr(q, '"', (size_t)(s->e - q));
    if (!q) { s->err = "missing closing '\"'"; s->p = s->e; return -1; }
    size_t bs = 0;
    while (q - bs > b && q[-1 - (ptrdiff_t)bs] == '\\') bs++;
    if (bs % 2 == 0) break;
    q++;
  }
  *out = b; *n = (size_t)(q - b);
  s->p = q + 1;
  return 0;
}

static int strncaseeq(const char *a, const char *b, size_t n) { return strncasecmp(a, b, n) == 0; }

/* parseRawNumber */
static int fj_raw_number(fjp *s, const char **out, size_t *n) {
  const char *b = s->p;
  size_t len = (size_t)(s->e - b);
  for (size_t i = 0; i < len; i++) {
    char ch = b[i];
    if ((ch >= '0' && ch <= '9') || ch == '.' || ch == '-' || ch == 'e' || ch == 'E' || ch == '+') continue;
    if (i == 0 || (i == 1 && (b[0] == '-' || b[0] == '+'))) {
      if (len - i >= 3 && (strncaseeq(b + i, "inf", 3) || strncaseeq(b + i, "nan", 3))) { *out = b; *n = i + 3; s->p = b + i + 3; return 0; }
      s->err = "unexpected char"; return -1;
    }
    *out = b; *n = i; s->p = b + i; return 0;
  }
  *out = b; *n = len; s->p = s->e;
  return 0;
}

static fj_value *fj_parse_value(fjp *s, int depth);

static fj_value *fj_parse_array(fjp *s, int depth) {
  fj_ws(s);
  if (s->p >= s->e) { s->err = "missing ']'"; return NULL; }
  fj_value *a = fj_new(FJ_ARR);
  if (*s->p == ']') { s->p++; return a; }
  for (;;) {
    fj_ws(s);
    fj_value *v = fj_parse_value(s, depth);
    if (!v) { fj_free(a); return NULL; }
    fj_push(a, v, NULL, 0);
    fj_ws(s);
    if (s->p >= s->e) { s->err = "unexpected end of array"; fj_free(a); return NULL; }
    if (*s->p == ',') { s->p++; continue; }
    if (*s->p == ']') { s->p++; return a; }
    s->err = "missing ',' after array value"; fj_free(a); return NULL;
  }
}

static fj_value *fj_parse_object(fjp *s, int depth) {
  fj_ws(s);
  if (s->p >= s->e) { s->err = "missing '}'"; return NULL; }
  fj_value *o = fj_new(FJ_OBJ);
  if (*s->p == '}') { s->p++; return o; }
  for (;;) {
    fj_ws(s);
    if (s->p >= s->e || *s->p != '"') { s->err = "cannot find opening '\"' for object key"; fj_free(o); return NULL; }
    s->p++;
    const char *k; size_t kl;
    if (fj_raw_string(s, &k, &kl)) { fj_free(o); return NULL; }  /* parseRawKey == parseRawString on every input */
    fj_ws(s);
    if (s->p >= s->e || *s->p != ':') { s->err = "missing ':' after object key"; fj_free(o); return NULL; }
    s->p++;
    fj_ws(s);
    fj_value *v = fj_parse_value(s, depth);
    if (!v) { fj_free(o); return NULL; }
    fj_push(o, v, k, kl);
    fj_ws(s);
    if (s->p >= s->e) { s->err = "unexpected end of object"; fj_free(o); return NULL; }
    if (*s->p == ',') { s->p++; continue; }
    if (*s->p == '}') { s->p++; return o; }
    s->err = "missing ',' after object value"; fj_free(o); return NULL;
  }
}

static fj_value *fj_parse_value(fjp *s, int depth) {
  if (s->p >= s->e) { s->err = "cannot parse empty string"; return NULL; }
  depth++;
  if (depth > FJ_MAXDEPTH) { s->err = "too big depth for the nested JSON"; return NULL; }
  char c = *s->p;
  size_t left = (size_t)(s->e - s->p);
  if (c == '{') { s->p++; return fj_parse_object(s, depth); }
  if (c == '[') { s->p++; return fj_parse_array(s, depth); }
  if (c == '"') {
    s->p++;
    fj_value *v = fj_new(FJ_STR);
    if (fj_raw_string(s, &v->s, &v->n)) { fj_free(v); return NULL; }
    return v;
  }
  if (c == 't') { if (left < 4 || memcmp(s->p, "true", 4)) { s->err = "unexpected value found"; return NULL; } s->p += 4; return fj_new(FJ_TRUE); }
  if (c == 'f') { if (left < 5 || memcmp(s->p, "false", 5)) { s->err = "unexpected value found"; return NULL; } s->p += 5; return fj_new(FJ_FALSE); }
  if (c == 'n') {
    if (left < 4 || memcmp(s->p, "null", 4)) {
      if (left >= 3 && strncaseeq(s->p, "nan", 3)) { fj_value *v = fj_new(FJ_NUM); v->s = s->p; v->n = 3; s->p += 3; return v; }
      s->err = "unexpected value found"; return NULL;
    }
    s->p += 4; return fj_new(FJ_NULL);
  }
  fj_value *v = fj_new(FJ_NUM);
  if (fj_raw_number(s, &v->s, &v->n)) { fj_free(v); s->err = "cannot parse number"; return NULL; }
  return v;
}

/* Parser.Parse */
static fj_value *fj_parse(const char *line, size_t n, const char **err) {
  fjp s = {line, line + n, NULL};
  fj_ws(&s);
  fj_value *v = fj_parse_value(&s, 0);
  if (!v) { *err = s.err ? s.err : "cannot parse JSON"; return NULL; }
  fj_ws(&s);
  if (s.p < s.e) { fj_free(v); *err = "unexpected tail"; return NULL; }
  return v;
}

/* ---- unescapeStringBestEffort ---- */
static size_t utf8_enc(char *d, unsigned cp) {  /* string(rune(x)); invalid → U+FFFD */
  if (cp > 0x10FFFF || (cp >= 0xD800 && cp < 0xE000)) cp = 0xFFFD;
  if (cp < 0x80) { d[0] = (char)cp; return 1; }
  if (cp < 0x800) { d[0] = (char)(0xC0 | cp >> 6); d[1] = (char)(0x80 | (cp & 0x3F)); return 2; }
  if (cp < 0x10000) { d[0] = (char)(0xE0 | cp >> 12); d[1] = (char)(0x80 | ((cp >> 6) & 0x3F)); d[2] = (char)(0x80 | (cp & 0x3F)); return 3; }
  d[0] = (char)(0xF0 | cp >> 18); d[1] = (char)(0x80 | ((cp >> 12) & 0x3F)); d[2] = (char)(0x80 | ((cp >> 6) & 0x3F)); d[3] = (char)(0x80 | (cp & 0x3F)); return 4;
}
static int hex4(const char *s, unsigned *out) {  /* strconv.ParseUint(xs, 16, 16) */
  unsigned v = 0;
  for (int i = 0; i < 4; i++) {
    char c = s[i]; unsigned d;
    if (c >= '0' && c <= '9') d = (unsigned)(c - '0'); else if (c >= 'a' && c <= 'f') d = (unsigned)(c - 'a' + 10); else if (c >= 'A' && c <= 'F') d = (unsigned)(c - 'A' + 10); else return -1;
    v = v * 16 + d;
  }
  *out = v; return 0;
}
static char *fj_unescape(const char *s, size_t n, size_t *olen) {
  char *out = (char *)malloc(n + 4);
  size_t w = 0, i = 0;
  const char *bs = (const char *)memchr(s, '\\', n);
  if (!bs) { memcpy(out, s, n); out[n] = 0; *olen = n; return out; }
  w = (size_t)(bs - s); memcpy(out, s, w); i = w + 1;
  while (i < n) {
    char ch = s[i++];
    switch (ch) {
      case '"': out[w++] = '"'; break; case '\\': out[w++] = '\\'; break; case '/': out[w++] = '/'; break;
      case 'b': out[w++] = '\b'; break; case 'f': out[w++] = '\f'; break; case 'n': out[w++] = '\n'; break;
      case 'r': out[w++] = '\r'; break; case 't': out[w++] = '\t'; break;
      case 'u': {
        unsigned x;
        if (n - i < 4 || hex4(s + i, &x)) { out[w++] = '\\'; out[w++] = 'u'; break; }
        const char *xs = s + i;
        i += 4;
        if (!(x >= 0xD800 && x < 0xE000)) { w += utf8_enc(out + w, x); break; }
        unsigned x1;
        if (n - i < 6 || s[i] != '\\' || s[i + 1] != 'u' || hex4(s + i + 2, &x1)) { out[w++] = '\\'; out[w++] = 'u'; memcpy(out + w, xs, 4); w += 4; break; }
        unsigned r = 0xFFFD;  /* utf16.DecodeRune */
        if (x >= 0xD800 && x < 0xDC00 && x1 >= 0xDC00 && x1 < 0xE000) r = ((x - 0xD800) << 10 | (x1 - 0xDC00)) + 0x10000;
        w += utf8_enc(out + w, r);
        i += 6;
        break;
      }
      default: out[w++] = '\\'; out[w++] = ch;
    }
    const char *nx = i < n ? (const char *)memchr(s + i, '\\', n - i) : NULL;
    if (!nx) { if (i < n) { memcpy(out + w, s + i, n - i); w += n - i; } break; }
    memcpy(out + w, s + i, (size_t)(nx - (s + i))); w += (size_t)(nx - (s + i));
    i = (size_t)(nx - s) + 1;
  }
  out[w] = 0; *olen = w;
  return out;
}

/* ---- fastfloat ---- */
static const double pow10tab_[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16};
static int ieq(const char *s, size_t n, const char *w) { return strlen(w) == n && strncasecmp(s, w, n) == 0; }

/* math.Pow10: table lookups (pow10tab 1e0..1e31, pow10postab32, pow10negtab32), i.e. the correctly rounded
 * decimal literal 1e<n>; n < -323 → 0, n > 308 → +Inf */
static double go_pow10(int n) {
  if (n < -323) return 0;
  if (n > 308) return INFINITY;
  char buf[16];
  snprintf(buf, sizeof buf, "1e%d", n);
  if (n >= 0) {  /* pow10postab32[n/32] * pow10tab[n%32] — a float64 product of two exact-literal table entries */
    char a[16], b[16];
    snprintf(a, sizeof a, "1e%d", (n / 32) * 32); snprintf(b, sizeof b, "1e%d", n % 32);
    return strtod(a, NULL) * strtod(b, NULL);
  }
  char a[16], b[16];
  snprintf(a, sizeof a, "1e-%d", ((-n) / 32) * 32); snprintf(b, sizeof b, "1e%d", (-n) % 32);
  return strtod(a, NULL) / strtod(b, NULL);
}

static double strconv_parse_float_or0(const char *s, size_t n) {  /* f, err := strconv.ParseFloat(s, 64); if err != nil && !IsInf(f) → 0 */
  double f;
  int rc = ora_parse_float(s, n, 64, &f);
  if (rc && !isinf(f)) return 0;
  return f;
}

double ora_fastfloat_parse_best_effort(const char *s, size_t n) {
  if (n == 0) return 0;
  size_t i = 0;
  int minus = s[0] == '-';
  if (minus) { i++; if (i >= n) return 0; }
  if (s[i] == '.' && (i + 1 >= n || s[i + 1] < '0' || s[i + 1] > '9')) return 0;
  uint64_t d = 0;
  size_t j = i;
  while (i < n) {
    if (s[i] >= '0' && s[i] <= '9') {
      d = d * 10 + (uint64_t)(s[i] - '0');
      i++;
      if (i > 18) return strconv_parse_float_or0(s, n);
      continue;
    }
    break;
  }
  if (i <= j && s[i] != '.') {
    const char *t = s + i; size_t tn = n - i;
    if (tn && t[0] == '+') { t++; tn--; }
    if (ieq(t, tn, "inf") || ieq(t, tn, "infinity")) return minus ? -INFINITY : INFINITY;
    if (ieq(t, tn, "nan")) return NAN;
    return 0;
  }
  double f = (double)d;
  if (i >= n) return minus ? -f : f;
  if (s[i] == '.') {
    i++;
    if (i >= n) return 0;
    size_t k = i;
    while (i < n) {
      if (s[i] >= '0' && s[i] <= '9') {
        d = d * 10 + (uint64_t)(s[i] - '0');
        i++;
        if (i - j >= sizeof pow10tab_ / sizeof pow10tab_[0]) return strconv_parse_float_or0(s, n);
        continue;
      }
      break;
    }
    if (i < k) return 0;
    f = (double)d / pow10tab_[i - k];
    if (i >= n) return minus ? -f : f;
  }
  if (s[i] == 'e' || s[i] == 'E') {
    i++;
    if (i >= n) return 0;
    int exp_minus = 0;
    if (s[i] == '+' || s[i] == '-') { exp_minus = s[i] == '-'; i++; if (i >= n) return 0; }
    int exp = 0;
    j = i;
    while (i < n) {
      if (s[i] >= '0' && s[i] <= '9') {
        exp = exp * 10 + (s[i] - '0');
        i++;
        if (exp > 300) return strconv_parse_float_or0(s, n);
        continue;
      }
      break;
    }
    if (i <= j) return 0;
    if (exp_minus) exp = -exp;
    f *= go_pow10(exp);
    if (i >= n) return minus ? -f : f;
  }
  return 0;
}

int64_t ora_fastfloat_parse_int64_best_effort(const char *s, size_t n) {
  if (n == 0) return 0;
  size_t i = 0;
  int minus = s[0] == '-';
  if (minus) { i++; if (i >= n) return 0; }
  int64_t d = 0;
  size_t j = i;
  while (i < n) {
    if (s[i] >= '0' && s[i] <= '9') {
      d = d * 10 + (s[i] - '0');
      i++;
      if (i > 18) { int64_t dd; if (ora_parse_int(s, n, 10, 64, &dd)) return 0; return dd; }
      continue;
    }
    break;
  }
  if (i <= j) return 0;
  if (i < n) return 0;
  return minus ? -d : d;
}
uint64_t ora_fastfloat_parse_uint64_best_effort(const char *s, size_t n) {
  if (n == 0) return 0;
  size_t i = 0;
  uint64_t d = 0;
  while (i < n) {
    if (s[i] >= '0' && s[i] <= '9') {
      d = d * 10 + (uint64_t)(s[i] - '0');
      i++;
      if (i > 18) { uint64_t dd; if (ora_parse_uint(s, n, 10, 64, &dd)) return 0; return dd; }
      continue;
    }
    break;
  }
  if (i == 0) return 0;
  if (i < n) return 0;
  return d;
}

/* ---- Value.MarshalTo (v.String()) ---- */
typedef struct { char *p; size_t n, cap; } sb;
static void sb_put(sb *b, const void *s, size_t n) {
  if (b->n + n + 1 > b->cap) { b->cap = (b->n + n + 1) * 2; b->p = (char *)realloc(b->p, b->cap); }
  if (n) memcpy(b->p + b->n, s, n);
  b->n += n; b->p[b->n] = 0;
}
static void sb_c(sb *b, char c) { sb_put(b, &c, 1); }
static void fj_marshal(sb *b, const fj_value *v, int top_keys_unescaped);
/* escapeString: plain if no byte < 0x20, '"' or '\\'; otherwise strconv.AppendQuote — returns -1 (not restated) */
static int fj_escape_string(sb *b, const char *s, size_t n) {
  for (size_t i = 0; i < n; i++) { unsigned char c = (unsigned char)s[i]; if (c < 0x20 || c == '"' || c == '\\') return -1; }
  sb_c(b, '"'); sb_put(b, s, n); sb_c(b, '"');
  return 0;
}
static int g_marshal_unsupported;
static void fj_marshal(sb *b, const fj_value *v, int keys_unescaped) {
  switch (v->t) {
    case FJ_NULL: sb_put(b, "null", 4); break;
    case FJ_TRUE: sb_put(b, "true", 4); break;
    case FJ_FALSE: sb_put(b, "false", 5); break;
    case FJ_NUM: sb_put(b, v->s, v->n); break;
    case FJ_STR: sb_c(b, '"'); sb_put(b, v->s, v->n); sb_c(b, '"'); break;  /* typeRawString: raw bytes kept */
    case FJ_ARR:
      sb_c(b, '[');
      for (int i = 0; i < v->nkids; i++) { fj_marshal(b, v->kids[i], 0); if (i != v->nkids - 1) sb_c(b, ','); }
      sb_c(b, ']');
      break;
    case FJ_OBJ:
      sb_c(b, '{');
      for (int i = 0; i < v->nkids; i++) {
        if (keys_unescaped) {
          size_t kl; char *k = fj_unescape(v->keys[i], v->klen[i], &kl);
          if (fj_escape_string(b, k, kl)) g_marshal_unsupported = 1;
          free(k);
        } else { sb_c(b, '"'); sb_put(b, v->keys[i], v->klen[i]); sb_c(b, '"'); }
        sb_c(b, ':');
        fj_marshal(b, v->kids[i], 0);
        if (i != v->nkids - 1) sb_c(b, ',');
      }
      sb_c(b, '}');
      break;
  }
}

/* ------------------------------------------------------------------ */
/* Go values of one parsed line (map[string]interface{})               */
/* ------------------------------------------------------------------ */
/* OV_JSON holds composite `any` values (map / slice) as encoding/json text with sorted keys; floats inside
 * use the json float format.  When the composite holds something this file does not restate, the line is
 * flagged ORA_JL_UNRESTATED so tests skip it. */
enum { ORA_JL_ROW = 0, ORA_JL_SKIPPED = 1, ORA_JL_UNPARSED = 2, ORA_JL_UNRESTATED = 3 };

static void json_float(sb *b, double f) {  /* encoding/json floatEncoder, 64-bit */
  char tmp[64];
  double a = fabs(f);
  char fmt = 'f';
  if (a != 0 && (a < 1e-6 || a >= 1e21)) fmt = 'e';
  if (fmt == 'f') { size_t n = ora_fmt_float(tmp, f, 'f', 64); sb_put(b, tmp, n); return; }
  /* 'e' with shortest digits: d.ddde±xx, then e-09 → e-9 */
  size_t n = ora_fmt_float(tmp, f, 'g', 64);  /* %v == 'g' with exponent threshold 21: for these magnitudes 'g' is in e-form */
  /* clean e-0X → e-X */
  if (n >= 4 && tmp[n - 4] == 'e' && tmp[n - 3] == '-' && tmp[n - 2] == '0') { tmp[n - 2] = tmp[n - 1]; n--; }
  sb_put(b, tmp, n);
}

static void go_json_string(sb *b, const char *s, size_t n) {  /* encoding/json string, escapeHTML = true (json.Marshal) */
  static const char HEXC[] = "0123456789abcdef";
  sb_c(b, '"');
  size_t i = 0;
  while (i < n) {
    unsigned char c = (unsigned char)s[i];
    if (c < 0x80) {
      if (c >= 0x20 && c != '"' && c != '\\' && c != '<' && c != '>' && c != '&') { sb_c(b, (char)c); i++; continue; }
      switch (c) {
        case '"': sb_put(b, "\\\"", 2); break; case '\\': sb_put(b, "\\\\", 2); break; case '\b': sb_put(b, "\\b", 2); break; case '\f': sb_put(b, "\\f", 2); break;
        case '\n': sb_put(b, "\\n", 2); break; case '\r': sb_put(b, "\\r", 2); break; case '\t': sb_put(b, "\\t", 2); break;
        default: sb_put(b, "\\u00", 4); sb_c(b, HEXC[c >> 4]); sb_c(b, HEXC[c & 15]);
      }
      i++; continue;
    }
    size_t need = 0; unsigned cp = 0; unsigned char lo = 0x80, hi = 0xBF;
    if (c >= 0xC2 && c <= 0xDF) { need = 1; cp = c & 0x1F; }
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; cp = c & 0x0F; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; cp = c & 0x07; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
    int ok = need > 0 && i + need < n;
    if (ok) for (size_t k = 1; k <= need; k++) {
      unsigned char d = (unsigned char)s[i + k], l = k == 1 ? lo : 0x80, h = k == 1 ? hi : 0xBF;
      if (d < l || d > h) { ok = 0; break; }
      cp = (cp << 6) | (d & 0x3F);
    }
    if (!ok) { sb_put(b, "\\ufffd", 6); i++; continue; }
    if (cp == 0x2028 || cp == 0x2029) { sb_put(b, cp == 0x2028 ? "\\u2028" : "\\u2029", 6); i += need + 1; continue; }
    sb_put(b, s + i, need + 1); i += need + 1;
  }
  sb_c(b, '"');
}

typedef struct { const char *k; size_t kl; int idx; } keyref;
static int keyref_cmp(const void *a, const void *b) {
  const keyref *x = (const keyref *)a, *y = (const keyref *)b;
  size_t m = x->kl < y->kl ? x->kl : y->kl;
  int c = memcmp(x->k, y->k, m);
  if (c) return c;
  if (x->kl != y->kl) return x->kl < y->kl ? -1 : 1;
  return x->idx - y->idx;
}

/* json.Marshal(wrapIntoEmptyInterface(v, useNumbers)) */
static void go_marshal_any(sb *b, const fj_value *v, int use_numbers) {
  switch (v->t) {
    case FJ_NULL: sb_put(b, "null", 4); break;
    case FJ_TRUE: sb_put(b, "true", 4); break;
    case FJ_FALSE: sb_put(b, "false", 5); break;
    case FJ_NUM:
      if (use_numbers) sb_put(b, v->s, v->n);  /* json.Number: written as is (validity is checked by the encoder: odd tokens are unrestated) */
      else {
        double f = ora_fastfloat_parse_best_effort(v->s, v->n);
        if (isnan(f) || isinf(f)) { g_marshal_unsupported = 1; break; }  /* json: unsupported value */
        json_float(b, f);
      }
      break;
    case FJ_STR: { size_t n; char *s = fj_unescape(v->s, v->n, &n); go_json_string(b, s, n); free(s); break; }
    case FJ_ARR:
      sb_c(b, '[');
      for (int i = 0; i < v->nkids; i++) { if (i) sb_c(b, ','); go_marshal_any(b, v->kids[i], use_numbers); }
      sb_c(b, ']');
      break;
    case FJ_OBJ: {
      /* map[string]interface{}: last duplicate wins, keys sorted */
      int n = v->nkids;
      keyref *ks = (keyref *)calloc((size_t)(n ? n : 1), sizeof(keyref));
      char **owned = (char **)calloc((size_t)(n ? n : 1), sizeof(char *));
      for (int i = 0; i < n; i++) { size_t kl; owned[i] = fj_unescape(v->keys[i], v->klen[i], &kl); ks[i].k = owned[i]; ks[i].kl = kl; ks[i].idx = i; }
      qsort(ks, (size_t)n, sizeof(keyref), keyref_cmp);
      sb_c(b, '{');
      int first = 1;
      for (int i = 0; i < n; i++) {
        if (i + 1 < n && ks[i + 1].kl == ks[i].kl && !memcmp(ks[i + 1].k, ks[i].k, ks[i].kl)) continue;  /* an earlier duplicate */
        if (!first) sb_c(b, ',');
        first = 0;
        go_json_string(b, ks[i].k, ks[i].kl);
        sb_c(b, ':');
        go_marshal_any(b, v->kids[ks[i].idx], use_numbers);
      }
      sb_c(b, '}');
      for (int i = 0; i < n; i++) free(owned[i]);
      free(owned); free(ks);
      break;
    }
  }
}

static ora_value val_str(int kind, const char *s, size_t n) {
  ora_value v; memset(&v, 0, sizeof v);
  v.kind = kind; v.slen = n; v.s = (char *)malloc(n + 1); if (n) memcpy(v.s, s, n); v.s[n] = 0;
  return v;
}
static ora_value val_nil(void) { ora_value v; memset(&v, 0, sizeof v); return v; }

/* Unmarshal (json) :672-731 — the Go value stored under one key, by the column type colTypeMap[key] (0 = unknown key) */
static ora_value unmarshal_value(const fj_value *v, int dtype, const tfgpu_json_options *o, int *unrestated) {
  ora_value r = val_nil();
  if (v->t == FJ_NULL) return r;
  if (v->t == FJ_STR) {
    size_t n; char *s = fj_unescape(v->s, v->n, &n);
    if (o->unescape_string_values && o->format != TFGPU_JFMT_TSKV) *unrestated = 1;  /* tryToUnescapeJSON: encoding/json string decoding, not restated (tskv: already applied) */
    r = val_str(OV_STRING, s, n); free(s);
    return r;
  }
  switch (dtype) {
    case TFGPU_T_UTF8: case TFGPU_T_BYTES: {  /* v.String() */
      sb b = {0}; g_marshal_unsupported = 0;
      fj_marshal(&b, v, 1);
      if (g_marshal_unsupported) *unrestated = 1;
      r = val_str(OV_STRING, b.p ? b.p : "", b.n); free(b.p);
      return r;
    }
    case TFGPU_T_FLOAT64: r.kind = OV_F64; r.v.f64 = v->t == FJ_NUM ? ora_fastfloat_parse_best_effort(v->s, v->n) : 0; return r;
    case TFGPU_T_BOOLEAN: r.kind = OV_BOOL; r.v.b = v->t == FJ_TRUE; return r;
    case TFGPU_T_INT8: case TFGPU_T_INT16: case TFGPU_T_INT32: case TFGPU_T_INT64: {
      int64_t n = v->t == FJ_NUM ? ora_fastfloat_parse_int64_best_effort(v->s, v->n) : 0;
      switch (dtype) {
        case TFGPU_T_INT8: r.kind = OV_I8; r.v.i = (int8_t)n; break;
        case TFGPU_T_INT16: r.kind = OV_I16; r.v.i = (int16_t)n; break;
        case TFGPU_T_INT32: r.kind = OV_I32; r.v.i = (int32_t)n; break;
        default: r.kind = OV_I64; r.v.i = n;
      }
      return r;
    }
    case TFGPU_T_UINT8: case TFGPU_T_UINT16: case TFGPU_T_UINT32: case TFGPU_T_UINT64: {
      uint64_t n = v->t == FJ_NUM ? ora_fastfloat_parse_uint64_best_effort(v->s, v->n) : 0;
      switch (dtype) {
        case TFGPU_T_UINT8: r.kind = OV_U8; r.v.u = (uint8_t)n; break;
        case TFGPU_T_UINT16: r.kind = OV_U16; r.v.u = (uint16_t)n; break;
        case TFGPU_T_UINT32: r.kind = OV_U32; r.v.u = (uint32_t)n; break;
        default: r.kind = OV_U64; r.v.u = n;
      }
      return r;
    }
    default:  /* wrapIntoEmptyInterface */
      switch (v->t) {
        case FJ_TRUE: case FJ_FALSE: r.kind = OV_BOOL; r.v.b = v->t == FJ_TRUE; return r;
        case FJ_NUM:
          if (o->use_numbers_in_any) return val_str(OV_JSONNUM, v->s, v->n);
          r.kind = OV_F64; r.v.f64 = ora_fastfloat_parse_best_effort(v->s, v->n); return r;
        default: {  /* map / slice */
          sb b = {0}; g_marshal_unsupported = 0;
          go_marshal_any(&b, v, o->use_numbers_in_any);
          /* a datetime column never marshals the container: extractTimeValue rejects any map / slice (:818-886) */
          if (g_marshal_unsupported && dtype != TFGPU_T_DATETIME) *unrestated = 1;
          r = val_str(OV_JSON, b.p ? b.p : "", b.n); free(b.p);
          return r;
        }
      }
  }
}

/* encoding/base64 StdEncoding.DecodeString: '\r' and '\n' are ignored, padding is mandatory and strict,
 * trailing garbage is an error, non-zero trailing bits are accepted (Strict mode is off) */
static char *b64_std_decode(const char *s, size_t n, size_t *olen) {
  char *out = (char *)malloc(n / 4 * 3 + 4);
  size_t w = 0; unsigned acc = 0; int nb = 0; int pad = 0;
  for (size_t i = 0; i < n; i++) {
    unsigned char c = (unsigned char)s[i];
    if (c == '\r' || c == '\n') continue;
    int d;
    if (c >= 'A' && c <= 'Z') d = c - 'A'; else if (c >= 'a' && c <= 'z') d = c - 'a' + 26; else if (c >= '0' && c <= '9') d = c - '0' + 52;
    else if (c == '+') d = 62; else if (c == '/') d = 63;
    else if (c == '=') { pad++; if (nb < 2 || nb + pad > 4) { free(out); return NULL; } continue; }
    else { free(out); return NULL; }
    if (pad) { free(out); return NULL; }
    acc = acc << 6 | (unsigned)d; nb++;
    if (nb == 4) { out[w++] = (char)(acc >> 16); out[w++] = (char)(acc >> 8); out[w++] = (char)acc; acc = 0; nb = 0; }
  }
  if (nb == 1 || (nb && nb + pad != 4)) { free(out); return NULL; }
  if (nb == 2) out[w++] = (char)(acc >> 4);
  else if (nb == 3) { out[w++] = (char)(acc >> 10); out[w++] = (char)(acc >> 2); }
  out[w] = 0; *olen = w;
  return out;
}

/* ParseVal :888-1123.  Returns 0 ok (value replaced in place), 1 error, 2 not restated. */
static int parse_val(ora_value *v, int dtype, const tfgpu_json_options *o) {
  if (dtype == TFGPU_T_DATETIME) {  /* strings.ToLower(typ) == "datetime" → extractTimeValue */
    switch (v->kind) {
      case OV_NIL: return 0;
      case OV_STRING: return 2;  /* dateparse (github.com/araddon/dateparse, unpinned) / TimeField.Format */
      case OV_JSONNUM: { int64_t n; if (ora_parse_int(v->s, v->slen, 10, 64, &n)) return 1; ora_value_free(v); v->kind = OV_TIME; v->v.t.sec = n; v->v.t.nsec = 0; return 0; }
      case OV_F64: {
        double a = fabs(v->v.f64);
        int64_t n = (a >= 9223372036854775808.0 || isnan(a)) ? INT64_MIN : (int64_t)a;  /* amd64 CVTTSD2SQ */
        v->kind = OV_TIME; v->v.t.sec = n; v->v.t.nsec = 0; return 0;
      }
      default: return 1;  /* "unable extract timestamp" */
    }
  }
  if (v->kind == OV_F64) {
    if (dtype == TFGPU_T_UTF8 || dtype == TFGPU_T_BYTES) return 2;  /* fmt.Sprintf("%v") — unreachable from JSON */
    return 0;  /* int targets are unreachable from the json format (Unmarshal already typed them) */
  }
  if (v->kind == OV_U64) return 0;
  if (v->kind == OV_JSONNUM) {
    switch (dtype) {
      case TFGPU_T_ANY: return 0;
      default: { double f = ora_fastfloat_parse_best_effort(v->s, v->slen); /* fastfloat.Parse errors are unrestated */ (void)f; return 2; }
    }
  }
  if (v->kind == OV_STRING) {
    const char *s = v->s; size_t n = v->slen;
    switch (dtype) {
      case TFGPU_T_FLOAT64: { double f; if (ora_parse_float(s, n, 64, &f)) return 1; ora_value_free(v); v->kind = OV_F64; v->v.f64 = f; return 0; }
      case TFGPU_T_BOOLEAN: { int b; if (ora_parse_bool(s, n, &b)) return 1; ora_value_free(v); v->kind = OV_BOOL; v->v.b = b; return 0; }
      case TFGPU_T_INT8: case TFGPU_T_INT16: case TFGPU_T_INT32: case TFGPU_T_INT64: {
        int bits = dtype == TFGPU_T_INT8 ? 8 : dtype == TFGPU_T_INT16 ? 16 : dtype == TFGPU_T_INT32 ? 32 : 64;
        int64_t i; if (ora_parse_int(s, n, 0, bits, &i)) return 1;
        ora_value_free(v); v->kind = bits == 8 ? OV_I8 : bits == 16 ? OV_I16 : bits == 32 ? OV_I32 : OV_I64; v->v.i = i; return 0;
      }
      case TFGPU_T_UINT8: case TFGPU_T_UINT16: case TFGPU_T_UINT32: case TFGPU_T_UINT64: {
        int bits = dtype == TFGPU_T_UINT8 ? 8 : dtype == TFGPU_T_UINT16 ? 16 : dtype == TFGPU_T_UINT32 ? 32 : 64;
        uint64_t u; if (ora_parse_uint(s, n, 0, bits, &u)) return 1;
        ora_value_free(v); v->kind = bits == 8 ? OV_U8 : bits == 16 ? OV_U16 : bits == 32 ? OV_U32 : OV_U64; v->v.u = u; return 0;
      }
      case TFGPU_T_BYTES: {
        if (!o->unpack_bytes_base64) return 0;
        size_t dn; char *d = b64_std_decode(s, n, &dn);  /* base64.StdEncoding.DecodeString → []byte */
        if (!d) return 1;
        ora_value_free(v); v->kind = OV_BYTES; v->s = d; v->slen = dn; return 0;
      }
      case TFGPU_T_ANY: {
        /* strings.ReplaceAll(vv, `\\\\`, `\\`) then json.Unmarshal into a map: a map only if the text is a JSON object */
        size_t i = 0;
        while (i < n && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r' || s[i] == '\n')) i++;
        if (i < n && s[i] == '{') return 2;
        return 0;  /* return vv — the ReplaceAll result */
      }
      default: return 0;
    }
  }
  /* ints / bools / composites: the trailing switch only touches int64/uint64 under timestamp/interval */
  if (dtype == TFGPU_T_TIMESTAMP && (v->kind == OV_I64 || v->kind == OV_U64)) return 2;
  if (dtype == TFGPU_T_INTERVAL && v->kind == OV_I64) return 2;
  return 0;
}

/* ------------------------------------------------------------------ */
/* the parser                                                          */
/* ------------------------------------------------------------------ */
typedef struct ora_json_lines {
  int64_t nlines;     /* non-empty lines over the whole batch */
  int32_t *status;    /* ORA_JL_* */
  int32_t *code;      /* tfgpu_rowerr for ORA_JL_UNPARSED */
  int32_t *column;    /* column the error refers to, or -1 */
  int32_t *msg;       /* message index */
  int32_t *idx;       /* 1-based index inside the message */
  int64_t *row;       /* index in the returned batch for ORA_JL_ROW, else -1 */
} ora_json_lines;

void ora_json_lines_free(ora_json_lines *l) {
  if (!l) return;
  free(l->status); free(l->code); free(l->column); free(l->msg); free(l->idx); free(l->row); free(l);
}

static char *dup_cstr(const char *s) { size_t n = strlen(s); char *r = (char *)malloc(n + 1); memcpy(r, s, n + 1); return r; }

static int has_name(const tfgpu_colschema *cols, int n, const char *name) {
  for (int i = 0; i < n; i++) if (!strcmp(cols[i].name, name)) return 1;
  return 0;
}
/* dedupColumnName :89-96 */
static char *dedup_name(const char *expected, const tfgpu_colschema *cols, int n) {
  char *cur = dup_cstr(expected);
  while (has_name(cols, n, cur)) {
    char *nx = (char *)malloc(strlen(cur) + 16);
    sprintf(nx, "_delivery_%s", cur);
    free(cur); cur = nx;
  }
  return cur;
}

/* addAuxFields :112-154 → malloc'd schema (free with ora_tschema_free) */
tfgpu_schema *ora_json_result_schema(const tfgpu_json_options *o, const tfgpu_schema *fields) {
  int cap = fields->ncols + 8;
  tfgpu_schema *s = (tfgpu_schema *)calloc(1, sizeof *s);
  s->cols = (tfgpu_colschema *)calloc((size_t)cap, sizeof(tfgpu_colschema));
  for (int i = 0; i < fields->ncols; i++) {
    s->cols[i].name = dup_cstr(fields->cols[i].name);
    s->cols[i].dtype = fields->cols[i].dtype;
    s->cols[i].flags = fields->cols[i].flags;
    s->cols[i].path = dup_cstr(fields->cols[i].path ? fields->cols[i].path : "");
    s->cols[i].original_type = dup_cstr(fields->cols[i].original_type ? fields->cols[i].original_type : "");
  }
  s->ncols = fields->ncols;
#define ADD(nm, dt, fl) do { char *n_ = dedup_name(nm, s->cols, s->ncols); s->cols[s->ncols].name = n_; s->cols[s->ncols].dtype = dt; s->cols[s->ncols].flags = fl; \
    s->cols[s->ncols].path = dup_cstr(""); s->cols[s->ncols].original_type = dup_cstr(""); s->ncols++; } while (0)
  if (o->add_rest) ADD("_rest", TFGPU_T_ANY, 0);
  if (o->add_dedupe_keys) {
    /* :127-136: with MarkDedupeKeysAsSystem the dedupe keys stop being keys once a user field is a key */
    int skip = 0;
    if (o->mark_dedupe_keys_as_system) for (int i = 0; i < fields->ncols; i++) if (fields->cols[i].flags & TFGPU_COL_KEY) skip = 1;
    uint32_t kf = skip ? 0 : (TFGPU_COL_KEY | TFGPU_COL_REQUIRED);  /* newColSchema: a key is also Required (:98-110) */
    ADD("_timestamp", TFGPU_T_TIMESTAMP, kf); ADD("_partition", TFGPU_T_BYTES, kf); ADD("_offset", TFGPU_T_UINT64, kf); ADD("_idx", TFGPU_T_UINT32, kf);
  }
#undef ADD
  return s;
}

static const char *col_path(const tfgpu_colschema *c) { return (c->path && c->path[0]) ? c->path : c->name; }

/* ---- GenericParser.Unmarshal, Format "tskv" (:732-746) + tryToUnescapeTSKV (:643-670) ----
 * strings.Split(line, "\t"), SplitN(field, "=", 2); a field without '=' is skipped; every value is a Go string.  The map
 * is handed on as a fastjson-shaped object whose keys and string values are JSON-escaped copies (", \ and control bytes),
 * so that the unescape every consumer below applies gives the raw bytes back.  `buf` owns the copies for the line. */
static size_t tskv_escape(char *dst, const char *s, size_t n) {
  static const char H[] = "0123456789abcdef";
  size_t w = 0;
  for (size_t i = 0; i < n; i++) {
    unsigned char c = (unsigned char)s[i];
    if (c == '"' || c == '\\') { dst[w++] = '\\'; dst[w++] = (char)c; }
    else if (c < 0x20) { dst[w++] = '\\'; dst[w++] = 'u'; dst[w++] = '0'; dst[w++] = '0'; dst[w++] = H[c >> 4]; dst[w++] = H[c & 15]; }
    else dst[w++] = (char)c;
  }
  return w;
}
static size_t tskv_unescape(char *dst, const char *s, size_t n) {  /* the input itself when an escape is broken */
  size_t w = 0;
  for (size_t i = 0; i < n; i++) {
    if (s[i] != '\\') { dst[w++] = s[i]; continue; }
    if (i == n - 1) { memcpy(dst, s, n); return n; }
    char c = s[i + 1], out;
    if (c == '\\') out = '\\'; else if (c == 'n') out = '\n'; else if (c == 'r') out = '\r'; else if (c == 't') out = '\t'; else if (c == '=') out = '=';
    else { memcpy(dst, s, n); return n; }
    dst[w++] = out; i++;
  }
  return w;
}
static fj_value *tskv_object(const char *line, size_t ln, int unescape, char **buf, size_t *cap) {
  if (*cap < ln * 7 + 16) { *cap = ln * 7 + 16; *buf = (char *)realloc(*buf, *cap); }
  char *w = *buf, *tmp = *buf + ln * 6 + 8;  /* tmp: ln bytes for the unescaped value */
  fj_value *root = fj_new(FJ_OBJ);
  size_t a = 0;
  while (a <= ln) {
    const char *t = (const char *)memchr(line + a, '\t', ln - a);
    const size_t z = t ? (size_t)(t - line) : ln;
    const char *eq = (const char *)memchr(line + a, '=', z - a);
    if (eq) {
      const size_t ke = (size_t)(eq - line);
      const char *k = w; size_t kl = tskv_escape(w, line + a, ke - a); w += kl;
      const char *vs = line + ke + 1; size_t vn = z - ke - 1;
      if (unescape) { vn = tskv_unescape(tmp, vs, vn); vs = tmp; }
      fj_value *v = fj_new(FJ_STR);
      v->s = w; v->n = tskv_escape(w, vs, vn); w += v->n;
      fj_push(root, v, k, kl);
    }
    if (!t) break;
    a = z + 1;
  }
  return root;
}

ora_batch *ora_json_parse(const tfgpu_json_options *o, const tfgpu_schema *fields, const void *bytes, uint64_t len,
                          const tfgpu_messages *msgs, ora_json_lines **lines_out) {
  ora_batch *out = ora_batch_new();
  ora_json_lines *L = (ora_json_lines *)calloc(1, sizeof *L);
  int64_t lcap = 0;
  tfgpu_schema *rs = ora_json_result_schema(o, fields);
  ora_schema *sch = ora_schema_from(rs);
  ora_names *nm = (ora_names *)calloc(1, sizeof *nm);
  nm->refs = 1; nm->n = rs->ncols; nm->names = (char **)calloc((size_t)rs->ncols + 1, sizeof(char *));
  for (int i = 0; i < rs->ncols; i++) nm->names[i] = dup_cstr(rs->cols[i].name);
  /* GenericParser.name = strings.ReplaceAll(Topic, "/", "_") :1236; tableName() :569-574 replaces '/' and '@' */
  char *tname = dup_cstr(o->topic ? o->topic : "");
  for (char *c = tname; *c; c++) if (*c == '/' || *c == '@') *c = '_';
  const char *part = o->partition ? o->partition : "";
  const int nraw = fields->ncols;
  const int rest_idx = o->add_rest ? nraw : -1;
  const int dd0 = o->add_dedupe_keys ? rs->ncols - 4 : -1;

  uint64_t one_start[2] = {0, len}; uint64_t one_off[1] = {0}; int64_t one_wt[1] = {0};
  tfgpu_messages single = {1, one_start, one_off, one_wt};
  if (!msgs) msgs = &single;

  const char *data = (const char *)bytes;
  const int tskv = o->format == TFGPU_JFMT_TSKV;
  char *tskv_buf = NULL; size_t tskv_cap = 0;
  for (int64_t m = 0; m < msgs->nmsg; m++) {
    const char *p = data + msgs->start[m], *e = data + msgs->start[m + 1];
    int idx = 0;
    while (p < e) {
      /* bufio.ScanLines: up to '\n' (dropped), a trailing '\r' dropped; the last line needs no '\n' */
      const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
      const char *le = nl ? nl : e;
      const char *next = nl ? nl + 1 : e;
      size_t ln = (size_t)(le - p);
      if (ln && p[ln - 1] == '\r') ln--;
      const char *line = p;
      p = next;
      if (ln == 0) continue;
      idx++;
      if (L->nlines == lcap) {
        lcap = lcap ? lcap * 2 : 64;
        L->status = (int32_t *)realloc(L->status, (size_t)lcap * 4); L->code = (int32_t *)realloc(L->code, (size_t)lcap * 4);
        L->column = (int32_t *)realloc(L->column, (size_t)lcap * 4); L->msg = (int32_t *)realloc(L->msg, (size_t)lcap * 4);
        L->idx = (int32_t *)realloc(L->idx, (size_t)lcap * 4); L->row = (int64_t *)realloc(L->row, (size_t)lcap * 8);
      }
      const int64_t ord = L->nlines++;
      L->status[ord] = ORA_JL_ROW; L->code[ord] = 0; L->column[ord] = -1; L->msg[ord] = (int32_t)m; L->idx[ord] = idx; L->row[ord] = -1;

      const char *perr = NULL;
      fj_value *root = tskv ? tskv_object(line, ln, o->unescape_string_values, &tskv_buf, &tskv_cap) : fj_parse(line, ln, &perr);
      if (!root) { L->status[ord] = ORA_JL_UNPARSED; L->code[ord] = TFGPU_ROW_JSON_SYNTAX; ora_batch_add_error(out, ord, TFGPU_ROW_JSON_SYNTAX, perr); continue; }
      if (root->t != FJ_OBJ || root->nkids == 0) { L->status[ord] = ORA_JL_SKIPPED; fj_free(root); continue; }

      /* the map: last duplicate of a key wins */
      int unrestated = 0;
      ora_value *vals = (ora_value *)calloc((size_t)rs->ncols + 1, sizeof(ora_value));
      int status = ORA_JL_ROW, code = 0, ecol = -1;
      int rest_nonempty = 0;
      /* unknown keys → _rest */
      sb restb = {0};
      if (o->add_rest) {
        /* collect unknown keys (p.known: ColumnName and ColPath of the raw fields) with their Unmarshal'ed values */
        int n = root->nkids;
        keyref *ks = (keyref *)calloc((size_t)n, sizeof(keyref));
        char **owned = (char **)calloc((size_t)n, sizeof(char *));
        int nk = 0;
        for (int i = 0; i < n; i++) {
          size_t kl; char *k = fj_unescape(root->keys[i], root->klen[i], &kl);
          int known = 0;
          for (int c = 0; c < nraw && !known; c++) {
            if (strlen(fields->cols[c].name) == kl && !memcmp(fields->cols[c].name, k, kl)) known = 1;
            if (!o->ignore_column_paths) { const char *cp = col_path(&fields->cols[c]); if (strlen(cp) == kl && !memcmp(cp, k, kl)) known = 1; }
          }
          /* colTypeMap also types the aux columns' names (:1218-1225): such a key would be Unmarshal'ed with that type */
          if (!strcmp(k, "_rest") || !strcmp(k, "_timestamp") || !strcmp(k, "_partition") || !strcmp(k, "_offset") || !strcmp(k, "_idx")) unrestated = 1;
          if (known) { free(k); continue; }
          owned[nk] = k; ks[nk].k = k; ks[nk].kl = kl; ks[nk].idx = i; nk++;
        }
        qsort(ks, (size_t)nk, sizeof(keyref), keyref_cmp);
        sb_c(&restb, '{');
        int first = 1;
        g_marshal_unsupported = 0;
        for (int i = 0; i < nk; i++) {
          if (i + 1 < nk && ks[i + 1].kl == ks[i].kl && !memcmp(ks[i + 1].k, ks[i].k, ks[i].kl)) continue;
          if (!first) sb_c(&restb, ',');
          first = 0; rest_nonempty = 1;
          go_json_string(&restb, ks[i].k, ks[i].kl);
          sb_c(&restb, ':');
          const fj_value *kv = root->kids[ks[i].idx];
          if (kv->t == FJ_STR) { size_t sn; char *s = fj_unescape(kv->s, kv->n, &sn); go_json_string(&restb, s, sn); free(s); if (o->unescape_string_values && !tskv) unrestated = 1; }
          else go_marshal_any(&restb, kv, o->use_numbers_in_any);
        }
        sb_c(&restb, '}');
        if (g_marshal_unsupported) unrestated = 1;
        for (int i = 0; i < nk; i++) free(owned[i]);
        free(owned); free(ks);
      }

      for (int c = 0; c < nraw && status == ORA_JL_ROW; c++) {
        const tfgpu_colschema *col = &fields->cols[c];
        const char *cp = col_path(col);
        /* Unmarshal types the key by colTypeMap[key]: ColPath (ColumnName with IgnoreColumnPaths) over the final
         * schema, last column wins (:1226-1233).  unmarshal_value below types it with THIS column's DataType, which
         * is the same thing only when the two agree; otherwise the cell is not restated. */
        {
          int kt = 0;
          for (int q = 0; q < rs->ncols; q++) {
            const char *kq = (o->ignore_column_paths || q >= nraw) ? rs->cols[q].name : col_path(&fields->cols[q]);
            if (!strcmp(kq, cp)) kt = rs->cols[q].dtype;
          }
          if (kt != col->dtype) { unrestated = 1; continue; }
        }
        if (strchr(cp, '.') || strchr(cp, '/')) {
          /* IsNestedKey (col_schema.go:95-97) → lookupComplex (generic_parser.go:323-347, lookup.go:10-38): strings.Split(path, "."),
           * or by "/" when that gives one part; the first name picks the top-level value, which must be a Go string holding JSON */
          char *pc = dup_cstr(cp);
          const char *segs[32]; int ns = 0;
          const char sepc = strchr(cp, '.') ? '.' : '/';
          for (char *q = pc, *start = pc;; q++) if (*q == sepc || !*q) { const int last = !*q; *q = 0; if (ns < 32) segs[ns++] = start; start = q + 1; if (last) break; }
          const fj_value *top = NULL;
          const size_t l0 = strlen(segs[0]);
          for (int i = 0; i < root->nkids; i++) {
            size_t kl; char *k = fj_unescape(root->keys[i], root->klen[i], &kl);
            if (kl == l0 && !memcmp(k, segs[0], kl)) top = root->kids[i];
            free(k);
          }
          const int is_key = (col->flags & TFGPU_COL_KEY) != 0, is_req = (col->flags & TFGPU_COL_REQUIRED) != 0;
          int rcl = ORA_LOOKUP_ERROR; char *sv = NULL; size_t svn = 0;
          if (ns >= 32 || (top && top->t != FJ_STR && top->t != FJ_NULL) || (top && top->t == FJ_STR && o->unescape_string_values && !tskv)) { unrestated = 1; free(pc); continue; }  /* a map at the top (Format json): not restated */
          if (top && top->t == FJ_STR) { size_t tn; char *ts = fj_unescape(top->s, top->n, &tn); rcl = ora_lookup_complex(ts, tn, segs, ns, &sv, &svn); free(ts); }
          /* (top absent: "unable to get field"; top nil: "unexpected value type: <nil>" — errors both) */
          free(pc);
          if (rcl == ORA_LOOKUP_OTHER) { unrestated = 1; continue; }
          if (rcl == ORA_LOOKUP_ERROR || rcl == ORA_LOOKUP_NIL) {
            if (!o->null_keys_allowed && (is_key || is_req)) { status = ORA_JL_UNPARSED; code = TFGPU_ROW_NIL_KEY; ecol = c; }  /* "lookupComplex error" / "lookupComplex nil … required" */
            continue;
          }
          ora_value v = val_str(OV_STRING, sv, svn); free(sv);
          const int rcp = parse_val(&v, col->dtype, o);
          if (rcp == 2) { unrestated = 1; ora_value_free(&v); continue; }
          if (rcp == 1) { ora_value_free(&v); if (!o->null_keys_allowed) { status = ORA_JL_UNPARSED; code = TFGPU_ROW_PARSE_VAL; ecol = c; } continue; }  /* ParseVal error: _unparsed whatever the column's flags */
          vals[c] = v;
          continue;
        }
        size_t cpl = strlen(cp);
        const fj_value *found = NULL;
        for (int i = 0; i < root->nkids; i++) {
          size_t kl; char *k = fj_unescape(root->keys[i], root->klen[i], &kl);
          if (kl == cpl && !memcmp(k, cp, kl)) found = root->kids[i];
          free(k);
        }
        /* colTypeMap is keyed by ColPath (or ColumnName when IgnoreColumnPaths) over the FINAL schema */
        ora_value v = found ? unmarshal_value(found, col->dtype, o, &unrestated) : val_nil();
        int rc = parse_val(&v, col->dtype, o);
        const int is_key = (col->flags & TFGPU_COL_KEY) != 0, is_req = (col->flags & TFGPU_COL_REQUIRED) != 0;
        if (rc == 2) { unrestated = 1; ora_value_free(&v); continue; }
        if (rc == 1) {
          ora_value_free(&v);
          if ((!o->null_keys_allowed && is_key) || is_req) { status = ORA_JL_UNPARSED; code = TFGPU_ROW_PARSE_VAL; ecol = c; }
          continue;
        }
        if (v.kind == OV_NIL && (is_key || is_req) && !o->null_keys_allowed) { status = ORA_JL_UNPARSED; code = TFGPU_ROW_NIL_KEY; ecol = c; continue; }
        vals[c] = v;
      }
      if (status == ORA_JL_ROW && unrestated) status = ORA_JL_UNRESTATED;  /* the row is still emitted: unrestated cells are nil */
      L->status[ord] = status; L->code[ord] = code; L->column[ord] = ecol;
      if (status != ORA_JL_ROW && status != ORA_JL_UNRESTATED) {
        if (status == ORA_JL_UNPARSED) ora_batch_add_error(out, ord, code, "");
        for (int c = 0; c < rs->ncols; c++) ora_value_free(&vals[c]);
        free(vals); free(restb.p); fj_free(root);
        continue;
      }
      (void)rest_nonempty;
      if (rest_idx >= 0) vals[rest_idx] = val_str(OV_JSON, restb.p, restb.n);
      if (dd0 >= 0) {
        /* extractTimestamp with TimeField == nil → msg.WriteTime */
        int64_t wt = msgs->write_time_ns[m];
        int64_t sec = wt / 1000000000, ns = wt % 1000000000;
        if (ns < 0) { ns += 1000000000; sec--; }
        vals[dd0].kind = OV_TIME; vals[dd0].v.t.sec = sec; vals[dd0].v.t.nsec = (int32_t)ns;
        vals[dd0 + 1] = val_str(OV_STRING, part, strlen(part));  /* changeItem.PartID: a Go string */
        vals[dd0 + 2].kind = OV_U64; vals[dd0 + 2].v.u = msgs->offset[m];
        vals[dd0 + 3].kind = OV_U32; vals[dd0 + 3].v.u = (uint32_t)idx;
      }
      free(restb.p);
      ora_item *it = ora_batch_push(out);
      it->kind = TFGPU_K_INSERT;
      it->ns = dup_cstr(""); it->table = dup_cstr(tname); it->part_id = dup_cstr(part);
      it->names = nm; nm->refs++;
      it->schema = sch; sch->refs++;
      it->nvalues = rs->ncols; it->values = vals;
      it->src_row = ord;
      L->row[ord] = out->n - 1;
      fj_free(root);
    }
  }
  free(tname); free(tskv_buf);
  /* drop our own references */
  if (--nm->refs == 0) { for (int i = 0; i < nm->n; i++) free(nm->names[i]); free(nm->names); free(nm); }
  ora_schema_unref(sch);
  ora_tschema_free(rs);
  if (lines_out) *lines_out = L; else ora_json_lines_free(L);
  return out;
}

/* row-wise read access for the tests (columns of one batch may hold different Go types per row) */
int ora_batch_value(const ora_batch *b, int64_t row, int col, int *kind, int64_t *i64, double *f64, const char **s, size_t *slen, int32_t *nsec) {
  if (row < 0 || row >= b->n || col < 0 || col >= b->items[row].nvalues) return -1;
  const ora_value *v = &b->items[row].values[col];
  *kind = v->kind; *i64 = 0; *f64 = 0; *s = NULL; *slen = 0; *nsec = 0;
  switch (v->kind) {
    case OV_I8: case OV_I16: case OV_I32: case OV_I64: case OV_DURATION: *i64 = v->v.i; break;
    case OV_U8: case OV_U16: case OV_U32: case OV_U64: *i64 = (int64_t)v->v.u; break;
    case OV_F32: *f64 = v->v.f32; break;
    case OV_F64: *f64 = v->v.f64; break;
    case OV_BOOL: *i64 = v->v.b; break;
    case OV_TIME: *i64 = v->v.t.sec; *nsec = v->v.t.nsec; break;
    case OV_STRING: case OV_BYTES: case OV_JSONNUM: case OV_JSON: *s = v->s; *slen = v->slen; break;
    default: break;
  }
  return 0;
}
