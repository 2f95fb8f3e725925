/*
 * oracle/ora_transformers.c — CPU restatement of the row transformers on the
 * hot path (SURVEY.md §8a rows a4–a13).  TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 * Each Apply keeps the reference's per-row behaviour, including the per-row
 * allocations that dominate its run time (a fresh TableSchema per row in
 * mask/to_string/to_datetime, hmac.New per value, AsMap per row in sharder).
 */
#define _GNU_SOURCE
#include <math.h>
#include <regex.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include "ora.h"
#include "ora_json.h"

static char *dupn(const char *s, size_t n) { char *r = (char *)malloc(n + 1); if (n) memcpy(r, s, n); r[n] = 0; return r; }
static char *dups(const char *s) { return s ? dupn(s, strlen(s)) : NULL; }

/* ======================================================================
 * a5  SerializeToString — pkg/transformer/registry/to_string/to_string.go:145-178
 * ====================================================================== */
static size_t fmt_percent_v(char *dst, size_t cap, const ora_value *v) {
  /* fmt.Sprintf("%v", value) for the dynamic types that occur in ColumnValues */
  switch (v->kind) {
    case OV_NIL: memcpy(dst, "<nil>", 5); return 5;
    case OV_I8: case OV_I16: case OV_I32: case OV_I64: return ora_fmt_int(dst, v->v.i);
    case OV_U8: case OV_U16: case OV_U32: case OV_U64: return ora_fmt_uint(dst, v->v.u);
    case OV_F32: return ora_fmt_float(dst, (double)v->v.f32, 'g', 32);
    case OV_F64: return ora_fmt_float(dst, v->v.f64, 'g', 64);
    case OV_BOOL: if (v->v.b) { memcpy(dst, "true", 4); return 4; } memcpy(dst, "false", 5); return 5;
    case OV_TIME: return ora_fmt_time_string(dst, v->v.t.sec, v->v.t.nsec);
    case OV_DURATION: return ora_fmt_duration(dst, v->v.i);
    case OV_BYTES: {
      /* %v of []byte prints the decimal byte values: [118 97 108] */
      size_t w = 0; dst[w++] = '[';
      for (size_t i = 0; i < v->slen && w + 8 < cap; i++) { if (i) dst[w++] = ' '; w += ora_fmt_uint(dst + w, (unsigned char)v->s[i]); }
      dst[w++] = ']'; return w;
    }
    default: return 0; /* strings handled by caller */
  }
}

char *ora_serialize_to_string(const ora_value *v, int dtype, size_t *len) {
  char buf[96];
  size_t n;
  switch (dtype) {
    case TFGPU_T_BYTES: /* to_string.go:151-155 */
      if (v->kind == OV_BYTES) { *len = v->slen; return dupn(v->s, v->slen); }
      break;
    case TFGPU_T_ANY: /* :156-160 json.Marshal(value) */
      if (v->kind == OV_JSON) { *len = v->slen; return dupn(v->s, v->slen); }
      if (v->kind == OV_NIL) { *len = 4; return dups("null"); }
      if (v->kind == OV_STRING) { /* json string, HTML-escaped by json.Marshal */
        size_t cap = v->slen * 6 + 3; char *o = (char *)malloc(cap); size_t w = 0; o[w++] = '"';
        for (size_t i = 0; i < v->slen; i++) {
          unsigned char c = (unsigned char)v->s[i];
          if (c == '"' || c == '\\') { o[w++] = '\\'; o[w++] = (char)c; }
          else if (c == '\n') { o[w++] = '\\'; o[w++] = 'n'; } else if (c == '\r') { o[w++] = '\\'; o[w++] = 'r'; } else if (c == '\t') { o[w++] = '\\'; o[w++] = 't'; }
          else if (c < 0x20 || c == '<' || c == '>' || c == '&') { w += (size_t)sprintf(o + w, "\\u%04x", c); }
          else o[w++] = (char)c;
        }
        o[w++] = '"'; o[w] = 0; *len = w; return o;
      }
      if (v->kind == OV_JSONNUM) { *len = v->slen; return dupn(v->s, v->slen); }
      if (v->kind == OV_BOOL || (v->kind >= OV_I8 && v->kind <= OV_U64)) { n = fmt_percent_v(buf, sizeof buf, v); *len = n; return dupn(buf, n); }
      break;
    case TFGPU_T_DATE: /* :161-164 time.DateOnly after UTC() */
      if (v->kind == OV_TIME) { n = ora_fmt_date(buf, v->v.t.sec); *len = n; return dupn(buf, n); }
      break;
    case TFGPU_T_DATETIME: case TFGPU_T_TIMESTAMP: /* :165-168 RFC3339Nano */
      if (v->kind == OV_TIME) { n = ora_fmt_rfc3339nano(buf, v->v.t.sec, v->v.t.nsec); *len = n; return dupn(buf, n); }
      break;
  }
  /* :170 fmt.Sprintf("%v", value) */
  if (v->kind == OV_STRING || v->kind == OV_JSONNUM || v->kind == OV_JSON) { *len = v->slen; return dupn(v->s, v->slen); }
  if (v->kind == OV_BYTES) {
    size_t cap = v->slen * 4 + 3; char *o = (char *)malloc(cap);
    n = fmt_percent_v(o, cap, v); o[n] = 0; *len = n; return o;
  }
  n = fmt_percent_v(buf, sizeof buf, v);
  *len = n; return dupn(buf, n);
}

/* ======================================================================
 * filter.Filter — pkg/transformer/registry/filter/filter.go:19-74,
 * transformer_common.go:9-33 (table name variants)
 * ====================================================================== */
typedef struct { int ninc, nexc; regex_t *inc, *exc; char **inc_src, **exc_src; } re_filter;

static char *re2_to_posix(const char *p) {
  /* translate the Perl classes RE2 accepts into POSIX ERE bracket classes */
  size_t n = strlen(p); char *o = (char *)malloc(n * 16 + 1); size_t w = 0;
  for (size_t i = 0; i < n; i++) {
    if (p[i] == '\\' && i + 1 < n) {
      char c = p[i + 1];
      const char *rep = NULL;
      if (c == 'd') rep = "[0-9]"; else if (c == 'D') rep = "[^0-9]";
      else if (c == 'w') rep = "[0-9A-Za-z_]"; else if (c == 'W') rep = "[^0-9A-Za-z_]";
      else if (c == 's') rep = "[ \t\n\f\r]"; else if (c == 'S') rep = "[^ \t\n\f\r]";
      if (rep) { strcpy(o + w, rep); w += strlen(rep); i++; continue; }
      o[w++] = p[i]; o[w++] = p[++i]; continue;
    }
    if (p[i] == '(' && i + 2 < n && p[i + 1] == '?' && p[i + 2] == ':') { o[w++] = '('; i += 2; continue; }
    o[w++] = p[i];
  }
  o[w] = 0; return o;
}

static int re_filter_init(re_filter *f, const jnode *inc, const jnode *exc, char *err, size_t errcap) {
  memset(f, 0, sizeof *f);
  const jnode *lists[2] = {inc, exc};
  for (int k = 0; k < 2; k++) {
    const jnode *l = lists[k];
    int n = (l && l->type == JN_ARR) ? l->n : 0;
    regex_t *arr = (regex_t *)calloc((size_t)(n ? n : 1), sizeof(regex_t));
    char **src = (char **)calloc((size_t)(n ? n : 1), sizeof(char *));
    for (int i = 0; i < n; i++) {
      const char *pat = jn_str(l->kids[i], "");
      char *px = re2_to_posix(pat);
      int rc = regcomp(&arr[i], px, REG_EXTENDED | REG_NOSUB);
      free(px);
      if (rc) { snprintf(err, errcap, "unable to compile %s regexp: %s", k ? "exclude" : "include", pat); return 1; }
      src[i] = dups(pat);
    }
    if (k == 0) { f->ninc = n; f->inc = arr; f->inc_src = src; } else { f->nexc = n; f->exc = arr; f->exc_src = src; }
  }
  return 0;
}
static void re_filter_free(re_filter *f) {
  for (int i = 0; i < f->ninc; i++) { regfree(&f->inc[i]); free(f->inc_src[i]); }
  for (int i = 0; i < f->nexc; i++) { regfree(&f->exc[i]); free(f->exc_src[i]); }
  free(f->inc); free(f->exc); free(f->inc_src); free(f->exc_src);
}
static int re_filter_match(const re_filter *f, const char *v) { /* filter.go:27-44 */
  for (int i = 0; i < f->nexc; i++) if (regexec(&f->exc[i], v, 0, NULL, 0) == 0) return 0;
  if (f->ninc == 0) return 1;
  for (int i = 0; i < f->ninc; i++) if (regexec(&f->inc[i], v, 0, NULL, 0) == 0) return 1;
  return 0;
}
static int re_filter_empty(const re_filter *f) { return f->ninc == 0 && f->nexc == 0; }

static int match_any_table_variant(const re_filter *f, const char *ns, const char *name) {
  /* transformer_common.go:9-33: "ns.name" and Fqtn() = "ns"."name" with "" escaping */
  if (re_filter_empty(f)) return 1;
  size_t cap = strlen(ns) * 2 + strlen(name) * 2 + 16;
  char *a = (char *)malloc(cap), *b = (char *)malloc(cap);
  if (!*ns) snprintf(a, cap, "%s", name); else snprintf(a, cap, "%s.%s", ns, name);
  size_t w = 0;
  if (*ns) { b[w++] = '"'; for (const char *p = ns; *p; p++) { if (*p == '"') b[w++] = '"'; b[w++] = *p; } b[w++] = '"'; b[w++] = '.'; }
  if (!strcmp(name, "*")) b[w++] = '*';
  else { b[w++] = '"'; for (const char *p = name; *p; p++) { if (*p == '"') b[w++] = '"'; b[w++] = *p; } b[w++] = '"'; }
  b[w] = 0;
  int r = re_filter_match(f, a) || re_filter_match(f, b);
  free(a); free(b);
  return r;
}

static int is_system_table(const char *name) {
  /* abstract.IsSystemTable: names registered by providers' init() */
  static const char *sys[] = {"__wal", "__table_transfer_progress", "__tm_gtid_keeper", "__tm_keeper", "__consumer_keeper",
                              "__data_transfer_lsn", "__data_transfer_signal_table", "__data_transfer", "__dt_cluster_time"};
  for (size_t i = 0; i < sizeof sys / sizeof *sys; i++) if (!strcmp(sys[i], name)) return 1;
  return 0;
}

/* ======================================================================
 * a12  filter grammar — library/go/yandex/cloud/filter/grammar/grammar.go:255-313
 * ====================================================================== */
enum { OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE, OP_IN, OP_NOTIN, OP_MATCH, OP_NOTMATCH };
enum { FV_STRING, FV_TIME, FV_BOOL, FV_FLOAT, FV_INT, FV_NULL, FV_LIST };

typedef struct fvalue {
  int type;
  char *s; size_t slen;
  int64_t i; double f; int b;
  int64_t tsec; int32_t tnsec;
  int nlist; struct fvalue *list;
} fvalue;

typedef struct { char *attr; int op; fvalue val; } fterm;
typedef struct { int nterms; fterm *terms; } fexpr;

enum { TK_OP, TK_STRING, TK_DATETIME, TK_IDENT, TK_FLOAT, TK_INT, TK_PUNCT, TK_WS, TK_EOF };
typedef struct { int type; const char *p; size_t n; } ftoken;

static int isdig(char c) { return c >= '0' && c <= '9'; }
static int isalpha_(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

/* one token at s (leftmost-first over the alternatives, grammar.go:256-265) */
static int lex_one(const char *s, size_t n, ftoken *t) {
  t->p = s;
  if (n == 0) { t->type = TK_EOF; t->n = 0; return 0; }
  /* Operator: != <= >= !~ | [=<>~] */
  if (n >= 2 && ((s[0] == '!' && s[1] == '=') || (s[0] == '<' && s[1] == '=') || (s[0] == '>' && s[1] == '=') || (s[0] == '!' && s[1] == '~'))) { t->type = TK_OP; t->n = 2; return 0; }
  if (s[0] == '=' || s[0] == '<' || s[0] == '>' || s[0] == '~') { t->type = TK_OP; t->n = 1; return 0; }
  /* String: '((\\'|[^']))*' | "(\\"|[^"])*" */
  if (s[0] == '\'' || s[0] == '"') {
    char q = s[0]; size_t i = 1; long last_pair_quote = -1;
    while (i < n) {
      if (s[i] == '\\' && i + 1 < n && s[i + 1] == q) { last_pair_quote = (long)i + 1; i += 2; continue; }
      if (s[i] == q) { t->type = TK_STRING; t->n = i + 1; return 0; }
      i++;
    }
    if (last_pair_quote >= 0) { t->type = TK_STRING; t->n = (size_t)last_pair_quote + 1; return 0; } /* regex backtrack */
    return 1;
  }
  /* DateTime: \d{4}-\d{2}-\d{2}(T\d{2}:\d{2}(:\d{2}(\.\d+)?)?(Z|[+-]\d+(:\d+)?)?)? */
  if (n >= 10 && isdig(s[0]) && isdig(s[1]) && isdig(s[2]) && isdig(s[3]) && s[4] == '-' && isdig(s[5]) && isdig(s[6]) && s[7] == '-' && isdig(s[8]) && isdig(s[9])) {
    size_t i = 10;
    if (n >= i + 6 && s[i] == 'T' && isdig(s[i + 1]) && isdig(s[i + 2]) && s[i + 3] == ':' && isdig(s[i + 4]) && isdig(s[i + 5])) {
      i += 6;
      if (n >= i + 3 && s[i] == ':' && isdig(s[i + 1]) && isdig(s[i + 2])) {
        i += 3;
        if (n >= i + 2 && s[i] == '.' && isdig(s[i + 1])) { i += 2; while (i < n && isdig(s[i])) i++; }
      }
      if (i < n && s[i] == 'Z') i++;
      else if (i + 1 < n && (s[i] == '+' || s[i] == '-') && isdig(s[i + 1])) {
        i += 2; while (i < n && isdig(s[i])) i++;
        if (i + 1 < n && s[i] == ':' && isdig(s[i + 1])) { i += 2; while (i < n && isdig(s[i])) i++; }
      }
    }
    t->type = TK_DATETIME; t->n = i; return 0;
  }
  /* Ident: [a-zA-Z][a-zA-Z0-9_.]* */
  if (isalpha_(s[0])) { size_t i = 1; while (i < n && (isalpha_(s[i]) || isdig(s[i]) || s[i] == '_' || s[i] == '.')) i++; t->type = TK_IDENT; t->n = i; return 0; }
  /* Float: [-+]?\d+\.\d+ ; Int: [-+]?\d+ */
  {
    size_t i = 0;
    if (s[0] == '-' || s[0] == '+') i = 1;
    if (i < n && isdig(s[i])) {
      size_t j = i; while (j < n && isdig(s[j])) j++;
      if (j + 1 < n && s[j] == '.' && isdig(s[j + 1])) { j += 2; while (j < n && isdig(s[j])) j++; t->type = TK_FLOAT; t->n = j; return 0; }
      t->type = TK_INT; t->n = j; return 0;
    }
  }
  if (s[0] == '(' || s[0] == ')' || s[0] == ',') { t->type = TK_PUNCT; t->n = 1; return 0; }
  if (s[0] == ' ' || s[0] == '\t' || s[0] == '\n' || s[0] == '\r' || s[0] == '\f' || s[0] == '\v') {
    size_t i = 1; while (i < n && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r' || s[i] == '\f' || s[i] == '\v')) i++;
    t->type = TK_WS; t->n = i; return 0;
  }
  return 1;
}

typedef struct { ftoken *toks; int n, pos; char *err; size_t errcap; } fparser;

static void perr(fparser *p, const char *msg) { if (p->err && !p->err[0]) snprintf(p->err, p->errcap, "%s", msg); }
static ftoken *peek(fparser *p) { return &p->toks[p->pos]; }
static void skip_ws(fparser *p) { if (peek(p)->type == TK_WS) p->pos++; }
static int ident_is(const ftoken *t, const char *kw) { return t->type == TK_IDENT && strlen(kw) == t->n && strncasecmp(t->p, kw, t->n) == 0; }

static void fvalue_free(fvalue *v) { free(v->s); for (int i = 0; i < v->nlist; i++) fvalue_free(&v->list[i]); free(v->list); memset(v, 0, sizeof *v); }

/* participle.Unquote → strconv.UnquoteChar loop */
static char *go_unquote(const char *s, size_t n, size_t *olen, int *bad) {
  char q = s[0]; s++; n -= 2;
  char *o = (char *)malloc(n * 4 + 1); size_t w = 0;
  for (size_t i = 0; i < n;) {
    unsigned char c = (unsigned char)s[i];
    if (c == (unsigned char)q) { *bad = 1; break; }
    if (c != '\\') { o[w++] = (char)c; i++; continue; }
    if (i + 1 >= n) { *bad = 1; break; }
    char e = s[i + 1]; i += 2;
    switch (e) {
      case 'a': o[w++] = '\a'; break; case 'b': o[w++] = '\b'; break; case 'f': o[w++] = '\f'; break;
      case 'n': o[w++] = '\n'; break; case 'r': o[w++] = '\r'; break; case 't': o[w++] = '\t'; break;
      case 'v': o[w++] = '\v'; break; case '\\': o[w++] = '\\'; break;
      case '\'': case '"': if (e != q) { *bad = 1; } o[w++] = e; break;
      case 'x': case 'u': case 'U': {
        int nd = e == 'x' ? 2 : e == 'u' ? 4 : 8; unsigned v = 0;
        if (i + (size_t)nd > n) { *bad = 1; break; }
        for (int k = 0; k < nd; k++) { char h = s[i + (size_t)k]; int d = isdig(h) ? h - '0' : (h >= 'a' && h <= 'f') ? h - 'a' + 10 : (h >= 'A' && h <= 'F') ? h - 'A' + 10 : -1; if (d < 0) { *bad = 1; d = 0; } v = v * 16 + (unsigned)d; }
        i += (size_t)nd;
        if (e == 'x') o[w++] = (char)v;
        else { if (v < 0x80) o[w++] = (char)v; else if (v < 0x800) { o[w++] = (char)(0xC0 | v >> 6); o[w++] = (char)(0x80 | (v & 0x3F)); } else if (v < 0x10000) { o[w++] = (char)(0xE0 | v >> 12); o[w++] = (char)(0x80 | ((v >> 6) & 0x3F)); o[w++] = (char)(0x80 | (v & 0x3F)); } else { o[w++] = (char)(0xF0 | v >> 18); o[w++] = (char)(0x80 | ((v >> 12) & 0x3F)); o[w++] = (char)(0x80 | ((v >> 6) & 0x3F)); o[w++] = (char)(0x80 | (v & 0x3F)); } }
        break;
      }
      default:
        if (e >= '0' && e <= '7') { if (i + 2 > n) { *bad = 1; break; } unsigned v = (unsigned)(e - '0'); for (int k = 0; k < 2; k++) { char h = s[i + (size_t)k]; if (h < '0' || h > '7') { *bad = 1; h = '0'; } v = v * 8 + (unsigned)(h - '0'); } i += 2; if (v > 255) *bad = 1; o[w++] = (char)v; }
        else *bad = 1;
    }
    if (*bad) break;
  }
  o[w] = 0; *olen = w; return o;
}

/* grammar.go:114-144 findTimeLayout */
static void find_time_layout(const char *v, char *layout) {
  strcpy(layout, "2006-01-02");
  const char *t = strchr(v, 'T');
  if (t && t > v) {
    const char *tz = strpbrk(t, "+-Z");
    char timepart[64];
    size_t tl = tz ? (size_t)(tz - t) : strlen(t);
    if (tl >= sizeof timepart) tl = sizeof timepart - 1;
    memcpy(timepart, t, tl); timepart[tl] = 0;
    int colons = 0; for (char *c = timepart; *c; c++) colons += *c == ':';
    strcat(layout, colons == 2 ? "T15:04:05" : "T15:04");
    if (tz) strcat(layout, strchr(tz, ':') ? "Z07:00" : "Z07");
  }
}

static int parse_value(fparser *p, fvalue *out, int depth) {
  memset(out, 0, sizeof *out);
  ftoken *t = peek(p);
  switch (t->type) {
    case TK_STRING: {
      int bad = 0; out->type = FV_STRING; out->s = go_unquote(t->p, t->n, &out->slen, &bad);
      if (bad) { perr(p, "invalid string literal"); return 1; }
      p->pos++; return 0;
    }
    case TK_DATETIME: {
      char v[96], layout[64];
      size_t n = t->n < sizeof v - 1 ? t->n : sizeof v - 1; memcpy(v, t->p, n); v[n] = 0;
      find_time_layout(v, layout);
      out->type = FV_TIME;
      if (ora_time_parse(layout, v, n, &out->tsec, &out->tnsec)) { perr(p, "cannot parse datetime"); return 1; }
      p->pos++; return 0;
    }
    case TK_IDENT:
      if (ident_is(t, "TRUE") || ident_is(t, "FALSE")) { out->type = FV_BOOL; out->b = ident_is(t, "TRUE"); p->pos++; return 0; }
      if (ident_is(t, "NULL") || ident_is(t, "NIL")) { out->type = FV_NULL; p->pos++; return 0; }
      perr(p, "unexpected token"); return 1;
    case TK_FLOAT: { out->type = FV_FLOAT; if (ora_parse_float(t->p, t->n, 64, &out->f)) { perr(p, "bad float"); return 1; } p->pos++; return 0; }
    case TK_INT: { out->type = FV_INT; if (ora_parse_int(t->p, t->n, 10, 64, &out->i)) { perr(p, "value out of range"); return 1; } p->pos++; return 0; }
    case TK_PUNCT:
      if (t->p[0] == '(') {
        p->pos++; skip_ws(p);
        out->type = FV_LIST;
        for (;;) {
          fvalue item;
          if (parse_value(p, &item, depth + 1)) { fvalue_free(&item); return 1; }
          out->list = (fvalue *)realloc(out->list, (size_t)(out->nlist + 1) * sizeof(fvalue));
          out->list[out->nlist++] = item;
          skip_ws(p); skip_ws(p);
          if (peek(p)->type == TK_PUNCT && peek(p)->p[0] == ',') { p->pos++; skip_ws(p); continue; }
          if (peek(p)->type == TK_PUNCT && peek(p)->p[0] == ')') { p->pos++; return 0; }
          perr(p, "unexpected token in list"); return 1;
        }
      }
      /* fallthrough */
    default: perr(p, "unexpected token"); return 1;
  }
}

static const char *fv_typeof(const fvalue *v) {
  switch (v->type) { case FV_STRING: return "string"; case FV_TIME: return "datetime"; case FV_BOOL: return "bool"; case FV_FLOAT: return "float"; case FV_INT: return "int"; case FV_NULL: return "null"; default: return "list"; }
}

static void fexpr_free(fexpr *e) { for (int i = 0; i < e->nterms; i++) { free(e->terms[i].attr); fvalue_free(&e->terms[i].val); } free(e->terms); e->terms = NULL; e->nterms = 0; }

/* filter.Parse (filters.go:293-311) = grammar.Parse + validateTerm (:240-272) */
static int filter_parse(const char *src, fexpr *out, char *err, size_t errcap) {
  memset(out, 0, sizeof *out);
  if (err && errcap) err[0] = 0;
  size_t n = strlen(src);
  if (n == 0) return 0;
  ftoken *toks = (ftoken *)malloc((n + 2) * sizeof(ftoken)); int nt = 0;
  for (size_t i = 0; i < n;) {
    ftoken t;
    if (lex_one(src + i, n - i, &t)) { snprintf(err, errcap, "invalid token at position %zu", i); free(toks); return 1; }
    toks[nt++] = t; i += t.n;
  }
  toks[nt].type = TK_EOF; toks[nt].p = src + n; toks[nt].n = 0; nt++;
  fparser p = {toks, nt, 0, err, errcap};
  int rc = 0;
  skip_ws(&p);
  int first = 1;
  while (peek(&p)->type != TK_EOF) {
    if (!first) {
      skip_ws(&p);
      if (!ident_is(peek(&p), "AND")) { perr(&p, "unexpected token (expected AND)"); rc = 1; break; }
      p.pos++; skip_ws(&p);
    }
    first = 0;
    fterm term; memset(&term, 0, sizeof term);
    ftoken *t = peek(&p);
    if (t->type != TK_IDENT) { perr(&p, "unexpected token (expected attribute)"); rc = 1; break; }
    term.attr = dupn(t->p, t->n); p.pos++; skip_ws(&p);
    t = peek(&p);
    if (t->type == TK_OP) {
      if (t->n == 1) term.op = t->p[0] == '=' ? OP_EQ : t->p[0] == '<' ? OP_LT : t->p[0] == '>' ? OP_GT : OP_MATCH;
      else term.op = (t->p[0] == '!' && t->p[1] == '=') ? OP_NE : (t->p[0] == '<') ? OP_LE : (t->p[0] == '>') ? OP_GE : OP_NOTMATCH;
      p.pos++;
    } else if (ident_is(t, "IN")) { term.op = OP_IN; p.pos++; }
    else if (ident_is(t, "NOT")) {
      p.pos++;
      while (peek(&p)->type == TK_WS) p.pos++;
      if (!ident_is(peek(&p), "IN")) { perr(&p, "unexpected token (expected IN)"); free(term.attr); rc = 1; break; }
      term.op = OP_NOTIN; p.pos++;
    } else { perr(&p, "unexpected token (expected operator)"); free(term.attr); rc = 1; break; }
    skip_ws(&p);
    if (parse_value(&p, &term.val, 0)) { free(term.attr); fvalue_free(&term.val); rc = 1; break; }
    skip_ws(&p);
    /* validateTerm */
    if (term.val.type == FV_LIST) {
      for (int i = 0; i < term.val.nlist && !rc; i++) {
        if (strcmp(fv_typeof(&term.val.list[i]), fv_typeof(&term.val.list[0]))) { perr(&p, "list items should have same type"); rc = 1; }
      }
      if (!rc && term.val.nlist > 0 && term.val.list[0].type == FV_LIST) { perr(&p, "nested list are not supported"); rc = 1; }
      if (!rc && term.op != OP_IN && term.op != OP_NOTIN) { perr(&p, "list values require [ NOT ] IN operator"); rc = 1; }
    } else if (term.op == OP_IN || term.op == OP_NOTIN) { perr(&p, "IN operator expect list value"); rc = 1; }
    if (!rc && term.val.type == FV_NULL && term.op != OP_EQ && term.op != OP_NE) { perr(&p, "NULL expects \"=\" or \"!=\" operator"); rc = 1; }
    out->terms = (fterm *)realloc(out->terms, (size_t)(out->nterms + 1) * sizeof(fterm));
    out->terms[out->nterms++] = term;
    if (rc) break;
  }
  free(toks);
  if (rc) { fexpr_free(out); return 1; }
  return 0;
}

int ora_filter_parse_check(const char *expr, char *err, size_t errcap) {
  fexpr e;
  if (filter_parse(expr, &e, err, errcap)) return -1;
  int n = e.nterms; fexpr_free(&e); return n;
}

/* ======================================================================
 * transformer object
 * ====================================================================== */
enum { T_MASK, T_RENAME, T_FILTER_COLUMNS, T_SKIP_EVENTS, T_FILTER_ROWS, T_TO_STRING, T_TO_DATETIME, T_SHARDER };

struct ora_transformer {
  int type;
  re_filter tables, columns;
  /* mask */
  char *salt; int ncolnames; char **colnames;
  /* rename */
  int nren; char **from_ns, **from_name, **to_ns, **to_name;
  /* skip_events */
  int skip[4];
  /* filter_rows */
  int nexpr; fexpr *exprs;
  /* to_string */
  int convert_to_bytes, skip_utc;
  /* sharder */
  int64_t shards; int is_random;
};

static const jnode *jn_geti(const jnode *o, const char *key) {
  /* encoding/json matches field names case-insensitively */
  const jnode *r = jn_get(o, key);
  if (r || !o || o->type != JN_OBJ) return r;
  for (int i = 0; i < o->n; i++) if (!strcasecmp(o->keys[i], key)) return o->kids[i];
  return NULL;
}

void ora_transformer_free(ora_transformer *t) {
  if (!t) return;
  re_filter_free(&t->tables); re_filter_free(&t->columns);
  free(t->salt);
  for (int i = 0; i < t->ncolnames; i++) free(t->colnames[i]);
  free(t->colnames);
  for (int i = 0; i < t->nren; i++) { free(t->from_ns[i]); free(t->from_name[i]); free(t->to_ns[i]); free(t->to_name[i]); }
  free(t->from_ns); free(t->from_name); free(t->to_ns); free(t->to_name);
  for (int i = 0; i < t->nexpr; i++) fexpr_free(&t->exprs[i]);
  free(t->exprs);
  free(t);
}

static int kind_from_string(const char *s) {
  if (!strcmp(s, "insert") || !strcmp(s, "Insert")) return TFGPU_K_INSERT;
  if (!strcmp(s, "update") || !strcmp(s, "Update")) return TFGPU_K_UPDATE;
  if (!strcmp(s, "delete") || !strcmp(s, "Delete")) return TFGPU_K_DELETE;
  return TFGPU_K_OTHER;
}

ora_transformer *ora_transformer_new(const char *type_name, const char *config_json, char *err, size_t errcap) {
  if (err && errcap) err[0] = 0;
  jnode *cfg = jn_parse(config_json && *config_json ? config_json : "{}", err, errcap);
  if (!cfg) return NULL;
  ora_transformer *t = (ora_transformer *)calloc(1, sizeof *t);
  const jnode *tables = jn_geti(cfg, "tables");
  const jnode *columns = jn_geti(cfg, "columns");
  int rc = 0;
  if (!strcmp(type_name, "mask_field")) { /* mask/mask.go:21-42 */
    t->type = T_MASK;
    rc = re_filter_init(&t->tables, jn_geti(tables, "includeTables"), jn_geti(tables, "excludeTables"), err, errcap);
    re_filter_init(&t->columns, NULL, NULL, err, errcap);
    t->salt = dups(jn_str(jn_geti(jn_geti(cfg, "maskFunctionHash"), "userDefinedSalt"), ""));
    if (columns && columns->type == JN_ARR) {
      t->ncolnames = columns->n; t->colnames = (char **)calloc((size_t)(columns->n ? columns->n : 1), sizeof(char *));
      for (int i = 0; i < columns->n; i++) t->colnames[i] = dups(jn_str(columns->kids[i], ""));
    }
  } else if (!strcmp(type_name, "rename_tables")) { /* rename/rename.go:20-41,85-93 */
    t->type = T_RENAME;
    re_filter_init(&t->tables, NULL, NULL, err, errcap); re_filter_init(&t->columns, NULL, NULL, err, errcap);
    const jnode *rt = jn_geti(cfg, "renameTables");
    int n = (rt && rt->type == JN_ARR) ? rt->n : 0;
    t->from_ns = (char **)calloc((size_t)(n ? n : 1), sizeof(char *)); t->from_name = (char **)calloc((size_t)(n ? n : 1), sizeof(char *));
    t->to_ns = (char **)calloc((size_t)(n ? n : 1), sizeof(char *)); t->to_name = (char **)calloc((size_t)(n ? n : 1), sizeof(char *));
    for (int i = 0; i < n; i++) {
      const jnode *o = jn_geti(rt->kids[i], "originalName"), *nn = jn_geti(rt->kids[i], "newName");
      const char *fns = jn_str(jn_geti(o, "nameSpace"), ""), *fnm = jn_str(jn_geti(o, "name"), "");
      /* map semantics: a later entry with the same key overwrites */
      int k; for (k = 0; k < t->nren; k++) if (!strcmp(t->from_ns[k], fns) && !strcmp(t->from_name[k], fnm)) break;
      if (k == t->nren) { t->from_ns[k] = dups(fns); t->from_name[k] = dups(fnm); t->nren++; } else { free(t->to_ns[k]); free(t->to_name[k]); }
      t->to_ns[k] = dups(jn_str(jn_geti(nn, "nameSpace"), "")); t->to_name[k] = dups(jn_str(jn_geti(nn, "name"), ""));
    }
  } else if (!strcmp(type_name, "filter_columns")) { /* filter/filter_columns_transformer.go:28-31,... */
    t->type = T_FILTER_COLUMNS;
    rc = re_filter_init(&t->tables, jn_geti(tables, "includeTables"), jn_geti(tables, "excludeTables"), err, errcap);
    if (!rc) rc = re_filter_init(&t->columns, jn_geti(columns, "includeColumns"), jn_geti(columns, "excludeColumns"), err, errcap);
  } else if (!strcmp(type_name, "skip_events")) { /* filter/skip_events.go:24-47 */
    t->type = T_SKIP_EVENTS;
    rc = re_filter_init(&t->tables, jn_geti(tables, "includeTables"), jn_geti(tables, "excludeTables"), err, errcap);
    re_filter_init(&t->columns, NULL, NULL, err, errcap);
    const jnode *ev = jn_geti(cfg, "events");
    /* kinds are compared as exact strings: "insert" only matches Kind("insert") */
    if (ev && ev->type == JN_ARR) for (int i = 0; i < ev->n; i++) {
      const char *e = jn_str(ev->kids[i], "");
      if (!strcmp(e, "insert")) t->skip[TFGPU_K_INSERT] = 1; else if (!strcmp(e, "update")) t->skip[TFGPU_K_UPDATE] = 1; else if (!strcmp(e, "delete")) t->skip[TFGPU_K_DELETE] = 1;
    }
  } else if (!strcmp(type_name, "filter_rows")) { /* filter_rows/filter_rows.go:42-96 */
    t->type = T_FILTER_ROWS;
    rc = re_filter_init(&t->tables, jn_geti(tables, "includeTables"), jn_geti(tables, "excludeTables"), err, errcap);
    re_filter_init(&t->columns, NULL, NULL, err, errcap);
    const char *single = jn_str(jn_geti(cfg, "filter"), "");
    const jnode *many = jn_geti(cfg, "filters");
    int nmany = (many && many->type == JN_ARR) ? many->n : 0;
    if (*single && nmany > 0) { snprintf(err, errcap, "Settings 'filters' and 'filter' cannot be enabled at the same time"); rc = 1; }
    if (!rc) {
      int n = nmany > 0 ? nmany : 1;
      t->exprs = (fexpr *)calloc((size_t)n, sizeof(fexpr));
      for (int i = 0; i < n && !rc; i++) {
        const char *src = nmany > 0 ? jn_str(many->kids[i], "") : single;
        char e2[200];
        if (filter_parse(src, &t->exprs[i], e2, sizeof e2)) { snprintf(err, errcap, "Unable to parse filter '%s': %s", src, e2); rc = 1; }
        t->nexpr = i + 1;
        /* valuesListToSet (util.go:84-112): bool lists are "not appropriate" */
        for (int k = 0; !rc && k < t->exprs[i].nterms; k++) {
          fterm *tm = &t->exprs[i].terms[k];
          if ((tm->op == OP_IN || tm->op == OP_NOTIN) && tm->val.type == FV_LIST && tm->val.nlist > 0 && (tm->val.list[0].type == FV_BOOL || tm->val.list[0].type == FV_NULL)) {
            snprintf(err, errcap, "Unable to prepare term values: not appropriate type of list values"); rc = 1;
          }
        }
      }
    }
  } else if (!strcmp(type_name, "convert_to_string")) { /* to_string/to_string.go:19-46 */
    t->type = T_TO_STRING;
    rc = re_filter_init(&t->tables, jn_geti(tables, "includeTables"), jn_geti(tables, "excludeTables"), err, errcap);
    if (!rc) rc = re_filter_init(&t->columns, jn_geti(columns, "includeColumns"), jn_geti(columns, "excludeColumns"), err, errcap);
    t->convert_to_bytes = jn_bool(jn_geti(cfg, "convert_to_bytes"), 0);
    t->skip_utc = jn_bool(jn_geti(cfg, "skip_utc_conversion"), 0);
  } else if (!strcmp(type_name, "convert_to_datetime")) { /* to_datetime/to_datetime.go:23-47 */
    t->type = T_TO_DATETIME;
    rc = re_filter_init(&t->tables, jn_geti(tables, "includeTables"), jn_geti(tables, "excludeTables"), err, errcap);
    if (!rc) rc = re_filter_init(&t->columns, jn_geti(columns, "includeColumns"), jn_geti(columns, "excludeColumns"), err, errcap);
  } else if (!strcmp(type_name, "sharder_transformer")) { /* sharder/sharder.go:20-68 */
    t->type = T_SHARDER;
    rc = re_filter_init(&t->tables, jn_geti(tables, "includeTables"), jn_geti(tables, "excludeTables"), err, errcap);
    t->is_random = jn_bool(jn_geti(cfg, "is_random"), 0);
    if (!rc) rc = t->is_random ? re_filter_init(&t->columns, NULL, NULL, err, errcap) : re_filter_init(&t->columns, jn_geti(columns, "includeColumns"), jn_geti(columns, "excludeColumns"), err, errcap);
    const char *sc = jn_str(jn_geti(cfg, "shardsCount"), "");
    if (!rc && ora_parse_int(sc, strlen(sc), 10, 64, &t->shards)) { snprintf(err, errcap, "cannot parse param as int: %s", sc); rc = 1; }
  } else {
    snprintf(err, errcap, "unknown transformer type %s", type_name); rc = 1;
  }
  jn_free(cfg);
  if (rc) { ora_transformer_free(t); return NULL; }
  return t;
}

static int mask_has_col(const ora_transformer *t, const char *name) {
  for (int i = 0; i < t->ncolnames; i++) if (!strcmp(t->colnames[i], name)) return 1;
  return 0;
}
static int todt_suitable_col(const ora_transformer *t, const char *name, int dtype) { /* to_datetime.go:59-61 */
  return re_filter_match(&t->columns, name) && (dtype == TFGPU_T_INT32 || dtype == TFGPU_T_UINT32);
}

/* checkColumnSuitable filter_rows.go:453-490 */
static int fr_col_suitable(const fvalue *v, int dt) {
  switch (v->type) {
    case FV_BOOL: return dt == TFGPU_T_BOOLEAN;
    case FV_FLOAT: case FV_INT: return (dt >= TFGPU_T_INT8 && dt <= TFGPU_T_UINT64) || dt == TFGPU_T_FLOAT32 || dt == TFGPU_T_FLOAT64;
    case FV_STRING: return dt == TFGPU_T_UTF8 || dt == TFGPU_T_BYTES || dt == TFGPU_T_ANY;
    case FV_TIME: return dt == TFGPU_T_TIMESTAMP || dt == TFGPU_T_DATE || dt == TFGPU_T_DATETIME;
    case FV_LIST: return v->nlist > 0 && v->list[0].type != FV_BOOL && v->list[0].type != FV_NULL;
    case FV_NULL: return 1;
  }
  return 0;
}

int ora_transformer_suitable(const ora_transformer *t, const char *ns, const char *table, const tfgpu_schema *s) {
  switch (t->type) {
    case T_RENAME: /* rename.go:63-66 */
      for (int i = 0; i < t->nren; i++) if (!strcmp(t->from_ns[i], ns) && !strcmp(t->from_name[i], table)) return 1;
      return 0;
    case T_MASK: /* hmac_hasher.go:76-89 */
      if (!match_any_table_variant(&t->tables, ns, table)) return 0;
      if (t->ncolnames == 0) return 1;
      for (int i = 0; i < s->ncols; i++) if (mask_has_col(t, s->cols[i].name)) return 1;
      return 0;
    case T_FILTER_COLUMNS: /* filter_columns_transformer.go:215-226 */
      if (!match_any_table_variant(&t->tables, ns, table)) return 0;
      for (int i = 0; i < s->ncols; i++) if (!re_filter_match(&t->columns, s->cols[i].name) && (s->cols[i].flags & TFGPU_COL_KEY)) return 0;
      return 1;
    case T_SKIP_EVENTS: return match_any_table_variant(&t->tables, ns, table);
    case T_FILTER_ROWS: /* filter_rows.go:418-451 */
      if (!match_any_table_variant(&t->tables, ns, table)) return 0;
      for (int e = 0; e < t->nexpr; e++)
        for (int k = 0; k < t->exprs[e].nterms; k++) {
          const fterm *tm = &t->exprs[e].terms[k];
          int found = 0;
          for (int i = 0; i < s->ncols; i++) if (!strcmp(tm->attr, s->cols[i].name)) { found = 1; if (!fr_col_suitable(&tm->val, s->cols[i].dtype)) return 0; break; }
          if (!found && s->ncols > 0) return 0;
        }
      return 1;
    case T_TO_STRING: case T_SHARDER: /* to_string.go:99-112, sharder.go:95-108 */
      if (!match_any_table_variant(&t->tables, ns, table)) return 0;
      if (re_filter_empty(&t->columns)) return 1;
      for (int i = 0; i < s->ncols; i++) if (re_filter_match(&t->columns, s->cols[i].name)) return 1;
      return 0;
    case T_TO_DATETIME: /* to_datetime.go:63-76 */
      if (!match_any_table_variant(&t->tables, ns, table)) return 0;
      if (re_filter_empty(&t->columns)) return 0;
      for (int i = 0; i < s->ncols; i++) if (todt_suitable_col(t, s->cols[i].name, s->cols[i].dtype)) return 1;
      return 0;
  }
  return 0;
}

static ora_schema *schema_clone(const ora_schema *s) {
  ora_schema *r = (ora_schema *)calloc(1, sizeof *r);
  r->refs = 1; r->ncols = s->ncols;
  r->cols = (ora_colschema *)calloc((size_t)(s->ncols ? s->ncols : 1), sizeof(ora_colschema));
  for (int i = 0; i < s->ncols; i++) { r->cols[i] = s->cols[i]; r->cols[i].name = dups(s->cols[i].name); r->cols[i].path = dups(s->cols[i].path); r->cols[i].original_type = dups(s->cols[i].original_type); }
  return r;
}

/* ResultSchema for each transformer, over the test-facing tfgpu_schema */
tfgpu_schema *ora_transformer_result_schema(const ora_transformer *t, const tfgpu_schema *s) {
  ora_schema *o = ora_schema_from(s);
  ora_schema *r = schema_clone(o);
  if (t->type == T_MASK) { /* hmac_hasher.go:35-46 */
    for (int i = 0; i < r->ncols; i++) if (mask_has_col(t, r->cols[i].name)) { r->cols[i].dtype = TFGPU_T_UTF8; free(r->cols[i].original_type); r->cols[i].original_type = dups(""); }
  } else if (t->type == T_TO_STRING) { /* to_string.go:114-127 */
    for (int i = 0; i < r->ncols; i++) if (re_filter_match(&t->columns, r->cols[i].name)) r->cols[i].dtype = t->convert_to_bytes ? TFGPU_T_BYTES : TFGPU_T_UTF8;
  } else if (t->type == T_TO_DATETIME) { /* to_datetime.go:125-133 */
    for (int i = 0; i < r->ncols; i++) if (todt_suitable_col(t, r->cols[i].name, r->cols[i].dtype)) r->cols[i].dtype = TFGPU_T_DATETIME;
  } else if (t->type == T_FILTER_COLUMNS) { /* filter_columns_transformer.go:228-239 */
    int w = 0;
    for (int i = 0; i < r->ncols; i++) {
      if (re_filter_match(&t->columns, r->cols[i].name)) r->cols[w++] = r->cols[i];
      else { free(r->cols[i].name); free(r->cols[i].path); free(r->cols[i].original_type); }
    }
    r->ncols = w;
  }
  ora_batch tmp; memset(&tmp, 0, sizeof tmp);
  ora_item it; memset(&it, 0, sizeof it); it.schema = r; tmp.items = &it; tmp.n = 1;
  tfgpu_schema *out = ora_batch_schema(&tmp);
  ora_schema_unref(r); ora_schema_unref(o);
  return out;
}

/* move an item from one batch to the end of another (Go: append(transformed, item) copies the struct) */
static ora_item *emit(ora_batch *out, ora_item *it) {
  ora_item *d = ora_batch_push(out);
  *d = *it;
  memset(it, 0, sizeof *it);
  return d;
}
static void emit_error(ora_batch *out, ora_item *it, int code, const char *msg) {
  ora_batch_add_error(out, it->src_row, code, msg);
  ora_item_clear(it);
}

static int schema_find(const ora_schema *s, const char *name) {
  for (int i = 0; i < s->ncols; i++) if (!strcmp(s->cols[i].name, name)) return i;
  return -1;
}

/* ---- a4 mask: hmac_hasher.go:52-74 ---- */
static void apply_mask(const ora_transformer *t, ora_batch *in, ora_batch *out) {
  static const char hexd[] = "0123456789abcdef";
  for (int64_t r = 0; r < in->n; r++) {
    ora_item *ci = &in->items[r];
    /* fastCols := ci.TableSchema.FastColumns() — lazy map built once per schema object */
    int nnames = ci->names ? ci->names->n : 0;
    for (int i = 0; i < nnames && i < ci->nvalues; i++) {
      const char *name = ci->names->names[i];
      if (!mask_has_col(t, name)) continue;
      int si = schema_find(ci->schema, name);
      int dtype = si >= 0 ? ci->schema->cols[si].dtype : TFGPU_T_INVALID;
      size_t sl; char *str = ora_serialize_to_string(&ci->values[i], dtype, &sl);
      uint8_t mac[32];
      ora_hmac_sha256(t->salt, strlen(t->salt), str, sl, mac); /* hmac.New per value */
      free(str);
      char *hex = (char *)malloc(65);
      for (int k = 0; k < 32; k++) { hex[2 * k] = hexd[mac[k] >> 4]; hex[2 * k + 1] = hexd[mac[k] & 15]; }
      hex[64] = 0;
      ora_value_free(&ci->values[i]);
      ci->values[i].kind = OV_STRING; ci->values[i].s = hex; ci->values[i].slen = 64;
    }
    /* ci.SetTableSchema(hh.schema(ci.TableSchema)) — a new schema per row */
    ora_schema *ns = schema_clone(ci->schema);
    for (int i = 0; i < ns->ncols; i++) if (mask_has_col(t, ns->cols[i].name)) { ns->cols[i].dtype = TFGPU_T_UTF8; free(ns->cols[i].original_type); ns->cols[i].original_type = dups(""); }
    ora_schema_unref(ci->schema); ci->schema = ns;
    emit(out, ci);
  }
}

/* ---- a8 rename: rename.go:46-61 ---- */
static void apply_rename(const ora_transformer *t, ora_batch *in, ora_batch *out) {
  for (int64_t r = 0; r < in->n; r++) {
    ora_item *ci = &in->items[r];
    for (int i = 0; i < t->nren; i++) if (!strcmp(t->from_ns[i], ci->ns) && !strcmp(t->from_name[i], ci->table)) {
      free(ci->ns); free(ci->table); ci->ns = dups(t->to_ns[i]); ci->table = dups(t->to_name[i]); break;
    }
    emit(out, ci);
  }
}

/* ---- a10 skip_events: skip_events.go:52-62 ---- */
static void apply_skip_events(const ora_transformer *t, ora_batch *in, ora_batch *out) {
  for (int64_t r = 0; r < in->n; r++) {
    ora_item *ci = &in->items[r];
    if (ci->kind < 3 && t->skip[ci->kind]) { ora_item_clear(ci); continue; }
    emit(out, ci);
  }
}

/* ---- a9 filter_columns: filter_columns_transformer.go:51-79,147-185 ---- */
static void apply_filter_columns(const ora_transformer *t, ora_batch *in, ora_batch *out) {
  for (int64_t r = 0; r < in->n; r++) {
    ora_item *ci = &in->items[r];
    /* getFilteredSchema → newFilteredTableSchema :105-123 (cached per schema hash in the reference) */
    int bad = 0, nkeep_schema = 0;
    for (int i = 0; i < ci->schema->ncols; i++) {
      if (!re_filter_match(&t->columns, ci->schema->cols[i].name)) { if (ci->schema->cols[i].key) { bad = 1; break; } }
      else nkeep_schema++;
    }
    if (bad) { emit_error(out, ci, TFGPU_ROW_CAST, "cannot exclude primary key column"); continue; }
    int nnames = ci->names ? ci->names->n : 0;
    int *idx = (int *)malloc(sizeof(int) * (size_t)(nnames ? nnames : 1)); int nk = 0;
    for (int i = 0; i < nnames; i++) {
      /* filteredColumns.Contains(name): the set of schema column names that matched */
      int si = schema_find(ci->schema, ci->names->names[i]);
      if (si >= 0 && re_filter_match(&t->columns, ci->names->names[i])) idx[nk++] = i;
    }
    if (nk == nnames && nkeep_schema == ci->schema->ncols) { free(idx); emit(out, ci); continue; } /* :169-171 untouched */
    ora_names *nn = (ora_names *)calloc(1, sizeof *nn); nn->refs = 1; nn->n = nk; nn->names = (char **)calloc((size_t)(nk ? nk : 1), sizeof(char *));
    ora_value *nv = (ora_value *)calloc((size_t)(nk ? nk : 1), sizeof(ora_value));
    for (int k = 0; k < nk; k++) { nn->names[k] = dups(ci->names->names[idx[k]]); nv[k] = ci->values[idx[k]]; ci->values[idx[k]].s = NULL; ci->values[idx[k]].kind = OV_NIL; }
    for (int i = 0; i < ci->nvalues; i++) ora_value_free(&ci->values[i]);
    free(ci->values); ci->values = nv; ci->nvalues = nk;
    ora_schema *ns = (ora_schema *)calloc(1, sizeof *ns); ns->refs = 1;
    ns->cols = (ora_colschema *)calloc((size_t)(nkeep_schema ? nkeep_schema : 1), sizeof(ora_colschema));
    for (int i = 0; i < ci->schema->ncols; i++) if (re_filter_match(&t->columns, ci->schema->cols[i].name)) {
      ora_colschema *d = &ns->cols[ns->ncols++]; *d = ci->schema->cols[i];
      d->name = dups(d->name); d->path = dups(d->path); d->original_type = dups(d->original_type);
    }
    /* names_unref */
    if (--ci->names->refs == 0) { for (int i = 0; i < ci->names->n; i++) free(ci->names->names[i]); free(ci->names->names); free(ci->names); }
    ci->names = nn;
    ora_schema_unref(ci->schema); ci->schema = ns;
    free(idx);
    emit(out, ci);
  }
}

/* ---- a11 filter_rows ---- */
/* toInt64E filter_rows/util.go:42-82: 0 ok, 1 not an int, 2 overflow */
static int to_int64e(const ora_value *v, int64_t *out) {
  switch (v->kind) {
    case OV_I8: case OV_I16: case OV_I32: case OV_I64: *out = v->v.i; return 0;
    case OV_U8: case OV_U16: case OV_U32: *out = (int64_t)v->v.u; return 0;
    case OV_U64: if (v->v.u > (uint64_t)INT64_MAX) return 2; *out = (int64_t)v->v.u; return 0;
    default: return 1;
  }
}
/* spf13/cast v1.7.1 ToFloat64E (third-party, go.mod:64; restated from its
 * published source): numeric types, string via ParseFloat, json.Number, bool, nil→0 */
static int to_float64e(const ora_value *v, double *out) {
  switch (v->kind) {
    case OV_F64: *out = v->v.f64; return 0;
    case OV_F32: *out = (double)v->v.f32; return 0;
    case OV_I8: case OV_I16: case OV_I32: case OV_I64: *out = (double)v->v.i; return 0;
    case OV_U8: case OV_U16: case OV_U32: case OV_U64: *out = (double)v->v.u; return 0;
    case OV_STRING: case OV_JSONNUM: { double d; int rc = ora_parse_float(v->s, v->slen, 64, &d); if (rc == 0) { *out = d; return 0; } return 1; }
    case OV_BOOL: *out = v->v.b ? 1 : 0; return 0;
    case OV_NIL: *out = 0; return 0;
    default: return 1;
  }
}
#define CMP_OP(a, b, op, res)                                                      \
  do {                                                                             \
    switch (op) {                                                                  \
      case OP_EQ: res = (a) == (b); break; case OP_NE: res = (a) != (b); break;    \
      case OP_LT: res = (a) < (b); break; case OP_LE: res = (a) <= (b); break;     \
      case OP_GT: res = (a) > (b); break; case OP_GE: res = (a) >= (b); break;     \
      default: res = -1;                                                           \
    }                                                                              \
  } while (0)

static int bytes_cmp(const char *a, size_t an, const char *b, size_t bn) {
  size_t m = an < bn ? an : bn; int c = m ? memcmp(a, b, m) : 0;
  if (c) return c < 0 ? -1 : 1;
  return an < bn ? -1 : an > bn ? 1 : 0;
}
static int contains(const char *h, size_t hn, const char *n, size_t nn) {
  if (nn == 0) return 1;
  if (nn > hn) return 0;
  return memmem(h, hn, n, nn) != NULL;
}

/* stringToTime filter_rows/util.go:15-39 */
static int string_to_time(const char *s, size_t n, int64_t *sec, int32_t *nsec) {
  static const char *layouts[] = {
      "2006-01-02 15:04:05 -0700 MST", "2006-01-02T15:04:05", "2006-01-02T15:04:05.000-0700",
      "01/02 03:04:05PM '06 -0700", "Mon Jan _2 15:04:05 2006", "Mon Jan _2 15:04:05 MST 2006", "Mon Jan 02 15:04:05 -0700 2006",
      "02 Jan 06 15:04 MST", "02 Jan 06 15:04 -0700", "Monday, 02-Jan-06 15:04:05 MST", "Mon, 02 Jan 2006 15:04:05 MST",
      "Mon, 02 Jan 2006 15:04:05 -0700", "2006-01-02T15:04:05Z07:00", "2006-01-02T15:04:05.999999999Z07:00", "3:04PM",
      "Jan _2 15:04:05", "Jan _2 15:04:05.000", "Jan _2 15:04:05.000000", "Jan _2 15:04:05.000000000",
      "2006-01-02 15:04:05", "2006-01-02", "15:04:05"};
  for (size_t i = 0; i < sizeof layouts / sizeof *layouts; i++) if (ora_time_parse(layouts[i], s, n, sec, nsec) == 0) return 1;
  return 0;
}
static int64_t unix_micro(int64_t sec, int32_t nsec) { return sec * 1000000 + nsec / 1000; }

/* matchValue filter_rows.go:180-365.  returns 1/0 match, or -code on error */
static int match_value(const ora_value *v1, const fterm *tm) {
  const fvalue *v2 = &tm->val; int op = tm->op;
  int is_set = (op == OP_IN || op == OP_NOTIN);
  int isInt1 = 0, isFloat1 = 0; int64_t int1 = 0; double float1 = 0;
  int rc = to_int64e(v1, &int1);
  if (rc == 0) isInt1 = 1; else if (rc == 2) return -TFGPU_ROW_INT_OVERFLOW;
  if (to_float64e(v1, &float1) == 0 && !isInt1) isFloat1 = 1;
  int t2 = v2->type == FV_LIST ? (v2->nlist ? v2->list[0].type : -1) : v2->type;
  int res = -1;
  switch (t2) {
    case FV_INT:
      if (isInt1) {
        if (is_set) { int f = 0; for (int i = 0; i < v2->nlist; i++) if (v2->list[i].i == int1) f = 1; return op == OP_IN ? f : !f; }
        CMP_OP(int1, v2->i, op, res); if (res < 0) return -TFGPU_ROW_TYPE_PAIR; return res;
      }
      if (isFloat1) {
        if (is_set) {
          int f = 0;
          if (trunc(float1) == float1) { int64_t iv = (int64_t)float1; for (int i = 0; i < v2->nlist; i++) if (v2->list[i].i == iv) f = 1; return op == OP_IN ? f : !f; }
          return 0; /* isMatched = false for both IN and NOT IN */
        }
        CMP_OP(float1, (double)v2->i, op, res); if (res < 0) return -TFGPU_ROW_TYPE_PAIR; return res;
      }
      break;
    case FV_FLOAT:
      if (isInt1 || isFloat1) {
        double a = isInt1 ? (double)int1 : float1;
        if (is_set) { int f = 0; for (int i = 0; i < v2->nlist; i++) if (v2->list[i].f == a) f = 1; return op == OP_IN ? f : !f; }
        CMP_OP(a, v2->f, op, res); if (res < 0) return -TFGPU_ROW_TYPE_PAIR; return res;
      }
      break;
    case FV_BOOL:
      if (v2->type == FV_LIST) return -TFGPU_ROW_TYPE_PAIR; /* "Unknown filter's value type" */
      if (v1->kind == OV_BOOL) { int a = v1->v.b ? 1 : 0, b = v2->b ? 1 : 0; CMP_OP(a, b, op, res); if (res < 0) return -TFGPU_ROW_TYPE_PAIR; return res; }
      break;
    case FV_STRING: {
      int isbytes = v1->kind == OV_BYTES;
      if (v2->type == FV_STRING && isbytes) {
        if (op == OP_MATCH) return contains(v1->s, v1->slen, v2->s, v2->slen);
        if (op == OP_NOTMATCH) return !contains(v1->s, v1->slen, v2->s, v2->slen);
        int c = bytes_cmp(v1->s, v1->slen, v2->s, v2->slen);
        CMP_OP(c, 0, op, res); if (res < 0) return -TFGPU_ROW_TYPE_PAIR; return res;
      }
      if (isbytes || v1->kind == OV_STRING) {
        if (op == OP_MATCH) return contains(v1->s, v1->slen, v2->s, v2->slen);
        if (op == OP_NOTMATCH) return !contains(v1->s, v1->slen, v2->s, v2->slen);
        if (is_set) { int f = 0; for (int i = 0; i < v2->nlist; i++) if (v2->list[i].slen == v1->slen && (v1->slen == 0 || !memcmp(v2->list[i].s, v1->s, v1->slen))) f = 1; return op == OP_IN ? f : !f; }
        int c = bytes_cmp(v1->s, v1->slen, v2->s, v2->slen);
        CMP_OP(c, 0, op, res); if (res < 0) return -TFGPU_ROW_TYPE_PAIR; return res;
      }
      break;
    }
    case FV_TIME: {
      int64_t us1; int have = 0;
      if (v1->kind == OV_TIME) { us1 = unix_micro(v1->v.t.sec, v1->v.t.nsec); have = 1; }
      else if (v1->kind == OV_STRING) { int64_t s; int32_t ns; if (string_to_time(v1->s, v1->slen, &s, &ns)) { us1 = unix_micro(s, ns); have = 1; } }
      if (have) {
        if (is_set) { int f = 0; for (int i = 0; i < v2->nlist; i++) if (unix_micro(v2->list[i].tsec, v2->list[i].tnsec) == us1) f = 1; return op == OP_IN ? f : !f; }
        int64_t us2 = unix_micro(v2->tsec, v2->tnsec);
        CMP_OP(us1, us2, op, res); if (res < 0) return -TFGPU_ROW_TYPE_PAIR; return res;
      }
      break;
    }
    case FV_NULL:
      if (op == OP_EQ) return v1->kind == OV_NIL;
      if (op == OP_NE) return v1->kind != OV_NIL;
      break;
    default: return -TFGPU_ROW_TYPE_PAIR;
  }
  return -TFGPU_ROW_TYPE_PAIR; /* "Unsupported type pair" */
}

static void apply_filter_rows(const ora_transformer *t, ora_batch *in, ora_batch *out) {
  for (int64_t r = 0; r < in->n; r++) {
    ora_item *it = &in->items[r];
    if (it->kind == TFGPU_K_UPDATE || it->kind == TFGPU_K_DELETE) { emit_error(out, it, TFGPU_ROW_UNSUPPORTED_KIND, "Found non-supported kind"); continue; }
    if (!match_any_table_variant(&t->tables, it->ns, it->table) || is_system_table(it->table) || it->kind != TFGPU_K_INSERT) { emit(out, it); continue; }
    int matched = 0, err = 0;
    for (int e = 0; e < t->nexpr && !matched && !err; e++) { /* matchItem :132-143 */
      const fexpr *ex = &t->exprs[e];
      int ok = 1;
      for (int k = 0; k < ex->nterms && ok && !err; k++) { /* matchExpression :145-178 */
        const fterm *tm = &ex->terms[k];
        int nn = it->names ? it->names->n : 0;
        for (int i = 0; i < nn; i++) {
          if (strcmp(it->names->names[i], tm->attr)) { if (i < nn - 1) continue; err = TFGPU_ROW_COLUMN_NOT_FOUND; break; }
          int m = match_value(&it->values[i], tm);
          if (m < 0) err = -m; else if (!m) ok = 0;
          break;
        }
      }
      if (!err && ok) matched = 1;
    }
    if (err) { emit_error(out, it, err, err == TFGPU_ROW_COLUMN_NOT_FOUND ? "Unable to find column" : err == TFGPU_ROW_INT_OVERFLOW ? "Provided value overflows int64" : "Unsupported type pair"); continue; }
    if (matched) emit(out, it); else ora_item_clear(it);
  }
}

/* ---- a6 convert_to_string: to_string.go:58-97 ---- */
static void apply_to_string(const ora_transformer *t, ora_batch *in, ora_batch *out) {
  for (int64_t r = 0; r < in->n; r++) {
    ora_item *it = &in->items[r];
    ora_schema *ns = schema_clone(it->schema);
    int *oldt = (int *)calloc((size_t)(ns->ncols ? ns->ncols : 1), sizeof(int)); /* oldTypes map */
    for (int i = 0; i < ns->ncols; i++) { oldt[i] = -1; if (re_filter_match(&t->columns, ns->cols[i].name)) { oldt[i] = ns->cols[i].dtype; ns->cols[i].dtype = t->convert_to_bytes ? TFGPU_T_BYTES : TFGPU_T_UTF8; } }
    int nn = it->names ? it->names->n : 0;
    ora_value *nv = (ora_value *)calloc((size_t)(it->nvalues ? it->nvalues : 1), sizeof(ora_value));
    for (int i = 0; i < nn && i < it->nvalues; i++) {
      const char *cn = it->names->names[i];
      if (re_filter_match(&t->columns, cn)) {
        int si = schema_find(it->schema, cn);
        int dt = si >= 0 ? oldt[si] : TFGPU_T_INVALID; /* oldTypes[columnName] ("" if absent) */
        size_t sl; char *s = ora_serialize_to_string(&it->values[i], dt, &sl);
        nv[i].kind = t->convert_to_bytes ? OV_BYTES : OV_STRING; nv[i].s = s; nv[i].slen = sl;
        ora_value_free(&it->values[i]);
      } else { nv[i] = it->values[i]; it->values[i].s = NULL; it->values[i].kind = OV_NIL; }
    }
    free(it->values); it->values = nv;
    ora_schema_unref(it->schema); it->schema = ns; free(oldt);
    emit(out, it);
  }
}

/* ---- a7 convert_to_datetime: to_datetime.go:89-123,135-149 ---- */
static void apply_to_datetime(const ora_transformer *t, ora_batch *in, ora_batch *out) {
  for (int64_t r = 0; r < in->n; r++) {
    ora_item *it = &in->items[r];
    ora_schema *ns = schema_clone(it->schema);
    for (int i = 0; i < ns->ncols; i++) if (todt_suitable_col(t, ns->cols[i].name, ns->cols[i].dtype)) ns->cols[i].dtype = TFGPU_T_DATETIME;
    int nn = it->names ? it->names->n : 0;
    for (int i = 0; i < nn && i < it->nvalues; i++) {
      int si = schema_find(it->schema, it->names->names[i]);
      if (si < 0) si = 0; /* colNameToIdx[missing] = 0 */
      if (it->schema->ncols == 0) continue;
      const ora_colschema *c = &it->schema->cols[si];
      if (!todt_suitable_col(t, c->name, c->dtype)) continue;
      /* oldTypes[columnName]: keyed by the ChangeItem column name */
      int sj = schema_find(it->schema, it->names->names[i]);
      int odt = (sj >= 0 && todt_suitable_col(t, it->schema->cols[sj].name, it->schema->cols[sj].dtype)) ? it->schema->cols[sj].dtype : TFGPU_T_INVALID;
      ora_value *v = &it->values[i];
      int64_t sec = 0;
      if (odt == TFGPU_T_INT32 && v->kind == OV_I32) sec = v->v.i;
      else if (odt == TFGPU_T_UINT32 && v->kind == OV_U32) sec = (int64_t)v->v.u;
      ora_value_free(v);
      v->kind = OV_TIME; v->v.t.sec = sec; v->v.t.nsec = 0;
    }
    ora_schema_unref(it->schema); it->schema = ns;
    emit(out, it);
  }
}

/* ---- a13 sharder: sharder.go:83-93,130-145 ---- */
static void apply_sharder(const ora_transformer *t, ora_batch *in, ora_batch *out) {
  for (int64_t r = 0; r < in->n; r++) {
    ora_item *it = &in->items[r];
    /* fieldNameToVal := item.AsMap() — a map allocated per row */
    size_t cap = 64, w = 0; char *sum = (char *)malloc(cap); int first = 1;
    for (int i = 0; i < it->schema->ncols; i++) {
      const char *cn = it->schema->cols[i].name;
      if (!re_filter_match(&t->columns, cn)) continue;
      ora_value nil; memset(&nil, 0, sizeof nil);
      const ora_value *v = &nil;
      int nn = it->names ? it->names->n : 0;
      for (int k = nn - 1; k >= 0; k--) if (!strcmp(it->names->names[k], cn)) { v = &it->values[k]; break; } /* map: last write wins */
      size_t sl; char *s = ora_serialize_to_string(v, it->schema->cols[i].dtype, &sl);
      if (w + sl + 2 > cap) { cap = (w + sl + 2) * 2; sum = (char *)realloc(sum, cap); }
      if (!first) sum[w++] = '.';
      first = 0;
      memcpy(sum + w, s, sl); w += sl; free(s);
    }
    uint32_t h = ora_crc32_ieee(sum, w) % (uint32_t)t->shards;
    free(sum);
    char buf[16]; size_t n = ora_fmt_uint(buf, h); buf[n] = 0;
    free(it->part_id); it->part_id = dups(buf);
    emit(out, it);
  }
}

ora_batch *ora_transformer_apply(const ora_transformer *t, ora_batch *in) {
  ora_batch *out = ora_batch_new();
  switch (t->type) {
    case T_MASK: apply_mask(t, in, out); break;
    case T_RENAME: apply_rename(t, in, out); break;
    case T_FILTER_COLUMNS: apply_filter_columns(t, in, out); break;
    case T_SKIP_EVENTS: apply_skip_events(t, in, out); break;
    case T_FILTER_ROWS: apply_filter_rows(t, in, out); break;
    case T_TO_STRING: apply_to_string(t, in, out); break;
    case T_TO_DATETIME: apply_to_datetime(t, in, out); break;
    case T_SHARDER: apply_sharder(t, in, out); break;
  }
  /* carry errors already recorded on the input batch */
  for (int64_t i = 0; i < in->nerr; i++) ora_batch_add_error(out, in->errs[i].row, in->errs[i].code, in->errs[i].msg);
  ora_batch_free(in);
  return out;
}
