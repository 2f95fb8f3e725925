/*
 * oracle/ora_jv.h — encoding/json restated for the oracle's JSON-envelope parsers (ora_srjson.c, ora_debezium.c): the
 * scanner grammar, Decoder.Decode with UseNumber into an interface{} tree (duplicate keys: the last one wins; strings
 * unquoted with surrogate pairs and invalid UTF-8 -> U+FFFD), and json.Marshal of that tree (sorted keys, escapeHTML).
 * TEST INFRASTRUCTURE ONLY (see ora.h).  Static functions: every includer gets its own copy.
 */
#ifndef ORA_JV_H
#define ORA_JV_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char *p; size_t n, cap; } sbuf;
static void sb_put(sbuf *b, const void *s, size_t n) {
  if (b->n + n + 1 > b->cap) { b->cap = (b->n + n + 1) * 2; b->p = (char *)realloc(b->p, b->cap); }
  memcpy(b->p + b->n, s, n); b->n += n; b->p[b->n] = 0;
}
static void sb_c(sbuf *b, char c) { sb_put(b, &c, 1); }
static void sb_s(sbuf *b, const char *s) { sb_put(b, s, strlen(s)); }

/* ---- the decoded value: interface{} as Decoder.Decode with UseNumber builds it ---- */
enum { JV_NULL, JV_FALSE, JV_TRUE, JV_NUM, JV_STR, JV_ARR, JV_OBJ };
typedef struct jv {
  int t;
  char *s; size_t n;            /* JV_NUM: literal text; JV_STR: unquoted bytes */
  struct jv **kids; char **keys; size_t *klen; int nk, cap;
} jv;
static void jv_free(jv *v) {
  if (!v) return;
  for (int i = 0; i < v->nk; i++) { jv_free(v->kids[i]); if (v->keys) free(v->keys[i]); }
  free(v->kids); free(v->keys); free(v->klen); free(v->s); free(v);
}
static int jv_keep_dups = 0;  /* 1: objects keep every member in document order (struct decoding looks at each occurrence); 0: a map */
static void jv_add(jv *c, char *key, size_t klen, jv *kid) {
  if (c->t == JV_OBJ && !jv_keep_dups) for (int i = 0; i < c->nk; i++) if (c->klen[i] == klen && !memcmp(c->keys[i], key, klen)) {  /* m[key] = v: the last one wins */
    jv_free(c->kids[i]); c->kids[i] = kid; free(key); return;
  }
  if (c->nk == c->cap) {
    c->cap = c->cap ? c->cap * 2 : 4;
    c->kids = (jv **)realloc(c->kids, sizeof(jv *) * (size_t)c->cap);
    if (c->t == JV_OBJ) { c->keys = (char **)realloc(c->keys, sizeof(char *) * (size_t)c->cap); c->klen = (size_t *)realloc(c->klen, sizeof(size_t) * (size_t)c->cap); }
  }
  c->kids[c->nk] = kid;
  if (c->t == JV_OBJ) { c->keys[c->nk] = key; c->klen[c->nk] = klen; }
  c->nk++;
}

typedef struct { const unsigned char *p, *e; int err; int depth; } jp;
static void ws(jp *s) { while (s->p < s->e && (*s->p == ' ' || *s->p == '\t' || *s->p == '\r' || *s->p == '\n')) s->p++; }
static int hexv(int c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
static void put_rune(sbuf *b, unsigned r) {
  if (r < 0x80) sb_c(b, (char)r);
  else if (r < 0x800) { sb_c(b, (char)(0xC0 | (r >> 6))); sb_c(b, (char)(0x80 | (r & 63))); }
  else if (r < 0x10000) { sb_c(b, (char)(0xE0 | (r >> 12))); sb_c(b, (char)(0x80 | ((r >> 6) & 63))); sb_c(b, (char)(0x80 | (r & 63))); }
  else { sb_c(b, (char)(0xF0 | (r >> 18))); sb_c(b, (char)(0x80 | ((r >> 12) & 63))); sb_c(b, (char)(0x80 | ((r >> 6) & 63))); sb_c(b, (char)(0x80 | (r & 63))); }
}
/* scanner.go stateInString / stateInStringEsc* + decode.go unquoteBytes; s->p at the opening quote */
static int scan_string(jp *s, sbuf *out) {
  s->p++;
  for (;;) {
    if (s->p >= s->e) return s->err = 1;
    unsigned c = *s->p;
    if (c == '"') { s->p++; return 0; }
    if (c < 0x20) return s->err = 1;  /* "invalid character in string literal" */
    if (c == '\\') {
      if (s->p + 1 >= s->e) return s->err = 1;
      unsigned d = s->p[1];
      s->p += 2;
      switch (d) {
        case '"': sb_c(out, '"'); break; case '\\': sb_c(out, '\\'); break; case '/': sb_c(out, '/'); break;
        case 'b': sb_c(out, '\b'); break; case 'f': sb_c(out, '\f'); break; case 'n': sb_c(out, '\n'); break;
        case 'r': sb_c(out, '\r'); break; case 't': sb_c(out, '\t'); break;
        case 'u': {
          if (s->e - s->p < 4) return s->err = 1;
          int r = 0;
          for (int i = 0; i < 4; i++) { int h = hexv(s->p[i]); if (h < 0) return s->err = 1; r = r * 16 + h; }
          s->p += 4;
          if (r >= 0xD800 && r < 0xE000) {  /* utf16.IsSurrogate: a valid pair combines, anything else is U+FFFD */
            int r2 = -1;
            if (s->e - s->p >= 6 && s->p[0] == '\\' && s->p[1] == 'u') { r2 = 0; for (int i = 0; i < 4; i++) { int h = hexv(s->p[2 + i]); if (h < 0) { r2 = -1; break; } r2 = r2 * 16 + h; } }
            if (r < 0xDC00 && r2 >= 0xDC00 && r2 < 0xE000) { put_rune(out, 0x10000 + (((unsigned)r - 0xD800) << 10) + ((unsigned)r2 - 0xDC00)); s->p += 6; }
            else put_rune(out, 0xFFFD);
          } else put_rune(out, (unsigned)r);
          break;
        }
        default: return s->err = 1;  /* "invalid character in string escape code" */
      }
      continue;
    }
    if (c < 0x80) { sb_c(out, (char)c); s->p++; continue; }
    /* utf8.DecodeRune: invalid bytes become U+FFFD one at a time */
    size_t need = 0; unsigned lo = 0x80, hi = 0xBF;
    if (c >= 0xC2 && c <= 0xDF) need = 1;
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
    int ok = need > 0 && (size_t)(s->e - s->p) > need;
    if (ok) for (size_t k = 1; k <= need; k++) { unsigned d = s->p[k], l = k == 1 ? lo : 0x80, h = k == 1 ? hi : 0xBF; if (d < l || d > h) { ok = 0; break; } }
    if (!ok) { put_rune(out, 0xFFFD); s->p++; continue; }
    sb_put(out, s->p, need + 1); s->p += need + 1;
  }
}
/* scanner.go stateNeg/state0/state1/stateDot/stateE...: -?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][+-]?[0-9]+)? */
static int scan_number(jp *s) {
  if (s->p < s->e && *s->p == '-') s->p++;
  if (s->p >= s->e) return s->err = 1;
  if (*s->p == '0') s->p++;
  else if (*s->p >= '1' && *s->p <= '9') { while (s->p < s->e && *s->p >= '0' && *s->p <= '9') s->p++; }
  else return s->err = 1;
  if (s->p < s->e && *s->p == '.') { s->p++; if (s->p >= s->e || *s->p < '0' || *s->p > '9') return s->err = 1; while (s->p < s->e && *s->p >= '0' && *s->p <= '9') s->p++; }
  if (s->p < s->e && (*s->p == 'e' || *s->p == 'E')) {
    s->p++;
    if (s->p < s->e && (*s->p == '+' || *s->p == '-')) s->p++;
    if (s->p >= s->e || *s->p < '0' || *s->p > '9') return s->err = 1;
    while (s->p < s->e && *s->p >= '0' && *s->p <= '9') s->p++;
  }
  return 0;
}
static jv *parse_value(jp *s) {
  ws(s);
  if (s->p >= s->e) { s->err = 1; return NULL; }
  jv *v = (jv *)calloc(1, sizeof *v);
  unsigned c = *s->p;
  if (c == '{' || c == '[') {
    if (++s->depth > 10000) { s->err = 1; return v; }  /* maxNestingDepth */
    const int obj = c == '{';
    v->t = obj ? JV_OBJ : JV_ARR;
    s->p++; ws(s);
    if (s->p < s->e && *s->p == (obj ? '}' : ']')) { s->p++; s->depth--; return v; }
    for (;;) {
      char *key = NULL; size_t klen = 0;
      if (obj) {
        ws(s);
        if (s->p >= s->e || *s->p != '"') { s->err = 1; return v; }
        sbuf kb = {0}; sb_put(&kb, "", 0);
        if (scan_string(s, &kb)) { free(kb.p); return v; }
        key = kb.p; klen = kb.n;
        ws(s);
        if (s->p >= s->e || *s->p != ':') { s->err = 1; free(key); return v; }
        s->p++;
      }
      jv *kid = parse_value(s);
      if (s->err) { jv_free(kid); free(key); return v; }
      jv_add(v, key, klen, kid);
      ws(s);
      if (s->p >= s->e) { s->err = 1; return v; }
      if (*s->p == ',') { s->p++; continue; }
      if (*s->p == (obj ? '}' : ']')) { s->p++; s->depth--; return v; }
      s->err = 1; return v;
    }
  }
  if (c == '"') { sbuf b = {0}; sb_put(&b, "", 0); v->t = JV_STR; scan_string(s, &b); v->s = b.p; v->n = b.n; return v; }
  if (c == '-' || (c >= '0' && c <= '9')) {
    const unsigned char *a = s->p;
    v->t = JV_NUM;
    if (!scan_number(s)) { v->n = (size_t)(s->p - a); v->s = (char *)malloc(v->n + 1); memcpy(v->s, a, v->n); v->s[v->n] = 0; }
    return v;
  }
  const char *lit = c == 't' ? "true" : c == 'f' ? "false" : c == 'n' ? "null" : NULL;
  if (!lit || (size_t)(s->e - s->p) < strlen(lit) || memcmp(s->p, lit, strlen(lit))) { s->err = 1; return v; }
  s->p += strlen(lit);
  v->t = c == 't' ? JV_TRUE : c == 'f' ? JV_FALSE : JV_NULL;
  return v;
}
/* Decoder.Decode(&map[string]interface{}): the next value of the stream; trailing bytes after an object / array are
 * left in the buffer, a scalar must be followed by white space or the end (stateEndTop).  Returns NULL on any error;
 * *is_null = 1 for a top-level null (the map stays nil).                                                          */
static jv *decode_map(const unsigned char *p, size_t n, int *is_null) {
  jp s = {p, p + n, 0, 0};
  *is_null = 0;
  jv *v = parse_value(&s);
  if (s.err || !v) { jv_free(v); return NULL; }
  if (v->t == JV_OBJ) return v;
  int scalar_ok = s.p >= s.e || *s.p == ' ' || *s.p == '\t' || *s.p == '\r' || *s.p == '\n';
  if (v->t == JV_NULL && scalar_ok) { *is_null = 1; jv_free(v); return NULL; }
  jv_free(v);  /* syntax error after the scalar, or UnmarshalTypeError: neither is a map */
  return NULL;
}

/* ---- json.Marshal(value) of the decoded interface{}: escapeHTML on, map keys sorted ---- */
static const char HEXD[] = "0123456789abcdef";
static void go_string(sbuf *b, const char *s, size_t n) {
  sb_c(b, '"');
  size_t i = 0;
  while (i < n) {
    unsigned char c = (unsigned char)s[i];
    if (c < 0x80) {
      if (c >= 0x20 && c != '"' && c != '\\' && c != '<' && c != '>' && c != '&') { sb_c(b, (char)c); i++; continue; }
      switch (c) {
        case '"': sb_s(b, "\\\""); break; case '\\': sb_s(b, "\\\\"); break; case '\b': sb_s(b, "\\b"); break; case '\f': sb_s(b, "\\f"); break;
        case '\n': sb_s(b, "\\n"); break; case '\r': sb_s(b, "\\r"); break; case '\t': sb_s(b, "\\t"); break;
        default: sb_s(b, "\\u00"); sb_c(b, HEXD[c >> 4]); sb_c(b, HEXD[c & 15]);
      }
      i++; continue;
    }
    /* the decoded strings are valid UTF-8 already; only U+2028 / U+2029 are rewritten */
    if (c == 0xE2 && i + 2 < n && (unsigned char)s[i + 1] == 0x80 && ((unsigned char)s[i + 2] & 0xFE) == 0xA8) { sb_s(b, "\\u202"); sb_c(b, HEXD[(unsigned char)s[i + 2] & 0xF]); i += 3; continue; }
    sb_c(b, (char)c); i++;
  }
  sb_c(b, '"');
}
typedef struct { const char *k; size_t n; int idx; } kref;
static int kcmp(const void *a, const void *b) {
  const kref *x = (const kref *)a, *y = (const kref *)b;
  size_t m = x->n < y->n ? x->n : y->n;
  int c = memcmp(x->k, y->k, m);
  return c ? c : (x->n < y->n ? -1 : x->n > y->n);
}
static void go_marshal(sbuf *b, const jv *v) {
  switch (v->t) {
    case JV_NULL: sb_s(b, "null"); return;
    case JV_FALSE: sb_s(b, "false"); return;
    case JV_TRUE: sb_s(b, "true"); return;
    case JV_NUM: sb_put(b, v->s, v->n); return;
    case JV_STR: go_string(b, v->s, v->n); return;
    case JV_ARR:
      sb_c(b, '[');
      for (int i = 0; i < v->nk; i++) { if (i) sb_c(b, ','); go_marshal(b, v->kids[i]); }
      sb_c(b, ']');
      return;
    default: {
      kref *ks = (kref *)malloc(sizeof(kref) * (size_t)(v->nk ? v->nk : 1));
      for (int i = 0; i < v->nk; i++) { ks[i].k = v->keys[i]; ks[i].n = v->klen[i]; ks[i].idx = i; }
      qsort(ks, (size_t)v->nk, sizeof(kref), kcmp);
      sb_c(b, '{');
      for (int i = 0; i < v->nk; i++) { if (i) sb_c(b, ','); go_string(b, ks[i].k, ks[i].n); sb_c(b, ':'); go_marshal(b, v->kids[ks[i].idx]); }
      sb_c(b, '}');
      free(ks);
    }
  }
}

#endif
