/* oracle/ora_json.c — minimal JSON tree reader for transformer configs
 * (the Go side remaps map[string]any → typed config through encoding/json,
 * pkg/transformer/registry.go:36-41).  TEST INFRASTRUCTURE ONLY. */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ora_json.h"

typedef struct { const char *p, *e; char *err; size_t errcap; int failed; } jp;

static void fail(jp *s, const char *m) { if (!s->failed && s->err) snprintf(s->err, s->errcap, "json: %s at offset %ld", m, (long)(s->e - s->p)); s->failed = 1; }
static void ws(jp *s) { while (s->p < s->e && (*s->p == ' ' || *s->p == '\t' || *s->p == '\n' || *s->p == '\r')) s->p++; }

static size_t utf8_put(char *d, unsigned cp) {
  if (cp < 0x80) { d[0] = (char)cp; return 1; }
  if (cp < 0x800) { d[0] = (char)(0xC0 | cp >> 6); d[1] = (char)(0x80 | (cp & 0x3F)); return 2; }
  if (cp < 0x10000) { d[0] = (char)(0xE0 | cp >> 12); d[1] = (char)(0x80 | ((cp >> 6) & 0x3F)); d[2] = (char)(0x80 | (cp & 0x3F)); return 3; }
  d[0] = (char)(0xF0 | cp >> 18); d[1] = (char)(0x80 | ((cp >> 12) & 0x3F)); d[2] = (char)(0x80 | ((cp >> 6) & 0x3F)); d[3] = (char)(0x80 | (cp & 0x3F)); return 4;
}

static char *parse_string(jp *s, size_t *olen) {
  if (s->p >= s->e || *s->p != '"') { fail(s, "expected string"); return NULL; }
  s->p++;
  size_t cap = 16, n = 0; char *out = (char *)malloc(cap);
  while (s->p < s->e && *s->p != '"') {
    if (n + 8 > cap) { cap *= 2; out = (char *)realloc(out, cap); }
    if (*s->p == '\\') {
      s->p++;
      if (s->p >= s->e) break;
      char c = *s->p++;
      switch (c) {
        case 'n': out[n++] = '\n'; break; case 't': out[n++] = '\t'; break; case 'r': out[n++] = '\r'; break;
        case 'b': out[n++] = '\b'; break; case 'f': out[n++] = '\f'; break;
        case 'u': {
          unsigned cp = 0;
          for (int i = 0; i < 4 && s->p < s->e; i++) { char h = *s->p++; cp = cp * 16 + (unsigned)(isdigit((unsigned char)h) ? h - '0' : (tolower(h) - 'a' + 10)); }
          if (cp >= 0xD800 && cp < 0xDC00 && s->p + 6 <= s->e && s->p[0] == '\\' && s->p[1] == 'u') {
            unsigned lo = 0; const char *q = s->p + 2;
            for (int i = 0; i < 4; i++) { char h = q[i]; lo = lo * 16 + (unsigned)(isdigit((unsigned char)h) ? h - '0' : (tolower(h) - 'a' + 10)); }
            if (lo >= 0xDC00 && lo < 0xE000) { cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); s->p += 6; }
          }
          n += utf8_put(out + n, cp); break;
        }
        default: out[n++] = c;
      }
    } else out[n++] = *s->p++;
  }
  if (s->p >= s->e) { fail(s, "unterminated string"); free(out); return NULL; }
  s->p++;
  out[n] = 0; *olen = n;
  return out;
}

static jnode *parse_value(jp *s);

static void push(jnode *n, jnode *kid, char *key) {
  n->kids = (jnode **)realloc(n->kids, (size_t)(n->n + 1) * sizeof(jnode *));
  n->kids[n->n] = kid;
  if (n->type == JN_OBJ) { n->keys = (char **)realloc(n->keys, (size_t)(n->n + 1) * sizeof(char *)); n->keys[n->n] = key; }
  n->n++;
}

static jnode *parse_value(jp *s) {
  ws(s);
  if (s->p >= s->e) { fail(s, "unexpected end"); return NULL; }
  jnode *n = (jnode *)calloc(1, sizeof *n);
  char c = *s->p;
  if (c == '{') {
    n->type = JN_OBJ; s->p++; ws(s);
    if (s->p < s->e && *s->p == '}') { s->p++; return n; }
    for (;;) {
      ws(s);
      size_t kl; char *k = parse_string(s, &kl);
      if (!k) break;
      ws(s);
      if (s->p >= s->e || *s->p != ':') { fail(s, "expected ':'"); free(k); break; }
      s->p++;
      jnode *v = parse_value(s);
      if (!v) { free(k); break; }
      push(n, v, k);
      ws(s);
      if (s->p < s->e && *s->p == ',') { s->p++; continue; }
      if (s->p < s->e && *s->p == '}') { s->p++; return n; }
      fail(s, "expected ',' or '}'"); break;
    }
    return n;
  }
  if (c == '[') {
    n->type = JN_ARR; s->p++; ws(s);
    if (s->p < s->e && *s->p == ']') { s->p++; return n; }
    for (;;) {
      jnode *v = parse_value(s);
      if (!v) break;
      push(n, v, NULL);
      ws(s);
      if (s->p < s->e && *s->p == ',') { s->p++; continue; }
      if (s->p < s->e && *s->p == ']') { s->p++; return n; }
      fail(s, "expected ',' or ']'"); break;
    }
    return n;
  }
  if (c == '"') { n->type = JN_STR; n->str = parse_string(s, &n->slen); return n; }
  if (!strncmp(s->p, "true", 4)) { n->type = JN_BOOL; n->b = 1; s->p += 4; return n; }
  if (!strncmp(s->p, "false", 5)) { n->type = JN_BOOL; n->b = 0; s->p += 5; return n; }
  if (!strncmp(s->p, "null", 4)) { n->type = JN_NULL; s->p += 4; return n; }
  if (c == '-' || isdigit((unsigned char)c)) {
    const char *b = s->p;
    while (s->p < s->e && (isdigit((unsigned char)*s->p) || strchr("+-.eE", *s->p))) s->p++;
    n->type = JN_NUM; n->slen = (size_t)(s->p - b);
    n->str = (char *)malloc(n->slen + 1); memcpy(n->str, b, n->slen); n->str[n->slen] = 0;
    n->num = strtod(n->str, NULL);
    return n;
  }
  fail(s, "unexpected character");
  return n;
}

jnode *jn_parse(const char *text, char *err, size_t errcap) {
  jp s = {text, text + strlen(text), err, errcap, 0};
  jnode *n = parse_value(&s);
  ws(&s);
  if (!s.failed && s.p != s.e) fail(&s, "trailing data");
  if (s.failed) { jn_free(n); return NULL; }
  return n;
}

void jn_free(jnode *n) {
  if (!n) return;
  for (int i = 0; i < n->n; i++) { jn_free(n->kids[i]); if (n->keys) free(n->keys[i]); }
  free(n->kids); free(n->keys); free(n->str); free(n);
}

const jnode *jn_get(const jnode *obj, const char *key) {
  if (!obj || obj->type != JN_OBJ) return NULL;
  for (int i = 0; i < obj->n; i++) if (!strcmp(obj->keys[i], key)) return obj->kids[i];
  return NULL;
}
const char *jn_str(const jnode *n, const char *dflt) { return (n && n->type == JN_STR) ? n->str : dflt; }
int jn_bool(const jnode *n, int dflt) { return (n && n->type == JN_BOOL) ? n->b : dflt; }
