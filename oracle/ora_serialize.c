/*
 * oracle/ora_serialize.c — CPU restatement of the sink-side marshalling
 * (SURVEY.md §8a rows a20, a21).  TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 *   TFGPU_FMT_CH_JSON_EACH_ROW  pkg/providers/clickhouse/httpuploader/marshal.go:82-419
 *   TFGPU_FMT_JSON              pkg/serializer/json.go:29-83 + encoding/json (sorted map keys,
 *                               SetEscapeHTML(false)), batch.go:206-219 (separator "\n")
 *   TFGPU_FMT_CSV               pkg/serializer/csv.go:22-74, csv_format.go:32-127 + encoding/csv
 *
 * Values are the Go dynamic types a strictified ChangeItem carries.  A value form whose
 * encoding is not restated here makes the whole call return NULL (the HIP side answers
 * TFGPU_ERR_UNSUPPORTED for the same inputs).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ora.h"

typedef struct { char *p; size_t n, cap; int unsupported; } sbuf;
static void sb_put(sbuf *b, const void *s, size_t n) {
  if (b->n + n + 1 > b->cap) { b->cap = (b->n + n + 1) * 2; b->p = (char *)realloc(b->p, b->cap); }
  memcpy(b->p + b->n, s, n); b->n += n;
}
static void sb_c(sbuf *b, char c) { sb_put(b, &c, 1); }
static void sb_s(sbuf *b, const char *s) { sb_put(b, s, strlen(s)); }

static const char HEXC[] = "0123456789abcdef";

/* writeQuoted (marshal.go:377-419): only ", \ and control bytes are escaped, raw bytes kept */
static void ch_write_quoted(sbuf *b, const char *s, size_t n) {
  sb_c(b, '"');
  for (size_t i = 0; i < n; i++) {
    unsigned char c = (unsigned char)s[i];
    if (c >= 0x20 && c != '"' && c != '\\') { sb_c(b, (char)c); continue; }
    switch (c) {
      case '"': sb_s(b, "\\\""); break; case '\\': sb_s(b, "\\\\"); break; case '\n': sb_s(b, "\\n"); break;
      case '\r': sb_s(b, "\\r"); break; case '\t': sb_s(b, "\\t"); break; case '\f': sb_s(b, "\\f"); break; case '\b': sb_s(b, "\\b"); break;
      default: sb_s(b, "\\u00"); sb_c(b, HEXC[c >> 4]); sb_c(b, HEXC[c & 15]);
    }
  }
  sb_c(b, '"');
}

/* encoding/json appendString with escapeHTML=false (encode.go): ", \, control bytes, invalid
 * UTF-8 -> �, U+2028/U+2029 always escaped */
static void json_string(sbuf *b, const char *s, size_t n, int escape_html) {
  sb_c(b, '"');
  size_t i = 0;
  while (i < n) {
    unsigned char c = (unsigned char)s[i];
    if (c < 0x80) {
      if (c >= 0x20 && c != '"' && c != '\\' && !(escape_html && (c == '<' || c == '>' || c == '&'))) { sb_c(b, (char)c); i++; continue; }
      switch (c) {
        case '"': sb_s(b, "\\\""); break; case '\\': sb_s(b, "\\\\"); break; case '\b': sb_s(b, "\\b"); break; case '\f': sb_s(b, "\\f"); break;
        case '\n': sb_s(b, "\\n"); break; case '\r': sb_s(b, "\\r"); break; case '\t': sb_s(b, "\\t"); break;
        default: sb_s(b, "\\u00"); sb_c(b, HEXC[c >> 4]); sb_c(b, HEXC[c & 15]);
      }
      i++; continue;
    }
    /* utf8.DecodeRuneInString */
    size_t need = 0; unsigned cp = 0; unsigned char lo = 0x80, hi = 0xBF;
    if (c >= 0xC2 && c <= 0xDF) { need = 1; cp = c & 0x1F; }
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; cp = c & 0x0F; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; cp = c & 0x07; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
    int ok = need > 0 && i + need < n;  /* all continuation bytes exist */
    if (ok) {
      for (size_t k = 1; k <= need; k++) {
        unsigned char d = (unsigned char)s[i + k];
        unsigned char l = k == 1 ? lo : 0x80, h = k == 1 ? hi : 0xBF;
        if (d < l || d > h) { ok = 0; break; }
        cp = (cp << 6) | (d & 0x3F);
      }
    }
    if (!ok) { sb_s(b, "\\ufffd"); i++; continue; }
    if (cp == 0x2028 || cp == 0x2029) { sb_s(b, cp == 0x2028 ? "\\u2028" : "\\u2029"); i += need + 1; continue; }
    sb_put(b, s + i, need + 1); i += need + 1;
  }
  sb_c(b, '"');
}

static void base64_std(sbuf *b, const unsigned char *s, size_t n) {
  static const char T[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  size_t i = 0;
  for (; i + 3 <= n; i += 3) { unsigned v = (s[i] << 16) | (s[i + 1] << 8) | s[i + 2]; sb_c(b, T[v >> 18]); sb_c(b, T[(v >> 12) & 63]); sb_c(b, T[(v >> 6) & 63]); sb_c(b, T[v & 63]); }
  if (n - i == 1) { unsigned v = s[i] << 16; sb_c(b, T[v >> 18]); sb_c(b, T[(v >> 12) & 63]); sb_s(b, "=="); }
  else if (n - i == 2) { unsigned v = (s[i] << 16) | (s[i + 1] << 8); sb_c(b, T[v >> 18]); sb_c(b, T[(v >> 12) & 63]); sb_c(b, T[(v >> 6) & 63]); sb_c(b, '='); }
}

static int is_int_kind(int k) { return k >= OV_I8 && k <= OV_U64; }
static void put_int(sbuf *b, const ora_value *v) {
  char t[32]; size_t n = (v->kind >= OV_U8 && v->kind <= OV_U64) ? ora_fmt_uint(t, v->v.u) : ora_fmt_int(t, v->v.i);
  sb_put(b, t, n);
}
static void put_float_f(sbuf *b, const ora_value *v) { /* strconv.FormatFloat(f, 'f', -1, bits) */
  char t[400]; size_t n = v->kind == OV_F32 ? ora_fmt_float(t, (double)v->v.f32, 'f', 32) : ora_fmt_float(t, v->v.f64, 'f', 64);
  sb_put(b, t, n);
}

/* strconv.ParseFloat range error for a JSON number literal: +1 / -1 when it overflows to ±Inf */
static int number_overflows(const char *s, size_t n) {
  double d; int rc = ora_parse_float(s, n, 64, &d);
  if (rc == 2) return d < 0 ? -1 : 1;
  return 0;
}

static uint32_t default_ch_flags(int dtype, uint8_t *prec) {
  *prec = 0;
  switch (dtype) { /* pkg/providers/clickhouse/typesystem.md */
    case TFGPU_T_BYTES: case TFGPU_T_UTF8: case TFGPU_T_ANY: return TFGPU_CH_STRING;
    case TFGPU_T_DATE: return TFGPU_CH_DATE;
    case TFGPU_T_TIMESTAMP: *prec = 9; return TFGPU_CH_DATETIME64;
    default: return 0;
  }
}

static int64_t pow10i(int k) { int64_t r = 1; while (k-- > 0) r *= 10; return r; }

/* marshalValue (marshal.go:140-253) for one non-nil value; returns 1 if the column must be skipped */
static int ch_value(sbuf *b, const ora_value *v, int dtype, uint32_t fl, uint8_t prec, int any_as_string) {
  char t[128]; size_t n;
  if ((fl & TFGPU_CH_DECIMAL) && (v->kind == OV_STRING || v->kind == OV_BYTES)) { sb_put(b, v->s, v->slen); return 0; }
  switch (dtype) {
    case TFGPU_T_INT8: case TFGPU_T_INT16: case TFGPU_T_INT32: case TFGPU_T_INT64: case TFGPU_T_UINT8: case TFGPU_T_UINT16:
    case TFGPU_T_UINT32: case TFGPU_T_UINT64: case TFGPU_T_FLOAT32: case TFGPU_T_FLOAT64: case TFGPU_T_INTERVAL: {
      /* marshalNumericValue :257-301 */
      int handled = 1;
      sbuf tmp = {0};
      if (is_int_kind(v->kind)) put_int(&tmp, v);
      else if (v->kind == OV_F32 || v->kind == OV_F64) put_float_f(&tmp, v);
      else if (v->kind == OV_JSONNUM) {
        int o = number_overflows(v->s, v->slen);
        if (o > 0) sb_s(&tmp, "inf"); else if (o < 0) sb_s(&tmp, "-inf"); else sb_put(&tmp, v->s, v->slen);
      } else handled = 0;
      if (handled) {
        if (fl & TFGPU_CH_STRING) sb_c(b, '"');
        sb_put(b, tmp.p ? tmp.p : "", tmp.n);
        if (fl & TFGPU_CH_STRING) sb_c(b, '"');
        free(tmp.p);
        return 0;
      }
      free(tmp.p);
      break;
    }
    case TFGPU_T_BYTES: case TFGPU_T_UTF8:
      if (v->kind == OV_STRING) { ch_write_quoted(b, v->s, v->slen); return 0; }
      if (v->kind == OV_BYTES) {
        if (fl & TFGPU_CH_ARRAY) { sb_c(b, '['); for (size_t i = 0; i < v->slen; i++) { if (i) sb_c(b, ','); n = ora_fmt_uint(t, (unsigned char)v->s[i]); sb_put(b, t, n); } sb_c(b, ']'); }
        else ch_write_quoted(b, v->s, v->slen);
        return 0;
      }
      break;
    case TFGPU_T_BOOLEAN:
      if (v->kind == OV_BOOL) { sb_s(b, v->v.b ? "true" : "false"); return 0; }
      break;
    case TFGPU_T_DATE: case TFGPU_T_DATETIME: case TFGPU_T_TIMESTAMP:
      if (v->kind == OV_TIME) { /* marshalTime :65-80 */
        if (fl & TFGPU_CH_STRING) { sb_c(b, '"'); n = ora_fmt_time_string(t, v->v.t.sec, v->v.t.nsec); sb_put(b, t, n); sb_c(b, '"'); }
        else if (fl & TFGPU_CH_DATETIME64) {
          int64_t full = v->v.t.sec * 1000000000LL + v->v.t.nsec;
          if (prec > 0 && prec < 9) full = full / pow10i(9 - prec);  /* Go integer division truncates toward zero */
          n = ora_fmt_int(t, full); sb_put(b, t, n);
        } else if (fl & TFGPU_CH_DATE) { sb_c(b, '"'); n = ora_fmt_date(t, v->v.t.sec); sb_put(b, t, n); sb_c(b, '"'); }
        else { n = ora_fmt_int(t, v->v.t.sec); sb_put(b, t, n); }
        return 0;
      }
      break;
    default: break;
  }
  /* marshalGeneric :318-359 */
  if (v->kind == OV_STRING) { ch_write_quoted(b, v->s, v->slen); return 0; }
  if (v->kind == OV_BYTES) {
    if (fl & TFGPU_CH_ARRAY) { sb_c(b, '['); for (size_t i = 0; i < v->slen; i++) { if (i) sb_c(b, ','); n = ora_fmt_uint(t, (unsigned char)v->s[i]); sb_put(b, t, n); } sb_c(b, ']'); }
    else ch_write_quoted(b, v->s, v->slen);
    return 0;
  }
  /* json.Marshal(v) of the remaining Go types */
  sbuf r = {0};
  if (is_int_kind(v->kind)) put_int(&r, v);
  else if (v->kind == OV_BOOL) sb_s(&r, v->v.b ? "true" : "false");
  else if (v->kind == OV_JSONNUM) { if (v->slen) sb_put(&r, v->s, v->slen); else sb_c(&r, '0'); }
  else if (v->kind == OV_JSON) sb_put(&r, v->s, v->slen);
  else if (v->kind == OV_DURATION) { n = ora_fmt_int(t, v->v.i); sb_put(&r, t, n); }
  else if (v->kind == OV_F32 || v->kind == OV_F64) { /* json.Marshal(float): goccy/go-json follows encoding/json's floatEncoder */
    n = v->kind == OV_F32 ? ora_json_float(t, (double)v->v.f32, 32) : ora_json_float(t, v->v.f64, 64);
    if (!n) { free(r.p); b->unsupported = 1; return 0; }
    sb_put(&r, t, n);
  }
  else { free(r.p); b->unsupported = 1; return 0; } /* time under a foreign DataType */
  if (r.n == 4 && !memcmp(r.p, "null", 4)) { free(r.p); return 1; }
  if (dtype != TFGPU_T_ANY || any_as_string || (fl & TFGPU_CH_STRING)) {
    /* DOUBLE_MARSHAL: json.Marshal(string(r)) — goccy/go-json escapes HTML like encoding/json */
    json_string(b, r.p, r.n, 1);
  } else sb_put(b, r.p, r.n);
  free(r.p);
  return 0;
}

static int dtype_of(const ora_item *it, int i) {
  /* MarshalCItoJSON looks the ColSchema up by column NAME (gfMap, marshal.go:92-98) */
  const char *name = it->names && i < it->names->n ? it->names->names[i] : NULL;
  if (it->schema && name) for (int k = 0; k < it->schema->ncols; k++) if (!strcmp(it->schema->cols[k].name, name)) return it->schema->cols[k].dtype;
  return TFGPU_T_INVALID;
}
static int dtype_by_index(const ora_item *it, int i) {
  /* the generic serializers pair values with columns[i] (json.go:35-39, csv.go:28-34) */
  if (it->schema && i < it->schema->ncols) return it->schema->cols[i].dtype;
  return TFGPU_T_INVALID;
}

static void ch_row(sbuf *b, const ora_item *it, const tfgpu_serialize_options *o) {
  sb_c(b, '{');
  int has = 0;
  for (int i = 0; i < it->nvalues; i++) {
    const ora_value *v = &it->values[i];
    if (v->kind == OV_NIL) continue;
    const char *name = it->names->names[i];
    size_t mark = b->n;
    sb_c(b, '"'); sb_s(b, name); sb_c(b, '"'); sb_c(b, ':');  /* writeColName: the name is NOT escaped */
    int dt = dtype_of(it, i);
    uint8_t prec = 0; uint32_t fl;
    if (o && o->ncols > i && o->ch_flags) { fl = o->ch_flags[i]; prec = o->ch_precision ? o->ch_precision[i] : 0; }
    else fl = default_ch_flags(dt, &prec);
    if (ch_value(b, v, dt, fl, prec, o ? o->any_as_string : 0)) { b->n = mark; continue; }
    sb_c(b, ','); has = 1;
  }
  if (has) b->n--;
  sb_c(b, '}'); sb_c(b, '\n');
}

/* ---- encoding/json of one value of a strictified row ---- */
static void json_value(sbuf *b, const ora_value *v, int dtype, int any_as_string, int strictified) {
  char t[128]; size_t n;
  /* strictified: the batch / stream serializers wrap jsonSerializer in strictifyingSerializer (batch_factory / strictify.go);
   * the queue JSON serializer calls NewJSONSerializer bare (queue/json_serializer.go:22-29) and writes values as they are */
  if (!strictified && (dtype == TFGPU_T_BYTES || dtype == TFGPU_T_FLOAT64)) dtype = TFGPU_T_INVALID;
  switch (v->kind) {
    case OV_NIL: sb_s(b, "null"); return;
    case OV_BOOL: sb_s(b, v->v.b ? "true" : "false"); return;
    case OV_STRING:
      if (dtype != TFGPU_T_BYTES) { json_string(b, v->s, v->slen, 0); return; }
      /* strictify first (serializer/strictify.go:24-36): castx.ToByteSliceE turns a Go string under "string" into []byte */
      /* fallthrough */
    case OV_BYTES: sb_c(b, '"'); base64_std(b, (const unsigned char *)v->s, v->slen); sb_c(b, '"'); return;
    case OV_JSONNUM: if (v->slen) sb_put(b, v->s, v->slen); else sb_c(b, '0'); return;
    case OV_TIME: sb_c(b, '"'); n = ora_fmt_rfc3339nano(t, v->v.t.sec, v->v.t.nsec); sb_put(b, t, n); sb_c(b, '"'); return; /* Time.MarshalJSON */
    case OV_DURATION: n = ora_fmt_int(t, v->v.i); sb_put(b, t, n); return;
    case OV_JSON:
      if (dtype == TFGPU_T_ANY && any_as_string) json_string(b, v->s, v->slen, 0); /* toJsonValue: string(valueData) */
      else sb_put(b, v->s, v->slen);
      return;
    default: break;
  }
  if (is_int_kind(v->kind)) { put_int(b, v); return; }
  if ((v->kind == OV_F32 || v->kind == OV_F64) && dtype == TFGPU_T_FLOAT64) {
    /* strictify first: "double" becomes json.Number(castx.ToStringE(v)) = FormatFloat(v, 'f', -1, bits) (strictify.go:119-124,
     * caste.go:36-49, 59-62), written as it stands; "NaN" / "+Inf" are no JSON number literals: Marshal fails */
    const double d = v->kind == OV_F32 ? (double)v->v.f32 : v->v.f64;
    if (d != d || d - d != 0) b->unsupported = 1; else put_float_f(b, v);
    return;
  }
  if (v->kind == OV_F32 || v->kind == OV_F64) { /* floatEncoder; NaN / Inf fail the whole Marshal (UnsupportedValueError) */
    n = v->kind == OV_F32 ? ora_json_float(t, (double)v->v.f32, 32) : ora_json_float(t, v->v.f64, 64);
    if (!n) b->unsupported = 1; else sb_put(b, t, n);
    return;
  }
  b->unsupported = 1;
}

typedef struct { const char *name; int idx; } keyref;
static int key_cmp(const void *a, const void *b) { return strcmp(((const keyref *)a)->name, ((const keyref *)b)->name); }

static void json_row(sbuf *b, const ora_item *it, const tfgpu_serialize_options *o, int strictified) {
  int n = it->nvalues;
  keyref *k = (keyref *)malloc(sizeof(keyref) * (size_t)(n ? n : 1));
  for (int i = 0; i < n; i++) { k[i].name = it->names->names[i]; k[i].idx = i; }
  qsort(k, (size_t)n, sizeof(keyref), key_cmp);  /* map keys are emitted sorted */
  sb_c(b, '{');
  for (int j = 0; j < n; j++) {
    if (j) sb_c(b, ',');
    json_string(b, k[j].name, strlen(k[j].name), 0);
    sb_c(b, ':');
    json_value(b, &it->values[k[j].idx], dtype_by_index(it, k[j].idx), o ? o->any_as_string : 0, strictified);
  }
  sb_c(b, '}');
  free(k);
}

/* what json.Marshal's compact(escape=true) does to bytes a Marshaler returned: <, >, & and U+2028/9 become \uXXXX */
static void put_html_compact(sbuf *b, const char *s, size_t n) {
  for (size_t i = 0; i < n; i++) {
    unsigned char c = (unsigned char)s[i];
    if (c == '<' || c == '>' || c == '&') { sb_s(b, "\\u00"); sb_c(b, HEXC[c >> 4]); sb_c(b, HEXC[c & 15]); }
    else if (c == 0xE2 && i + 2 < n && (unsigned char)s[i + 1] == 0x80 && ((unsigned char)s[i + 2] & 0xFE) == 0xA8) { sb_s(b, "\\u202"); sb_c(b, HEXC[(unsigned char)s[i + 2] & 0xF]); i += 2; }
    else sb_c(b, (char)c);
  }
}

/* encoding/csv Writer.fieldNeedsQuotes + quoted write (Comma ',', UseCRLF false) */
static void csv_field(sbuf *b, const char *s, size_t n) {
  int need = 0;
  if (n == 0) need = 0;
  else if (n == 2 && s[0] == '\\' && s[1] == '.') need = 1;
  else {
    for (size_t i = 0; i < n; i++) if (s[i] == ',' || s[i] == '"' || s[i] == '\r' || s[i] == '\n') { need = 1; break; }
    if (!need) { /* unicode.IsSpace(first rune) */
      unsigned char c = (unsigned char)s[0];
      if (c == ' ' || (c >= 9 && c <= 13)) need = 1;
      else if (c == 0xC2 && n >= 2 && ((unsigned char)s[1] == 0x85 || (unsigned char)s[1] == 0xA0)) need = 1;
      else if (n >= 3 && (c == 0xE1 || c == 0xE2 || c == 0xE3)) {
        unsigned char d = (unsigned char)s[1], e = (unsigned char)s[2];
        if (c == 0xE1 && d == 0x9A && e == 0x80) need = 1;
        if (c == 0xE3 && d == 0x80 && e == 0x80) need = 1;
        if (c == 0xE2 && ((d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) || (d == 0x81 && e == 0x9F))) need = 1;
      }
    }
  }
  if (!need) { sb_put(b, s, n); return; }
  sb_c(b, '"');
  for (size_t i = 0; i < n; i++) { if (s[i] == '"') sb_s(b, "\"\""); else sb_c(b, s[i]); }
  sb_c(b, '"');
}

static void csv_row(sbuf *b, const ora_item *it) {
  for (int i = 0; i < it->nvalues; i++) {
    if (i) sb_c(b, ',');
    const ora_value *v = &it->values[i];
    int dt = dtype_by_index(it, i);
    sbuf c = {0}; char t[128]; size_t n;
    if (v->kind == OV_NIL) { /* "" */ }
    else if (dt == TFGPU_T_BYTES) { if (v->kind == OV_BYTES || v->kind == OV_STRING) base64_std(&c, (const unsigned char *)v->s, v->slen); else b->unsupported = 1; }
    else if (dt == TFGPU_T_ANY) { /* json.Marshal(value) */
      if (v->kind == OV_JSON) put_html_compact(&c, v->s, v->slen);  /* Marshal escapes HTML; the JSON serializer's Encoder does not */
      else if (v->kind == OV_JSONNUM) sb_put(&c, v->s, v->slen);
      else if (is_int_kind(v->kind)) put_int(&c, v);
      else if (v->kind == OV_BOOL) sb_s(&c, v->v.b ? "true" : "false");
      else if (v->kind == OV_STRING) json_string(&c, v->s, v->slen, 1);
      else b->unsupported = 1;
    } else switch (v->kind) { /* castx.ToStringE caste.go:57-106 */
      case OV_STRING: case OV_BYTES: case OV_JSONNUM: sb_put(&c, v->s, v->slen); break;
      case OV_BOOL: sb_s(&c, v->v.b ? "true" : "false"); break;
      case OV_F32: case OV_F64: put_float_f(&c, v); break;
      case OV_TIME: n = ora_fmt_time_string(t, v->v.t.sec, v->v.t.nsec); sb_put(&c, t, n); break; /* fmt.Stringer */
      case OV_DURATION: n = ora_fmt_duration(t, v->v.i); sb_put(&c, t, n); break;
      case OV_JSON: b->unsupported = 1; break; /* maps/slices under a non-any type: cast.ToStringE error */
      default: if (is_int_kind(v->kind)) put_int(&c, v); else b->unsupported = 1;
    }
    csv_field(b, c.p ? c.p : "", c.n);
    free(c.p);
  }
  sb_c(b, '\n');
}

char *ora_serialize_ex(int format, const ora_batch *bt, const tfgpu_serialize_options *o, uint64_t *len) {
  sbuf b = {0};
  b.p = (char *)malloc(64); b.cap = 64;
  for (int64_t r = 0; r < bt->n; r++) {
    const ora_item *it = &bt->items[r];
    if (it->kind > TFGPU_K_DELETE) continue; /* !IsRowEvent */
    switch (format) {
      case TFGPU_FMT_CH_JSON_EACH_ROW: ch_row(&b, it, o); break;
      case TFGPU_FMT_JSON:
        json_row(&b, it, o, 1);
        if (o && o->add_closing_newline) sb_c(&b, '\n');   /* per item, json.go:66-72; batch separator is nil then */
        else if (r + 1 < bt->n) sb_c(&b, '\n');            /* separator between items, none after the last */
        break;
      case TFGPU_FMT_CSV: csv_row(&b, it); break;
      case TFGPU_FMT_RAW: { /* rawSerializer.Serialize raw.go:24-39 under batchSerializer (separator "\n" unless AddClosingNewLine) */
        static const char *const MIRROR[] = {"topic", "partition", "seq_no", "write_time", "data", "meta", "sequence_key"};
        int mirror = it->names && it->names->n == 7;
        for (int c = 0; mirror && c < 7; c++) mirror = !strcmp(it->names->names[c], MIRROR[c]);
        if (!mirror || !(it->values[4].kind == OV_BYTES || it->values[4].kind == OV_STRING)) { b.unsupported = 1; break; }
        sb_put(&b, it->values[4].s, it->values[4].slen);
        if ((o && o->add_closing_newline) || r + 1 < bt->n) sb_c(&b, '\n');
        break;
      }
      default: b.unsupported = 1;
    }
  }
  if (b.unsupported) { free(b.p); *len = 0; return NULL; }
  *len = b.n;
  return b.p;
}
char *ora_serialize(int format, const ora_batch *b, uint64_t *len) { return ora_serialize_ex(format, b, NULL, len); }

/* ======================================================================================================
 * a24  abstract.Collapse — pkg/abstract/changeitem/change_item_collapse.go:48-134, with
 *      OldOrCurrentKeysString / CurrentKeysString (change_item.go:314-358), MakeMapKeys, compareColumns (:7-35).
 * Go maps are restated as a string-keyed open-addressing table; toDelete, which the reference emits in map
 * (= unspecified) order, is emitted in the input order of the delete items.
 * ====================================================================================================== */
typedef struct { char **keys; int *vals; int cap, n; } smap;  /* vals: >= 0 live, -1 empty, -2 deleted */
static uint32_t smap_hash(const char *k) { uint32_t h = 2166136261u; for (; *k; k++) h = (h ^ (unsigned char)*k) * 16777619u; return h; }
static void smap_init(smap *m, int cap) { m->cap = 16; while (m->cap < cap * 4) m->cap <<= 1; m->n = 0; m->keys = (char **)calloc((size_t)m->cap, sizeof(char *)); m->vals = (int *)malloc((size_t)m->cap * sizeof(int)); for (int i = 0; i < m->cap; i++) m->vals[i] = -1; }
static void smap_free(smap *m) { for (int i = 0; i < m->cap; i++) free(m->keys[i]); free(m->keys); free(m->vals); }
static int smap_slot(const smap *m, const char *k, int for_insert) {
  int first_free = -1;
  for (uint32_t s = smap_hash(k) & (uint32_t)(m->cap - 1);; s = (s + 1) & (uint32_t)(m->cap - 1)) {
    if (m->vals[s] == -1) return for_insert ? (first_free >= 0 ? first_free : (int)s) : -1;
    if (m->keys[s] && !strcmp(m->keys[s], k)) { if (m->vals[s] >= 0) return (int)s; if (for_insert && first_free < 0) first_free = (int)s; }
    else if (m->vals[s] == -2 && for_insert && first_free < 0) first_free = (int)s;
  }
}
static int smap_get(const smap *m, const char *k) { int s = smap_slot(m, k, 0); return s < 0 ? -1 : m->vals[s]; }
static void smap_set(smap *m, const char *k, int v) { int s = smap_slot(m, k, 1); if (!m->keys[s] || strcmp(m->keys[s], k)) { free(m->keys[s]); m->keys[s] = strdup(k); } m->vals[s] = v; }
static void smap_del(smap *m, const char *k) { int s = smap_slot(m, k, 0); if (s >= 0) m->vals[s] = -2; }

static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return x < y ? -1 : x > y; }
static int cmp_cstr(const void *a, const void *b) { return strcmp(*(const char *const *)a, *(const char *const *)b); }

/* json.Marshal([]interface{}{…}) of the key values in sorted key-name order; values looked up by name */
static char *keys_string(char **keynames, int nkeys, char **names, int nnames, const ora_value *values) {
  sbuf b = {0};
  sb_c(&b, '[');
  for (int k = 0; k < nkeys; k++) {
    if (k) sb_c(&b, ',');
    const ora_value *v = NULL;
    for (int i = 0; i < nnames; i++) if (!strcmp(names[i], keynames[k])) v = &values[i];  /* a later duplicate name wins, like the map write */
    if (!v || v->kind == OV_NIL) sb_s(&b, "null"); else json_value(&b, v, TFGPU_T_INVALID, 0, 0);
  }
  sb_c(&b, ']');
  sb_c(&b, 0);
  return b.p;
}
static char *current_keys_string(const ora_item *c, char **kn, int nk) { return keys_string(kn, nk, c->names ? c->names->names : NULL, c->names ? c->names->n : 0, c->values); }
static char *old_or_current_keys_string(const ora_item *c, char **kn, int nk) {
  if ((c->kind == TFGPU_K_UPDATE || c->kind == TFGPU_K_DELETE) && c->n_old > 0) return keys_string(kn, nk, c->old_names->names, c->n_old, c->old_values);
  return current_keys_string(c, kn, nk);
}
static void item_clone(ora_item *d, const ora_item *s) {
  memset(d, 0, sizeof *d);
  d->kind = s->kind; d->ns = strdup(s->ns ? s->ns : ""); d->table = strdup(s->table ? s->table : ""); d->part_id = strdup(s->part_id ? s->part_id : "");
  d->names = s->names; if (d->names) d->names->refs++;
  d->schema = s->schema; if (d->schema) d->schema->refs++;
  d->nvalues = s->nvalues; d->values = (ora_value *)calloc((size_t)(s->nvalues ? s->nvalues : 1), sizeof(ora_value));
  for (int i = 0; i < s->nvalues; i++) d->values[i] = ora_value_clone(&s->values[i]);
  d->src_row = s->src_row;
  d->old_names = s->old_names; if (d->old_names) d->old_names->refs++;
  d->n_old = s->n_old;
  if (s->n_old) { d->old_values = (ora_value *)calloc((size_t)s->n_old, sizeof(ora_value)); for (int i = 0; i < s->n_old; i++) d->old_values[i] = ora_value_clone(&s->old_values[i]); }
}
static void item_set_values_from(ora_item *cur, const ora_item *c) {  /* compareColumns + merge (:86-102) */
  int same = cur->names == c->names;
  if (!same && cur->names && c->names && cur->names->n == c->names->n) { same = 1; for (int i = 0; i < c->names->n; i++) if (strcmp(cur->names->names[i], c->names->names[i])) { same = 0; break; } }
  if (same) {
    for (int i = 0; i < cur->nvalues; i++) ora_value_free(&cur->values[i]);
    free(cur->values);
    cur->nvalues = c->nvalues; cur->values = (ora_value *)calloc((size_t)(c->nvalues ? c->nvalues : 1), sizeof(ora_value));
    for (int i = 0; i < c->nvalues; i++) cur->values[i] = ora_value_clone(&c->values[i]);
    return;
  }
  /* total = old names, then the new ones not seen; values: old, overwritten by new */
  int no = cur->names ? cur->names->n : 0, nn = c->names ? c->names->n : 0;
  ora_names *tot = (ora_names *)calloc(1, sizeof *tot);
  tot->refs = 1; tot->names = (char **)calloc((size_t)(no + nn + 1), sizeof(char *));
  for (int i = 0; i < no; i++) tot->names[tot->n++] = strdup(cur->names->names[i]);
  for (int i = 0; i < nn; i++) { int seen = 0; for (int j = 0; j < no; j++) if (!strcmp(cur->names->names[j], c->names->names[i])) seen = 1; if (!seen) tot->names[tot->n++] = strdup(c->names->names[i]); }
  ora_value *vals = (ora_value *)calloc((size_t)(tot->n ? tot->n : 1), sizeof(ora_value));
  for (int t = 0; t < tot->n; t++) {
    for (int j = 0; j < no; j++) if (!strcmp(cur->names->names[j], tot->names[t])) { ora_value_free(&vals[t]); vals[t] = ora_value_clone(&cur->values[j]); }  /* oldM: last index wins */
    for (int j = 0; j < nn; j++) if (!strcmp(c->names->names[j], tot->names[t])) { ora_value_free(&vals[t]); vals[t] = ora_value_clone(&c->values[j]); }
  }
  for (int i = 0; i < cur->nvalues; i++) ora_value_free(&cur->values[i]);
  free(cur->values);
  if (cur->names && --cur->names->refs == 0) { for (int i = 0; i < cur->names->n; i++) free(cur->names->names[i]); free(cur->names->names); free(cur->names); }
  cur->names = tot; cur->nvalues = tot->n; cur->values = vals;
}

ora_batch *ora_collapse(const ora_batch *in) {
  ora_batch *out = ora_batch_new();
  int passthrough = in->n < 2;
  if (!passthrough) { passthrough = 1; for (int64_t i = 0; i < in->n; i++) if (in->items[i].kind != TFGPU_K_INSERT && in->items[i].kind != TFGPU_K_SYNCHRONIZE) { passthrough = 0; break; } }  /* InsertsOnly :37-44 */
  /* keyCols := input[0].MakeMapKeys(): the PrimaryKey columns of its TableSchema */
  char **kn = NULL; int nk = 0;
  if (!passthrough && in->items[0].schema) {
    const ora_schema *sc = in->items[0].schema;
    kn = (char **)calloc((size_t)(sc->ncols ? sc->ncols : 1), sizeof(char *));
    for (int i = 0; i < sc->ncols; i++) if (sc->cols[i].key) { int dup = 0; for (int j = 0; j < nk; j++) if (!strcmp(kn[j], sc->cols[i].name)) dup = 1; if (!dup) kn[nk++] = sc->cols[i].name; }
    qsort(kn, (size_t)nk, sizeof(char *), cmp_cstr);  /* util.MapKeysInOrder */
  }
  if (!passthrough && nk == 0) passthrough = 1;
  if (passthrough) { for (int64_t i = 0; i < in->n; i++) item_clone(ora_batch_push(out), &in->items[i]); free(kn); return out; }

  const int n = (int)in->n;
  smap rows, to_delete, k2idx;
  smap_init(&rows, n); smap_init(&to_delete, n); smap_init(&k2idx, n);
  ora_item *store = (ora_item *)calloc((size_t)n, sizeof(ora_item));      /* the ChangeItem values held by `rows` / `toDelete` (slot = creating row) */
  char **idx2k = (char **)calloc((size_t)n, sizeof(char *));
  for (int i = 0; i < n; i++) {
    const ora_item *c = &in->items[i];
    char *hashK = old_or_current_keys_string(c, kn, nk);
    switch (c->kind) {
      case TFGPU_K_INSERT:
        smap_del(&to_delete, hashK);
        item_clone(&store[i], c);
        smap_set(&rows, hashK, i); smap_set(&k2idx, hashK, i); idx2k[i] = strdup(hashK);
        break;
      case TFGPU_K_UPDATE: {
        smap_del(&to_delete, hashK);
        int cur = smap_get(&rows, hashK);
        if (cur < 0) {
          char *nk_ = current_keys_string(c, kn, nk);
          item_clone(&store[i], c);
          smap_set(&rows, nk_, i); smap_set(&k2idx, nk_, i); idx2k[i] = nk_;
          break;
        }
        item_set_values_from(&store[cur], c);
        char *nk_ = old_or_current_keys_string(&store[cur], kn, nk);
        if (strcmp(nk_, hashK)) smap_del(&rows, hashK);
        smap_set(&rows, nk_, cur); smap_set(&k2idx, nk_, i); idx2k[i] = nk_;
        break;
      }
      case TFGPU_K_DELETE: {
        int cur = smap_get(&rows, hashK);
        smap_del(&rows, hashK);
        item_clone(&store[i], c);
        if (cur >= 0 && store[cur].n_old > 0) {  /* c.OldKeys = current.OldKeys */
          for (int q = 0; q < store[i].n_old; q++) ora_value_free(&store[i].old_values[q]);
          free(store[i].old_values);
          if (store[i].old_names && --store[i].old_names->refs == 0) { for (int q = 0; q < store[i].old_names->n; q++) free(store[i].old_names->names[q]); free(store[i].old_names->names); free(store[i].old_names); }
          store[i].old_names = store[cur].old_names; store[i].old_names->refs++;
          store[i].n_old = store[cur].n_old;
          store[i].old_values = (ora_value *)calloc((size_t)store[cur].n_old, sizeof(ora_value));
          for (int q = 0; q < store[cur].n_old; q++) store[i].old_values[q] = ora_value_clone(&store[cur].old_values[q]);
          free(hashK);
          hashK = old_or_current_keys_string(&store[i], kn, nk);
        }
        smap_set(&to_delete, hashK, i);
        break;
      }
      default:
        item_clone(ora_batch_push(out), c);
    }
    free(hashK);
  }
  /* rows in the order of hashKToIdx; an entity filed under key k sits in store[rows[k]] */
  int *order = (int *)malloc((size_t)n * sizeof(int) * 2);
  int no = 0;
  for (int s = 0; s < rows.cap; s++) if (rows.vals[s] >= 0) { order[2 * no] = smap_get(&k2idx, rows.keys[s]); order[2 * no + 1] = rows.vals[s]; no++; }
  qsort(order, (size_t)no, 2 * sizeof(int), cmp_int);
  for (int a = 0; a < no; a++) item_clone(ora_batch_push(out), &store[order[2 * a + 1]]);
  /* toDelete: map order in the reference; here by the delete item's position */
  int nd = 0;
  for (int s = 0; s < to_delete.cap; s++) if (to_delete.vals[s] >= 0) order[nd++] = to_delete.vals[s];
  qsort(order, (size_t)nd, sizeof(int), cmp_int);
  for (int a = 0; a < nd; a++) item_clone(ora_batch_push(out), &store[order[a]]);
  free(order);
  for (int i = 0; i < n; i++) { if (store[i].values || store[i].ns) ora_item_clear(&store[i]); free(idx2k[i]); }
  free(store); free(idx2k); free(kn);
  smap_free(&rows); smap_free(&to_delete); smap_free(&k2idx);
  return out;
}

/* ---- ChangeItem.KeysChanged (change_item.go:237-286) and SplitUpdatedPKeys (utils.go:75-128) ---- */
static int value_deep_equal(const ora_value *a, const ora_value *b) {  /* reflect.DeepEqual of two boxed scalars */
  const int an = !a || a->kind == OV_NIL, bn = !b || b->kind == OV_NIL;
  if (an || bn) return an && bn;
  if (a->kind != b->kind) return 0;  /* different dynamic types */
  switch (a->kind) {
    case OV_STRING: case OV_BYTES: case OV_JSONNUM: case OV_JSON: return a->slen == b->slen && !memcmp(a->s, b->s, a->slen);
    case OV_F64: return a->v.f64 == b->v.f64;
    case OV_F32: return a->v.f32 == b->v.f32;
    case OV_BOOL: return !a->v.b == !b->v.b;
    case OV_TIME: return a->v.t.sec == b->v.t.sec && a->v.t.nsec == b->v.t.nsec;
    default: return a->v.u == b->v.u;
  }
}
int ora_item_keys_changed(const ora_item *c) {
  if (c->kind != TFGPU_K_UPDATE) return 0;
  /* the len(KeyNames) != len(KeyValues) / len(ColumnNames) != len(ColumnValues) guards cannot fail in this row model */
  const ora_schema *sc = c->schema;
  for (int i = 0; sc && i < sc->ncols; i++) {
    if (!sc->cols[i].key) continue;
    const ora_value *ov = NULL, *nv = NULL;
    for (int j = 0; j < c->n_old; j++) if (!strcmp(c->old_names->names[j], sc->cols[i].name)) { ov = &c->old_values[j]; break; }
    for (int j = 0; c->names && j < c->names->n && j < c->nvalues; j++) if (!strcmp(c->names->names[j], sc->cols[i].name)) { nv = &c->values[j]; break; }
    if (!value_deep_equal(ov, nv)) return 1;
  }
  return 0;
}
void ora_keys_changed(const ora_batch *b, uint8_t *out) { for (int64_t i = 0; i < b->n; i++) out[i] = (uint8_t)ora_item_keys_changed(&b->items[i]); }
/* All sublists back to back in one batch; lens[k] = length of sublist k (caller frees with free()); returns the count */
ora_batch *ora_split_updated_pkeys(const ora_batch *in, int64_t **lens_out, int64_t *nlists) {
  ora_batch *out = ora_batch_new();
  int64_t *lens = (int64_t *)calloc((size_t)(2 * in->n + 2), sizeof(int64_t)), nl = 0, cur = 0;
  for (int64_t i = 0; i < in->n; i++) {
    const ora_item *c = &in->items[i];
    if (!ora_item_keys_changed(c)) { item_clone(ora_batch_push(out), c); cur++; continue; }
    if (cur > 0) { lens[nl++] = cur; cur = 0; }
    ora_item *d = ora_batch_push(out);  /* Delete: OldKeys kept, ColumnNames / ColumnValues nil */
    item_clone(d, c);
    d->kind = TFGPU_K_DELETE;
    for (int q = 0; q < d->nvalues; q++) ora_value_free(&d->values[q]);
    d->nvalues = 0;
    if (d->names && --d->names->refs == 0) { for (int q = 0; q < d->names->n; q++) free(d->names->names[q]); free(d->names->names); free(d->names); }
    d->names = (ora_names *)calloc(1, sizeof(ora_names)); d->names->refs = 1; d->names->names = (char **)calloc(1, sizeof(char *));
    ora_item *n = ora_batch_push(out);  /* Insert: values kept, OldKeys = EmptyOldKeys() */
    item_clone(n, c);
    n->kind = TFGPU_K_INSERT;
    for (int q = 0; q < n->n_old; q++) ora_value_free(&n->old_values[q]);
    free(n->old_values); n->old_values = NULL; n->n_old = 0;
    if (n->old_names && --n->old_names->refs == 0) { for (int q = 0; q < n->old_names->n; q++) free(n->old_names->names[q]); free(n->old_names->names); free(n->old_names); }
    n->old_names = NULL;
    lens[nl++] = 2;
  }
  if (cur > 0) lens[nl++] = cur;
  *lens_out = lens; *nlists = nl;
  return out;
}

/* ---- row-wise test I/O for Collapse: items whose ColumnNames differ (TOAST updates) have no columnar form ---- */
#include "ora_json.h"
static ora_value value_from_json(const jnode *pair) {  /* ["gotype", value] */
  ora_value v; memset(&v, 0, sizeof v);
  if (!pair || pair->type != JN_ARR || pair->n != 2) return v;
  const char *g = jn_str(pair->kids[0], "nil");
  const jnode *x = pair->kids[1];
  if (!strcmp(g, "nil")) return v;
  if (!strcmp(g, "string") || !strcmp(g, "json")) { v.kind = !strcmp(g, "json") ? OV_JSON : OV_STRING; v.slen = x->slen; v.s = (char *)malloc(x->slen + 1); memcpy(v.s, x->str, x->slen); v.s[x->slen] = 0; return v; }
  if (!strcmp(g, "bool")) { v.kind = OV_BOOL; v.v.b = jn_bool(x, 0); return v; }
  if (!strcmp(g, "float64")) { v.kind = OV_F64; v.v.f64 = x->num; return v; }
  int64_t i = 0; ora_parse_int(x->str, x->slen, 10, 64, &i);  /* JN_NUM keeps its text */
  if (!strcmp(g, "int") || !strcmp(g, "int64")) { v.kind = OV_I64; v.v.i = i; }
  else if (!strcmp(g, "int32")) { v.kind = OV_I32; v.v.i = i; }
  else if (!strcmp(g, "uint64")) { v.kind = OV_U64; v.v.u = (uint64_t)i; }
  return v;
}
static ora_names *names_from_json(const jnode *arr) {
  ora_names *nm = (ora_names *)calloc(1, sizeof *nm);
  nm->refs = 1; nm->n = arr ? arr->n : 0; nm->names = (char **)calloc((size_t)(nm->n ? nm->n : 1), sizeof(char *));
  for (int i = 0; i < nm->n; i++) nm->names[i] = strdup(jn_str(arr->kids[i], ""));
  return nm;
}
/* {"items": [{"kind": "insert|update|delete|other", "keys": ["pk names of its TableSchema"], "names": [...], "values": [["int",1],...],
 *             "old_names": [...], "old_values": [...]}]} */
ora_batch *ora_batch_from_json(const char *text) {
  char err[128];
  jnode *root = jn_parse(text, err, sizeof err);
  if (!root) return NULL;
  ora_batch *b = ora_batch_new();
  const jnode *items = jn_get(root, "items");
  for (int i = 0; items && i < items->n; i++) {
    const jnode *it = items->kids[i];
    ora_item *o = ora_batch_push(b);
    const char *k = jn_str(jn_get(it, "kind"), "insert");
    o->kind = !strcmp(k, "insert") ? TFGPU_K_INSERT : !strcmp(k, "update") ? TFGPU_K_UPDATE : !strcmp(k, "delete") ? TFGPU_K_DELETE : TFGPU_K_OTHER;
    o->ns = strdup(""); o->table = strdup("t"); o->part_id = strdup("");
    o->names = names_from_json(jn_get(it, "names"));
    const jnode *vals = jn_get(it, "values");
    o->nvalues = o->names->n; o->values = (ora_value *)calloc((size_t)(o->nvalues ? o->nvalues : 1), sizeof(ora_value));
    for (int c = 0; c < o->nvalues; c++) o->values[c] = value_from_json(vals && c < vals->n ? vals->kids[c] : NULL);
    const jnode *keys = jn_get(it, "keys");
    ora_schema *sc = (ora_schema *)calloc(1, sizeof *sc);
    sc->refs = 1; sc->ncols = keys ? keys->n : 0; sc->cols = (ora_colschema *)calloc((size_t)(sc->ncols ? sc->ncols : 1), sizeof(ora_colschema));
    for (int c = 0; c < sc->ncols; c++) { sc->cols[c].name = strdup(jn_str(keys->kids[c], "")); sc->cols[c].key = 1; sc->cols[c].path = strdup(""); sc->cols[c].original_type = strdup(""); }
    o->schema = sc;
    const jnode *on = jn_get(it, "old_names"), *ov = jn_get(it, "old_values");
    if (on && on->n) {
      o->old_names = names_from_json(on);
      o->n_old = on->n; o->old_values = (ora_value *)calloc((size_t)on->n, sizeof(ora_value));
      for (int c = 0; c < on->n; c++) o->old_values[c] = value_from_json(ov && c < ov->n ? ov->kids[c] : NULL);
    }
    o->src_row = i;
  }
  jn_free(root);
  return b;
}
int ora_batch_item_info(const ora_batch *b, int64_t row, int *kind, int *nvalues, int *n_old, int64_t *src_row) {
  if (row < 0 || row >= b->n) return -1;
  *kind = b->items[row].kind; *nvalues = b->items[row].nvalues; *n_old = b->items[row].n_old; *src_row = b->items[row].src_row;
  return 0;
}
const char *ora_batch_item_name(const ora_batch *b, int64_t row, int col, int old) {
  const ora_names *nm = old ? b->items[row].old_names : b->items[row].names;
  return (nm && col < nm->n) ? nm->names[col] : "";
}
int ora_batch_old_value(const ora_batch *b, int64_t row, int col, int *kind, int64_t *i64, const char **s, size_t *slen) {
  if (row < 0 || row >= b->n || col < 0 || col >= b->items[row].n_old) return -1;
  const ora_value *v = &b->items[row].old_values[col];
  *kind = v->kind; *i64 = v->v.i; *s = v->s; *slen = v->slen;
  return 0;
}
int64_t ora_batch_len(const ora_batch *b) { return b->n; }

/* ======================================================================================================
 * §8f.4  queue serializers — pkg/serializer/queue:
 *   NativeSerializer.serializeOneTableID (native_serializer.go:14-26), BatchNative (native_batcher.go:10-63),
 *   ChangeItem.ToJSONString / MarshalJSON (change_item.go:452-455, 568-616) under json.Marshal's compact();
 *   JSONSerializer.serializeOneTableID (json_serializer.go:38-59), BatchJSON (json_batcher.go:11-66) over
 *   jsonSerializer.Serialize (pkg/serializer/json.go:50-94) with UnsupportedItemKinds = {update, delete}.
 * Pinned to gotest/canondata TestNativeSerializerTopicName + the batching table of queue/test.go (commonTest).
 * ====================================================================================================== */
static const char *DTYPE_NAMES[] = {"", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float", "double",
                                    "boolean", "string", "utf8", "date", "datetime", "timestamp", "interval", "any"};
static const char *KIND_NAMES[] = {"insert", "update", "delete"};

/* encoding/json of one boxed ColumnValues / KeyValues element, escapeHTML on */
static void native_value(sbuf *b, const ora_value *v) {
  char t[128]; size_t n;
  switch (v->kind) {
    case OV_NIL: sb_s(b, "null"); return;
    case OV_BOOL: sb_s(b, v->v.b ? "true" : "false"); return;
    case OV_STRING: json_string(b, v->s, v->slen, 1); return;
    case OV_BYTES: sb_c(b, '"'); base64_std(b, (const unsigned char *)v->s, v->slen); sb_c(b, '"'); return;
    case OV_JSONNUM: if (v->slen) sb_put(b, v->s, v->slen); else sb_c(b, '0'); return;
    case OV_TIME: sb_c(b, '"'); n = ora_fmt_rfc3339nano(t, v->v.t.sec, v->v.t.nsec); sb_put(b, t, n); sb_c(b, '"'); return;
    case OV_DURATION: n = ora_fmt_int(t, v->v.i); sb_put(b, t, n); return;
    case OV_JSON: put_html_compact(b, v->s, v->slen); return;
    case OV_F32: case OV_F64:
      n = v->kind == OV_F32 ? ora_json_float(t, (double)v->v.f32, 32) : ora_json_float(t, v->v.f64, 64);
      if (!n) b->unsupported = 1; else sb_put(b, t, n);  /* NaN / Inf: Marshal fails, ToJSONString drops the error */
      return;
    default: if (is_int_kind(v->kind)) put_int(b, v); else b->unsupported = 1;
  }
}

static void native_string_array(sbuf *b, char *const *names, int n) {
  sb_c(b, '[');
  for (int i = 0; i < n; i++) { if (i) sb_c(b, ','); json_string(b, names[i], strlen(names[i]), 1); }
  sb_c(b, ']');
}

/* json.Marshal([]ColSchema) — col_schema.go:14-29 field tags; Properties is omitempty */
static void native_table_schema(sbuf *b, const tfgpu_schema *ts, const ora_schema *os) {
  int n = ts ? ts->ncols : os->ncols;
  sb_c(b, '[');
  for (int i = 0; i < n; i++) {
    const char *name = ts ? ts->cols[i].name : os->cols[i].name;
    int dtype = ts ? ts->cols[i].dtype : os->cols[i].dtype;
    int key = ts ? !!(ts->cols[i].flags & TFGPU_COL_KEY) : os->cols[i].key;
    int req = ts ? !!(ts->cols[i].flags & TFGPU_COL_REQUIRED) : os->cols[i].required;
    int fake = ts ? !!(ts->cols[i].flags & TFGPU_COL_FAKE_KEY) : os->cols[i].fake_key;
    const char *path = ts ? ts->cols[i].path : os->cols[i].path;
    const char *ot = ts ? ts->cols[i].original_type : os->cols[i].original_type;
    const char *tsn = ts ? ts->cols[i].table_schema : os->cols[i].table_schema, *tbn = ts ? ts->cols[i].table_name : os->cols[i].table_name;
    const char *ex = ts ? ts->cols[i].expression : os->cols[i].expression, *pj = ts ? ts->cols[i].properties_json : os->cols[i].properties_json;
    if (i) sb_c(b, ',');
    sb_s(b, "{\"table_schema\":"); json_string(b, tsn ? tsn : "", tsn ? strlen(tsn) : 0, 1);
    sb_s(b, ",\"table_name\":"); json_string(b, tbn ? tbn : "", tbn ? strlen(tbn) : 0, 1);
    sb_s(b, ",\"path\":"); json_string(b, path ? path : "", path ? strlen(path) : 0, 1);
    sb_s(b, ",\"name\":"); json_string(b, name ? name : "", name ? strlen(name) : 0, 1);
    sb_s(b, ",\"type\":"); { const char *tn = dtype > 0 && dtype < TFGPU_T__COUNT ? DTYPE_NAMES[dtype] : ""; json_string(b, tn, strlen(tn), 1); }
    sb_s(b, ",\"key\":"); sb_s(b, key ? "true" : "false");
    sb_s(b, ",\"fake_key\":"); sb_s(b, fake ? "true" : "false");
    sb_s(b, ",\"required\":"); sb_s(b, req ? "true" : "false");
    sb_s(b, ",\"expression\":"); json_string(b, ex ? ex : "", ex ? strlen(ex) : 0, 1);
    sb_s(b, ",\"original_type\":"); json_string(b, ot ? ot : "", ot ? strlen(ot) : 0, 1);
    if (pj && pj[0]) { sb_s(b, ",\"properties\":"); sb_s(b, pj); } /* omitempty */
    sb_c(b, '}');
  }
  sb_c(b, ']');
}

static void meta_string(sbuf *b, const uint32_t *off, const uint8_t *data, int64_t i) {
  if (!off) { sb_s(b, "\"\""); return; }
  json_string(b, (const char *)data + off[i], off[i + 1] - off[i], 1);
}

/* ChangeItem.ToJSONString(): MarshalJSON's putItem sequence, compacted by json.Marshal */
static void native_item(sbuf *b, const ora_item *it, const tfgpu_queue_options *o, const tfgpu_row_meta *m, const char *part) {
  char t[32]; size_t n;
  const int64_t k = it->src_row;
  const int form = (m && m->names_form) ? m->names_form[k] : 0;
  sb_s(b, "{\"id\":"); n = ora_fmt_uint(t, m && m->id ? m->id[k] : 0); sb_put(b, t, n);
  sb_s(b, ",\"nextlsn\":"); n = ora_fmt_uint(t, m && m->lsn ? m->lsn[k] : 0); sb_put(b, t, n);
  sb_s(b, ",\"commitTime\":"); n = ora_fmt_uint(t, m && m->commit_time ? m->commit_time[k] : 0); sb_put(b, t, n);
  sb_s(b, ",\"txPosition\":"); n = ora_fmt_int(t, m && m->counter ? m->counter[k] : 0); sb_put(b, t, n);
  sb_s(b, ",\"kind\":"); { const char *kn = it->kind >= 0 && it->kind <= 2 ? KIND_NAMES[it->kind] : ""; if (it->kind > 2) b->unsupported = 1; json_string(b, kn, strlen(kn), 1); }
  sb_s(b, ",\"schema\":"); json_string(b, it->ns, strlen(it->ns), 1);
  sb_s(b, ",\"table\":"); json_string(b, it->table, strlen(it->table), 1);
  if (!part) part = it->part_id ? it->part_id : "";
  sb_s(b, ",\"part\":"); json_string(b, part, strlen(part), 1);
  sb_s(b, ",\"columnnames\":");
  if (form == 1) sb_s(b, "null");
  else if (form == 2) sb_s(b, "[]");
  else native_string_array(b, it->names ? it->names->names : NULL, it->names ? it->names->n : 0);
  if (form == 0 && it->nvalues > 0) {  /* len(c.ColumnValues) > 0 */
    sb_s(b, ",\"columnvalues\":[");
    for (int i = 0; i < it->nvalues; i++) { if (i) sb_c(b, ','); native_value(b, &it->values[i]); }
    sb_c(b, ']');
  }
  if (!o->omit_table_schema) {
    if (o->table_schema_json) { sb_s(b, ",\"table_schema\":"); sb_s(b, o->table_schema_json); }
    else if ((o->table_schema && o->table_schema->ncols > 0) || (!o->table_schema && it->schema && it->schema->ncols > 0)) {
      sb_s(b, ",\"table_schema\":"); native_table_schema(b, o->table_schema, it->schema);
    }
  }
  sb_s(b, ",\"oldkeys\":{");  /* OldKeysType: every field omitempty (old_keys.go:3-7) */
  if (it->n_old > 0) {
    sb_s(b, "\"keynames\":"); native_string_array(b, it->old_names->names, it->old_names->n);
    if (o->old_key_types) { sb_s(b, ",\"keytypes\":"); native_string_array(b, (char *const *)o->old_key_types, it->n_old); }
    sb_s(b, ",\"keyvalues\":[");
    for (int i = 0; i < it->n_old; i++) { if (i) sb_c(b, ','); native_value(b, &it->old_values[i]); }
    sb_c(b, ']');
  }
  sb_c(b, '}');
  sb_s(b, ",\"tx_id\":"); meta_string(b, m ? m->tx_id_offsets : NULL, m ? m->tx_id_data : NULL, k);
  sb_s(b, ",\"query\":"); meta_string(b, m ? m->query_offsets : NULL, m ? m->query_data : NULL, k);
  sb_c(b, '}');
}

/* one element as the format's batcher sees it: returns 0, or 1 = the reference's Serialize returns an error */
static int queue_element(sbuf *e, int format, const ora_item *it, const tfgpu_queue_options *o, const tfgpu_row_meta *m, const char *part) {
  if (format == TFGPU_QFMT_NATIVE) { native_item(e, it, o, m, part); return 0; }
  if (it->kind > TFGPU_K_DELETE) return 0;                                    /* !IsRowEvent: empty bytes (json.go:51-53) */
  if (it->kind == TFGPU_K_UPDATE || it->kind == TFGPU_K_DELETE) return 1;    /* UnsupportedItemKinds (json.go:54-56)     */
  json_row(e, it, NULL, 0);
  return 0;
}

typedef struct { uint64_t *start; int64_t *row; int64_t n, cap; } qmsgs;
static void qm_push(qmsgs *q, uint64_t start, int64_t row) {
  if (q->n == q->cap) { q->cap = q->cap ? q->cap * 2 : 16; q->start = (uint64_t *)realloc(q->start, sizeof(uint64_t) * (size_t)(q->cap + 1)); q->row = (int64_t *)realloc(q->row, sizeof(int64_t) * (size_t)(q->cap + 1)); }
  q->start[q->n] = start; q->row[q->n] = row; q->n++;
}

/* Returns the message values back to back (malloc'd, *len bytes) or NULL when the reference's Serialize fails / a value
 * form is outside the restatement.  msg_start / msg_row: malloc'd, *nmsg + 1 entries each.                        */
char *ora_queue_serialize(const tfgpu_queue_options *o, const ora_batch *bt, const tfgpu_row_meta *m, uint64_t *len,
                          uint64_t **msg_start, int64_t **msg_row, int64_t *nmsg) {
  sbuf out = {0}; out.p = (char *)malloc(64); out.cap = 64;
  qmsgs q = {0};
  const int native = o->format == TFGPU_QFMT_NATIVE;
  const char *join = native ? "," : "\n";
  const uint64_t wrap = native ? 2 : 0;  /* "[" "]" */
  int failed = 0;
  int64_t g0 = 0;
  const int ngroups = o->group_rows ? o->ngroups : 1;
  for (int g = 0; g < ngroups && !failed; g++) {
    const int64_t gn = o->group_rows ? o->group_rows[g] : bt->n, g1 = g0 + gn;
    /* serializedElements / lenElements */
    sbuf *el = (sbuf *)calloc((size_t)(gn ? gn : 1), sizeof(sbuf));
    for (int64_t i = 0; i < gn; i++) {
      if (queue_element(&el[i], o->format, &bt->items[g0 + i], o, m, o->group_part_ids ? o->group_part_ids[g] : NULL)) failed = 1;
      if (el[i].unsupported) { if (native) { el[i].n = 0; out.unsupported = 1; } else failed = 1; }
    }
#define EMIT(a, z) do { qm_push(&q, out.n, g0 + (a)); if (native) sb_c(&out, '['); \
      for (int64_t e_ = (a); e_ < (z); e_++) { if (e_ > (a)) sb_s(&out, join); sb_put(&out, el[e_].p ? el[e_].p : "", el[e_].n); } \
      if (native) sb_c(&out, ']'); } while (0)
    if (!failed) {
      if (!o->batching_enabled) { for (int64_t i = 0; i < gn; i++) EMIT(i, i + 1); }
      else {
        int64_t start = 0; uint64_t sum = 0;
        for (int64_t i = 0; i < gn; i++) {
          const int64_t num_new = i - start + 1;
          int violates = 0;
          if (o->max_message_size != 0 && sum + (uint64_t)(num_new - 1) + el[i].n + wrap > (uint64_t)o->max_message_size) violates = 1;
          if (o->max_change_items != 0 && num_new > o->max_change_items) violates = 1;
          if (violates) {
            if (i - start == 0) { EMIT(start, start + 1); start = i + 1; sum = 0; }
            else { EMIT(start, i); start = i; sum = el[i].n; }
          } else sum += el[i].n;
        }
        if (start != gn) EMIT(start, gn);
      }
    }
#undef EMIT
    for (int64_t i = 0; i < gn; i++) free(el[i].p);
    free(el);
    g0 = g1;
  }
  if (failed || out.unsupported) { free(out.p); free(q.start); free(q.row); *len = 0; *nmsg = 0; *msg_start = NULL; *msg_row = NULL; return NULL; }
  qm_push(&q, out.n, bt->n); q.n--;
  *len = out.n; *msg_start = q.start; *msg_row = q.row; *nmsg = q.n;
  return out.p;
}
