/*
 * oracle/ora_debezium.c — CPU restatement of the Debezium parser with inline schemas (the source of
 * BASELINE.json configs[4]).  TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 *   DebeziumImpl.DoBatch / Do / DoBuf / DoOne     pkg/parsers/registry/debezium/engine/parser.go:33-57, 59-71, 98-104, 120-130
 *   IncludeSchema.Unpack                          pkg/debezium/unpacker/include_schema.go:13-25
 *   Receiver.Receive / receive / receiveSchema /
 *   receiveTableSchema / add                      pkg/debezium/receiver.go:142-235, 60-96, 45-59, 98-121
 *   receiveFieldColSchema / receiveField /
 *   extractVal / convertVal                       pkg/debezium/receiver_engine.go:108-146, 148-219, 221-287, 289-371
 *   the default receivers (no original types)     pkg/debezium/common/field_receiver_default.go:14-31, 40-355
 *   Payload / Source / Schema structs             pkg/debezium/common/debezium_schema.go:12-56
 *   opToKind                                      pkg/debezium/kind.go:34-46
 *   Base64ToNumeric                               pkg/debezium/typeutil/helpers.go:966-996, 367-377
 *
 * Scope = NewDebeziumImpl(logger, nil, threads): no schema registry (every message is one event with "schema" and
 * "payload"), NewReceiver(nil, nil): no original-type table, so every field takes the default receiver of its Kafka
 * type unless the schema carries `__dt_original_type_info` (then: TFGPU_ROW_HOST_FALLBACK, like everything else the
 * restatement leaves to the stock code: arrays, Go panics, case-folded keys).
 * encoding/json semantics (struct decoding: keys match exactly or ASCII-case-insensitively, the last match wins, null
 * leaves a field alone, a value of the wrong JSON type is an UnmarshalTypeError) are restated with ora_jv.h's decoder.
 * Pinned to the reference's canon: engine/gotest/canondata/result.json (TestParser) over engine/parser_test.jsonl, and
 * to receiver_test.go's TestDelete vector; golden file tests/golden/debezium.json (tools/extract_golden.py).
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ora.h"
#include "ora_jv.h"

/* ---- struct-field lookup of encoding/json: exact name or ASCII case fold; the LAST matching member wins ---- */
static const jv *field_of(const jv *obj, const char *name, int *folded) {
  const jv *hit = NULL;
  size_t n = strlen(name);
  if (folded) *folded = 0;
  for (int i = 0; i < obj->nk; i++) {
    if (obj->klen[i] != n) continue;
    if (!memcmp(obj->keys[i], name, n)) { hit = obj->kids[i]; continue; }
    if (!strncasecmp(obj->keys[i], name, n)) { hit = obj->kids[i]; if (folded) *folded = 1; }
  }
  return hit;
}

/* how many members of obj bind struct field `name` (exactly or by case folding) */
static int occurrences(const jv *obj, const char *name) {
  size_t n = strlen(name); int c = 0;
  for (int i = 0; i < obj->nk; i++) if (obj->klen[i] == n && !strncasecmp(obj->keys[i], name, n)) c++;
  return c;
}

/* strconv.ParseUint(text, 10, bits) of a JSON number literal as encoding/json applies it to unsigned fields */
static int lit_uint(const jv *v, int bits, uint64_t *out) {
  if (v->n == 0) return 0;
  uint64_t x = 0;
  for (size_t i = 0; i < v->n; i++) {
    if (v->s[i] < '0' || v->s[i] > '9') return 0;
    uint64_t d = (uint64_t)(v->s[i] - '0');
    if (x > (UINT64_MAX - d) / 10) return 0;
    x = x * 10 + d;
  }
  if (bits < 64 && x >> bits) return 0;
  *out = x;
  return 1;
}

/* ---- base64.StdEncoding.DecodeString: '\r' / '\n' skipped, padding required, stray bits tolerated ---- */
static int b64v(int c) {
  if (c >= 'A' && c <= 'Z') return c - 'A';
  if (c >= 'a' && c <= 'z') return c - 'a' + 26;
  if (c >= '0' && c <= '9') return c - '0' + 52;
  return c == '+' ? 62 : c == '/' ? 63 : -1;
}
static int b64_decode(const char *s, size_t n, unsigned char **out, size_t *on) {
  unsigned char *o = (unsigned char *)malloc(n + 4);
  size_t k = 0; int q[4], nq = 0, pad = 0;
  for (size_t i = 0; i < n; i++) {
    int c = (unsigned char)s[i];
    if (c == '\r' || c == '\n') continue;
    if (c == '=') { pad++; if (pad > 2) { free(o); return 0; } continue; }
    if (pad) { free(o); return 0; }
    int v = b64v(c);
    if (v < 0) { free(o); return 0; }
    q[nq++] = v;
    if (nq == 4) { o[k++] = (unsigned char)(q[0] << 2 | q[1] >> 4); o[k++] = (unsigned char)(q[1] << 4 | q[2] >> 2); o[k++] = (unsigned char)(q[2] << 6 | q[3]); nq = 0; }
  }
  if (nq == 1 || (nq == 0 && pad) || (nq == 2 && pad != 2) || (nq == 3 && pad != 1)) { free(o); return 0; }
  if (nq >= 2) o[k++] = (unsigned char)(q[0] << 2 | q[1] >> 4);
  if (nq == 3) o[k++] = (unsigned char)(q[1] << 4 | q[2] >> 2);
  *out = o; *on = k;
  return 1;
}

/* typeutil.Base64ToNumeric: 0 error, 1 ok, 2 the reference panics (empty buffer / negative scale slices) */
static int base64_to_numeric(const char *s, size_t n, int scale, sbuf *res) {
  unsigned char *buf; size_t bn;
  if (!b64_decode(s, n, &buf, &bn)) return 0;
  if (bn == 0) { free(buf); return 2; }                          /* isHighestBitSet(in[0]) on an empty slice */
  int neg = buf[0] & 0x80;
  if (neg) {                                                      /* makeNegativeNum: invert, add one */
    for (size_t i = 0; i < bn; i++) buf[i] = (unsigned char)~buf[i];
    for (size_t i = bn; i-- > 0;) if (++buf[i]) break;           /* a carry out of the top byte cannot happen: buf[0] < 0x80 after the inversion */
  }
  /* big.Int.String(): base-256 → decimal by repeated division */
  char *dig = (char *)malloc(bn * 3 + 2); size_t nd = 0;
  size_t first = 0;
  while (first < bn && buf[first] == 0) first++;
  while (first < bn) {
    unsigned rem = 0;
    for (size_t i = first; i < bn; i++) { unsigned cur = rem * 256 + buf[i]; buf[i] = (unsigned char)(cur / 10); rem = cur % 10; }
    dig[nd++] = (char)('0' + rem);
    while (first < bn && buf[first] == 0) first++;
  }
  free(buf);
  if (nd == 0) { free(dig); sb_s(res, "0"); return 1; }           /* resultStr == "0": returned without a sign */
  sbuf t = {0};
  for (size_t i = nd; i-- > 0;) sb_c(&t, dig[i]);
  free(dig);
  if (scale < 0) { free(t.p); return 2; }
  if (neg) sb_c(res, '-');
  if (scale != 0) {
    size_t len = t.n;
    sbuf u = {0};
    if ((size_t)scale > len) for (size_t i = 0; i < (size_t)scale - len + 1; i++) sb_c(&u, '0');
    sb_put(&u, t.p, t.n);
    sb_put(res, u.p, u.n - (size_t)scale); sb_c(res, '.'); sb_put(res, u.p + u.n - (size_t)scale, (size_t)scale);
    free(u.p);
  } else sb_put(res, t.p, t.n);
  free(t.p);
  return 1;
}

/* ---- the Schema struct (debezium_schema.go:12-22) decoded from a jv: 0 = json.Unmarshal fails ---- */
typedef struct dschema {
  char *field, *name, *type; int optional; char *scale; int has_params;
  struct dschema *fields; int nfields; int has_items; int has_dt_info;
} dschema;
static void dschema_free(dschema *s) {
  for (int i = 0; i < s->nfields; i++) dschema_free(&s->fields[i]);
  free(s->fields); free(s->field); free(s->name); free(s->type); free(s->scale);
}
static int str_field(const jv *o, const char *name, char **out, int *fold) {
  int f; const jv *v = field_of(o, name, &f);
  if (f) *fold = 1;
  if (!v || v->t == JV_NULL) return 1;
  if (v->t != JV_STR) return 0;
  free(*out); *out = strndup(v->s, v->n);
  return 1;
}
static int decode_schema(const jv *o, dschema *s, int *fold, int depth) {
  memset(s, 0, sizeof *s);
  s->field = strdup(""); s->name = strdup(""); s->type = strdup(""); s->scale = strdup("");
  if (o->t == JV_NULL) return 1;
  if (o->t != JV_OBJ || depth > 64) return 0;
  if (!str_field(o, "field", &s->field, fold) || !str_field(o, "name", &s->name, fold) || !str_field(o, "type", &s->type, fold)) return 0;
  int f; const jv *v;
  if ((v = field_of(o, "optional", &f))) { if (f) *fold = 1; if (v->t == JV_TRUE) s->optional = 1; else if (v->t == JV_FALSE) s->optional = 0; else if (v->t != JV_NULL) return 0; }
  if ((v = field_of(o, "version", &f))) {
    if (f) *fold = 1;
    if (v->t == JV_NUM) { int64_t x; if (ora_parse_int(v->s, v->n, 10, 64, &x)) return 0; } else if (v->t != JV_NULL) return 0;
  }
  if ((v = field_of(o, "parameters", &f))) {
    if (f) *fold = 1;
    if (v->t == JV_OBJ) {
      s->has_params = 1;
      char *dummy = strdup("");
      int ok = str_field(v, "length", &dummy, fold) && str_field(v, "connect.decimal.precision", &dummy, fold) && str_field(v, "scale", &s->scale, fold) && str_field(v, "allowed", &dummy, fold);
      free(dummy);
      if (!ok) return 0;
    } else if (v->t != JV_NULL) return 0;
  }
  if ((v = field_of(o, "items", &f))) { if (f) *fold = 1; if (v->t == JV_OBJ) { s->has_items = 1; dschema it; int ok = decode_schema(v, &it, fold, depth + 1); dschema_free(&it); if (!ok) return 0; } else if (v->t != JV_NULL) return 0; }
  if ((v = field_of(o, "__dt_original_type_info", &f))) { if (f) *fold = 1; if (v->t != JV_NULL) s->has_dt_info = 1; }
  if ((v = field_of(o, "fields", &f))) {
    if (f) *fold = 1;
    if (v->t == JV_ARR) {
      s->fields = (dschema *)calloc((size_t)(v->nk ? v->nk : 1), sizeof(dschema));
      for (int i = 0; i < v->nk; i++) { s->nfields = i + 1; if (!decode_schema(v->kids[i], &s->fields[i], fold, depth + 1)) return 0; }
    } else if (v->t != JV_NULL) return 0;
  }
  return 1;
}

/* receiveFieldColSchema with an empty original type: the receiver of one field.  op 0 = no receiver (error) */
enum { OP_NONE = 0, OP_BOOL = TFGPU_DBZ_BOOLEAN, OP_I8 = TFGPU_DBZ_INT8, OP_I16 = TFGPU_DBZ_INT16, OP_I32 = TFGPU_DBZ_INT32, OP_I64 = TFGPU_DBZ_INT64,
       OP_F64 = TFGPU_DBZ_FLOAT64, OP_STR = TFGPU_DBZ_STRING, OP_BYTES = TFGPU_DBZ_BYTES, OP_DEC = TFGPU_DBZ_DECIMAL, OP_POINT = TFGPU_DBZ_POINT,
       OP_VSD = TFGPU_DBZ_VSD, OP_HOST = TFGPU_DBZ_HOST };
static int field_op(const dschema *f, int *dtype) {
  const char *t = f->type;
  if (f->has_dt_info) { *dtype = TFGPU_T_ANY; return OP_HOST; }
  if (!strcmp(t, "array")) { *dtype = TFGPU_T_ANY; return OP_HOST; }
  if (!strcmp(t, "int8")) { *dtype = TFGPU_T_INT8; return OP_I8; }
  if (!strcmp(t, "int16")) { *dtype = TFGPU_T_INT16; return OP_I16; }
  if (!strcmp(t, "int32")) { *dtype = TFGPU_T_INT32; return OP_I32; }
  if (!strcmp(t, "int64")) { *dtype = TFGPU_T_INT64; return OP_I64; }
  if (!strcmp(t, "boolean")) { *dtype = TFGPU_T_BOOLEAN; return OP_BOOL; }
  if (!strcmp(t, "string")) { *dtype = TFGPU_T_UTF8; return OP_STR; }
  if (!strcmp(t, "float") || !strcmp(t, "double")) { *dtype = TFGPU_T_FLOAT64; return OP_F64; }
  if (!strcmp(t, "struct")) {
    if (!strcmp(f->name, "io.debezium.data.geometry.Point")) { *dtype = TFGPU_T_UTF8; return OP_POINT; }
    if (!strcmp(f->name, "io.debezium.data.VariableScaleDecimal")) { *dtype = TFGPU_T_FLOAT64; return OP_VSD; }
    return OP_NONE;
  }
  if (!strcmp(t, "bytes")) {
    if (!strcmp(f->name, "org.apache.kafka.connect.data.Decimal")) { *dtype = TFGPU_T_UTF8; return OP_DEC; }
    *dtype = TFGPU_T_BYTES; return OP_BYTES;
  }
  return OP_NONE;  /* "map" and unknown Kafka types have no default receiver */
}

static void set_str(ora_value *v, int kind, const char *s, size_t n) {
  v->kind = kind; v->s = (char *)malloc(n + 1); memcpy(v->s, s, n); v->s[n] = 0; v->slen = n;
}

/* receiveField for one present value.  0 ok, 1 error (the item becomes `_unparsed`), 2 host (panic / absent marker / unrestated) */
static int receive_value(const dschema *f, int op, const jv *val, ora_value *out) {
  memset(out, 0, sizeof *out);
  if (val->t == JV_NULL) { out->kind = OV_NIL; return 0; }
  if (val->t == JV_STR && val->n == 28 && !memcmp(val->s, "__debezium_unavailable_value", 28)) return 2;  /* absent: the column list becomes ragged */
  switch (op) {
    case OP_I8: case OP_I16: case OP_I32: case OP_I64: {
      if (val->t != JV_NUM) return 1;                             /* "assert no one value extracted" */
      int64_t x;
      if (ora_parse_int(val->s, val->n, 10, 64, &x)) return 1;    /* json.Number.Int64 */
      out->kind = op == OP_I8 ? OV_I8 : op == OP_I16 ? OV_I16 : op == OP_I32 ? OV_I32 : OV_I64;
      out->v.i = op == OP_I8 ? (int8_t)x : op == OP_I16 ? (int16_t)x : op == OP_I32 ? (int32_t)x : x;  /* Go's truncating conversions */
      return 0;
    }
    case OP_BOOL:
      if (val->t != JV_TRUE && val->t != JV_FALSE) return 1;
      out->kind = OV_BOOL; out->v.b = val->t == JV_TRUE;
      return 0;
    case OP_F64: {
      if (val->t != JV_NUM) return 1;
      double d;
      if (ora_parse_float(val->s, val->n, 64, &d)) return 1;      /* json.Number.Float64: a range error is an error */
      out->kind = OV_F64; out->v.f64 = d;
      return 0;
    }
    case OP_STR:
      if (val->t != JV_STR && val->t != JV_NUM) return 1;         /* a json.Number is taken by its text */
      set_str(out, OV_STRING, val->s, val->n);
      return 0;
    case OP_BYTES: {
      if (val->t != JV_STR && val->t != JV_NUM) return 1;
      unsigned char *b; size_t bn;
      if (!b64_decode(val->s, val->n, &b, &bn)) return 1;
      set_str(out, OV_BYTES, (const char *)b, bn);
      free(b);
      return 0;
    }
    case OP_DEC: {
      if (val->t != JV_STR && val->t != JV_NUM) return 1;
      int scale = 0;
      if (f->has_params && f->scale[0]) {                         /* strconv.Atoi */
        int64_t x;
        if (ora_parse_int(f->scale, strlen(f->scale), 10, 64, &x) || x > 2147483647LL || x < -2147483648LL) return 1;
        scale = (int)x;
      }
      sbuf r = {0};
      int rc = base64_to_numeric(val->s, val->n, scale, &r);
      if (rc != 1) { free(r.p); return rc == 0 ? 1 : 2; }
      set_str(out, OV_STRING, r.p, r.n);
      free(r.p);
      return 0;
    }
    case OP_POINT: {
      if (val->t != JV_OBJ) return 2;                             /* in.(map[string]interface{}) panics */
      const jv *xy[2] = {NULL, NULL};
      for (int i = 0; i < val->nk; i++) {
        if (val->klen[i] == 1 && val->keys[i][0] == 'x') xy[0] = val->kids[i];
        if (val->klen[i] == 1 && val->keys[i][0] == 'y') xy[1] = val->kids[i];
      }
      if (!xy[0] || !xy[1]) return 1;
      sbuf r = {0};
      sb_c(&r, '(');
      for (int k = 0; k < 2; k++) {                               /* fmt %v */
        const jv *e = xy[k];
        if (e->t == JV_NUM || e->t == JV_STR) sb_put(&r, e->s, e->n);
        else if (e->t == JV_NULL) sb_s(&r, "<nil>");
        else if (e->t == JV_TRUE) sb_s(&r, "true");
        else if (e->t == JV_FALSE) sb_s(&r, "false");
        else { free(r.p); return 2; }                             /* maps / slices: Go's %v of a container, left to the host */
        sb_c(&r, k == 0 ? ',' : ')');
      }
      set_str(out, OV_STRING, r.p, r.n);
      free(r.p);
      return 0;
    }
    case OP_VSD: {
      if (val->t != JV_OBJ) return 2;
      const jv *value = NULL, *scale = NULL;
      for (int i = 0; i < val->nk; i++) {
        if (val->klen[i] == 5 && !memcmp(val->keys[i], "value", 5)) value = val->kids[i];
        if (val->klen[i] == 5 && !memcmp(val->keys[i], "scale", 5)) scale = val->kids[i];
      }
      if (!value) return 1;
      if (value->t != JV_STR) return 2;                           /* .(string) panics */
      int sc = 0;
      if (scale) {
        if (scale->t != JV_NUM) return 2;                         /* .(json.Number) panics */
        int64_t x;
        if (ora_parse_int(scale->s, scale->n, 10, 64, &x)) return 1;
        if (x > 2147483647LL || x < -2147483648LL) return 2;
        sc = (int)x;
      }
      sbuf r = {0};
      int rc = base64_to_numeric(value->s, value->n, sc, &r);
      if (rc != 1) { free(r.p); return rc == 0 ? 1 : 2; }
      set_str(out, OV_JSONNUM, r.p, r.n);
      free(r.p);
      return 0;
    }
  }
  return 2;
}

/* one message → one item or a code */
static int receive_message(const unsigned char *p, size_t n, ora_batch *out, uint32_t *id, uint64_t *lsn, uint64_t *commit, uint8_t *names_form) {
  if (n == 0) return TFGPU_ROW_DBZ_UNPACK;                        /* "debezium parser received empty message" */
  /* DoOne: a leading 0x00 cuts the event at the next 0x00 after the 5-byte prefix; either way json.Unmarshal fails on it */
  if (p[0] == 0) return n < 5 ? TFGPU_ROW_HOST_FALLBACK /* buf[5:] panics */ : TFGPU_ROW_DBZ_UNPACK;
  /* IncludeSchema.Unpack: json.Unmarshal(message, &struct{Schema, Payload json.RawMessage}) — the whole message must be one valid value */
  jp s = {p, p + n, 0, 0};
  jv_keep_dups = 1;   /* struct fields see every occurrence of their key; the map lookups below take the last one */
  jv *msg = parse_value(&s);
  jv_keep_dups = 0;
  if (!s.err) { ws(&s); if (s.p != s.e) s.err = 1; }
  if (s.err || !msg) { jv_free(msg); return TFGPU_ROW_DBZ_UNPACK; }
  int rc = TFGPU_ROW_OK, fold = 0;
  dschema sch; int have_sch = 0;
  if (msg->t == JV_NULL) { jv_free(msg); return TFGPU_ROW_DBZ_PAYLOAD; }   /* both RawMessages nil → Decode of no bytes: EOF */
  if (msg->t != JV_OBJ) { jv_free(msg); return TFGPU_ROW_DBZ_UNPACK; }      /* UnmarshalTypeError */
  int f1, f2;
  const jv *schema = field_of(msg, "schema", &f1), *payload = field_of(msg, "payload", &f2);
  if (f1 || f2) { jv_free(msg); return TFGPU_ROW_HOST_FALLBACK; }
  if (!payload) { jv_free(msg); return TFGPU_ROW_DBZ_PAYLOAD; }              /* EOF */
  /* UnmarshalPayload: Decoder(UseNumber).Decode(&Payload) */
  const jv *after = NULL, *before = NULL, *source = NULL;
  char *op = strdup(""), *sschema = strdup(""), *stable = strdup("");
  uint64_t v_lsn = 0, v_ts = 0, v_tx = 0;
  int dup = 0;   /* a struct key that repeats: encoding/json decodes every occurrence INTO the same field (maps and structs merge) — not restated */
  if (payload->t == JV_OBJ) {
    const jv *v; int f;
    int bad = 0;
    static const char *ptop[] = {"after", "before", "op", "source", "ts_ms", "transaction"};
    for (int i = 0; i < 6; i++) if (occurrences(payload, ptop[i]) > 1) dup = 1;
    if ((v = field_of(payload, "after", &f))) { fold |= f; if (v->t == JV_OBJ) after = v; else if (v->t != JV_NULL) bad = 1; }
    if ((v = field_of(payload, "before", &f))) { fold |= f; if (v->t == JV_OBJ) before = v; else if (v->t != JV_NULL) bad = 1; }
    if (!str_field(payload, "op", &op, &fold)) bad = 1;
    if ((v = field_of(payload, "ts_ms", &f))) { fold |= f; uint64_t x; if (v->t == JV_NUM) { if (!lit_uint(v, 64, &x)) bad = 1; } else if (v->t != JV_NULL) bad = 1; }
    (void)field_of(payload, "transaction", &f); fold |= f;
    if ((v = field_of(payload, "source", &f))) {
      fold |= f;
      if (v->t == JV_OBJ) {
        source = v;
        static const char *sf[] = {"connector", "db", "lsn", "name", "schema", "sequence", "snapshot", "table", "ts_ms", "txId", "version", "xmin"};
        for (int i = 0; i < 12; i++) if (occurrences(v, sf[i]) > 1) dup = 1;
        char *d = strdup("");
        static const char *strs[] = {"connector", "db", "name", "sequence", "snapshot", "version"};
        for (int i = 0; i < 6; i++) if (!str_field(v, strs[i], &d, &fold)) bad = 1;
        free(d);
        if (!str_field(v, "schema", &sschema, &fold) || !str_field(v, "table", &stable, &fold)) bad = 1;
        const jv *w;
        if ((w = field_of(v, "lsn", &f))) { fold |= f; if (w->t == JV_NUM) { if (!lit_uint(w, 64, &v_lsn)) bad = 1; } else if (w->t != JV_NULL) bad = 1; }
        if ((w = field_of(v, "ts_ms", &f))) { fold |= f; if (w->t == JV_NUM) { if (!lit_uint(w, 64, &v_ts)) bad = 1; } else if (w->t != JV_NULL) bad = 1; }
        if ((w = field_of(v, "txId", &f))) { fold |= f; if (w->t == JV_NUM) { if (!lit_uint(w, 32, &v_tx)) bad = 1; } else if (w->t != JV_NULL) bad = 1; }
        if ((w = field_of(v, "xmin", &f))) { fold |= f; if (w->t == JV_NUM) { int64_t x; if (ora_parse_int(w->s, w->n, 10, 64, &x)) bad = 1; } else if (w->t != JV_NULL) bad = 1; }
      } else if (v->t != JV_NULL) bad = 1;
    }
    if (bad) rc = TFGPU_ROW_DBZ_PAYLOAD;
    if (dup) rc = TFGPU_ROW_HOST_FALLBACK;   /* decided before the type errors, like the device */
  } else if (payload->t != JV_NULL) rc = TFGPU_ROW_DBZ_PAYLOAD;    /* UnmarshalTypeError */
  if (rc == TFGPU_ROW_OK && fold) rc = TFGPU_ROW_HOST_FALLBACK;
  int kind = -1;
  if (rc == TFGPU_ROW_OK) {
    if (!strcmp(op, "c") || !strcmp(op, "r")) kind = TFGPU_K_INSERT; else if (!strcmp(op, "u")) kind = TFGPU_K_UPDATE; else if (!strcmp(op, "d")) kind = TFGPU_K_DELETE;
    else rc = TFGPU_ROW_DBZ_OP;
  }
  /* receiveSchema: json.Unmarshal(schema, &Schema); before and after table schemas, both, whatever the kind */
  const dschema *bs = NULL, *as = NULL;
  if (rc == TFGPU_ROW_OK) {
    if (!schema) rc = TFGPU_ROW_DBZ_SCHEMA;                        /* json.Unmarshal(nil): unexpected end of JSON input */
    else {
      fold = 0;
      have_sch = 1;
      if (!decode_schema(schema, &sch, &fold, 0) || (schema->t != JV_OBJ && schema->t != JV_NULL)) rc = TFGPU_ROW_DBZ_SCHEMA;
      else if (fold) rc = TFGPU_ROW_HOST_FALLBACK;
      else {
        for (int i = 0; i < sch.nfields && !bs; i++) if (!strcmp(sch.fields[i].field, "before")) bs = &sch.fields[i];
        for (int i = 0; i < sch.nfields && !as; i++) if (!strcmp(sch.fields[i].field, "after")) as = &sch.fields[i];
        if (!bs || !as) rc = TFGPU_ROW_HOST_FALLBACK;              /* receiveTableSchema(nil): nil pointer dereference */
      }
    }
  }
  int host = 0;
  if (rc == TFGPU_ROW_OK) {
    const dschema *both[2] = {bs, as};
    for (int k = 0; k < 2 && rc == TFGPU_ROW_OK; k++) for (int i = 0; i < both[k]->nfields; i++) {
      int dt, o = field_op(&both[k]->fields[i], &dt);
      if (o == OP_NONE) { rc = TFGPU_ROW_DBZ_SCHEMA; break; }
      if (o == OP_HOST) host = 1;
    }
    if (rc == TFGPU_ROW_OK && host) rc = TFGPU_ROW_HOST_FALLBACK;
  }
  if (rc == TFGPU_ROW_OK) {
    const dschema *cur = kind == TFGPU_K_DELETE ? bs : as;
    const jv *vals = kind == TFGPU_K_DELETE ? before : after;
    ora_item it; memset(&it, 0, sizeof it);
    it.kind = kind; it.ns = strdup(sschema); it.table = strdup(stable); it.part_id = strdup("");
    it.schema = (ora_schema *)calloc(1, sizeof(ora_schema)); it.schema->refs = 1; it.schema->ncols = cur->nfields;
    it.schema->cols = (ora_colschema *)calloc((size_t)(cur->nfields ? cur->nfields : 1), sizeof(ora_colschema));
    it.names = (ora_names *)calloc(1, sizeof(ora_names)); it.names->refs = 1; it.names->names = (char **)calloc((size_t)(cur->nfields ? cur->nfields : 1), sizeof(char *));
    it.old_names = (ora_names *)calloc(1, sizeof(ora_names)); it.old_names->refs = 1; it.old_names->names = (char **)calloc((size_t)(cur->nfields ? cur->nfields : 1), sizeof(char *));
    it.values = (ora_value *)calloc((size_t)(cur->nfields ? cur->nfields : 1), sizeof(ora_value));
    it.old_values = (ora_value *)calloc((size_t)(cur->nfields ? cur->nfields : 1), sizeof(ora_value));
    for (int i = 0; i < cur->nfields; i++) {
      const dschema *f = &cur->fields[i];
      ora_colschema *c = &it.schema->cols[i];
      int dt = 0, o = field_op(f, &dt);
      c->name = strdup(f->field); c->dtype = dt; c->key = !f->optional; c->path = strdup(""); c->original_type = strdup("");
      c->table_schema = ora_intern(sschema); c->table_name = ora_intern(stable); c->expression = ora_intern(""); c->properties_json = ora_intern("");
      if (rc != TFGPU_ROW_OK) continue;
      const jv *val = NULL;
      if (vals) for (int k = 0; k < vals->nk; k++) if (vals->klen[k] == strlen(f->field) && !memcmp(vals->keys[k], f->field, vals->klen[k])) val = vals->kids[k];
      if (!val) { rc = TFGPU_ROW_DBZ_FIELD; continue; }            /* "unable to get field %s from 'after'" */
      ora_value v;
      if (f->optional && val->t == JV_STR && val->n == 28 && !memcmp(val->s, "__debezium_unavailable_value", 28)) continue;  /* isAbsent (receiver.go:98-105): the item does not list the column */
      int r = receive_value(f, o, val, &v);
      if (r) { rc = r == 1 ? TFGPU_ROW_DBZ_FIELD : TFGPU_ROW_HOST_FALLBACK; continue; }
      if (kind != TFGPU_K_DELETE) { it.names->names[it.names->n++] = strdup(f->field); it.values[it.nvalues++] = ora_value_clone(&v); }
      if (kind != TFGPU_K_INSERT && c->key) { it.old_names->names[it.old_names->n++] = strdup(f->field); it.old_values[it.n_old++] = ora_value_clone(&v); }
      ora_value_free(&v);
    }
    if (rc == TFGPU_ROW_OK) {
      ora_item *dst = ora_batch_push(out);
      it.src_row = dst->src_row;
      *dst = it;
      *id = (uint32_t)v_tx; *lsn = v_lsn; *commit = v_ts * 1000000ull;
      *names_form = kind == TFGPU_K_DELETE ? 1 : 0;                /* ColumnNames stays nil */
    } else ora_item_clear(&it);
  }
  (void)source;
  if (have_sch) dschema_free(&sch);
  free(op); free(sschema); free(stable);
  jv_free(msg);
  return rc;
}

/* DebeziumImpl.DoBatch without a schema registry.  code[m] = TFGPU_ROW_OK or why message m became an `_unparsed` item
 * (or TFGPU_ROW_HOST_FALLBACK); row_of[m] = its item in the returned batch or -1; per item: ID, LSN, CommitTime and
 * whether ColumnNames is nil.  All arrays malloc'd, nmsg entries (the per-item ones use the first `n items` entries). */
ora_batch *ora_debezium_parse(const void *bytes, uint64_t len, const tfgpu_messages *msgs, int32_t **code, int64_t **row_of, uint32_t **id, uint64_t **lsn,
                              uint64_t **commit_time, uint8_t **names_form) {
  const unsigned char *base = (const unsigned char *)bytes;
  int64_t nmsg = msgs ? msgs->nmsg : 1;
  ora_batch *out = ora_batch_new();
  size_t cap = (size_t)(nmsg ? nmsg : 1);
  *code = (int32_t *)calloc(cap, 4); *row_of = (int64_t *)calloc(cap, 8); *id = (uint32_t *)calloc(cap, 4); *lsn = (uint64_t *)calloc(cap, 8);
  *commit_time = (uint64_t *)calloc(cap, 8); *names_form = (uint8_t *)calloc(cap, 1);
  for (int64_t m = 0; m < nmsg; m++) {
    uint64_t a = msgs ? msgs->start[m] : 0, z = msgs ? msgs->start[m + 1] : len;
    int64_t at = out->n;
    uint32_t i = 0; uint64_t l = 0, c = 0; uint8_t nf = 0;
    int rc = receive_message(base + a, (size_t)(z - a), out, &i, &l, &c, &nf);
    (*code)[m] = rc;
    (*row_of)[m] = rc == TFGPU_ROW_OK ? at : -1;
    if (rc == TFGPU_ROW_OK) { out->items[at].src_row = m; (*id)[at] = i; (*lsn)[at] = l; (*commit_time)[at] = c; (*names_form)[at] = nf; }
  }
  return out;
}
