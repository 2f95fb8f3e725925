/*
 * oracle/ora_chnative.c — the v2 ClickHouse sink's row marshalling + the driver's Native column layout, restated row by
 * row.  TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 * Reference side (in /root/reference): marshalChangeItemInto (pkg/providers/clickhouse/async/marshaller.go:62-166) —
 * nil stays nil (:80-81); String columns take string / []byte as they are and anything else as its json.Marshal text
 * (:84-99); other values go through columntypes.Restore (types.go:74-115), which clamps time.Time values of YT `date` /
 * `datetime` columns to [1970-01-01, 2106-01-01] (types.go:15-29, 92-95) and leaves the rest to abstract.Restore
 * (restore.go:20-56: a time.Time of a date/datetime/timestamp column is returned as is).
 * Dependency side (NOT in /root/reference: github.com/ClickHouse/clickhouse-go/v2 v2.46.0 with ch-go v0.71.0, go.mod:9,121):
 * each []any row is appended column-wise and the block is encoded in ClickHouse's Native layout —
 *     varuint(ncolumns) varuint(nrows) { string(name) string(type) [Nullable: one byte per row, 1 = NULL] data }*
 * fixed-width data little-endian, String data varuint(len) + bytes per row, a nil in a non-Nullable column = the zero
 * value; time.Time → Date: Unix()/86400, DateTime: Unix(), DateTime64(p): UnixNano()/10^(9-p), Go's truncating
 * division; out-of-range times fail Append (DateOverflowError).  PARITY UNPINNED: this half restates the published
 * format of a dependency; tests/test_chnative.py additionally decodes blocks with an independent numpy reader.
 */
#include <stdlib.h>
#include <string.h>
#include "ora.h"

typedef struct { char *p; size_t n, cap; } buf;
static void put(buf *b, const void *s, size_t n) {
  if (b->n + n > b->cap) { b->cap = (b->n + n) * 2 + 64; b->p = (char *)realloc(b->p, b->cap); }
  memcpy(b->p + b->n, s, n); b->n += n;
}
static void put_varuint(buf *b, uint64_t v) {
  while (v >= 0x80) { unsigned char c = (unsigned char)(v | 0x80); put(b, &c, 1); v >>= 7; }
  unsigned char c = (unsigned char)v; put(b, &c, 1);
}
static void put_str(buf *b, const char *s, size_t n) { put_varuint(b, n); put(b, s, n); }

enum { B_I8, B_I16, B_I32, B_I64, B_U8, B_U16, B_U32, B_U64, B_F32, B_F64, B_BOOL, B_STR, B_DATE, B_DATE32, B_DT, B_DT64, B_BAD };

static int parse_type(const char *t, int *nullable, int *prec) {
  char s[128]; size_t n = strlen(t);
  if (n >= sizeof s) return B_BAD;
  memcpy(s, t, n + 1);
  *nullable = 0; *prec = 0;
  char *q = s;
  if (!strncmp(q, "Nullable(", 9) && q[n - 1] == ')') { *nullable = 1; q[n - 1] = 0; q += 9; }
  char *par = strchr(q, '(');
  char args[64] = "";
  if (par) { size_t m = strlen(par); if (par[m - 1] != ')' || m >= sizeof args) return B_BAD; memcpy(args, par + 1, m - 2); args[m - 2] = 0; *par = 0; }
  static const char *names[] = {"Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16", "UInt32", "UInt64", "Float32", "Float64", "Bool", "String", "Date", "Date32", "DateTime", "DateTime64"};
  int base = B_BAD;
  for (int i = 0; i < 16; i++) if (!strcmp(q, names[i])) base = i;
  if (base == B_BAD) return B_BAD;
  if (base == B_DT64) { if (args[0] < '0' || args[0] > '9' || (args[1] && args[1] != ',')) return B_BAD; *prec = args[0] - '0'; }
  else if (args[0] && base != B_DT) return B_BAD;
  return base;
}

static int kind_fits(int k, int base) {
  switch (base) {
    case B_I8: return k == OV_I8; case B_I16: return k == OV_I16; case B_I32: return k == OV_I32; case B_I64: return k == OV_I64 || k == OV_DURATION;
    case B_U8: return k == OV_U8 || k == OV_BOOL; case B_U16: return k == OV_U16; case B_U32: return k == OV_U32; case B_U64: return k == OV_U64;
    case B_F32: return k == OV_F32; case B_F64: return k == OV_F64; case B_BOOL: return k == OV_BOOL;
    case B_STR: return k == OV_STRING || k == OV_BYTES || k == OV_JSON;
    default: return k == OV_TIME;
  }
}

/* one value of a fixed-width column; returns 0 on a range error */
static int put_fixed(buf *b, const ora_value *v, int base, int prec, int dtype) {
  int nil = v->kind == OV_NIL;
  switch (base) {
    case B_I8: case B_U8: case B_BOOL: { uint8_t x = nil ? 0 : (v->kind == OV_BOOL ? (uint8_t)(v->v.b != 0) : (uint8_t)v->v.i); put(b, &x, 1); return 1; }
    case B_I16: case B_U16: { uint16_t x = nil ? 0 : (uint16_t)v->v.i; put(b, &x, 2); return 1; }
    case B_I32: case B_U32: { uint32_t x = nil ? 0 : (uint32_t)v->v.i; put(b, &x, 4); return 1; }
    case B_I64: case B_U64: { uint64_t x = nil ? 0 : v->v.u; put(b, &x, 8); return 1; }
    case B_F32: { float x = nil ? 0 : v->v.f32; put(b, &x, 4); return 1; }
    case B_F64: { double x = nil ? 0 : v->v.f64; put(b, &x, 8); return 1; }
  }
  int64_t s = nil ? 0 : v->v.t.sec; int32_t ns = nil ? 0 : v->v.t.nsec;
  if (!nil && (dtype == TFGPU_T_DATE || dtype == TFGPU_T_DATETIME)) { /* applyClickhouseDateBoundaries */
    const int64_t maxd = 4291747200ll;
    if (s > maxd || (s == maxd && ns > 0)) { s = maxd; ns = 0; }
    if (s < 0) { s = 0; ns = 0; }
  }
#define AFTER(lim) (s > (lim) || (s == (lim) && ns > 0))
  switch (base) {
    case B_DATE: { if (!nil && (s < 0 || AFTER(5662224000ll))) return 0; uint16_t x = (uint16_t)(s / 86400); put(b, &x, 2); return 1; }
    case B_DATE32: { if (!nil && (s < -2208988800ll || AFTER(10413705600ll))) return 0; int32_t x = (int32_t)(s / 86400); put(b, &x, 4); return 1; }
    case B_DT: { if (!nil && (s < 0 || AFTER(4294967295ll))) return 0; uint32_t x = (uint32_t)s; put(b, &x, 4); return 1; }
    default: {
      if (!nil && (s < -2208988800ll || AFTER(9223372036ll))) return 0;
      int64_t scale = 1; for (int k = prec; k < 9; k++) scale *= 10;
      int64_t x = nil ? 0 : (s * 1000000000ll + ns) / scale;
      put(b, &x, 8); return 1;
    }
  }
}

/* returns malloc'd block, or NULL (a type without encoder, a value that does not append as is, a time out of range) */
char *ora_ch_native_block(const ora_batch *rows, const char *const *names, const char *const *types, int ncols, uint64_t *len) {
  buf b = {0};
  put_varuint(&b, (uint64_t)ncols);
  put_varuint(&b, (uint64_t)rows->n);
  for (int c = 0; c < ncols; c++) {
    int nullable, prec, base = parse_type(types[c], &nullable, &prec);
    if (base == B_BAD) { free(b.p); return NULL; }
    put_str(&b, names[c], strlen(names[c]));
    put_str(&b, types[c], strlen(types[c]));
    /* the column's position in each item (ColumnNames are per item in Go) */
    if (nullable) for (int64_t r = 0; r < rows->n; r++) {
      const ora_item *it = &rows->items[r]; int at = -1;
      for (int k = 0; k < it->names->n; k++) if (!strcmp(it->names->names[k], names[c])) at = k;
      if (at < 0) { free(b.p); return NULL; }
      unsigned char z = it->values[at].kind == OV_NIL; put(&b, &z, 1);
    }
    for (int64_t r = 0; r < rows->n; r++) {
      const ora_item *it = &rows->items[r]; int at = -1;
      for (int k = 0; k < it->names->n; k++) if (!strcmp(it->names->names[k], names[c])) at = k;
      if (at < 0) { free(b.p); return NULL; }
      const ora_value *v = &it->values[at];
      if (v->kind != OV_NIL && !kind_fits(v->kind, base)) { free(b.p); return NULL; }
      if (base == B_STR) {
        if (v->kind == OV_NIL) put_varuint(&b, 0);
        else if (v->kind == OV_JSON && v->slen && v->s[0] == '"') { free(b.p); return NULL; } /* a Go string inside `any`: sent unquoted, not restated */
        else put_str(&b, v->s, v->slen);
      } else {
        int dtype = TFGPU_T_INVALID;
        for (int k = 0; it->schema && k < it->schema->ncols; k++) if (!strcmp(it->schema->cols[k].name, names[c])) dtype = it->schema->cols[k].dtype;
        if (!put_fixed(&b, v, base, prec, dtype)) { free(b.p); return NULL; }
      }
    }
  }
  *len = b.n;
  if (!b.p) b.p = (char *)malloc(1);
  return b.p;
}
