/*
 * oracle/ora_csv.c — CPU restatement of the CSV ingest path (SURVEY.md §8a
 * rows a15, a16, a18, a19):
 *   pkg/csv/reader.go:89-324            csv.Reader.ReadLine / splitString / sanitizeElement
 *   pkg/providers/s3/reader/registry/csv/reader_csv.go:186-247,266-452
 *                                       parseCSVRows / doParse / constructCI / getCorrespondingValue
 *   pkg/abstract/changeitem/strictify/strictify.go:18-181
 *   pkg/util/castx/caste.go:16-106
 * and of the third-party github.com/spf13/cast v1.7.1 (go.mod:64) conversions
 * those call (restated from its published source; value parity for these is
 * only indirectly pinned by the reference's tests — SURVEY.md §8c).
 * TEST INFRASTRUCTURE ONLY (see ora.h).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include "ora.h"

static char *dupn(const char *s, size_t n) { char *r = (char *)malloc(n + 1); if (n) memcpy(r, s, n); r[n] = 0; return r; }
static char *dups(const char *s) { return s ? dupn(s, strlen(s)) : NULL; }

/* ---- strings.TrimSpace (unicode.IsSpace over UTF-8) ---- */
static size_t space_prefix(const unsigned char *s, size_t n) {
  if (n == 0) return 0;
  unsigned char c = s[0];
  if (c == ' ' || (c >= 9 && c <= 13)) return 1;
  if (c == 0xC2 && n >= 2 && (s[1] == 0x85 || s[1] == 0xA0)) return 2;
  if (c == 0xE1 && n >= 3 && s[1] == 0x9A && s[2] == 0x80) return 3;                 /* U+1680 */
  if (c == 0xE2 && n >= 3) {
    if (s[1] == 0x80 && ((s[2] >= 0x80 && s[2] <= 0x8A) || s[2] == 0xA8 || s[2] == 0xA9 || s[2] == 0xAF)) return 3;
    if (s[1] == 0x81 && s[2] == 0x9F) return 3;                                       /* U+205F */
  }
  if (c == 0xE3 && n >= 3 && s[1] == 0x80 && s[2] == 0x80) return 3;                 /* U+3000 */
  return 0;
}
static size_t space_suffix(const unsigned char *s, size_t n) {
  if (n == 0) return 0;
  unsigned char c = s[n - 1];
  if (c == ' ' || (c >= 9 && c <= 13)) return 1;
  if (n >= 2 && s[n - 2] == 0xC2 && (c == 0x85 || c == 0xA0)) return 2;
  if (n >= 3) { size_t k = space_prefix(s + n - 3, 3); if (k == 3) return 3; }
  return 0;
}
static void trim_space(const char **ps, size_t *pn) {
  const unsigned char *s = (const unsigned char *)*ps; size_t n = *pn, k;
  while ((k = space_prefix(s, n)) > 0) { s += k; n -= k; }
  while ((k = space_suffix(s, n)) > 0) n -= k;
  *ps = (const char *)s; *pn = n;
}

/* sanitizeElement reader.go:273-324; returns malloc'd string or NULL + *code */
static char *sanitize(const tfgpu_csv_options *o, const char *s, size_t n, size_t *olen, int *code) {
  trim_space(&s, &n);
  *code = 0;
  if (o->quote_char == 0) { *olen = n; return dupn(s, n); }
  /* unquote :293-307 */
  if (n > 0) {
    char q = (char)o->quote_char;
    if (n == 1 && s[0] == q) { *code = TFGPU_ROW_QUOTE; return NULL; }
    if (s[0] == q && s[n - 1] == q) { s++; n -= 2; }
  }
  /* swapToSingleQuotes :311-324: DoubleQuoteStr = quote+quote → `"` */
  char q = (char)o->quote_char;
  if (!o->double_quote) {
    for (size_t i = 0; i + 1 < n; i++) if (s[i] == q && s[i + 1] == q) { *code = TFGPU_ROW_DOUBLE_QUOTE; return NULL; }
    *olen = n; return dupn(s, n);
  }
  char *out = (char *)malloc(n + 1); size_t w = 0;
  for (size_t i = 0; i < n;) {
    if (i + 1 < n && s[i] == q && s[i + 1] == q) { out[w++] = '"'; i += 2; }
    else out[w++] = s[i++];
  }
  out[w] = 0; *olen = w;
  return out;
}

typedef struct { char **f; size_t *l; int n, cap; } fields;
static void fpush(fields *fs, char *s, size_t l) {
  if (fs->n == fs->cap) { fs->cap = fs->cap ? fs->cap * 2 : 8; fs->f = (char **)realloc(fs->f, sizeof(char *) * (size_t)fs->cap); fs->l = (size_t *)realloc(fs->l, sizeof(size_t) * (size_t)fs->cap); }
  fs->f[fs->n] = s; fs->l[fs->n] = l; fs->n++;
}

/* splitString reader.go:220-271 (byte-wise: delimiter/quote/escape are ASCII,
 * UTF-8 continuation bytes can never equal them) */
static int split_string(const tfgpu_csv_options *o, const char *line, size_t n, fields *out) {
  int prev = -1, in_quotes = 0;
  size_t last_delim = 0, prev_delim = 0;
  for (size_t i = 0; i < n; i++) {
    int ch = (unsigned char)line[i];
    if (o->escape_char != 0 && o->escape_char == prev) {
      if (in_quotes) { prev = ch; continue; }
    }
    if (o->quote_char != 0 && ch == o->quote_char) { in_quotes = !in_quotes; prev = ch; continue; }
    if (ch == o->delimiter && !in_quotes) {
      last_delim = i;
      size_t l; int code;
      char *e = sanitize(o, line + prev_delim, last_delim - prev_delim, &l, &code);
      if (!e) return code;
      fpush(out, e, l);
      prev_delim = last_delim + 1;
    }
    prev = ch;
  }
  /* lastElement := line[lastDelimPosition+1:]  — NB: with no delimiter in the
   * line lastDelimPosition is 0, so the first byte is dropped (reader.go:263) */
  size_t from = last_delim + 1; if (from > n) from = n;
  size_t l; int code;
  char *e = sanitize(o, line + from, n - from, &l, &code);
  if (!e) return code;
  fpush(out, e, l);
  return 0;
}

/* checkCompleteQuotes reader.go:190-218 */
static int complete_quotes(const tfgpu_csv_options *o, const char *s, size_t n) {
  int prev = -1, in_quotes = 0;
  for (size_t i = 0; i < n; i++) {
    int ch = (unsigned char)s[i];
    if (o->escape_char != 0 && o->escape_char == prev) { if (in_quotes) { prev = ch; continue; } }
    if (o->quote_char != 0 && ch == o->quote_char) { in_quotes = !in_quotes; prev = ch; continue; }
    prev = ch;
  }
  return !in_quotes;
}

/* One ReadLine (reader.go:89-108).  Returns: 0 = got line (fields or nil),
 * 1 = EOF, >1 = tfgpu_rowerr for this line.  *pos advances. */
/* readAndDecodeLine :157-183: the line is cut at the raw '\n' byte, then decoded by the charmap decoder
 * (golang.org/x/text/encoding/charmap: one table entry per byte, U+FFFD for an undefined one).  Returns a
 * malloc'd UTF-8 copy (or NULL when no Encoding is set: the bytes are used as they are). */
static char *decode_line(const tfgpu_csv_options *o, const char *line, size_t ln, size_t *oln) {
  if (!o->encoding_table) return NULL;
  char *d = (char *)malloc(ln * 3 + 1); size_t w = 0;
  for (size_t i = 0; i < ln; i++) {
    uint32_t r = o->encoding_table[(unsigned char)line[i]];
    if (r < 0x80) d[w++] = (char)r;
    else if (r < 0x800) { d[w++] = (char)(0xC0 | (r >> 6)); d[w++] = (char)(0x80 | (r & 0x3F)); }
    else { if (r > 0xFFFF) r = 0xFFFD; d[w++] = (char)(0xE0 | (r >> 12)); d[w++] = (char)(0x80 | ((r >> 6) & 0x3F)); d[w++] = (char)(0x80 | (r & 0x3F)); }
  }
  *oln = w; return d;
}

static int read_line(const tfgpu_csv_options *o, const char *buf, uint64_t len, uint64_t *pos, fields *out, int *is_nil) {
  *is_nil = 0;
  if (o->newlines_in_value && o->quote_char != 0) { /* readMultiline :110-137 */
    char *full = NULL; size_t fl = 0;
    for (;;) {
      if (*pos >= len) { free(full); return 1; }
      const char *nl = (const char *)memchr(buf + *pos, '\n', len - *pos);
      if (!nl) { free(full); return 1; } /* incomplete last line dropped */
      size_t ln = (size_t)(nl - (buf + *pos)) + 1;
      const char *part = buf + *pos; *pos += ln;
      size_t dl = 0; char *dec = decode_line(o, part, ln, &dl);
      if (dec) { part = dec; ln = dl; }
      if (ln <= 1) { free(dec); continue; } /* "\n": skip empty line ("\r" alone cannot occur: lines end in \n) */
      full = (char *)realloc(full, fl + ln + 1); memcpy(full + fl, part, ln); fl += ln;
      free(dec);
      if (complete_quotes(o, full, fl)) break;
    }
    int rc = split_string(o, full, fl, out);
    free(full);
    return rc ? rc : 0;
  }
  if (*pos >= len) return 1;
  const char *nl = (const char *)memchr(buf + *pos, '\n', len - *pos);
  if (!nl) return 1;
  size_t ln = (size_t)(nl - (buf + *pos)) + 1;
  const char *line = buf + *pos; *pos += ln;
  size_t dl = 0; char *dec = decode_line(o, line, ln, &dl);
  if (dec) { line = dec; ln = dl; }
  int rc;
  if (o->quote_char == 0 && memchr(line, '"', ln)) rc = TFGPU_ROW_QUOTING_DISABLED; /* :182-184 */
  else if (ln <= 1) { *is_nil = 1; rc = 0; } /* readSingleLine :146-150 */
  else rc = split_string(o, line, ln, out);
  free(dec);
  return rc;
}

static void fields_free(fields *f) { for (int i = 0; i < f->n; i++) free(f->f[i]); free(f->f); free(f->l); memset(f, 0, sizeof *f); }

ora_csv_table *ora_csv_read_all(const tfgpu_csv_options *o, const void *bytes, uint64_t len) {
  ora_csv_table *t = (ora_csv_table *)calloc(1, sizeof *t);
  const char *buf = (const char *)bytes; uint64_t pos = 0;
  int64_t cap = 0, fcap = 0;
  for (;;) {
    fields f = {0}; int is_nil;
    int rc = read_line(o, buf, len, &pos, &f, &is_nil);
    if (rc == 1) { fields_free(&f); break; }
    if (t->nlines == cap) { cap = cap ? cap * 2 : 64; t->nfields = (int32_t *)realloc(t->nfields, sizeof(int32_t) * (size_t)cap); t->line_err = (int32_t *)realloc(t->line_err, sizeof(int32_t) * (size_t)cap); }
    t->line_err[t->nlines] = rc > 1 ? rc : 0;
    t->nfields[t->nlines] = (rc > 1 || is_nil) ? -1 : f.n;
    if (rc == 0 && !is_nil) {
      for (int i = 0; i < f.n; i++) {
        if (t->ntotal == fcap) { fcap = fcap ? fcap * 2 : 256; t->fields = (char **)realloc(t->fields, sizeof(char *) * (size_t)fcap); t->lens = (size_t *)realloc(t->lens, sizeof(size_t) * (size_t)fcap); }
        t->fields[t->ntotal] = f.f[i]; t->lens[t->ntotal] = f.l[i]; t->ntotal++;
      }
      free(f.f); free(f.l);
    } else fields_free(&f);
    t->nlines++;
  }
  t->consumed = pos;
  return t;
}
void ora_csv_table_free(ora_csv_table *t) {
  if (!t) return;
  for (int64_t i = 0; i < t->ntotal; i++) free(t->fields[i]);
  free(t->fields); free(t->lens); free(t->nfields); free(t->line_err); free(t);
}

/* ---- abstract.DefaultValue (pkg/abstract/change_item_builders.go:88-109) ---- */
static ora_value default_value(int dtype) {
  ora_value v; memset(&v, 0, sizeof v);
  switch (dtype) {
    case TFGPU_T_INT8: v.kind = OV_I8; break; case TFGPU_T_INT16: v.kind = OV_I16; break;
    case TFGPU_T_INT32: v.kind = OV_I32; break; case TFGPU_T_INT64: v.kind = OV_I64; break;
    case TFGPU_T_UINT8: v.kind = OV_U8; break; case TFGPU_T_UINT16: v.kind = OV_U16; break;
    case TFGPU_T_UINT32: v.kind = OV_U32; break; case TFGPU_T_UINT64: v.kind = OV_U64; break;
    case TFGPU_T_FLOAT32: v.kind = OV_F32; break; case TFGPU_T_FLOAT64: v.kind = OV_F64; break;
    case TFGPU_T_BYTES: case TFGPU_T_UTF8: v.kind = OV_STRING; v.s = dups(""); break;
    case TFGPU_T_BOOLEAN: v.kind = OV_BOOL; break;
    case TFGPU_T_ANY: v.kind = OV_JSON; v.s = dups("{}"); v.slen = 2; break;
    case TFGPU_T_DATE: case TFGPU_T_DATETIME: case TFGPU_T_TIMESTAMP: v.kind = OV_TIME; break;
    case TFGPU_T_INTERVAL: v.kind = OV_DURATION; break;
    default: v.kind = OV_NIL;
  }
  return v;
}

static int in_list(int n, const char *const *l, const char *s, size_t sl) {
  for (int i = 0; i < n; i++) if (strlen(l[i]) == sl && memcmp(l[i], s, sl) == 0) return 1;
  return 0;
}
static ora_value str_value(const char *s, size_t n) { ora_value v; memset(&v, 0, sizeof v); v.kind = OV_STRING; v.s = dupn(s, n); v.slen = n; return v; }

/* getCorrespondingValue reader_csv.go:345-452 */
static ora_value corresponding_value(const tfgpu_csv_options *o, const char *s, size_t n, int dtype) {
  ora_value v; memset(&v, 0, sizeof v);
  switch (dtype) {
    case TFGPU_T_BOOLEAN: /* parseBooleanValue :431-452 */
      if (o->strings_can_be_null && in_list(o->n_null_values, o->null_values, s, n)) { v.kind = OV_BOOL; v.v.b = 0; return v; }
      if (in_list(o->n_true_values, o->true_values, s, n)) { v.kind = OV_BOOL; v.v.b = 1; return v; }
      if (in_list(o->n_false_values, o->false_values, s, n)) { v.kind = OV_BOOL; v.v.b = 0; return v; }
      { int b; if (ora_parse_bool(s, n, &b) == 0) { v.kind = OV_BOOL; v.v.b = b; return v; } }
      return str_value(s, n);
    case TFGPU_T_DATE: case TFGPU_T_DATETIME: /* parseDateValue :405-415 */
      for (int i = 0; i < o->n_timestamp_parsers; i++) {
        int64_t sec; int32_t ns;
        if (ora_time_parse(o->timestamp_parsers[i], s, n, &sec, &ns) == 0) { v.kind = OV_TIME; v.v.t.sec = sec; v.v.t.nsec = ns; return v; }
      }
      return str_value(s, n);
    case TFGPU_T_TIMESTAMP: { /* parseTimestampValue :419-426 */
      int64_t iv;
      if (ora_parse_int(s, n, 10, 64, &iv) == 0) { v.kind = OV_TIME; v.v.t.sec = iv; v.v.t.nsec = 0; return v; }
      return str_value(s, n);
    }
    case TFGPU_T_FLOAT32: case TFGPU_T_FLOAT64: /* parseFloatValue :363-378 */
      if (o->decimal_point && o->decimal_point[0]) {
        const char *dp = o->decimal_point; size_t dl = strlen(dp);
        const char *hit = (const char *)memmem(s, n, dp, dl);
        char *rep; size_t rl;
        if (hit) { rl = n - dl + 1; rep = (char *)malloc(rl + 1); size_t k = (size_t)(hit - s); memcpy(rep, s, k); rep[k] = '.'; memcpy(rep + k + 1, hit + dl, n - k - dl); rep[rl] = 0; }
        else { rep = dupn(s, n); rl = n; }
        double d;
        if (ora_parse_float(rep, rl, 64, &d) == 0) { v.kind = OV_STRING; v.s = rep; v.slen = rl; return v; } /* err == nil */
        free(rep);
      }
      return str_value(s, n);
    default: /* parseNullValues :384-401 */
      if (o->quoted_strings_can_be_null) {
        const char *t = s; size_t tn = n;
        if (n >= 1 && s[0] == '"' && s[n - 1] == '"') { t = s + 1; tn = n >= 2 ? n - 2 : 0; }
        else if (n >= 1 && s[0] == '\'' && s[n - 1] == '\'') { t = s + 1; tn = n >= 2 ? n - 2 : 0; }
        if (in_list(o->n_null_values, o->null_values, t, tn)) return default_value(dtype);
      } else if (o->strings_can_be_null) {
        if (in_list(o->n_null_values, o->null_values, s, n)) return default_value(dtype);
      }
      return str_value(s, n);
  }
}

/* spf13/cast trimZeroDecimal */
static size_t trim_zero_decimal(const char *s, size_t n) {
  int found_zero = 0;
  for (size_t i = n; i > 0; i--) {
    char c = s[i - 1];
    if (c == '.') { if (found_zero) return i - 1; }
    else if (c == '0') found_zero = 1;
    else return n;
  }
  return n;
}

/* spf13/cast StringToDate: timeFormats list, v1.7.1 */
static int cast_string_to_date(const char *s, size_t n, int64_t *sec, int32_t *nsec) {
  static const char *fm[] = {
      "2006-01-02", "2006-01-02T15:04:05Z07:00", "2006-01-02T15:04:05", "Mon, 02 Jan 2006 15:04:05 -0700", "Mon, 02 Jan 2006 15:04:05 MST",
      "02 Jan 06 15:04 -0700", "02 Jan 06 15:04 MST", "Monday, 02-Jan-06 15:04:05 MST", "2006-01-02 15:04:05.999999999 -0700 MST",
      "2006-01-02T15:04:05-0700", "2006-01-02 15:04:05Z0700", "2006-01-02 15:04:05", "Mon Jan _2 15:04:05 2006", "Mon Jan _2 15:04:05 MST 2006",
      "Mon Jan 02 15:04:05 -0700 2006", "2006-01-02 15:04:05Z07:00", "02 Jan 2006", "2006-01-02 15:04:05 -07:00", "2006-01-02 15:04:05 -0700",
      "3:04PM", "Jan _2 15:04:05", "Jan _2 15:04:05.000", "Jan _2 15:04:05.000000", "Jan _2 15:04:05.000000000"};
  for (size_t i = 0; i < sizeof fm / sizeof *fm; i++) if (ora_time_parse(fm[i], s, n, sec, nsec) == 0) return 0;
  return 1;
}

/* castx.ToStringE caste.go:58-106 for the dynamic types we carry */
static char *castx_to_string(const ora_value *v, size_t *len) {
  char b[512]; size_t n;  /* FormatFloat(f, 'f', -1, 64) of 5e-324 is 326 bytes, of MaxFloat64 309 */
  switch (v->kind) {
    case OV_STRING: case OV_JSONNUM: case OV_BYTES: *len = v->slen; return dupn(v->s, v->slen);
    case OV_BOOL: return v->v.b ? (*len = 4, dups("true")) : (*len = 5, dups("false"));
    case OV_F64: n = ora_fmt_float(b, v->v.f64, 'f', 64); break;
    case OV_F32: n = ora_fmt_float(b, (double)v->v.f32, 'f', 32); break;
    case OV_I8: case OV_I16: case OV_I32: case OV_I64: n = ora_fmt_int(b, v->v.i); break;
    case OV_U8: case OV_U16: case OV_U32: case OV_U64: n = ora_fmt_uint(b, v->v.u); break;
    case OV_NIL: *len = 0; return dups("");
    case OV_TIME: n = ora_fmt_time_string(b, v->v.t.sec, v->v.t.nsec); break; /* fmt.Stringer */
    case OV_DURATION: n = ora_fmt_duration(b, v->v.i); break;
    default: return NULL;
  }
  *len = n; return dupn(b, n);
}

/* strictifyValue strictify.go:44-157; returns 0 or a tfgpu_rowerr */
static int strictify_value(ora_value *v, int dtype) {
  if (v->kind == OV_NIL) return 0;
  int is_str = v->kind == OV_STRING || v->kind == OV_JSONNUM;
  switch (dtype) {
    case TFGPU_T_BOOLEAN: {
      if (v->kind == OV_BOOL) return 0;
      if (is_str) { int b; if (ora_parse_bool(v->s, v->slen, &b)) return TFGPU_ROW_CAST; ora_value_free(v); v->kind = OV_BOOL; v->v.b = b; return 0; }
      if (v->kind >= OV_I8 && v->kind <= OV_I64) { int b = v->v.i != 0; v->kind = OV_BOOL; v->v.b = b; return 0; }
      if (v->kind >= OV_U8 && v->kind <= OV_U64) { int b = v->v.u != 0; v->kind = OV_BOOL; v->v.b = b; return 0; }
      if (v->kind == OV_F64) { int b = v->v.f64 != 0; v->kind = OV_BOOL; v->v.b = b; return 0; }
      if (v->kind == OV_F32) { int b = v->v.f32 != 0; v->kind = OV_BOOL; v->v.b = b; return 0; }
      return TFGPU_ROW_CAST;
    }
    case TFGPU_T_INT8: case TFGPU_T_INT16: case TFGPU_T_INT32: case TFGPU_T_INT64: {
      static const int64_t lo[] = {INT8_MIN, INT16_MIN, INT32_MIN, INT64_MIN}, hi[] = {INT8_MAX, INT16_MAX, INT32_MAX, INT64_MAX};
      int k = dtype - TFGPU_T_INT8; int64_t x;
      if (is_str) { /* cast.ToIntNE(string): ParseInt(trimZeroDecimal(s), 0, 0) */
        size_t tn = trim_zero_decimal(v->s, v->slen);
        if (ora_parse_int(v->s, tn, 0, 64, &x)) return TFGPU_ROW_CAST;
      } else if (v->kind >= OV_I8 && v->kind <= OV_I64) x = v->v.i;
      else if (v->kind >= OV_U8 && v->kind <= OV_U64) x = (int64_t)v->v.u;
      else if (v->kind == OV_F64) x = (int64_t)v->v.f64; else if (v->kind == OV_F32) x = (int64_t)v->v.f32;
      else if (v->kind == OV_BOOL) x = v->v.b;
      else return TFGPU_ROW_CAST;
      if (x < lo[k] || x > hi[k]) return TFGPU_ROW_RANGE; /* toSignedInt :159-169 */
      ora_value_free(v); v->kind = OV_I8 + k; v->v.i = x; return 0;
    }
    case TFGPU_T_UINT8: case TFGPU_T_UINT16: case TFGPU_T_UINT32: case TFGPU_T_UINT64: {
      static const uint64_t hi[] = {UINT8_MAX, UINT16_MAX, UINT32_MAX, UINT64_MAX};
      int k = dtype - TFGPU_T_UINT8; uint64_t x;
      if (is_str) {
        size_t tn = trim_zero_decimal(v->s, v->slen);
        if (dtype == TFGPU_T_UINT64) { if (ora_parse_uint(v->s, tn, 0, 64, &x)) return TFGPU_ROW_CAST; } /* ToUint64E: ParseUint */
        else { int64_t sx; if (ora_parse_int(v->s, tn, 0, 64, &sx)) return TFGPU_ROW_CAST; if (sx < 0) return TFGPU_ROW_CAST; x = (uint64_t)sx; }
      } else if (v->kind >= OV_I8 && v->kind <= OV_I64) { if (v->v.i < 0) return TFGPU_ROW_CAST; x = (uint64_t)v->v.i; }
      else if (v->kind >= OV_U8 && v->kind <= OV_U64) x = v->v.u;
      else if (v->kind == OV_F64) { if (v->v.f64 < 0) return TFGPU_ROW_CAST; x = (uint64_t)v->v.f64; }
      else if (v->kind == OV_F32) { if (v->v.f32 < 0) return TFGPU_ROW_CAST; x = (uint64_t)v->v.f32; }
      else if (v->kind == OV_BOOL) x = (uint64_t)v->v.b;
      else return TFGPU_ROW_CAST;
      if (x > hi[k]) return TFGPU_ROW_RANGE; /* toUnsignedInt :171-181 */
      ora_value_free(v); v->kind = OV_U8 + k; v->v.u = x; return 0;
    }
    case TFGPU_T_FLOAT32: {
      double d;
      if (v->kind == OV_F32) return 0;
      if (is_str) { int rc = ora_parse_float(v->s, v->slen, 32, &d); if (rc) return TFGPU_ROW_CAST; }
      else if (v->kind == OV_F64) d = v->v.f64;
      else if (v->kind >= OV_I8 && v->kind <= OV_I64) d = (double)v->v.i; else if (v->kind >= OV_U8 && v->kind <= OV_U64) d = (double)v->v.u;
      else if (v->kind == OV_BOOL) d = v->v.b; else return TFGPU_ROW_CAST;
      ora_value_free(v); v->kind = OV_F32; v->v.f32 = (float)d; return 0;
    }
    case TFGPU_T_FLOAT64: { /* castx.ToJSONNumberE caste.go:41-55 */
      size_t sl; char *s = castx_to_string(v, &sl);
      if (!s) return TFGPU_ROW_CAST;
      /* fastfloat.Parse (valyala/fastjson v1.6.4): decimal/exponent/inf/nan grammar; or Int64 */
      double d; int64_t iv; int ok = 0;
      {
        /* fastfloat.Parse accepts [-+]digits[.digits][e[+-]digits], "inf"/"nan" (case-insens.) */
        const char *p = s; const char *e = s + sl;
        if (p < e && (*p == '-' || *p == '+')) p++;
        const char *d0 = p; while (p < e && *p >= '0' && *p <= '9') p++;
        int nd = (int)(p - d0);
        if (p < e && *p == '.') { p++; const char *f0 = p; while (p < e && *p >= '0' && *p <= '9') p++; if (p == f0) nd = 0; else nd += (int)(p - f0); }
        if (nd > 0 && p < e && (*p == 'e' || *p == 'E')) { p++; if (p < e && (*p == '-' || *p == '+')) p++; const char *x0 = p; while (p < e && *p >= '0' && *p <= '9') p++; if (p == x0) nd = 0; }
        if (nd > 0 && p == e) ok = 1;
        if (!ok) { const char *q = s; if (*q == '-' || *q == '+') q++; if (!strcasecmp(q, "inf") || !strcasecmp(q, "infinity") || !strcasecmp(q, "nan")) ok = 1; }
        (void)d;
      }
      if (!ok && ora_parse_int(s, sl, 10, 64, &iv) == 0) ok = 1;
      if (!ok) { free(s); return TFGPU_ROW_CAST; }
      ora_value_free(v); v->kind = OV_JSONNUM; v->s = s; v->slen = sl; return 0;
    }
    case TFGPU_T_BYTES: /* castx.ToByteSliceE */
      if (v->kind == OV_BYTES) return 0;
      if (v->kind == OV_STRING) { v->kind = OV_BYTES; return 0; }
      return TFGPU_ROW_CAST;
    case TFGPU_T_UTF8: { /* castx.ToStringE */
      if (v->kind == OV_STRING) return 0;
      size_t sl; char *s = castx_to_string(v, &sl);
      if (!s) return TFGPU_ROW_CAST;
      ora_value_free(v); v->kind = OV_STRING; v->s = s; v->slen = sl; return 0;
    }
    case TFGPU_T_DATE: case TFGPU_T_DATETIME: case TFGPU_T_TIMESTAMP: { /* cast.ToTimeE */
      if (v->kind == OV_TIME) return 0;
      int64_t sec; int32_t ns = 0;
      if (v->kind == OV_STRING) { if (cast_string_to_date(v->s, v->slen, &sec, &ns)) return TFGPU_ROW_CAST; }
      else if (v->kind == OV_JSONNUM) { if (ora_parse_int(v->s, v->slen, 10, 64, &sec)) return TFGPU_ROW_CAST; }
      else if (v->kind >= OV_I8 && v->kind <= OV_I64) sec = v->v.i;
      else if (v->kind >= OV_U8 && v->kind <= OV_U64) sec = (int64_t)v->v.u;
      else return TFGPU_ROW_CAST;
      ora_value_free(v); v->kind = OV_TIME; v->v.t.sec = sec; v->v.t.nsec = ns; return 0;
    }
    case TFGPU_T_INTERVAL:
      if (v->kind == OV_DURATION) return 0;
      if (v->kind >= OV_I8 && v->kind <= OV_I64) { v->kind = OV_DURATION; return 0; }
      if (v->kind == OV_STRING) { /* cast.ToDurationE(string): time.ParseDuration, "ns" appended to a text without unit letters */
        int64_t d;
        if (ora_cast_string_to_duration(v->s, v->slen, &d)) return TFGPU_ROW_CAST;
        ora_value_free(v); v->kind = OV_DURATION; v->v.i = d; return 0;
      }
      return TFGPU_ROW_HOST_FALLBACK; /* the other Go kinds under an interval column (floats, json.Number): not restated */
    case TFGPU_T_ANY: return 0;
  }
  return TFGPU_ROW_CAST;
}

/* s3_reader.SystemColumnNames (pkg/providers/s3/reader/util.go:24-32): 1 = __file_name, 2 = __row_index */
static int system_col(const char *name) {
  if (!name) return 0;
  if (strcmp(name, "__file_name") == 0) return 1;
  if (strcmp(name, "__row_index") == 0) return 2;
  return 0;
}

/* Test hook: getCorrespondingValue alone (before Strictify), as TestParse{Float,Null,Date,Boolean}Value call it
 * (reader_csv_test.go:281-430).  Returns a one-item batch holding the value. */
ora_batch *ora_csv_corresponding_value(const tfgpu_csv_options *o, const char *s, uint64_t n, int dtype) {
  ora_batch *b = ora_batch_new();
  ora_item *it = ora_batch_push(b);
  it->kind = TFGPU_K_INSERT; it->ns = dups(""); it->table = dups(""); it->part_id = dups("");
  it->nvalues = 1; it->values = (ora_value *)calloc(1, sizeof(ora_value));
  it->values[0] = corresponding_value(o, s, (size_t)n, dtype);
  return b;
}

ora_batch *ora_csv_parse(const tfgpu_csv_options *o, const tfgpu_schema *schema, const char *ns, const char *table,
                         const void *bytes, uint64_t len, uint64_t *consumed) {
  ora_batch *b = ora_batch_new();
  ora_schema *sch = ora_schema_from(schema);
  ora_names *nm = (ora_names *)calloc(1, sizeof *nm);
  nm->refs = 1; nm->n = schema->ncols; nm->names = (char **)calloc((size_t)(schema->ncols ? schema->ncols : 1), sizeof(char *));
  for (int i = 0; i < schema->ncols; i++) nm->names[i] = dups(schema->cols[i].name);
  int *index = (int *)calloc((size_t)(schema->ncols ? schema->ncols : 1), sizeof(int));
  for (int i = 0; i < schema->ncols; i++) index[i] = system_col(schema->cols[i].name) ? 0 : atoi(schema->cols[i].path ? schema->cols[i].path : "0");
  const char *buf = (const char *)bytes; uint64_t pos = 0;
  int64_t row = 0;
  for (int64_t k = 0; k < o->skip_rows; k++) { /* skipRows: lines consumed with ReadLine */
    fields f = {0}; int is_nil; int rc = read_line(o, buf, len, &pos, &f, &is_nil); fields_free(&f); if (rc == 1) break;
  }
  for (;; row++) { /* parseCSVRows :196-231 */
    fields f = {0}; int is_nil;
    int rc = read_line(o, buf, len, &pos, &f, &is_nil);
    if (rc == 1) { fields_free(&f); break; }
    if (rc > 1) { ora_batch_add_error(b, row, rc, "csv.ReadLine"); fields_free(&f); continue; }
    /* constructCI :266-341 (system columns not modelled) */
    ora_value *vals = (ora_value *)calloc((size_t)(schema->ncols ? schema->ncols : 1), sizeof(ora_value));
    int err = 0;
    for (int i = 0; i < schema->ncols && !err; i++) {
      int dt = schema->cols[i].dtype;
      int sys = system_col(schema->cols[i].name);
      if (sys) { /* :275-290 */
        if (o->hide_system_cols) continue; /* nil */
        if (sys == 1) vals[i] = str_value(o->file_name ? o->file_name : "", o->file_name ? strlen(o->file_name) : 0);
        else { vals[i].kind = OV_U64; vals[i].v.u = o->row_number_base + (uint64_t)row; }
        continue;
      }
      if (index[i] < 0) vals[i] = default_value(dt);
      else if (index[i] >= f.n) { if (o->include_missing_columns) vals[i] = default_value(dt); else err = TFGPU_ROW_MISSING_CELL; }
      else vals[i] = corresponding_value(o, f.f[index[i]], f.l[index[i]], dt);
    }
    /* strictify.Strictify :18-42 */
    for (int i = 0; i < schema->ncols && !err; i++) err = strictify_value(&vals[i], schema->cols[i].dtype);
    fields_free(&f);
    if (err) { for (int i = 0; i < schema->ncols; i++) ora_value_free(&vals[i]); free(vals); ora_batch_add_error(b, row, err, "csv.doParse"); continue; }
    ora_item *it = ora_batch_push(b);
    it->kind = TFGPU_K_INSERT; it->ns = dups(ns ? ns : ""); it->table = dups(table ? table : ""); it->part_id = dups("");
    it->names = nm; nm->refs++; it->schema = sch; sch->refs++;
    it->nvalues = schema->ncols; it->values = vals; it->src_row = row;
  }
  if (consumed) *consumed = pos;
  free(index);
  if (--nm->refs == 0) { for (int i = 0; i < nm->n; i++) free(nm->names[i]); free(nm->names); free(nm); }
  ora_schema_unref(sch);
  return b;
}

/* csv.Splitter.ConsumeRow / updateState (pkg/csv/splitter.go:37-85): offsets one past the '\n' that ends each entry.
 * Returns the number of entries; ends is malloc'd. */
int64_t ora_csv_split_rows(const void *bytes, uint64_t len, uint64_t **ends) {
  enum { OUTSIDE, OPEN, CLOSING } st = OUTSIDE;
  const unsigned char *p = (const unsigned char *)bytes;
  uint64_t *e = NULL; int64_t n = 0, cap = 0;
  for (uint64_t i = 0; i < len; i++) {
    unsigned char c = p[i];
    switch (st) {
      case OUTSIDE: if (c == '"') st = OPEN; break;
      case OPEN: if (c == '"') st = CLOSING; break;
      case CLOSING: st = c == '"' ? OPEN : OUTSIDE; break;
    }
    if (c == '\n' && st == OUTSIDE) { /* ReadSlice returned a whole line and the state is outside: the row is complete */
      if (n == cap) { cap = cap ? cap * 2 : 64; e = (uint64_t *)realloc(e, sizeof(uint64_t) * (size_t)cap); }
      e[n++] = i + 1;
    }
  }
  *ends = e;
  return n;
}

/* strictify.Strictify (pkg/abstract/changeitem/strictify/strictify.go:17-42) over every item of a batch, the way the
 * strictifying serializers call it (pkg/serializer/strictify.go:24-36): ColumnValues become the strict Go type of the column's
 * DataType in the item's TableSchema; a column the schema does not name keeps its value; an item one of whose values cannot be
 * converted is left unchanged ("all values remain unchanged") and reported as (row index, tfgpu_rowerr).  In place. */
ora_batch *ora_strictify(ora_batch *b) {
  for (int64_t r = 0; r < b->n; r++) {
    ora_item *it = &b->items[r];
    if (!it->schema || !it->names) continue;
    ora_value *tmp = (ora_value *)calloc((size_t)(it->nvalues ? it->nvalues : 1), sizeof(ora_value));
    int err = 0;
    for (int i = 0; i < it->nvalues; i++) tmp[i] = ora_value_clone(&it->values[i]);
    for (int i = 0; i < it->nvalues && !err; i++) {
      const char *nm = i < it->names->n ? it->names->names[i] : NULL;
      for (int k = 0; nm && k < it->schema->ncols; k++)
        if (strcmp(it->schema->cols[k].name, nm) == 0) { err = strictify_value(&tmp[i], it->schema->cols[k].dtype); break; }
    }
    if (err) {
      for (int i = 0; i < it->nvalues; i++) ora_value_free(&tmp[i]);
      free(tmp);
      ora_batch_add_error(b, r, err, "strictify");
      continue;
    }
    for (int i = 0; i < it->nvalues; i++) ora_value_free(&it->values[i]);
    free(it->values);
    it->values = tmp;
  }
  return b;
}
