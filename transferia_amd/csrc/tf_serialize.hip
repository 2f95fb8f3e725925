// tf_serialize.hip — sink-side marshalling from device columns:
//
//   TFGPU_FMT_CH_JSON_EACH_ROW  httpuploader.MarshalCItoJSON / marshalValue / marshalNumericValue /
//                               marshalTime / writeQuoted (pkg/providers/clickhouse/httpuploader/marshal.go:65-419)
//   TFGPU_FMT_JSON              jsonSerializer (pkg/serializer/json.go:29-83, json_format.go:32-82) under
//                               batchSerializer (batch.go:206-219): encoding/json of a map — keys sorted,
//                               SetEscapeHTML(false)
//   TFGPU_FMT_CSV               csvSerializer (csv.go:22-74, csv_format.go:32-127) + encoding/csv quoting
//
// Columns in, row-major text out: (1) one lane per CELL computes the cell's byte length, (2) one lane per
// ROW turns the lengths into in-row offsets, (3) prefix scan of the row lengths, (4) one lane per cell
// writes key + value at its final position.  Every value is rendered by the same templated emitter in
// passes (1) and (4) — a counting sink, then a writing sink — so lengths and bytes cannot disagree.
// HBM-bound byte kernels: algorithmic traffic B_bin + B_text per row.
#include <algorithm>

#include "tf_emit.hpp"

namespace tf {

struct SCol {
  DCol c;
  uint32_t pre_off, pre_len;  // key / separator bytes emitted before the value
  uint32_t ch_flags;          // TFGPU_CH_*
  uint32_t prec;              // DateTime64 precision
};

struct SerParams {
  const SCol *cols;
  int32_t ncols;
  int64_t nrows;
  int32_t format, any_as_string, closing_newline;
  const uint8_t *blob;     // prefixes
  uint32_t *cell;          // [ncols][nrows]: length, then offset inside the row
  uint32_t *row_len;       // [nrows+1] → row offsets after the scan
  int32_t *last_present;   // CH: index of the last emitted column per row, -1 = none
  unsigned long long *total64;  // sum of the row lengths (the offsets are 32-bit)
  uint8_t *out;
};

// strconv.ParseFloat(s, 64) overflows to ±Inf (err = ErrRange) exactly when |s| >= 2^1024 - 2^970
// (the value half-way between MaxFloat64 and 2^1024; ties round to the even neighbour = overflow).
__constant__ char F64_OVERFLOW_DIGITS[310] =
    "179769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207096330286416692887910946555547851"
    "940402630657488671505820681908902000708383676273854845817711531764475730270069855571366959622842914819860834936475292719074168444365"
    "510704342711559699508093042880177904174497792";
// +1 / -1 if the JSON number text overflows float64, 0 otherwise (inf/nan words and malformed text: 0)
__device__ int jsonnum_overflow(const uint8_t *p, uint32_t n) {
  uint32_t i = 0; int sign = 1;
  if (n && p[0] == '-') { sign = -1; i = 1; } else if (n && p[0] == '+') i = 1;
  // mantissa: value = 0.d1d2d3… × 10^lead, d1 = first non-zero digit
  int int_digits = 0, frac_zeros = 0; uint32_t first = 0;
  bool seen_dot = false, started = false, bad = false;
  while (i < n) {
    const uint32_t c = p[i];
    if (c == '.') { if (seen_dot) bad = true; seen_dot = true; }
    else if (c >= '0' && c <= '9') {
      if (!started && c != '0') { started = true; first = i; }
      if (started) { if (!seen_dot) int_digits++; } else if (seen_dot) frac_zeros++;
    } else break;
    i++;
  }
  int e10 = 0;
  if (i < n) {
    const uint32_t c = p[i];
    if (c != 'e' && c != 'E') bad = true;
    i++;
    int es = 1;
    if (i < n && p[i] == '-') { es = -1; i++; } else if (i < n && p[i] == '+') i++;
    if (i >= n) bad = true;
    while (i < n) {
      const uint32_t d = p[i];
      if (d < '0' || d > '9') bad = true; else if (e10 < 100000) e10 = e10 * 10 + (int)(d - '0');
      i++;
    }
    e10 *= es;
  }
  if (bad || !started) return 0;
  const int lead = (int_digits > 0 ? int_digits : -frac_zeros) + e10;
  if (lead > 309) return sign;
  if (lead < 309) return 0;
  // same magnitude as the threshold: compare digit strings
  uint32_t k = 0; int verdict = 2;  // 2 = undecided
  for (uint32_t q = first; q < n && k < 309 && verdict == 2; q++) {
    const uint32_t c = p[q];
    if (c == '.') continue;
    if (c < '0' || c > '9') break;
    const uint32_t t = (uint8_t)F64_OVERFLOW_DIGITS[k++];
    if (c != t) verdict = c > t ? 1 : 0;
  }
  if (verdict == 2) { verdict = 1; for (; k < 309; k++) if (F64_OVERFLOW_DIGITS[k] != '0') verdict = 0; }  // input ran out: it is smaller
  return verdict ? sign : 0;
}

__device__ __forceinline__ int64_t pow10_i64(int k) { int64_t r = 1; while (k-- > 0) r *= 10; return r; }

// ---- one value, ClickHouse JSONEachRow (marshalValue).  Returns false when the column is skipped. ----
template <class S> __device__ bool emit_ch_value(S &s, const SCol &sc, int64_t r, int any_as_string) {
  const DCol &c = sc.c;
  const bool is_text = c.repr == TFGPU_R_STRING || c.repr == TFGPU_R_BYTES;
  const uint8_t *vp = nullptr; uint32_t vn = 0;
  if (c.offsets) { vp = c.data + c.offsets[r]; vn = c.offsets[r + 1] - c.offsets[r]; }
  const uint32_t fl = sc.ch_flags;
  if ((fl & TFGPU_CH_DECIMAL) && is_text) { put_bytes(s, vp, vn); return true; }
  uint8_t t[64];
  switch (c.dtype) {
    case TFGPU_T_INT8: case TFGPU_T_INT16: case TFGPU_T_INT32: case TFGPU_T_INT64: case TFGPU_T_UINT8: case TFGPU_T_UINT16:
    case TFGPU_T_UINT32: case TFGPU_T_UINT64: case TFGPU_T_FLOAT32: case TFGPU_T_FLOAT64: case TFGPU_T_INTERVAL: {
      const bool q = fl & TFGPU_CH_STRING;
      if (c.repr >= TFGPU_R_INT8 && c.repr <= TFGPU_R_UINT64) { if (q) s.put('"'); emit_int(s, c, r); if (q) s.put('"'); return true; }
      if (c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) { if (q) s.put('"'); emit_float_f(s, c, r); if (q) s.put('"'); return true; }
      if (c.repr == TFGPU_R_JSONNUM) {
        if (q) s.put('"');
        const int o = jsonnum_overflow(vp, vn);
        if (o > 0) put_lit(s, "inf"); else if (o < 0) put_lit(s, "-inf"); else put_bytes(s, vp, vn);
        if (q) s.put('"');
        return true;
      }
      break;
    }
    case TFGPU_T_BYTES: case TFGPU_T_UTF8:
      if (c.repr == TFGPU_R_STRING) { emit_ch_quoted(s, vp, vn); return true; }
      if (c.repr == TFGPU_R_BYTES) {
        if (fl & TFGPU_CH_ARRAY) { s.put('['); for (uint32_t i = 0; i < vn; i++) { if (i) s.put(','); int n = dev::fmt_u64(t, vp[i]); emit_small(s, t, n); } s.put(']'); }
        else emit_ch_quoted(s, vp, vn);
        return true;
      }
      break;
    case TFGPU_T_BOOLEAN:
      if (c.repr == TFGPU_R_BOOL) { put_lit(s, ((const uint8_t *)c.values)[r] ? "true" : "false"); return true; }
      break;
    case TFGPU_T_DATE: case TFGPU_T_DATETIME: case TFGPU_T_TIMESTAMP:
      if (c.repr == TFGPU_R_TIME) {  // marshalTime :65-80
        const int64_t sec = ((const int64_t *)c.values)[r]; const int32_t ns = c.nanos ? c.nanos[r] : 0;
        if (fl & TFGPU_CH_STRING) { s.put('"'); int n = dev::fmt_time_string(t, sec, ns); emit_small(s, t, n); s.put('"'); }
        else if (fl & TFGPU_CH_DATETIME64) {
          int64_t full = sec * 1000000000LL + ns;
          if (sc.prec > 0 && sc.prec < 9) full = full / pow10_i64(9 - (int)sc.prec);
          int n = dev::fmt_i64(t, full); emit_small(s, t, n);
        } else if (fl & TFGPU_CH_DATE) { s.put('"'); int n = dev::fmt_date(t, sec); emit_small(s, t, n); s.put('"'); }
        else { int n = dev::fmt_i64(t, sec); emit_small(s, t, n); }
        return true;
      }
      break;
    default: break;
  }
  // marshalGeneric :318-359
  if (c.repr == TFGPU_R_STRING) { emit_ch_quoted(s, vp, vn); return true; }
  if (c.repr == TFGPU_R_BYTES) {
    if (fl & TFGPU_CH_ARRAY) { s.put('['); for (uint32_t i = 0; i < vn; i++) { if (i) s.put(','); int n = dev::fmt_u64(t, vp[i]); emit_small(s, t, n); } s.put(']'); }
    else emit_ch_quoted(s, vp, vn);
    return true;
  }
  // json.Marshal(v) of the remaining Go types, double-marshalled when the target is a string
  const bool dbl = c.dtype != TFGPU_T_ANY || any_as_string || (fl & TFGPU_CH_STRING);
  if (c.repr >= TFGPU_R_INT8 && c.repr <= TFGPU_R_UINT64) { if (dbl) s.put('"'); emit_int(s, c, r); if (dbl) s.put('"'); return true; }
  if (c.repr == TFGPU_R_BOOL) { if (dbl) s.put('"'); put_lit(s, ((const uint8_t *)c.values)[r] ? "true" : "false"); if (dbl) s.put('"'); return true; }
  if (c.repr == TFGPU_R_DURATION) { if (dbl) s.put('"'); int n = dev::fmt_i64(t, ((const int64_t *)c.values)[r]); emit_small(s, t, n); if (dbl) s.put('"'); return true; }
  if (c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) { if (dbl) s.put('"'); emit_float_json(s, c, r); if (dbl) s.put('"'); return true; }
  if (c.repr == TFGPU_R_JSONNUM || c.repr == TFGPU_R_JSON) {
    if (c.repr == TFGPU_R_JSONNUM && vn == 0) { if (dbl) s.put('"'); s.put('0'); if (dbl) s.put('"'); return true; }  // json.Number("") encodes as 0
    if (vn == 4 && vp[0] == 'n' && vp[1] == 'u' && vp[2] == 'l' && vp[3] == 'l') return false;  // value is null: skip the column
    if (dbl) emit_json_string(s, vp, vn, true); else put_bytes(s, vp, vn);
    return true;
  }
  return true;  // unreachable: the host rejects the remaining (dtype, repr) pairs
}

// ---- one value, encoding/json (jsonSerializer) ----
template <class S> __device__ __forceinline__ void emit_json_value(S &s, const SCol &sc, int64_t r, int any_as_string) {
  emit_json_cell(s, sc.c, r, any_as_string, false);  // SetEscapeHTML(false), json_format.go
}

// ---- one field, encoding/csv over toCsvValue ----
__device__ bool csv_needs_quotes(const uint8_t *p, uint32_t n) {  // Writer.fieldNeedsQuotes, Comma ','
  if (n == 0) return false;
  if (n == 2 && p[0] == '\\' && p[1] == '.') return true;
  for (uint32_t i = 0; i < n; i++) { const uint32_t c = p[i]; if (c == ',' || c == '"' || c == '\r' || c == '\n') return true; }
  const uint32_t c = p[0];
  if (c == ' ' || (c >= 9 && c <= 13)) return true;
  if (c == 0xC2 && n >= 2 && (p[1] == 0x85 || p[1] == 0xA0)) return true;
  if (n >= 3 && (c == 0xE1 || c == 0xE2 || c == 0xE3)) {
    const uint32_t d = p[1], e = p[2];
    if (c == 0xE1 && d == 0x9A && e == 0x80) return true;
    if (c == 0xE3 && d == 0x80 && e == 0x80) return true;
    if (c == 0xE2 && ((d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) || (d == 0x81 && e == 0x9F))) return true;
  }
  return false;
}
template <class S> __device__ void emit_csv_field(S &s, const SCol &sc, int64_t r) {
  const DCol &c = sc.c;
  if (!is_valid(c, r)) return;  // nil → ""
  const uint8_t *vp = nullptr; uint32_t vn = 0;
  if (c.offsets) { vp = c.data + c.offsets[r]; vn = c.offsets[r + 1] - c.offsets[r]; }
  uint8_t t[64];
  if (c.dtype == TFGPU_T_BYTES) { emit_base64(s, vp, vn); return; }  // repr is []byte (host-checked)
  if (c.dtype == TFGPU_T_ANY && c.repr == TFGPU_R_STRING) {           // json.Marshal(string): always holds '"'
    s.put('"'); CsvQuoteSink<S> q{s}; emit_json_string(q, vp, vn, true); s.put('"');
    return;
  }
  switch (c.repr) {
    case TFGPU_R_STRING: case TFGPU_R_BYTES: case TFGPU_R_JSONNUM: case TFGPU_R_JSON:
      if (csv_needs_quotes(vp, vn)) { s.put('"'); CsvQuoteSink<S> q{s}; put_bytes(q, vp, vn); s.put('"'); }
      else put_bytes(s, vp, vn);
      return;
    case TFGPU_R_BOOL: put_lit(s, ((const uint8_t *)c.values)[r] ? "true" : "false"); return;
    case TFGPU_R_TIME: { int n = dev::fmt_time_string(t, ((const int64_t *)c.values)[r], c.nanos ? c.nanos[r] : 0); emit_small(s, t, n); return; }  // fmt.Stringer
    case TFGPU_R_DURATION: { int n = dev::fmt_duration(t, ((const int64_t *)c.values)[r]); emit_small(s, t, n); return; }
    case TFGPU_R_FLOAT32: case TFGPU_R_FLOAT64: emit_float_f(s, c, r); return;
    default: emit_int(s, c, r);
  }
}

template <class S> __device__ __forceinline__ bool emit_cell(S &s, const SerParams &p, const SCol &sc, int64_t r) {
  switch (p.format) {
    case TFGPU_FMT_CH_JSON_EACH_ROW: {
      if (!is_valid(sc.c, r)) return false;  // nil values are omitted (marshal.go:100-102)
      // key, value, then one byte for ',' or '}' written by the caller
      CountSink probe;  // a column whose value marshals to null is dropped together with its key
      (void)probe;
      put_bytes(s, p.blob + sc.pre_off, sc.pre_len);
      return emit_ch_value(s, sc, r, p.any_as_string);
    }
    case TFGPU_FMT_JSON:
      put_bytes(s, p.blob + sc.pre_off, sc.pre_len);
      emit_json_value(s, sc, r, p.any_as_string);
      return true;
    default:
      put_bytes(s, p.blob + sc.pre_off, sc.pre_len);
      emit_csv_field(s, sc, r);
      return true;
  }
}

// (1) cell lengths; item = column * nrows + row
__global__ void __launch_bounds__(256) ser_cell_len(SerParams p) {
  const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= (int64_t)p.ncols * p.nrows) return;
  const int32_t ci = (int32_t)(it / p.nrows); const int64_t r = it - (int64_t)ci * p.nrows;
  CountSink s;
  const bool present = emit_cell(s, p, p.cols[ci], r);
  uint32_t n = present ? s.n : 0;
  if (p.format == TFGPU_FMT_CH_JSON_EACH_ROW && present) n += 1;  // its ',' (or the closing '}' for the last one)
  p.cell[it] = n;
}
// (2) per row: lengths → offsets inside the row, row length, last emitted column
__global__ void __launch_bounds__(256) ser_row_layout(SerParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  uint32_t off = (p.format == TFGPU_FMT_CSV) ? 0u : 1u;  // '{'
  int32_t last = -1;
  for (int32_t c = 0; c < p.ncols; c++) {
    const uint32_t n = p.cell[(int64_t)c * p.nrows + r];
    p.cell[(int64_t)c * p.nrows + r] = off;
    if (n) last = c;
    off += n;
  }
  uint32_t tail;
  if (p.format == TFGPU_FMT_CH_JSON_EACH_ROW) tail = last < 0 ? 2u : 1u;                       // "}\n" or "\n" (the '}' replaced a ',')
  else if (p.format == TFGPU_FMT_JSON) tail = 1u + ((p.closing_newline || r + 1 < p.nrows) ? 1u : 0u);  // '}' + newline / separator
  else tail = 1u;                                                                                // '\n'
  p.row_len[r] = off + tail;
  if (p.last_present) p.last_present[r] = last;
  atomicAdd(p.total64, (unsigned long long)(off + tail));
}
// (4) cells at their final position
__global__ void __launch_bounds__(256) ser_cell_write(SerParams p) {
  const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= (int64_t)p.ncols * p.nrows) return;
  const int32_t ci = (int32_t)(it / p.nrows); const int64_t r = it - (int64_t)ci * p.nrows;
  const SCol &sc = p.cols[ci];
  if (p.format == TFGPU_FMT_CH_JSON_EACH_ROW && !is_valid(sc.c, r)) return;
  const uint64_t base = (uint64_t)p.row_len[r] + p.cell[it];
  if (p.format == TFGPU_FMT_CH_JSON_EACH_ROW) {
    // a null-marshalling value must leave no bytes: probe first (rare: only `any` columns can do it)
    if (sc.c.repr == TFGPU_R_JSON || sc.c.repr == TFGPU_R_JSONNUM) { CountSink probe; if (!emit_cell(probe, p, sc, r)) return; }
    WriteSink w{p.out + base};
    emit_cell(w, p, sc, r);
    w.put((p.last_present[r] == ci) ? '}' : ',');
    w.flush();
    return;
  }
  WriteSink w{p.out + base};
  emit_cell(w, p, sc, r);
  w.flush();
}
// row frame: '{' and the tail
__global__ void __launch_bounds__(256) ser_row_frame(SerParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  uint8_t *row = p.out + p.row_len[r];
  const uint32_t len = p.row_len[r + 1] - p.row_len[r];
  if (p.format == TFGPU_FMT_CSV) { row[len - 1] = '\n'; return; }
  row[0] = '{';
  if (p.format == TFGPU_FMT_CH_JSON_EACH_ROW) { if (p.last_present[r] < 0) row[len - 2] = '}'; row[len - 1] = '\n'; return; }
  const bool nl = p.closing_newline || r + 1 < p.nrows;
  row[len - 1 - (nl ? 1 : 0)] = '}';
  if (nl) row[len - 1] = '\n';
}

static inline unsigned blocks(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + 255) / 256); }

// host-side escaping of a column name the way encoding/json writes a map key (escapeHTML=false)
static void json_key(std::string &out, const std::string &k) {
  static const char *hex = "0123456789abcdef";
  out += '"';
  for (size_t i = 0; i < k.size();) {
    unsigned char c = (unsigned char)k[i];
    if (c < 0x80) {
      if (c >= 0x20 && c != '"' && c != '\\') out += (char)c;
      else switch (c) {
        case '"': out += "\\\""; break; case '\\': out += "\\\\"; break; case '\b': out += "\\b"; break; case '\f': out += "\\f"; break;
        case '\n': out += "\\n"; break; case '\r': out += "\\r"; break; case '\t': out += "\\t"; break;
        default: out += "\\u00"; out += hex[c >> 4]; out += hex[c & 15];
      }
      i++; continue;
    }
    size_t need = 0; unsigned cp = 0, lo = 0x80, hi = 0xBF;
    if (c >= 0xC2 && c <= 0xDF) { need = 1; cp = c & 0x1F; }
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; cp = c & 0x0F; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; cp = c & 0x07; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
    bool ok = need > 0 && i + need < k.size();
    if (ok) for (size_t q = 1; q <= need; q++) {
      unsigned d = (unsigned char)k[i + q], l = q == 1 ? lo : 0x80u, h = q == 1 ? hi : 0xBFu;
      if (d < l || d > h) { ok = false; break; }
      cp = (cp << 6) | (d & 0x3F);
    }
    if (!ok) { out += "\\ufffd"; i++; continue; }
    if (cp == 0x2028 || cp == 0x2029) { out += cp == 0x2028 ? "\\u2028" : "\\u2029"; i += need + 1; continue; }
    out.append(k, i, need + 1); i += need + 1;
  }
  out += '"';
}

// Which (DataType, Go type) pairs each format renders on device; the rest is the host's (stock Go) business.
static void require_supported(int format, const DColumn &c, int any_as_string) {
  auto bad = [&](const char *why) {
    throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_serialize: column " + c.name + ": " + why);
  };
  if (c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) {
    // strconv.FormatFloat(f, 'f', -1, bits): marshalNumericValue (marshal.go:257-301), castx.ToStringE (caste.go:57-106);
    // json.Marshal(float) = encoding/json's floatEncoder elsewhere (NaN / Inf there fail the whole call: checked on device)
    if (format == TFGPU_FMT_CSV && (c.dtype == TFGPU_T_ANY || c.dtype == TFGPU_T_BYTES)) bad("Go float under `any` / \"string\" in the CSV serializer");
  }
  if (format == TFGPU_FMT_CH_JSON_EACH_ROW) {
    const bool temporal = c.dtype == TFGPU_T_DATE || c.dtype == TFGPU_T_DATETIME || c.dtype == TFGPU_T_TIMESTAMP;
    if (c.repr == TFGPU_R_TIME && !temporal) bad("time.Time under a non-temporal DataType (generic JSON fallback)");
  } else if (format == TFGPU_FMT_CSV) {
    if (c.dtype == TFGPU_T_BYTES && c.repr != TFGPU_R_BYTES) bad("\"string\" (bytes) column not holding []byte");
    if (c.dtype == TFGPU_T_ANY && (c.repr == TFGPU_R_TIME || c.repr == TFGPU_R_DURATION || c.repr == TFGPU_R_BYTES)) bad("`any` column holding a non-JSON-native Go value");
    if (c.dtype != TFGPU_T_ANY && c.repr == TFGPU_R_JSON) bad("map/slice value under a scalar DataType");
  }
  (void)any_as_string;
}

// json.Marshal fails the whole call on NaN / ±Inf (UnsupportedValueError): find them before any text is produced
__global__ void float_nonfinite_kernel(const void *values, int64_t n, int is32, const uint8_t *validity, uint32_t *flag) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || (validity && !((validity[r >> 3] >> (r & 7)) & 1))) return;
  const double v = is32 ? (double)((const float *)values)[r] : ((const double *)values)[r];
  if (v != v || v == INFINITY || v == -INFINITY) *flag = 1;
}

}  // namespace tf

using namespace tf;

extern "C" int tfgpu_serialize_ex(int format, const tfgpu_dbatch *b, const tfgpu_serialize_options *opts, tfgpu_dbuf **out) {
  try {
    if (!b || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_serialize: null argument");
    if (format != TFGPU_FMT_CH_JSON_EACH_ROW && format != TFGPU_FMT_JSON && format != TFGPU_FMT_CSV)
      return tf::fail(TFGPU_ERR_INVALID, "tfgpu_serialize: unknown format");
    Context &cx = ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    hipStream_t st = cx.stream;
    const int64_t n = b->nrows;
    const int ncols = (int)b->cols.size();
    const int any_as_string = opts ? opts->any_as_string : 0;
    materialize(*b);
    for (auto &c : b->cols) require_supported(format, c, any_as_string);
    {
      Buf flag;
      for (auto &c : b->cols) {
        if (c.repr != TFGPU_R_FLOAT32 && c.repr != TFGPU_R_FLOAT64) continue;
        const bool numeric = (c.dtype >= TFGPU_T_INT8 && c.dtype <= TFGPU_T_FLOAT64) || c.dtype == TFGPU_T_INTERVAL;
        if (format == TFGPU_FMT_CSV || (format == TFGPU_FMT_CH_JSON_EACH_ROW && numeric)) continue;  // FormatFloat prints NaN / +Inf
        if (!flag) flag = dalloc_zero(4);
        if (n) float_nonfinite_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(c.values->p, n, c.repr == TFGPU_R_FLOAT32, ptr<uint8_t>(c.validity), ptr<uint32_t>(flag));
      }
      if (flag) {
        const uint32_t *h = d2h_u32(flag->p);
        tf::sync();
        if (*h) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_serialize: json: unsupported value: NaN or Inf (json.Marshal fails the batch in the reference)");
      }
    }
    if (opts && opts->ncols && opts->ncols != ncols) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_serialize: options describe another column count");

    // column order: CH and CSV keep ColumnNames order; encoding/json sorts map keys
    std::vector<int> order((size_t)ncols);
    for (int i = 0; i < ncols; i++) order[(size_t)i] = i;
    if (format == TFGPU_FMT_JSON) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return b->cols[(size_t)x].name < b->cols[(size_t)y].name; });
    if (format == TFGPU_FMT_JSON)  // a map holds each key once: the last value of a duplicated name wins
      for (int i = 0; i + 1 < ncols; i++) if (b->cols[(size_t)order[(size_t)i]].name == b->cols[(size_t)order[(size_t)i + 1]].name)
        return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_serialize: duplicate column names");

    std::string blob;
    std::vector<SCol> sc((size_t)ncols);
    for (int j = 0; j < ncols; j++) {
      const DColumn &c = b->cols[(size_t)order[(size_t)j]];
      SCol &s = sc[(size_t)j];
      s.c = dcol_of(c);
      s.pre_off = (uint32_t)blob.size();
      if (format == TFGPU_FMT_CH_JSON_EACH_ROW) { blob += '"'; blob += c.name; blob += "\":"; }  // writeColName: raw name
      else if (format == TFGPU_FMT_JSON) { if (j) blob += ','; json_key(blob, c.name); blob += ':'; }
      else if (j) blob += ',';
      s.pre_len = (uint32_t)blob.size() - s.pre_off;
      s.ch_flags = 0; s.prec = 0;
      if (format == TFGPU_FMT_CH_JSON_EACH_ROW) {
        if (opts && opts->ncols && opts->ch_flags) { s.ch_flags = opts->ch_flags[order[(size_t)j]]; s.prec = opts->ch_precision ? opts->ch_precision[order[(size_t)j]] : 0; }
        else switch (c.dtype) {  // pkg/providers/clickhouse/typesystem.md
          case TFGPU_T_BYTES: case TFGPU_T_UTF8: case TFGPU_T_ANY: s.ch_flags = TFGPU_CH_STRING; break;
          case TFGPU_T_DATE: s.ch_flags = TFGPU_CH_DATE; break;
          case TFGPU_T_TIMESTAMP: s.ch_flags = TFGPU_CH_DATETIME64; s.prec = 9; break;
          default: break;
        }
      }
    }
    auto res = std::make_unique<tfgpu_dbuf>();
    if (n == 0 || (ncols == 0 && format == TFGPU_FMT_CSV && n == 0)) { res->mem = dalloc(64); res->size = 0; *out = res.release(); return TFGPU_OK; }

    Buf bsc = upload_small(sc.data(), sc.size() * sizeof(SCol)), bblob = upload_small(blob.data(), blob.size());
    Buf cell = dalloc((size_t)std::max(ncols, 1) * (size_t)n * 4), row_len = dalloc((size_t)(n + 1) * 4 + 16), last = dalloc((size_t)n * 4 + 16);
    SerParams p{};
    p.cols = ptr<SCol>(bsc); p.ncols = ncols; p.nrows = n; p.format = format; p.any_as_string = any_as_string;
    p.closing_newline = opts ? opts->add_closing_newline : 0;
    p.blob = ptr<uint8_t>(bblob); p.cell = ptr<uint32_t>(cell); p.row_len = ptr<uint32_t>(row_len); p.last_present = ptr<int32_t>(last);
    Buf tot64 = dalloc_zero(8);
    p.total64 = reinterpret_cast<unsigned long long *>(tot64->p);
    const int64_t ncell = (int64_t)ncols * n;
    { KernelTimer t("ser_cell_len"); if (ncell) ser_cell_len<<<blocks(ncell), 256, 0, st>>>(p); }
    { KernelTimer t("ser_row_layout"); ser_row_layout<<<blocks(n), 256, 0, st>>>(p); }
    exclusive_scan_u32(p.row_len, p.row_len, n, true);
    const uint32_t *htot = d2h_u32(p.row_len + n);
    const uint32_t *h64 = d2h_u32(tot64->p, 2);
    tf::sync();
    if (h64[1] != 0) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_serialize: output exceeds 4 GiB; split the batch by rows (the reference serialises 25 000-row chunks, batch.go:18)");
    // a uint32 offset space: batches whose text exceeds 4 GiB must be split by rows (batch.go chunks at 25 000 rows)
    res->size = *htot;
    res->mem = dalloc(res->size + 64);
    p.out = ptr<uint8_t>(res->mem);
    { KernelTimer t("ser_cell_write"); if (ncell) ser_cell_write<<<blocks(ncell), 256, 0, st>>>(p); }
    { KernelTimer t("ser_row_frame"); ser_row_frame<<<blocks(n), 256, 0, st>>>(p); }
    *out = res.release();
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}

extern "C" int tfgpu_serialize(int format, const tfgpu_dbatch *b, tfgpu_dbuf **out) { return tfgpu_serialize_ex(format, b, nullptr, out); }
