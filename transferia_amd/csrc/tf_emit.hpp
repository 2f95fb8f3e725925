// tf_emit.hpp — Go value text on device: byte sinks and the emitters shared by the serializers (tf_serialize.hip) and by
// Collapse's key strings (tf_collapse.hip): encoding/json strings, base64, integers, shortest floats.  Every emitter is a
// template over a sink with put(byte), so the same code counts, writes or hashes.
#pragma once
#include <type_traits>
#include "tf_devcol.hpp"
#include "tf_devfmt.hpp"
#include "tf_devfloat.hpp"

namespace tf {

// ---- sinks -----------------------------------------------------------------------
struct CountSink {
  uint32_t n = 0;
  __device__ __forceinline__ void put(uint32_t) { n++; }
  __device__ __forceinline__ void put_word(uint64_t, uint32_t k) { n += k; }
};
// Bytes gathered eight at a time and stored with ONE (possibly unaligned) 8-byte store: a cell's text lands in the
// middle of its row, so byte stores would cost one memory transaction per character.
struct WriteSink {
  uint8_t *p; uint64_t acc = 0; uint32_t n = 0;
  struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
  __device__ __forceinline__ void put(uint32_t c) {
    acc |= (uint64_t)(c & 0xFFu) << (8 * n);
    if (++n == 8) { reinterpret_cast<U64 *>(p)->v = acc; p += 8; acc = 0; n = 0; }
  }
  // the low k (1..8) bytes of w, the bytes above them zero: one merge instead of k puts
  __device__ __forceinline__ void put_word(uint64_t w, uint32_t k) {
    acc |= w << (8 * n);
    const uint32_t t = n + k;
    if (t >= 8) {
      reinterpret_cast<U64 *>(p)->v = acc; p += 8;
      acc = n ? w >> (8 * (8 - n)) : 0;  // what did not fit (n == 0: everything did)
      n = t - 8;
    } else n = t;
  }
  __device__ __forceinline__ void flush() {
    if (n >= 4) { struct __attribute__((packed, aligned(1))) U32 { uint32_t v; }; reinterpret_cast<U32 *>(p)->v = (uint32_t)acc; p += 4; acc >>= 32; n -= 4; }
    for (; n; n--) { *p++ = (uint8_t)acc; acc >>= 8; }
  }
};
template <class S> struct CsvQuoteSink {  // doubles '"' (encoding/csv quoted field body)
  S &s;
  __device__ __forceinline__ void put(uint32_t c) { if (c == '"') s.put('"'); s.put(c); }
};

// s.put_word(w, k) where the sink has one, k puts otherwise (wrapping sinks that look at every byte)
template <class S> __device__ __forceinline__ auto sink_word(S &s, uint64_t w, uint32_t k, int) -> decltype(s.put_word(w, k), void()) { s.put_word(w, k); }
template <class S> __device__ __forceinline__ void sink_word(S &s, uint64_t w, uint32_t k, long) { for (uint32_t i = 0; i < k; i++) s.put((uint32_t)(w >> (8 * i)) & 0xFFu); }
struct __attribute__((packed, aligned(1))) UnalignedU64 { uint64_t v; };
__device__ __forceinline__ uint64_t load8(const uint8_t *p) { return reinterpret_cast<const UnalignedU64 *>(p)->v; }
// SWAR byte tests over a 64-bit word (the flags may be inexact only ABOVE a true hit, so "no flag" is exact)
__device__ __forceinline__ uint64_t swar_has_zero(uint64_t v) { return (v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull; }
__device__ __forceinline__ uint64_t swar_has(uint64_t w, uint32_t c) { return swar_has_zero(w ^ (0x0101010101010101ull * c)); }
__device__ __forceinline__ uint64_t swar_has_less(uint64_t w, uint32_t c) { return (w - 0x0101010101010101ull * c) & ~w & 0x8080808080808080ull; }

// A lane's walk over one text cell, eight bytes per step.  The loads run TWO steps ahead of the bytes being looked at: the step's
// branches depend on the bytes, so the compiler cannot start the next load before they are taken, and a cell of n bytes would
// otherwise cost n / 8 memory round trips in a row.
struct TextWords {
  const uint8_t *p; uint32_t n, i = 0; uint64_t w0 = 0, w1 = 0;
  __device__ __forceinline__ TextWords(const uint8_t *p_, uint32_t n_) : p(p_), n(n_) {
    if (n >= 8) w0 = load8(p);
    if (n >= 16) w1 = load8(p + 8);
  }
  __device__ __forceinline__ bool more() const { return i + 8 <= n; }
  __device__ __forceinline__ uint64_t next() {  // the eight bytes at i; i moves on
    const uint64_t w = w0;
    w0 = w1;
    w1 = i + 24 <= n ? load8(p + i + 16) : 0;
    i += 8;
    return w;
  }
  // the last n - i (< 8) bytes as one word; `len` receives their count
  __device__ __forceinline__ uint64_t tail(uint32_t &len) {
    len = n - i;
    uint64_t w = 0;
#pragma unroll
    for (uint32_t k = 0; k < 7; k++) if (k < len) w |= (uint64_t)p[i + k] << (8 * k);
    i = n;
    return w;
  }
};
template <class S> __device__ __forceinline__ void put_bytes(S &s, const uint8_t *p, uint32_t n) {
  TextWords t(p, n);
  while (t.more()) sink_word(s, t.next(), 8, 0);
  uint32_t k; const uint64_t w = t.tail(k);
  if (k) sink_word(s, w, k, 0);
}
template <class S> __device__ __forceinline__ void put_lit(S &s, const char *p) { while (*p) s.put((uint8_t)*p++); }
__device__ __forceinline__ uint32_t hexc(uint32_t n) { return n + (n < 10 ? '0' : 'a' - 10); }

// writeQuoted (marshal.go:377-419)
template <class S> __device__ __forceinline__ void ch_quoted_bytes(S &s, uint64_t w, uint32_t k) {  // the low k bytes of w, one by one
  for (uint32_t b = 0; b < k; b++) {
    const uint32_t c = (uint32_t)(w >> (8 * b)) & 0xFFu;
    if (c >= 0x20 && c != '"' && c != '\\') { s.put(c); continue; }
    s.put('\\');
    switch (c) {
      case '"': s.put('"'); break; case '\\': s.put('\\'); break; case '\n': s.put('n'); break; case '\r': s.put('r'); break;
      case '\t': s.put('t'); break; case '\f': s.put('f'); break; case '\b': s.put('b'); break;
      default: s.put('u'); s.put('0'); s.put('0'); s.put(hexc(c >> 4)); s.put(hexc(c & 15));
    }
  }
}
template <class S> __device__ __forceinline__ void emit_ch_quoted(S &s, const uint8_t *p, uint32_t n) {
  s.put('"');
  TextWords t(p, n);
  while (t.more()) {
    const uint64_t w = t.next();
    if (!(swar_has_less(w, 0x20) | swar_has(w, '"') | swar_has(w, '\\'))) sink_word(s, w, 8, 0);  // eight bytes that need no escaping at once
    else ch_quoted_bytes(s, w, 8);
  }
  uint32_t k; const uint64_t w = t.tail(k);
  if (k) { if (!(swar_has_less(w | (~0ull << (8 * k)), 0x20) | swar_has(w, '"') | swar_has(w, '\\'))) sink_word(s, w, k, 0); else ch_quoted_bytes(s, w, k); }
  s.put('"');
}

// encoding/json appendString: ", \, control bytes, invalid UTF-8 → �, U+2028/9; optional HTML escaping
template <class S> __device__ __forceinline__ void emit_json_string(S &s, const uint8_t *p, uint32_t n, bool html) {
  s.put('"');
  uint32_t i = 0;
  while (i < n) {
    if (i + 8 <= n) {  // eight printable ASCII bytes that need no escaping at once
      const uint64_t w = load8(p + i);
      uint64_t bad = (w & 0x8080808080808080ull) | swar_has_less(w, 0x20) | swar_has(w, '"') | swar_has(w, '\\');
      if (html) bad |= swar_has(w, '<') | swar_has(w, '>') | swar_has(w, '&');
      if (!bad) { sink_word(s, w, 8, 0); i += 8; continue; }
    }
    const uint32_t c = p[i];
    if (c < 0x80) {
      if (c >= 0x20 && c != '"' && c != '\\' && !(html && (c == '<' || c == '>' || c == '&'))) { s.put(c); i++; continue; }
      s.put('\\');
      switch (c) {
        case '"': s.put('"'); break; case '\\': s.put('\\'); break; case '\b': s.put('b'); break; case '\f': s.put('f'); break;
        case '\n': s.put('n'); break; case '\r': s.put('r'); break; case '\t': s.put('t'); break;
        default: s.put('u'); s.put('0'); s.put('0'); s.put(hexc(c >> 4)); s.put(hexc(c & 15));
      }
      i++; continue;
    }
    uint32_t need = 0, cp = 0, lo = 0x80, hi = 0xBF;  // utf8.DecodeRune
    if (c >= 0xC2 && c <= 0xDF) { need = 1; cp = c & 0x1F; }
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; cp = c & 0x0F; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; cp = c & 0x07; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
    bool ok = need > 0 && i + need < n;
    if (ok) for (uint32_t k = 1; k <= need; k++) {
      const uint32_t d = p[i + k], l = k == 1 ? lo : 0x80u, h = k == 1 ? hi : 0xBFu;
      if (d < l || d > h) { ok = false; break; }
      cp = (cp << 6) | (d & 0x3F);
    }
    if (!ok) { put_lit(s, "\\ufffd"); i++; continue; }
    if (cp == 0x2028 || cp == 0x2029) { put_lit(s, cp == 0x2028 ? "\\u2028" : "\\u2029"); i += need + 1; continue; }
    for (uint32_t k = 0; k <= need; k++) s.put(p[i + k]);
    i += need + 1;
  }
  s.put('"');
}

template <class S> __device__ __forceinline__ void emit_base64(S &s, const uint8_t *p, uint32_t n) {  // base64.StdEncoding
  const char *T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  uint32_t i = 0;
  for (; i + 3 <= n; i += 3) { uint32_t v = (p[i] << 16) | (p[i + 1] << 8) | p[i + 2]; s.put(T[v >> 18]); s.put(T[(v >> 12) & 63]); s.put(T[(v >> 6) & 63]); s.put(T[v & 63]); }
  if (n - i == 1) { uint32_t v = p[i] << 16; s.put(T[v >> 18]); s.put(T[(v >> 12) & 63]); s.put('='); s.put('='); }
  else if (n - i == 2) { uint32_t v = (p[i] << 16) | (p[i + 1] << 8); s.put(T[v >> 18]); s.put(T[(v >> 12) & 63]); s.put(T[(v >> 6) & 63]); s.put('='); }
}

template <class S> __device__ __forceinline__ void emit_small(S &s, const uint8_t *t, int n) { for (int i = 0; i < n; i++) s.put(t[i]); }
// decimal text of v < 10^8 built in a register (32-bit arithmetic, no scratch array) and handed over as one word
// decimal digits of v < 10^8
__device__ __forceinline__ uint32_t ndigits8(uint32_t v) {
  return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u);
}
template <class S> __device__ __forceinline__ void emit_dec8(S &s, uint32_t v, bool neg) {
  if constexpr (std::is_same_v<S, CountSink>) { s.n += ndigits8(v) + (neg ? 1u : 0u); return; }  // the length pass only counts
  uint64_t w = 0; uint32_t nd = 0;
  do { const uint32_t q = v / 10u; w = (w << 8) | ('0' + (v - q * 10u)); v = q; nd++; } while (v);
  if (neg) s.put('-');
  sink_word(s, w, nd, 0);
}

// exactly `width` (1..8) decimal digits of v, zero padded, as one word
template <class S> __device__ __forceinline__ void emit_dec_pad(S &s, uint32_t v, uint32_t width) {
  if constexpr (std::is_same_v<S, CountSink>) { s.n += width; return; }
  uint64_t w = 0;
  for (uint32_t i = 0; i < width; i++) { const uint32_t q = v / 10u; w = (w << 8) | ('0' + (v - q * 10u)); v = q; }
  sink_word(s, w, width, 0);
}
// strconv.FormatUint / FormatInt(v, 10): eight-digit groups, each built in a register (no byte array: arrays indexed at run
// time live in scratch memory, and a 19-digit hash column then costs ~80 scratch accesses per value)
template <class S> __device__ __forceinline__ void emit_u64(S &s, uint64_t v) {
  if (v < 100000000ull) { emit_dec8(s, (uint32_t)v, false); return; }
  const uint64_t hi = v / 100000000ull;
  const uint32_t lo = (uint32_t)(v - hi * 100000000ull);
  if (hi < 100000000ull) emit_dec8(s, (uint32_t)hi, false);
  else { const uint64_t top = hi / 100000000ull; emit_dec8(s, (uint32_t)top, false); emit_dec_pad(s, (uint32_t)(hi - top * 100000000ull), 8); }
  emit_dec_pad(s, lo, 8);
}
template <class S> __device__ __forceinline__ void emit_i64(S &s, int64_t v) {
  if (v < 0) { s.put('-'); emit_u64(s, (uint64_t)(-(v + 1)) + 1u); } else emit_u64(s, (uint64_t)v);
}
// time.Format pieces (UTC): "2006-01-02", "15:04:05", ".999999999" with trailing zeros trimmed — dev::fmt_date / fmt_clock /
// fmt_frac9_trim without the byte window
template <class S> __device__ __forceinline__ void emit_date(S &s, int64_t sec) {
  int64_t y; int m, d;
  dev::civil_from_days(dev::floordiv(sec, 86400), y, m, d);
  uint64_t u = y < 0 ? (uint64_t)(-y) : (uint64_t)y;
  if (y < 0) s.put('-');
  if (u < 10000) emit_dec_pad(s, (uint32_t)u, 4); else emit_u64(s, u);
  // "-MM-DD" as one word
  const uint32_t mm = (uint32_t)m, dd = (uint32_t)d;
  const uint64_t w = (uint64_t)'-' | (uint64_t)('0' + mm / 10) << 8 | (uint64_t)('0' + mm % 10) << 16 | (uint64_t)'-' << 24 | (uint64_t)('0' + dd / 10) << 32 | (uint64_t)('0' + dd % 10) << 40;
  sink_word(s, w, 6, 0);
}
template <class S> __device__ __forceinline__ void emit_clock(S &s, int64_t sec) {
  const uint32_t sod = (uint32_t)(sec - dev::floordiv(sec, 86400) * 86400);
  const uint32_t h = sod / 3600u, mi = (sod / 60u) % 60u, ss = sod % 60u;
  const uint64_t w = (uint64_t)('0' + h / 10) | (uint64_t)('0' + h % 10) << 8 | (uint64_t)':' << 16 | (uint64_t)('0' + mi / 10) << 24 | (uint64_t)('0' + mi % 10) << 32 |
                     (uint64_t)':' << 40 | (uint64_t)('0' + ss / 10) << 48 | (uint64_t)('0' + ss % 10) << 56;
  sink_word(s, w, 8, 0);
}
template <class S> __device__ __forceinline__ void emit_frac9_trim(S &s, int32_t nsec) {
  if (nsec == 0) return;
  uint32_t v = (uint32_t)nsec, n = 9;
  while (v % 10u == 0) { v /= 10u; n--; }
  s.put('.');
  if (n == 9) { emit_dec_pad(s, v / 100000000u, 1); emit_dec_pad(s, v % 100000000u, 8); } else emit_dec_pad(s, v, n);
}
template <class S> __device__ __forceinline__ void emit_rfc3339nano(S &s, int64_t sec, int32_t nsec) {  // time.RFC3339Nano, UTC
  emit_date(s, sec); s.put('T'); emit_clock(s, sec); emit_frac9_trim(s, nsec); s.put('Z');
}
template <class S> __device__ __forceinline__ void emit_time_string(S &s, int64_t sec, int32_t nsec) {  // Time.String(), UTC
  emit_date(s, sec); s.put(' '); emit_clock(s, sec); emit_frac9_trim(s, nsec);
  const char tail[] = " +0000 UTC";
  for (int i = 0; i < 10; i++) s.put((uint32_t)tail[i]);
}

// The fixed-size part of one cell — value bits (or the text cell's two offsets), nanoseconds, validity — as one load group.
// The serializers' chunk walks issue the groups of a whole chunk before the first cell is formatted: formatting is a chain of
// data-dependent branches the compiler cannot move a load across, and a lane that loads, formats, loads, formats pays a full
// HBM round trip per cell.
struct CellBits {
  uint64_t v;     // fixed-width reprs: the value as stored, zero-extended; text reprs: begin | end << 32
  int32_t ns;     // TIME: nanoseconds
  bool valid;
};
__device__ __forceinline__ CellBits load_cell(const DCol &c, int64_t r) {
  CellBits b; b.v = 0; b.ns = 0;
  b.valid = is_valid(c, r);
  if (c.offsets) b.v = (uint64_t)c.offsets[r] | ((uint64_t)c.offsets[r + 1] << 32);
  else switch (c.repr) {
    case TFGPU_R_INT8: case TFGPU_R_UINT8: case TFGPU_R_BOOL: b.v = ((const uint8_t *)c.values)[r]; break;
    case TFGPU_R_INT16: case TFGPU_R_UINT16: b.v = ((const uint16_t *)c.values)[r]; break;
    case TFGPU_R_INT32: case TFGPU_R_UINT32: case TFGPU_R_FLOAT32: b.v = ((const uint32_t *)c.values)[r]; break;
    case TFGPU_R_INT64: case TFGPU_R_UINT64: case TFGPU_R_FLOAT64: case TFGPU_R_DURATION: b.v = ((const uint64_t *)c.values)[r]; break;
    case TFGPU_R_TIME:
      b.v = ((const uint64_t *)c.values)[r];
      if (c.nanos) b.ns = c.nanos[r];
      break;
    default: break;
  }
  return b;
}
__device__ __forceinline__ const uint8_t *cell_text(const DCol &c, const CellBits &b, uint32_t &n) {
  if (!c.offsets) { n = 0; return nullptr; }
  const uint32_t o = (uint32_t)b.v;
  n = (uint32_t)(b.v >> 32) - o;
  return c.data + o;
}
__device__ __forceinline__ double cell_f64(const DCol &c, const CellBits &b) {
  return c.repr == TFGPU_R_FLOAT32 ? (double)__uint_as_float((uint32_t)b.v) : __longlong_as_double((long long)b.v);
}

// strconv.FormatFloat(f, 'f', -1, bits) of a float column's value
template <class S> __device__ __forceinline__ void emit_float_f(S &s, const DCol &c, const CellBits &b) {
  dev::fmt_float(s, cell_f64(c, b), 'f', c.repr == TFGPU_R_FLOAT32 ? 32 : 64);
}

// json.Marshal(float): encoding/json's floatEncoder
template <class S> __device__ __forceinline__ void emit_float_json(S &s, const DCol &c, const CellBits &b) {
  dev::fmt_json_float(s, cell_f64(c, b), c.repr == TFGPU_R_FLOAT32 ? 32 : 64);
}

// The integer Go types as decimal text; returns false if the column is not an integer repr.
template <class S> __device__ __forceinline__ bool emit_int(S &s, const DCol &c, const CellBits &b) {
  int64_t v;
  switch (c.repr) {
    case TFGPU_R_INT8: v = (int8_t)b.v; break;
    case TFGPU_R_INT16: v = (int16_t)b.v; break;
    case TFGPU_R_INT32: v = (int32_t)b.v; break;
    case TFGPU_R_INT64: v = (int64_t)b.v; break;
    case TFGPU_R_UINT8: case TFGPU_R_UINT16: case TFGPU_R_UINT32: v = (int64_t)b.v; break;  // zero-extended by the load
    case TFGPU_R_UINT64: emit_u64(s, b.v); return true;
    default: return false;
  }
  if (v > -100000000ll && v < 100000000ll) emit_dec8(s, (uint32_t)(v < 0 ? -v : v), v < 0);  // the usual case: fewer than nine digits
  else emit_i64(s, v);
  return true;
}

// json.Marshal of one Go value held in a column cell (the encoding/json serializer's values, and the elements of
// Collapse's key arrays): nil → null, []byte → base64, time.Time → RFC 3339, json.Number / pre-marshalled `any` verbatim.
template <class S> __device__ __forceinline__ void emit_json_cell(S &s, const DCol &c, const CellBits &b, int any_as_string, bool html) {
  if (!b.valid) { put_lit(s, "null"); return; }
  uint32_t vn; const uint8_t *vp = cell_text(c, b, vn);
  switch (c.repr) {
    case TFGPU_R_BOOL: put_lit(s, (uint8_t)b.v ? "true" : "false"); return;
    case TFGPU_R_STRING: emit_json_string(s, vp, vn, html); return;
    case TFGPU_R_BYTES: s.put('"'); emit_base64(s, vp, vn); s.put('"'); return;
    case TFGPU_R_JSONNUM: if (vn) put_bytes(s, vp, vn); else s.put('0'); return;
    case TFGPU_R_TIME: s.put('"'); emit_rfc3339nano(s, (int64_t)b.v, b.ns); s.put('"'); return;
    case TFGPU_R_DURATION: emit_i64(s, (int64_t)b.v); return;
    case TFGPU_R_JSON:
      if (c.dtype == TFGPU_T_ANY && any_as_string) emit_json_string(s, vp, vn, html);
      else put_bytes(s, vp, vn);
      return;
    case TFGPU_R_FLOAT32: case TFGPU_R_FLOAT64: emit_float_json(s, c, b); return;
    default: emit_int(s, c, b);
  }
}
template <class S> __device__ __forceinline__ void emit_json_cell(S &s, const DCol &c, int64_t r, int any_as_string, bool html) {
  emit_json_cell(s, c, load_cell(c, r), any_as_string, html);
}

}  // namespace tf
