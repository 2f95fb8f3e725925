// tf_devparse.hpp — device-side byte readers and the strconv restatements (ParseInt / ParseUint /
// ParseBool, underscoreOK) shared by the CSV and JSON ingest kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tf {

// ---------------------------------------------------------------------------
// byte readers
// ---------------------------------------------------------------------------
// Sequential reader over HBM with an 8-byte register window.
struct MemBytes {
  const uint8_t *base;  // 8-byte aligned buffer start
  uint64_t win; uint64_t widx;
  __device__ __forceinline__ explicit MemBytes(const uint8_t *b) : base(b), win(0), widx(~0ull) {}
  __device__ __forceinline__ uint32_t at(uint64_t pos) {
    uint64_t i = pos >> 3;
    if (i != widx) { win = reinterpret_cast<const uint64_t *>(base)[i]; widx = i; }
    return (uint32_t)(win >> ((pos & 7) * 8)) & 0xFFu;
  }
  // the 8 bytes at [pos, pos + 8), little-endian, for any alignment (buffers are padded past their payload)
  uint64_t win2 = 0; uint64_t w2idx = ~0ull;
  __device__ __forceinline__ uint64_t word(uint64_t pos) {
    const uint64_t i = pos >> 3;
    const uint32_t sh = (uint32_t)(pos & 7) * 8;
    if (i != widx) {
      if (i == w2idx) win = win2; else win = reinterpret_cast<const uint64_t *>(base)[i];
      widx = i;
    }
    if (!sh) return win;
    if (w2idx != i + 1) { win2 = reinterpret_cast<const uint64_t *>(base)[i + 1]; w2idx = i + 1; }
    return (win >> sh) | (win2 << (64 - sh));
  }
  // the n (<= 8) bytes at [pos, pos + n), zero-extended: the following word is loaded only when the value reaches into it, so no
  // load starts beyond the aligned 8-byte word that holds the value's last byte (a caller's device buffer carries no padding)
  __device__ __forceinline__ uint64_t bytes(uint64_t pos, int n) {
    const uint32_t sh = (uint32_t)(pos & 7) * 8;
    if (sh + (uint32_t)n * 8 > 64) { const uint64_t w = word(pos); return n == 8 ? w : (w & ((1ull << (n * 8)) - 1)); }
    (void)at(pos);
    const uint64_t w = win >> sh;
    return n == 8 ? w : (w & ((1ull << (n * 8)) - 1));
  }
};
// A field view: absolute [start, start+n) through a MemBytes reader (HBM, slow path).
struct Field {
  MemBytes *m; uint64_t start; uint32_t n;
  __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return m->at(start + i); }
};
// The same over a tile staged in LDS (fast path): one aligned ds_read_b64 per 8 bytes walked.
struct LdsBytes {
  const uint8_t *base;  // 8-byte aligned LDS tile
  uint64_t win; uint32_t widx;
  __device__ __forceinline__ explicit LdsBytes(const uint8_t *b) : base(b), win(0), widx(~0u) {}
  __device__ __forceinline__ uint32_t at(uint32_t pos) {
    uint32_t i = pos >> 3;
    if (i != widx) { win = reinterpret_cast<const uint64_t *>(base)[i]; widx = i; }
    return (uint32_t)(win >> ((pos & 7) * 8)) & 0xFFu;
  }
};
struct LField {
  LdsBytes *m; uint32_t start; uint32_t n;
  __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return m->at(start + i); }
};

__device__ __forceinline__ uint32_t lower_(uint32_t c) { return c | 0x20u; }
__device__ __forceinline__ bool dg(uint32_t c) { return c >= '0' && c <= '9'; }

// strconv/atoi.go underscoreOK over s = f[a..b)
template <class F> __device__ bool underscore_ok(const F &f, uint32_t a, uint32_t b) {
  uint32_t saw = '^';
  uint32_t i = a;
  if (b - i >= 1 && (f[i] == '-' || f[i] == '+')) i++;
  bool hex = false;
  if (b - i >= 2 && f[i] == '0' && (lower_(f[i + 1]) == 'b' || lower_(f[i + 1]) == 'o' || lower_(f[i + 1]) == 'x')) { hex = lower_(f[i + 1]) == 'x'; i += 2; saw = '0'; }
  for (; i < b; i++) {
    uint32_t c = f[i];
    if (dg(c) || (hex && lower_(c) >= 'a' && lower_(c) <= 'f')) { saw = '0'; continue; }
    if (c == '_') { if (saw != '0') return false; saw = '_'; continue; }
    if (saw == '_') return false;
    saw = '!';
  }
  return saw != '_';
}

// strconv.ParseUint(s, base 0 or 10, 64): 0 ok, 1 syntax, 2 range
template <class F> __device__ int parse_uint64(const F &f, uint32_t a, uint32_t b, bool base0, uint64_t *out) {
  *out = 0;
  if (a >= b) return 1;
  uint32_t s0 = a;
  uint32_t base = 10;
  if (base0 && f[a] == '0') {
    uint32_t n = b - a;
    uint32_t c1 = n >= 3 ? lower_(f[a + 1]) : 0;
    if (c1 == 'b') { base = 2; a += 2; } else if (c1 == 'o') { base = 8; a += 2; } else if (c1 == 'x') { base = 16; a += 2; } else { base = 8; a += 1; }
  }
  const uint64_t cutoff = 0xFFFFFFFFFFFFFFFFull / base + 1;
  bool underscores = false;
  uint64_t v = 0;
  for (uint32_t i = a; i < b; i++) {
    uint32_t c = f[i], d;
    if (c == '_' && base0) { underscores = true; continue; }
    else if (dg(c)) d = c - '0';
    else if (lower_(c) >= 'a' && lower_(c) <= 'z') d = lower_(c) - 'a' + 10;
    else return 1;
    if (d >= base) return 1;
    if (v >= cutoff) { *out = ~0ull; return 2; }
    v *= base;
    uint64_t v1 = v + d;
    if (v1 < v) { *out = ~0ull; return 2; }
    v = v1;
  }
  if (underscores && !underscore_ok(f, s0, b)) return 1;
  *out = v;
  return 0;
}
// strconv.ParseInt(s, base, 64)
template <class F> __device__ int parse_int64(const F &f, uint32_t a, uint32_t b, bool base0, int64_t *out) {
  *out = 0;
  if (a >= b) return 1;
  bool neg = false;
  if (f[a] == '+') a++; else if (f[a] == '-') { neg = true; a++; }
  uint64_t un;
  int rc = parse_uint64(f, a, b, base0, &un);
  if (rc == 1) return 1;
  const uint64_t cutoff = 1ull << 63;
  if (!neg && un >= cutoff) { *out = (int64_t)(cutoff - 1); return 2; }
  if (neg && un > cutoff) { *out = (int64_t)cutoff; return 2; }
  if (rc == 2) return 2;
  *out = neg ? (int64_t)(0 - un) : (int64_t)un;
  return 0;
}

// strconv.ParseBool
template <class F> __device__ __forceinline__ int parse_bool(const F &f, uint32_t a, uint32_t b, int *out) {
  uint32_t n = b - a;
  if (n == 1) { uint32_t c = f[a]; if (c == '1' || c == 't' || c == 'T') { *out = 1; return 0; } if (c == '0' || c == 'f' || c == 'F') { *out = 0; return 0; } return 1; }
  if (n == 4) {
    uint32_t c0 = f[a], c1 = f[a + 1], c2 = f[a + 2], c3 = f[a + 3];
    if ((c0 == 'T' && c1 == 'R' && c2 == 'U' && c3 == 'E') || ((c0 == 't' || c0 == 'T') && c1 == 'r' && c2 == 'u' && c3 == 'e')) { *out = 1; return 0; }
    return 1;
  }
  if (n == 5) {
    uint32_t c0 = f[a], c1 = f[a + 1], c2 = f[a + 2], c3 = f[a + 3], c4 = f[a + 4];
    if ((c0 == 'F' && c1 == 'A' && c2 == 'L' && c3 == 'S' && c4 == 'E') || ((c0 == 'f' || c0 == 'F') && c1 == 'a' && c2 == 'l' && c3 == 's' && c4 == 'e')) { *out = 0; return 0; }
    return 1;
  }
  return 1;
}

// Decimal text → integer without a per-digit loop.  `t` holds up to 8 digit VALUES (0..9), one per byte, most
// significant digit in byte 0, zero-padded at the front: pairs → fours → eight, all in full-rate 24-bit multiplies.
__device__ __forceinline__ uint32_t four_digits(uint32_t x) {
  const uint32_t pairs = __umul24(x & 0x00FF00FFu, 10u) + ((x >> 8) & 0x00FF00FFu);  // b0*10+b1 | (b2*10+b3) << 16
  return __umul24(pairs & 0xFFFFu, 100u) + (pairs >> 16);
}
__device__ __forceinline__ uint32_t eight_digits(uint64_t t) { return __umul24(four_digits((uint32_t)t), 10000u) + four_digits((uint32_t)(t >> 32)); }
// every byte of d (= text ^ '0' per byte) is a digit value
__device__ __forceinline__ bool all_digits(uint64_t d) { return (((d + 0x7676767676767676ull) | d) & 0x8080808080808080ull) == 0; }
// n characters at bytes [0, n) of the 24-byte little-endian window (b0,b1,b2); the first i0 (0 or 1) are a sign,
// the other nd = n - i0 (1..19) must be decimal digits.  false if a non-digit shows up.
__device__ __forceinline__ bool digits_u64(uint64_t b0, uint64_t b1, uint64_t b2, uint32_t i0, uint32_t n, uint64_t *out) {
  const uint64_t K = 0x3030303030303030ull;
  const bool sane = n > i0 && n - i0 <= 19;
  const uint32_t nd = sane ? n - i0 : 1u;
  const uint64_t s0 = i0 ? (b0 >> 8) | (b1 << 56) : b0;
  if (!__any(nd > 8)) {  // the common case: the whole wave parses short numbers
    uint64_t d = (s0 ^ K) & (nd >= 8 ? ~0ull : (1ull << (8 * nd)) - 1);
    const bool ok = all_digits(d);
    d <<= 8 * (8 - (nd > 8 ? 8 : nd));
    *out = eight_digits(d);
    return ok && sane;
  }
  const uint64_t s1 = i0 ? (b1 >> 8) | (b2 << 56) : b1, s2 = i0 ? b2 >> 8 : b2;
  // groups of 8 digits from the right; the leftmost group has g1 = nd - 8 * (groups - 1) digits
  const uint32_t ng = nd > 16 ? 3u : nd > 8 ? 2u : 1u;
  const uint32_t g1 = nd - 8 * (ng - 1);  // 1..8
  uint64_t d1 = (s0 ^ K) & (g1 >= 8 ? ~0ull : (1ull << (8 * g1)) - 1);
  bool ok = all_digits(d1);
  d1 <<= 8 * (8 - g1);
  uint64_t v = eight_digits(d1);
  if (ng >= 2) {
    const uint32_t sh = 8 * g1;  // 8..64
    const uint64_t d2 = (sh == 64 ? s1 : (s0 >> sh) | (s1 << (64 - sh))) ^ K;
    ok = ok && all_digits(d2);
    v = v * 100000000ull + eight_digits(d2);
    if (ng == 3) {
      const uint64_t d3 = (sh == 64 ? s2 : (s1 >> sh) | (s2 << (64 - sh))) ^ K;
      ok = ok && all_digits(d3);
      v = v * 100000000ull + eight_digits(d3);
    }
  }
  *out = v;
  return ok && sane;
}

// strconv.eiselLemire64 (Go strconv/eisel_lemire.go): man * 10^exp10 → the correctly rounded float64 bits,
// or false when the 128-bit approximation cannot decide (half-way cases, subnormals, overflow): the caller
// then needs the arbitrary-precision path.  `tab` = {lo, hi} of floor(10^e * 2^k), e = -348..347
// (tf_pow10_128.inc).  man != 0.
__device__ __forceinline__ bool eisel_lemire64(uint64_t man, int exp10, const uint64_t *tab, uint64_t *bits_out) {
  if (exp10 < -348 || exp10 > 347) return false;
  const int clz = __clzll((long long)man);
  man <<= clz;
  uint64_t ret_exp2 = (uint64_t)((217706 * exp10 >> 16) + 64 + 1023) - (uint64_t)clz;
  const uint64_t plo = tab[2 * (exp10 + 348)], phi = tab[2 * (exp10 + 348) + 1];
  uint64_t x_hi = __umul64hi(man, phi), x_lo = man * phi;
  if ((x_hi & 0x1FF) == 0x1FF && x_lo + man < man) {  // wider approximation
    const uint64_t y_hi = __umul64hi(man, plo), y_lo = man * plo;
    uint64_t m_hi = x_hi;
    const uint64_t m_lo = x_lo + y_hi;
    if (m_lo < x_lo) m_hi++;
    if ((m_hi & 0x1FF) == 0x1FF && m_lo + 1 == 0 && y_lo + man < man) return false;
    x_hi = m_hi; x_lo = m_lo;
  }
  const uint64_t msb = x_hi >> 63;
  uint64_t ret_man = x_hi >> (msb + 9);
  ret_exp2 -= 1 ^ msb;
  if (x_lo == 0 && (x_hi & 0x1FF) == 0 && (ret_man & 3) == 1) return false;  // half-way ambiguity
  ret_man += ret_man & 1;
  ret_man >>= 1;
  if (ret_man >> 53) { ret_man >>= 1; ret_exp2 += 1; }
  if (ret_exp2 - 1 >= 0x7FF - 1) return false;  // subnormal or Inf/NaN space
  *bits_out = ret_exp2 << 52 | (ret_man & 0x000FFFFFFFFFFFFFull);
  return true;
}

// strconv.ParseFloat(s, 64) as atof64 decides it (strconv/atof.go): special(), readFloat, the exact
// float path (atof64exact), then Eisel-Lemire — also on mantissa+1 when digits were truncated.  What is left
// for Go's arbitrary-precision fallback (half-way cases, subnormals) is not decided here.
// 0 ok, 1 syntax error, 2 range error (*out = ±Inf), 3 not decided (also hex floats and '_').
template <class F> __device__ int parse_float_go(const F &f, uint32_t a, const uint32_t b, const double *p10, const uint64_t *p128, double *out) {
  uint32_t i = a;
  *out = 0;
  if (i >= b) return 1;
  bool neg = false, sign = false;
  if (f[i] == '+') { i++; sign = true; } else if (f[i] == '-') { neg = true; sign = true; i++; }
  {  // special(): [+-]?inf(inity)? | nan, case-insensitive
    const uint32_t n = b - i;
    auto eq = [&](const char *w, uint32_t wl) { if (n != wl) return false; for (uint32_t k = 0; k < wl; k++) if (lower_(f[i + k]) != (uint32_t)w[k]) return false; return true; };
    if (eq("inf", 3) || eq("infinity", 8)) { *out = neg ? -INFINITY : INFINITY; return 0; }
    if (!sign && eq("nan", 3)) { *out = NAN; return 0; }
  }
  if (b - i >= 2 && f[i] == '0' && lower_(f[i + 1]) == 'x') return 3;
  uint64_t mant = 0;
  int nd = 0, ndm = 0, dp = 0;
  bool sawdot = false, sawdigits = false, trunc = false;
  for (; i < b; i++) {
    const uint32_t c = f[i];
    if (c == '_') return 3;
    if (c == '.') { if (sawdot) break; sawdot = true; dp = nd; continue; }
    if (dg(c)) {
      sawdigits = true;
      if (c == '0' && nd == 0) { dp--; continue; }
      nd++;
      if (ndm < 19) { mant = mant * 10 + (c - '0'); ndm++; } else if (c != '0') trunc = true;
      continue;
    }
    break;
  }
  if (!sawdigits) return 1;
  if (!sawdot) dp = nd;
  if (i < b && lower_(f[i]) == 'e') {
    i++;
    if (i >= b) return 1;
    int esign = 1;
    if (f[i] == '+') i++; else if (f[i] == '-') { i++; esign = -1; }
    if (i >= b || !dg(f[i])) return 1;
    int e = 0;
    for (; i < b && (dg(f[i]) || f[i] == '_'); i++) {
      if (f[i] == '_') return 3;
      if (e < 10000) e = e * 10 + (int)(f[i] - '0');
    }
    dp += e * esign;
  }
  if (i != b) return 1;
  if (mant == 0) { *out = neg ? -0.0 : 0.0; return 0; }
  // decimal.floatBits: beyond these the slow path answers without looking at the digits
  if (dp > 310) { *out = neg ? -INFINITY : INFINITY; return 2; }
  if (dp < -330) { *out = neg ? -0.0 : 0.0; return 0; }
  int exp10 = dp - ndm;
  // 12300000000000000000e-3 and 123e13 are the same number: without trailing zeros the exact path decides
  // integers Eisel-Lemire cannot (its 10^-k rows are rounded down), with the same correctly rounded result
  if (!trunc && (mant >> 52)) while (mant % 10 == 0) { mant /= 10; exp10++; }
  if (!trunc && !(mant >> 52)) {  // atof64exact
    double v = (double)mant;
    bool exact = true;
    if (exp10 > 0 && exp10 <= 15 + 22) {
      int e = exp10;
      if (e > 22) { v *= p10[323 + e - 22]; e = 22; }
      if (v > 1e15) exact = false; else v *= p10[323 + e];
    } else if (exp10 < 0 && exp10 >= -22) v /= p10[323 - exp10];
    else if (exp10 != 0) exact = false;
    if (exact) { *out = neg ? -v : v; return 0; }
  }
  uint64_t bits;
  if (!eisel_lemire64(mant, exp10, p128, &bits)) return 3;
  if (trunc) {  // the truncated digits may still matter: confirm with the upper bound
    uint64_t up;
    if (!eisel_lemire64(mant + 1, exp10, p128, &up) || up != bits) return 3;
  }
  if (neg) bits |= 0x8000000000000000ull;
  *out = __longlong_as_double((long long)bits);
  return 0;
}

// strconv.eiselLemire32: the same with float32's widths (23 mantissa bits, bias 127, 38 discarded bits)
__device__ __forceinline__ bool eisel_lemire32(uint64_t man, int exp10, const uint64_t *tab, uint32_t *bits_out) {
  if (exp10 < -348 || exp10 > 347) return false;
  const int clz = __clzll((long long)man);
  man <<= clz;
  uint64_t ret_exp2 = (uint64_t)((217706 * exp10 >> 16) + 64 + 127) - (uint64_t)clz;
  const uint64_t plo = tab[2 * (exp10 + 348)], phi = tab[2 * (exp10 + 348) + 1];
  uint64_t x_hi = __umul64hi(man, phi), x_lo = man * phi;
  constexpr uint64_t M = 0x3FFFFFFFFFull;
  if ((x_hi & M) == M && x_lo + man < man) {  // wider approximation
    const uint64_t y_hi = __umul64hi(man, plo), y_lo = man * plo;
    uint64_t m_hi = x_hi;
    const uint64_t m_lo = x_lo + y_hi;
    if (m_lo < x_lo) m_hi++;
    if ((m_hi & M) == M && m_lo + 1 == 0 && y_lo + man < man) return false;
    x_hi = m_hi; x_lo = m_lo;
  }
  const uint64_t msb = x_hi >> 63;
  uint64_t ret_man = x_hi >> (msb + 38);
  ret_exp2 -= 1 ^ msb;
  if (x_lo == 0 && (x_hi & M) == 0 && (ret_man & 3) == 1) return false;  // half-way ambiguity
  ret_man += ret_man & 1;
  ret_man >>= 1;
  if (ret_man >> 24) { ret_man >>= 1; ret_exp2 += 1; }
  if (ret_exp2 - 1 >= 0xFF - 1) return false;  // subnormal or Inf/NaN space
  *bits_out = (uint32_t)(ret_exp2 << 23 | (ret_man & 0x007FFFFFull));
  return true;
}

// strconv.ParseFloat(s, 32) as atof32 decides it: special(), readFloat, atof32exact (float32 arithmetic, mantissa < 2^24,
// powers up to 10^10), then eiselLemire32 — on mantissa + 1 as well when digits were truncated.  Same return codes as
// parse_float_go; what Go leaves to its decimal slow path (half-way cases, subnormals, the overflow boundary) is 3.
template <class F> __device__ int parse_float32_go(const F &f, uint32_t a, const uint32_t b, const uint64_t *p128, float *out) {
  uint32_t i = a;
  *out = 0;
  if (i >= b) return 1;
  bool neg = false, sign = false;
  if (f[i] == '+') { i++; sign = true; } else if (f[i] == '-') { neg = true; sign = true; i++; }
  {
    const uint32_t n = b - i;
    auto eq = [&](const char *w, uint32_t wl) { if (n != wl) return false; for (uint32_t k = 0; k < wl; k++) if (lower_(f[i + k]) != (uint32_t)w[k]) return false; return true; };
    if (eq("inf", 3) || eq("infinity", 8)) { *out = neg ? -INFINITY : INFINITY; return 0; }
    if (!sign && eq("nan", 3)) { *out = NAN; return 0; }
  }
  if (b - i >= 2 && f[i] == '0' && lower_(f[i + 1]) == 'x') return 3;
  uint64_t mant = 0;
  int nd = 0, ndm = 0, dp = 0;
  bool sawdot = false, sawdigits = false, trunc = false;
  for (; i < b; i++) {
    const uint32_t c = f[i];
    if (c == '_') return 3;
    if (c == '.') { if (sawdot) break; sawdot = true; dp = nd; continue; }
    if (dg(c)) {
      sawdigits = true;
      if (c == '0' && nd == 0) { dp--; continue; }
      nd++;
      if (ndm < 19) { mant = mant * 10 + (c - '0'); ndm++; } else if (c != '0') trunc = true;
      continue;
    }
    break;
  }
  if (!sawdigits) return 1;
  if (!sawdot) dp = nd;
  if (i < b && lower_(f[i]) == 'e') {
    i++;
    if (i >= b) return 1;
    int esign = 1;
    if (f[i] == '+') i++; else if (f[i] == '-') { i++; esign = -1; }
    if (i >= b || !dg(f[i])) return 1;
    int e = 0;
    for (; i < b && (dg(f[i]) || f[i] == '_'); i++) {
      if (f[i] == '_') return 3;
      if (e < 10000) e = e * 10 + (int)(f[i] - '0');
    }
    dp += e * esign;
  }
  if (i != b) return 1;
  if (mant == 0) { *out = neg ? -0.0f : 0.0f; return 0; }
  // the value is in [10^(dp-1), 10^dp): above float32's largest (3.4e38) from dp = 40 on — ±Inf and ErrRange out of Go's slow
  // path — and below half its smallest subnormal (7e-46) up to dp = -46, which rounds to ±0 without an error
  if (dp > 39) { *out = neg ? -INFINITY : INFINITY; return 2; }
  if (dp < -45) { *out = neg ? -0.0f : 0.0f; return 0; }
  const int exp10 = dp - ndm;
  if (!trunc && !(mant >> 23)) {  // atof32exact
    const float P[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
    float v = (float)mant;
    bool exact = true;
    if (exp10 > 0 && exp10 <= 7 + 10) {
      int e = exp10;
      if (e > 10) { v = __fmul_rn(v, P[e - 10]); e = 10; }
      if (v > 1e7f) exact = false; else v = __fmul_rn(v, P[e]);
    } else if (exp10 < 0 && exp10 >= -10) v = __fdiv_rn(v, P[-exp10]);
    else if (exp10 != 0) exact = false;
    if (exact) { *out = neg ? -v : v; return 0; }
  }
  uint32_t bits;
  if (!eisel_lemire32(mant, exp10, p128, &bits)) return 3;
  if (trunc) {
    uint32_t up;
    if (!eisel_lemire32(mant + 1, exp10, p128, &up) || up != bits) return 3;
  }
  if (neg) bits |= 0x80000000u;
  *out = __uint_as_float(bits);
  return 0;
}

// time.ParseDuration (time/format.go) behind spf13/cast v1.7.1's ToDurationE(string) (caste.go: a string holding none of
// "nsuµmh" gets "ns" appended): [-+]?([0-9]*(\.[0-9]*)?unit)+ with leadingInt / leadingFraction's overflow rules and the
// fraction added through float64 exactly as Go does.  0 = ok, 1 = error.
template <class F> __device__ int parse_duration_go(const F &f, uint32_t a, uint32_t b, int64_t *out) {
  *out = 0;
  bool has_unit_char = false;
  for (uint32_t i = a; i < b; i++) {
    const uint32_t c = f[i];
    if (c == 'n' || c == 's' || c == 'u' || c == 'm' || c == 'h' || (c == 0xC2 && i + 1 < b && f[i + 1] == 0xB5)) { has_unit_char = true; break; }
  }
  uint32_t i = a;
  bool neg = false;
  if (i < b && (f[i] == '-' || f[i] == '+')) { neg = f[i] == '-'; i++; }
  // (ParseDuration's `s == "0"` shortcut cannot be reached: "0" holds no unit character, so it arrives as "0ns")
  if (i >= b) return 1;  // "" or a bare sign: the text is "ns", which starts with no digit
  constexpr uint64_t TOP = 1ull << 63;
  uint64_t d = 0;
  bool appended_done = false;  // the appended "ns" is the unit of the last (and only unit-less) group
  while (i < b) {
    if (!(f[i] == '.' || dg(f[i]))) return 1;
    uint64_t v = 0, fr = 0; double scale = 1;
    const uint32_t pl = i;
    for (; i < b && dg(f[i]); i++) {  // leadingInt
      if (v > TOP / 10) return 1;
      v = v * 10 + (f[i] - '0');
      if (v > TOP) return 1;
    }
    const bool pre = i != pl;
    bool post = false;
    if (i < b && f[i] == '.') {
      i++;
      const uint32_t pf = i;
      bool over = false;
      for (; i < b && dg(f[i]); i++) {  // leadingFraction
        if (over) continue;
        if (fr > (TOP - 1) / 10) { over = true; continue; }
        const uint64_t y = fr * 10 + (f[i] - '0');
        if (y > TOP) { over = true; continue; }
        fr = y; scale *= 10;
      }
      post = i != pf;
    }
    if (!pre && !post) return 1;
    uint32_t u0 = i;
    for (; i < b; i++) { const uint32_t c = f[i]; if (c == '.' || dg(c)) break; }
    uint64_t unit = 0;
    const uint32_t ul = i - u0;
    if (ul == 0) {
      if (has_unit_char || appended_done || i != b) return 1;  // missing unit
      unit = 1; appended_done = true;                            // the appended "ns"
    } else {
      auto is = [&](const char *w, uint32_t wl) { if (ul != wl) return false; for (uint32_t k = 0; k < wl; k++) if (f[u0 + k] != (uint32_t)(uint8_t)w[k]) return false; return true; };
      // without a unit character in the text "ns" is appended to the LAST unit: "5x" reads the unit "xns"
      if (!has_unit_char) return 1;
      if (is("ns", 2)) unit = 1; else if (is("us", 2) || is("\xC2\xB5s", 3) || is("\xCE\xBCs", 3)) unit = 1000;
      else if (is("ms", 2)) unit = 1000000; else if (is("s", 1)) unit = 1000000000ull; else if (is("m", 1)) unit = 60000000000ull;
      else if (is("h", 1)) unit = 3600000000000ull; else return 1;
    }
    if (v > TOP / unit) return 1;
    v *= unit;
    if (fr > 0) {
      v += (uint64_t)((double)fr * ((double)unit / scale));
      if (v > TOP) return 1;
    }
    d += v;
    if (d > TOP) return 1;
  }
  if (neg) { *out = (int64_t)(0 - d); return 0; }
  if (d > TOP - 1) return 1;
  *out = (int64_t)d;
  return 0;
}

// math.Pow10(n), n = -323..308 (632 doubles) followed by the 128-bit powers of Eisel-Lemire (696 x 2 words): HBM-resident,
// built once per lane (tf_json.hip)
const double *pow10_table();

}  // namespace tf
