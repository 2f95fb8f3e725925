// tf_devfloat.hpp — shortest round-trip decimal text of float64 / float32 on device, and the Go layouts built
// on it: strconv.FormatFloat(f, 'f'|'g'|'e', -1, bits), fmt's %v for floats, encoding/json's float encoder.
//
// The digits are what strconv's shortest mode (ryuFtoaShortest) yields: the shortest decimal that parses back
// to the same float, the closest one to the true value among those.  They are computed with Raffaello
// Giulietti's Schubfach algorithm ("The Schubfach way to render doubles", 2020) — one multiply of the
// significand's interval by a 126-bit approximation of a power of ten (the table tf_pow10_128.inc shares with
// the Eisel-Lemire parser), without the at-least-two-digits rule Java adds on top: one digit is dropped whenever
// a shorter decimal still lies inside the rounding interval.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tf {
namespace dev {

__device__ static const uint64_t POW10_128[696][2] = {
#include "tf_pow10_128.inc"
};

struct Dec { uint64_t f; int32_t e; };  // value = f * 10^e

// round-to-odd of cp * g * 2^-127, g = g1 * 2^63 + g0 (both 63-bit)
__device__ __forceinline__ uint64_t schub_rop(uint64_t g1, uint64_t g0, uint64_t cp) {
  const uint64_t x1 = __umul64hi(g0, cp);
  const uint64_t y0 = g1 * cp, y1 = __umul64hi(g1, cp);
  const uint64_t z = (y0 >> 1) + x1;
  const uint64_t vbp = y1 + (z >> 63);
  return vbp | (((z & 0x7FFFFFFFFFFFFFFFull) + 0x7FFFFFFFFFFFFFFFull) >> 63);
}
// v = c * 2^q (c > 0).  `irregular`: c is the smallest normal significand and q is not the smallest exponent,
// so the gap below v is half the gap above it.
__device__ __forceinline__ Dec schubfach(int q, uint64_t c, bool irregular) {
  const uint64_t out = c & 1;
  const uint64_t cb = c << 2, cbr = cb + 2;
  uint64_t cbl;
  int k;
  if (!irregular) { cbl = cb - 2; k = (int)(((int64_t)q * 661971961083ll) >> 41); }               // floor(log10(2^q))
  else { cbl = cb - 1; k = (int)(((int64_t)q * 661971961083ll - 274743187321ll) >> 41); }        // floor(log10(3/4 * 2^q))
  const int h = q + (int)(((int64_t)(-k) * 913124641741ll) >> 38) + 2;                             // q + floor(log2(10^-k)) + 2
  // g = floor(10^-k * 2^r) + 1 with 2^125 <= g < 2^126: the table row, two bits narrower, plus one
  const uint64_t tlo = POW10_128[348 - k][0], thi = POW10_128[348 - k][1];
  uint64_t glo = (tlo >> 2) | (thi << 62), ghi = thi >> 2;
  glo += 1; ghi += glo == 0 ? 1u : 0u;
  const uint64_t g1 = (ghi << 1) | (glo >> 63), g0 = glo & 0x7FFFFFFFFFFFFFFFull;
  const uint64_t vb = schub_rop(g1, g0, cb << h), vbl = schub_rop(g1, g0, cbl << h), vbr = schub_rop(g1, g0, cbr << h);
  const uint64_t s = vb >> 2;
  if (s >= 10) {  // is a decimal one digit shorter inside the rounding interval?
    const uint64_t sp10 = 10 * (s / 10), tp10 = sp10 + 10;
    const bool upin = vbl + out <= sp10 << 2, wpin = (tp10 << 2) + out <= vbr;
    if (upin != wpin) return Dec{upin ? sp10 : tp10, k};
  }
  const uint64_t t = s + 1;
  const bool uin = vbl + out <= s << 2, win = (t << 2) + out <= vbr;
  if (uin != win) return Dec{uin ? s : t, k};
  const int64_t cmp = (int64_t)(vb - ((s + t) << 1));
  return Dec{(cmp < 0 || (cmp == 0 && (s & 1) == 0)) ? s : t, k};
}
// finite, non-zero |v|
__device__ __forceinline__ Dec shortest_f64(double v) {
  const uint64_t bits = (uint64_t)__double_as_longlong(v);
  const uint64_t t = bits & 0x000FFFFFFFFFFFFFull;
  const int bq = (int)(bits >> 52) & 0x7FF;
  if (bq != 0) {
    const int mq = 1075 - bq;
    const uint64_t c = 0x0010000000000000ull | t;
    if (mq > 0 && mq < 53) { const uint64_t f = c >> mq; if (f << mq == c) return Dec{f, 0}; }  // an integer below 2^53
    return schubfach(-mq, c, t == 0 && bq > 1);
  }
  return schubfach(-1074, t, false);
}
__device__ __forceinline__ Dec shortest_f32(float v) {
  const uint32_t bits = (uint32_t)__float_as_int(v);
  const uint32_t t = bits & 0x007FFFFFu;
  const int bq = (int)(bits >> 23) & 0xFF;
  if (bq != 0) {
    const int mq = 150 - bq;
    const uint64_t c = 0x00800000u | t;
    if (mq > 0 && mq < 24) { const uint64_t f = c >> mq; if (f << mq == c) return Dec{f, 0}; }
    return schubfach(-mq, c, t == 0 && bq > 1);
  }
  return schubfach(-149, t, false);
}

// decimal digits of |f| (finite, non-zero) without trailing zeros into d[]; value = 0.d1d2… * 10^dp
__device__ __forceinline__ int float_digits(double f, int bits, uint8_t d[20], int *dp) {
  Dec x = bits == 32 ? shortest_f32((float)f) : shortest_f64(f);
  while (x.f % 10 == 0) { x.f /= 10; x.e++; }
  uint8_t tmp[20];
  int n = 0;
  do { tmp[n++] = (uint8_t)('0' + x.f % 10); x.f /= 10; } while (x.f);
  for (int i = 0; i < n; i++) d[i] = tmp[n - 1 - i];
  *dp = n + x.e;
  return n;
}

struct CountOut { uint32_t n = 0; __device__ __forceinline__ void put(uint32_t) { n++; } };
struct StoreOut { uint8_t *d; uint32_t n = 0; __device__ __forceinline__ void put(uint32_t c) { d[n++] = (uint8_t)c; } };

template <class S> __device__ __forceinline__ void put_exp(S &o, int exp) {  // e±dd, at least two digits
  o.put('e');
  if (exp < 0) { o.put('-'); exp = -exp; } else o.put('+');
  if (exp < 10) { o.put('0'); o.put('0' + exp); }
  else if (exp < 100) { o.put('0' + exp / 10); o.put('0' + exp % 10); }
  else { o.put('0' + exp / 100); o.put('0' + (exp / 10) % 10); o.put('0' + exp % 10); }
}
// strconv.FormatFloat(f, fmt, -1, bits) for fmt 'f', 'e' and 'g' — 'g' with the shortest-mode threshold
// (exponent < -4 || >= 21 is encoding/json's rule, not this one: here eprec = 6, or the digit count when larger).
template <class S> __device__ __forceinline__ void fmt_float(S &o, double f, char fmt, int bits) {
  if (f != f) { o.put('N'); o.put('a'); o.put('N'); return; }
  const bool neg = __double_as_longlong(f) < 0;
  if (f == INFINITY || f == -INFINITY) { o.put(neg ? '-' : '+'); o.put('I'); o.put('n'); o.put('f'); return; }
  if (neg) { o.put('-'); f = -f; }
  uint8_t d[20];
  int dp = 1, nd = 1;
  if (f == 0) d[0] = '0'; else nd = float_digits(f, bits, d, &dp);
  bool use_e = fmt == 'e';
  int exp = f == 0 ? 0 : dp - 1;
  if (fmt == 'g') {  // %e is used if the exponent is < -4 or >= eprec; shortest: eprec = 6 (ftoa.go %g)
    int eprec = 6;
    if (eprec > nd && nd >= dp) eprec = nd;
    eprec = 6;  // "if shortest { eprec = 6 }"
    use_e = exp < -4 || exp >= eprec;
  }
  if (use_e) {
    o.put(d[0]);
    if (nd > 1) { o.put('.'); for (int i = 1; i < nd; i++) o.put(d[i]); }
    put_exp(o, exp);
    return;
  }
  if (f == 0) { o.put('0'); return; }
  if (dp > 0) {
    int m = nd < dp ? nd : dp;
    for (int i = 0; i < m; i++) o.put(d[i]);
    for (; m < dp; m++) o.put('0');
  } else o.put('0');
  if (nd > dp) {
    o.put('.');
    for (int i = dp; i < 0; i++) o.put('0');
    for (int i = dp > 0 ? dp : 0; i < nd; i++) o.put(d[i]);
  }
}
// encoding/json floatEncoder: 'f' unless |f| < 1e-6 or >= 1e21, then 'e' with e-0X cleaned to e-X.  NaN / Inf are
// an error in encoding/json: the caller must not come here with them.
template <class S> __device__ __forceinline__ void fmt_json_float(S &o, double f, int bits) {
  const double a = f < 0 ? -f : f;
  bool e_form = false;
  if (a != 0) {
    if (bits == 64) e_form = a < 1e-6 || a >= 1e21;
    else { const float af = (float)a; e_form = af < 1e-6f || af >= 1e21f; }
  }
  if (!e_form) { fmt_float(o, f, 'f', bits); return; }
  uint8_t tmp[32];
  StoreOut t{tmp};
  fmt_float(t, f, 'e', bits);
  uint32_t n = t.n;
  if (n >= 4 && tmp[n - 4] == 'e' && tmp[n - 3] == '-' && tmp[n - 2] == '0') { tmp[n - 2] = tmp[n - 1]; n--; }
  for (uint32_t i = 0; i < n; i++) o.put(tmp[i]);
}

}  // namespace dev
}  // namespace tf
