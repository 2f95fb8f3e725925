// tf_f64range.hpp — does a decimal number text overflow float64?  Shared by the serializers (json.Number values) and the CSV
// ingest (parseFloatValue keeps the original text when strconv.ParseFloat fails, ErrRange included).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tf {

// strconv.ParseFloat(s, 64) overflows to ±Inf (err = ErrRange) exactly when |s| >= 2^1024 - 2^970
// (the value half-way between MaxFloat64 and 2^1024; ties round to the even neighbour = overflow).
static __constant__ char F64_OVERFLOW_DIGITS[310] =
    "179769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207096330286416692887910946555547851"
    "940402630657488671505820681908902000708383676273854845817711531764475730270069855571366959622842914819860834936475292719074168444365"
    "510704342711559699508093042880177904174497792";
// +1 / -1 if the JSON number text overflows float64, 0 otherwise (inf/nan words and malformed text: 0)
template <class P> __device__ int jsonnum_overflow(const P &p, uint32_t n) {
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


}  // namespace tf
