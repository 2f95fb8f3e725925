// tf_devfmt.hpp — device-side restatements of the Go formatting the reference's
// SerializeToString relies on (to_string.go:145-178): strconv.FormatInt/Uint,
// time.Format(DateOnly | RFC3339Nano), Time.String() and Duration.String().
// All helpers write into a caller-provided byte window and return the length.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tf {
namespace dev {

// decimal digits of v without a division: compares against the powers of ten (32-bit ones while v fits 32 bits)
__device__ __forceinline__ int digits_u64(uint64_t v) {
  if (!(v >> 32)) {
    const uint32_t x = (uint32_t)v;
    return 1 + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) + (x >= 10000000u) + (x >= 100000000u) + (x >= 1000000000u);
  }
  return 10 + (v >= 10000000000ull) + (v >= 100000000000ull) + (v >= 1000000000000ull) + (v >= 10000000000000ull) + (v >= 100000000000000ull) +
         (v >= 1000000000000000ull) + (v >= 10000000000000000ull) + (v >= 100000000000000000ull) + (v >= 1000000000000000000ull) + (v >= 10000000000000000000ull);
}
// strconv.FormatUint(v, 10): the digits are written from the back in 9-digit pieces of 32-bit arithmetic (a 64-bit `/ 10` a digit
// and a reversed copy through a second local array is what this replaced: 147 us of sharder_crc32 per 2^20 int64 keys)
__device__ __forceinline__ int fmt_u64(uint8_t *dst, uint64_t v) {
  const int n = digits_u64(v);
  int i = n;
  uint32_t top;
  if (v >> 32) {
    const uint64_t q = v / 1000000000ull;
    uint32_t lo = (uint32_t)(v - q * 1000000000ull);
#pragma unroll
    for (int k = 0; k < 9; k++) { dst[--i] = (uint8_t)('0' + lo % 10u); lo /= 10u; }
    if (q >> 32) {
      const uint32_t q2 = (uint32_t)(q / 1000000000ull);
      uint32_t mid = (uint32_t)(q - (uint64_t)q2 * 1000000000ull);
#pragma unroll
      for (int k = 0; k < 9; k++) { dst[--i] = (uint8_t)('0' + mid % 10u); mid /= 10u; }
      top = q2;
    } else top = (uint32_t)q;
  } else top = (uint32_t)v;
  do { dst[--i] = (uint8_t)('0' + top % 10u); top /= 10u; } while (top);
  return n;
}
__device__ __forceinline__ int fmt_i64(uint8_t *dst, int64_t v) {
  if (v < 0) { dst[0] = '-'; return 1 + fmt_u64(dst + 1, (uint64_t)(-(v + 1)) + 1u); }
  return fmt_u64(dst, (uint64_t)v);
}

__device__ __forceinline__ int64_t floordiv(int64_t a, int64_t b) {
  int64_t q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
  return q;
}

__device__ __forceinline__ void civil_from_days(int64_t z, int64_t &y, int &m, int &d) {
  z += 719468;
  int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  int64_t doe = z - era * 146097;
  int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  int64_t yy = yoe + era * 400;
  int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  int64_t mp = (5 * doy + 2) / 153;
  d = (int)(doy - (153 * mp + 2) / 5 + 1);
  m = (int)(mp < 10 ? mp + 3 : mp - 9);
  y = yy + (m <= 2);
}
__device__ __forceinline__ int64_t days_from_civil(int64_t y, int m, int d) {
  y -= m <= 2;
  int64_t era = (y >= 0 ? y : y - 399) / 400;
  int64_t yoe = y - era * 400;
  int64_t doy = (153 * (m > 2 ? m - 3 : m + 9) + 2) / 5 + d - 1;
  int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + doe - 719468;
}
__device__ __forceinline__ int days_in_month(int m, int64_t y) {
  const int t[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  return (m == 2 && (y % 4 == 0 && (y % 100 != 0 || y % 400 == 0))) ? 29 : t[m - 1];
}

// time.appendInt(b, v, width): zero padded, '-' prefix for negatives
__device__ __forceinline__ int fmt_pad(uint8_t *dst, int64_t v, int width) {
  int w = 0;
  uint64_t u;
  if (v < 0) { dst[w++] = '-'; u = (uint64_t)(-v); } else u = (uint64_t)v;
  int nd = digits_u64(u);
  for (int i = nd; i < width; i++) dst[w++] = '0';
  w += fmt_u64(dst + w, u);
  return w;
}
__device__ __forceinline__ int fmt_date(uint8_t *dst, int64_t sec) {  // "2006-01-02" (UTC)
  int64_t y; int m, d;
  civil_from_days(floordiv(sec, 86400), y, m, d);
  int w = fmt_pad(dst, y, 4);
  dst[w++] = '-'; dst[w++] = (uint8_t)('0' + m / 10); dst[w++] = (uint8_t)('0' + m % 10);
  dst[w++] = '-'; dst[w++] = (uint8_t)('0' + d / 10); dst[w++] = (uint8_t)('0' + d % 10);
  return w;
}
__device__ __forceinline__ int fmt_clock(uint8_t *dst, int64_t sec) {  // "15:04:05"
  int sod = (int)(sec - floordiv(sec, 86400) * 86400);
  int h = sod / 3600, mi = (sod / 60) % 60, s = sod % 60;
  dst[0] = (uint8_t)('0' + h / 10); dst[1] = (uint8_t)('0' + h % 10); dst[2] = ':';
  dst[3] = (uint8_t)('0' + mi / 10); dst[4] = (uint8_t)('0' + mi % 10); dst[5] = ':';
  dst[6] = (uint8_t)('0' + s / 10); dst[7] = (uint8_t)('0' + s % 10);
  return 8;
}
__device__ __forceinline__ int fmt_frac9_trim(uint8_t *dst, int32_t nsec) {  // ".999999999"
  if (nsec == 0) return 0;
  uint8_t b[9];
  int v = nsec;
  for (int i = 8; i >= 0; i--) { b[i] = (uint8_t)('0' + v % 10); v /= 10; }
  int n = 9;
  while (n > 0 && b[n - 1] == '0') n--;
  dst[0] = '.';
  for (int i = 0; i < n; i++) dst[1 + i] = b[i];
  return n + 1;
}
__device__ __forceinline__ int fmt_rfc3339nano(uint8_t *dst, int64_t sec, int32_t nsec) {
  int w = fmt_date(dst, sec);
  dst[w++] = 'T';
  w += fmt_clock(dst + w, sec);
  w += fmt_frac9_trim(dst + w, nsec);
  dst[w++] = 'Z';
  return w;
}
__device__ __forceinline__ int fmt_time_string(uint8_t *dst, int64_t sec, int32_t nsec) {  // Time.String(), UTC
  int w = fmt_date(dst, sec);
  dst[w++] = ' ';
  w += fmt_clock(dst + w, sec);
  w += fmt_frac9_trim(dst + w, nsec);
  const char tail[] = " +0000 UTC";
  for (int i = 0; i < 10; i++) dst[w++] = (uint8_t)tail[i];
  return w;
}

// Duration.String() (time/time.go), at most 25 bytes ("-2562047h47m16.854775808s")
__device__ __forceinline__ int fmt_duration(uint8_t *dst, int64_t ns) {
  uint8_t buf[32];
  int w = 32;
  uint64_t u = ns < 0 ? (uint64_t)(-(ns + 1)) + 1u : (uint64_t)ns;
  bool neg = ns < 0;
  auto frac = [&](int prec) {
    bool print = false;
    for (int i = 0; i < prec; i++) {
      int digit = (int)(u % 10);
      print = print || digit != 0;
      if (print) buf[--w] = (uint8_t)('0' + digit);
      u /= 10;
    }
    if (print) buf[--w] = '.';
  };
  auto integer = [&](uint64_t v) {
    if (v == 0) buf[--w] = '0';
    else while (v > 0) { buf[--w] = (uint8_t)('0' + v % 10); v /= 10; }
  };
  if (u < 1000000000ull) {
    int prec;
    buf[--w] = 's';
    if (u == 0) { dst[0] = '0'; dst[1] = 's'; return 2; }
    else if (u < 1000ull) { prec = 0; buf[--w] = 'n'; }
    else if (u < 1000000ull) { prec = 3; buf[--w] = 0xB5; buf[--w] = 0xC2; }
    else { prec = 6; buf[--w] = 'm'; }
    frac(prec);
    integer(u);
  } else {
    buf[--w] = 's';
    frac(9);
    integer(u % 60);
    u /= 60;
    if (u > 0) {
      buf[--w] = 'm';
      integer(u % 60);
      u /= 60;
      if (u > 0) { buf[--w] = 'h'; integer(u); }
    }
  }
  if (neg) buf[--w] = '-';
  int n = 32 - w;
  for (int i = 0; i < n; i++) dst[i] = buf[w + i];
  return n;
}

}  // namespace dev
}  // namespace tf
