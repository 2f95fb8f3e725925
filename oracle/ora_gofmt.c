/*
 * oracle/ora_gofmt.c — Go standard-library behaviours the reference relies on,
 * restated from the Go language/library specification (the Go stdlib is not in
 * /root/reference; SURVEY.md §8c lists these as "must be restated").
 * TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 *   strconv.FormatInt/FormatUint/FormatFloat('g'|'f', -1), ParseInt/ParseUint
 *   (base 0 and 10), ParseFloat, ParseBool; fmt %v for floats;
 *   time.Format(DateOnly|RFC3339Nano), Time.String, time.Parse(layout),
 *   Duration.String.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ora.h"

/* ---------------- integers ---------------- */
size_t ora_fmt_uint(char *dst, uint64_t v) {
  char tmp[24];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  for (int i = 0; i < n; i++) dst[i] = tmp[n - 1 - i];
  return (size_t)n;
}
size_t ora_fmt_int(char *dst, int64_t v) {
  if (v < 0) {
    dst[0] = '-';
    return 1 + ora_fmt_uint(dst + 1, (uint64_t)(-(v + 1)) + 1u);
  }
  return ora_fmt_uint(dst, (uint64_t)v);
}

/* ---------------- floats ---------------- */
/* Shortest decimal digits that round-trip (strconv 'shortest' mode): try
 * increasing precision until parsing the correctly-rounded p-digit decimal
 * gives the value back.  digits[] gets nd digits, *dp = decimal point position
 * (value = 0.d1d2... * 10^dp). */
static int shortest_digits(double f, int bits, char *digits, int *dp) {
  char buf[64];
  int maxp = bits == 32 ? 9 : 17;
  for (int p = 1; p <= maxp; p++) {
    snprintf(buf, sizeof buf, "%.*e", p - 1, f);
    int ok;
    if (bits == 32) ok = (strtof(buf, NULL) == (float)f);
    else ok = (strtod(buf, NULL) == f);
    if (!ok && p < maxp) {
      /* An exact power of two has a rounding interval half as wide below as above: the correctly rounded p-digit decimal can fall
       * outside it while its neighbour in the last digit lies inside.  strconv's shortest mode (ryuFtoaShortest) takes the
       * shortest decimal INSIDE the interval, so the neighbours are candidates too (at most one of them parses back). */
      char *e0 = strchr(buf, 'e');
      int ex0 = atoi(e0 + 1);
      unsigned long long D = 0;
      for (char *c = buf; c < e0; c++) if (*c >= '0' && *c <= '9') D = D * 10 + (unsigned long long)(*c - '0');
      unsigned long long lim = 1;
      for (int k = 0; k < p; k++) lim *= 10;
      for (int delta = -1; delta <= 1 && !ok; delta += 2) {
        unsigned long long C = D + (unsigned long long)(long long)delta;
        int ex = ex0;
        if (C >= lim) { C /= 10; ex++; }
        if (C < lim / 10) continue;   /* (fewer digits: an earlier p would have taken it) */
        char cand[64], dg[32];
        snprintf(dg, sizeof dg, "%llu", C);
        if (p > 1) snprintf(cand, sizeof cand, "%c.%se%d", dg[0], dg + 1, ex); else snprintf(cand, sizeof cand, "%ce%d", dg[0], ex);
        const int good = bits == 32 ? (strtof(cand, NULL) == (float)f) : (strtod(cand, NULL) == f);
        if (good) { snprintf(buf, sizeof buf, "%s", cand); ok = 1; }
      }
    }
    if (ok || p == maxp) {
      /* buf = d.ddddde[+-]XX */
      int nd = 0;
      char *e = strchr(buf, 'e');
      for (char *c = buf; c < e; c++)
        if (*c >= '0' && *c <= '9') digits[nd++] = *c;
      int ex = atoi(e + 1);
      /* strip trailing zeros (can appear at p == maxp only) */
      while (nd > 1 && digits[nd - 1] == '0') nd--;
      *dp = ex + 1;
      return nd;
    }
  }
  return 0;
}

size_t ora_fmt_float(char *dst, double f, char fmt, int bits) {
  size_t w = 0;
  if (isnan(f)) { memcpy(dst, "NaN", 3); return 3; }
  if (isinf(f)) { memcpy(dst, f > 0 ? "+Inf" : "-Inf", 4); return 4; }
  if (signbit(f)) { dst[w++] = '-'; f = -f; }
  char d[32];
  int dp, nd;
  if (f == 0) { d[0] = '0'; nd = 1; dp = 1; /* Go: digs.nd=0, dp=0 → prints "0" */ }
  else nd = shortest_digits(f, bits, d, &dp);
  int use_e = 0;
  if (fmt == 'g') {
    /* strconv/ftoa.go %g shortest: eprec = 6; exp = dp-1; %e if exp < -4 || exp >= eprec.
     * Go additionally: if eprec > nd && nd >= dp → eprec = nd, overridden by shortest → 6. */
    int exp = (f == 0) ? 0 : dp - 1;
    int eprec = 6; /* "if shortest { eprec = 6 }" — the 1e21 switch-over belongs to encoding/json, not fmt */
    use_e = (exp < -4 || exp >= eprec);
    if (use_e) {
      dst[w++] = d[0];
      if (nd > 1) { dst[w++] = '.'; memcpy(dst + w, d + 1, (size_t)nd - 1); w += (size_t)nd - 1; }
      dst[w++] = 'e';
      if (exp < 0) { dst[w++] = '-'; exp = -exp; } else dst[w++] = '+';
      if (exp < 10) { dst[w++] = '0'; dst[w++] = (char)('0' + exp); }
      else if (exp < 100) { dst[w++] = (char)('0' + exp / 10); dst[w++] = (char)('0' + exp % 10); }
      else { dst[w++] = (char)('0' + exp / 100); dst[w++] = (char)('0' + (exp / 10) % 10); dst[w++] = (char)('0' + exp % 10); }
      return w;
    }
  }
  /* %f with shortest digits */
  if (f == 0) { dst[w++] = '0'; return w; }
  if (dp > 0) {
    int m = nd < dp ? nd : dp;
    memcpy(dst + w, d, (size_t)m); w += (size_t)m;
    for (; m < dp; m++) dst[w++] = '0';
  } else dst[w++] = '0';
  if (nd > dp) {
    dst[w++] = '.';
    for (int i = dp; i < 0; i++) dst[w++] = '0';
    int from = dp > 0 ? dp : 0;
    memcpy(dst + w, d + from, (size_t)(nd - from)); w += (size_t)(nd - from);
  }
  return w;
}

/* encoding/json floatEncoder (encode.go): 'f' unless |f| < 1e-6 or >= 1e21 (compared in float32 for 32-bit values),
 * then 'e' with a two-digit negative exponent e-0X cleaned to e-X.  NaN / ±Inf are an UnsupportedValueError: returns 0. */
size_t ora_json_float(char *dst, double f, int bits) {
  if (isnan(f) || isinf(f)) return 0;
  double a = fabs(f);
  int e_form = 0;
  if (a != 0) {
    if (bits == 64) e_form = a < 1e-6 || a >= 1e21;
    else { float af = (float)a; e_form = af < 1e-6f || af >= 1e21f; }
  }
  if (!e_form) return ora_fmt_float(dst, f, 'f', bits);
  size_t n = ora_fmt_float(dst, f, 'g', bits);  /* at these magnitudes %g is in e-form and equals %e with shortest digits */
  if (n >= 4 && dst[n - 4] == 'e' && dst[n - 3] == '-' && dst[n - 2] == '0') { dst[n - 2] = dst[n - 1]; n--; }
  return n;
}

/* ---------------- ParseUint / ParseInt (strconv/atoi.go) ---------------- */
static int lower_(int c) { return c | ('x' - 'X'); }

static int underscore_ok(const char *s, size_t n) {
  /* strconv/atoi.go underscoreOK */
  char saw = '^';
  size_t i = 0;
  if (n >= 1 && (s[0] == '-' || s[0] == '+')) { s++; n--; }
  int hex = 0;
  if (n >= 2 && s[0] == '0' && (lower_(s[1]) == 'b' || lower_(s[1]) == 'o' || lower_(s[1]) == 'x')) {
    i = 2; saw = '0'; hex = lower_(s[1]) == 'x';
  }
  for (; i < n; i++) {
    unsigned char c = (unsigned char)s[i];
    if ((c >= '0' && c <= '9') || (hex && lower_(c) >= 'a' && lower_(c) <= 'f')) { saw = '0'; continue; }
    if (c == '_') { if (saw != '0') return 0; saw = '_'; continue; }
    if (saw == '_') return 0;
    saw = '!';
  }
  return saw != '_';
}

int ora_parse_uint(const char *s, size_t n, int base, int bits, uint64_t *out) {
  *out = 0;
  if (n == 0) return 1;
  const char *s0 = s; size_t n0 = n;
  int base0 = base == 0;
  if (base == 0) {
    base = 10;
    if (s[0] == '0') {
      if (n >= 3 && lower_(s[1]) == 'b') { base = 2; s += 2; n -= 2; }
      else if (n >= 3 && lower_(s[1]) == 'o') { base = 8; s += 2; n -= 2; }
      else if (n >= 3 && lower_(s[1]) == 'x') { base = 16; s += 2; n -= 2; }
      else { base = 8; s += 1; n -= 1; }
    }
  } else if (base < 2 || base > 36) return 1;
  if (bits == 0) bits = 64;
  uint64_t maxv = bits == 64 ? UINT64_MAX : ((1ull << bits) - 1);
  uint64_t cutoff = UINT64_MAX / (uint64_t)base + 1;
  int underscores = 0;
  uint64_t v = 0;
  for (size_t i = 0; i < n; i++) {
    unsigned char c = (unsigned char)s[i];
    unsigned d;
    if (c == '_' && base0) { underscores = 1; continue; }
    else if (c >= '0' && c <= '9') d = c - '0';
    else if (lower_(c) >= 'a' && lower_(c) <= 'z') d = (unsigned)(lower_(c) - 'a' + 10);
    else return 1;
    if (d >= (unsigned)base) return 1;
    if (v >= cutoff) { *out = maxv; return 2; }
    v *= (uint64_t)base;
    uint64_t v1 = v + d;
    if (v1 < v || v1 > maxv) { *out = maxv; return 2; }
    v = v1;
  }
  if (underscores && !underscore_ok(s0, n0)) return 1;
  *out = v;
  return 0;
}

int ora_parse_int(const char *s, size_t n, int base, int bits, int64_t *out) {
  *out = 0;
  if (n == 0) return 1;
  int neg = 0;
  if (s[0] == '+') { s++; n--; }
  else if (s[0] == '-') { neg = 1; s++; n--; }
  uint64_t un;
  int rc = ora_parse_uint(s, n, base, bits, &un);
  if (rc == 1) return 1;
  if (bits == 0) bits = 64;
  uint64_t cutoff = 1ull << (bits - 1);
  if (!neg && un >= cutoff) { *out = (int64_t)(cutoff - 1); return 2; }
  if (neg && un > cutoff) { *out = (int64_t)(0 - cutoff); return 2; }
  if (rc == 2) return 2;
  *out = neg ? (int64_t)(0 - un) : (int64_t)un;
  return 0;
}

int ora_parse_bool(const char *s, size_t n, int *out) {
  /* strconv.ParseBool: 1 t T TRUE true True / 0 f F FALSE false False */
  static const char *T[] = {"1", "t", "T", "TRUE", "true", "True"};
  static const char *F[] = {"0", "f", "F", "FALSE", "false", "False"};
  for (int i = 0; i < 6; i++) {
    if (strlen(T[i]) == n && memcmp(T[i], s, n) == 0) { *out = 1; return 0; }
    if (strlen(F[i]) == n && memcmp(F[i], s, n) == 0) { *out = 0; return 0; }
  }
  return 1;
}

int ora_parse_float(const char *s, size_t n, int bits, double *out) {
  /* strconv.ParseFloat syntax: [+-] (inf|infinity|nan | decimal | 0x hex) with
   * optional exponent; underscores only legal with a base prefix (hex).  We
   * validate the decimal grammar and delegate rounding to strtod/strtof
   * (correctly rounded in glibc).  Returns 0 ok, 1 syntax, 2 range. */
  *out = 0;
  if (n == 0 || n > 4000) return 1;
  char buf[4096];
  memcpy(buf, s, n); buf[n] = 0;
  const char *p = buf;
  if (*p == '+' || *p == '-') p++;
  if (!strcasecmp(p, "inf") || !strcasecmp(p, "infinity")) { *out = buf[0] == '-' ? -INFINITY : INFINITY; return 0; }
  if (!strcasecmp(p, "nan")) { if (p != buf) return 1; *out = NAN; return 0; }
  int hex = (p[0] == '0' && lower_(p[1]) == 'x');
  if (hex) {
    p += 2;
    int nd = 0;
    while (isxdigit((unsigned char)*p) || *p == '_') { nd += *p != '_'; p++; }
    if (*p == '.') { p++; while (isxdigit((unsigned char)*p) || *p == '_') { nd += *p != '_'; p++; } }
    if (!nd) return 1;
    if (lower_(*p) != 'p') return 1; /* hex floats need p exponent */
    p++;
    if (*p == '+' || *p == '-') p++;
    if (!isdigit((unsigned char)*p)) return 1;
    while (isdigit((unsigned char)*p) || *p == '_') p++;
    if (*p) return 1;
    if (strchr(buf, '_')) { if (!underscore_ok(buf, n)) return 1; /* strip */ char *w = buf; for (char *r = buf; *r; r++) if (*r != '_') *w++ = *r; *w = 0; }
  } else {
    /* readFloat (strconv/atof.go): '_' may separate digits anywhere in the mantissa and the exponent, provided
     * underscoreOK(s) holds (atof_test.go: "1_23.50_0_0e+1_2" parses as 1.235e+14) */
    int nd = 0;
    while (isdigit((unsigned char)*p) || *p == '_') { nd += *p != '_'; p++; }
    if (*p == '.') { p++; while (isdigit((unsigned char)*p) || *p == '_') { nd += *p != '_'; p++; } }
    if (!nd) return 1;
    if (lower_(*p) == 'e') {
      p++;
      if (*p == '+' || *p == '-') p++;
      if (!isdigit((unsigned char)*p)) return 1;
      while (isdigit((unsigned char)*p) || *p == '_') p++;
    }
    if (*p) return 1;
    if (strchr(buf, '_')) { if (!underscore_ok(buf, n)) return 1; char *w = buf; for (char *r = buf; *r; r++) if (*r != '_') *w++ = *r; *w = 0; }
  }
  if (bits == 32) { float f = strtof(buf, NULL); *out = f; if (isinf(f)) return 2; }
  else { double f = strtod(buf, NULL); *out = f; if (isinf(f)) return 2; }
  return 0;
}

/* ---------------- civil time ---------------- */
void ora_civil_from_days(int64_t z, int64_t *y, int *m, int *d) {
  /* days since 1970-01-01 → proleptic Gregorian y-m-d */
  z += 719468;
  int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  int64_t doe = z - era * 146097;
  int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  int64_t yy = yoe + era * 400;
  int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  int64_t mp = (5 * doy + 2) / 153;
  *d = (int)(doy - (153 * mp + 2) / 5 + 1);
  *m = (int)(mp < 10 ? mp + 3 : mp - 9);
  *y = yy + (*m <= 2);
}
int64_t ora_days_from_civil(int64_t y, int m, int d) {
  y -= m <= 2;
  int64_t era = (y >= 0 ? y : y - 399) / 400;
  int64_t yoe = y - era * 400;
  int64_t doy = (153 * (m > 2 ? m - 3 : m + 9) + 2) / 5 + d - 1;
  int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + doe - 719468;
}

static int64_t floordiv(int64_t a, int64_t b) { int64_t q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) q--; return q; }

static size_t fmt_pad(char *dst, int64_t v, int width) {
  /* time.appendInt: zero-padded to width, '-' prefix for negatives */
  size_t w = 0;
  uint64_t u;
  if (v < 0) { dst[w++] = '-'; u = (uint64_t)(-v); } else u = (uint64_t)v;
  char tmp[24]; int n = 0;
  do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
  for (int i = n; i < width; i++) dst[w++] = '0';
  for (int i = 0; i < n; i++) dst[w++] = tmp[n - 1 - i];
  return w;
}

size_t ora_fmt_date(char *dst, int64_t sec) {
  int64_t days = floordiv(sec, 86400);
  int64_t y; int m, d;
  ora_civil_from_days(days, &y, &m, &d);
  size_t w = fmt_pad(dst, y, 4);
  dst[w++] = '-'; w += fmt_pad(dst + w, m, 2);
  dst[w++] = '-'; w += fmt_pad(dst + w, d, 2);
  return w;
}

static size_t fmt_clock(char *dst, int64_t sec) {
  int64_t sod = sec - floordiv(sec, 86400) * 86400;
  size_t w = fmt_pad(dst, sod / 3600, 2);
  dst[w++] = ':'; w += fmt_pad(dst + w, (sod / 60) % 60, 2);
  dst[w++] = ':'; w += fmt_pad(dst + w, sod % 60, 2);
  return w;
}

static size_t fmt_frac9_trim(char *dst, int32_t nsec) {
  /* layout ".999999999": omitted if zero, trailing zeros trimmed */
  if (nsec == 0) return 0;
  char b[10];
  snprintf(b, sizeof b, "%09d", nsec);
  int n = 9;
  while (n > 0 && b[n - 1] == '0') n--;
  dst[0] = '.';
  memcpy(dst + 1, b, (size_t)n);
  return (size_t)n + 1;
}

size_t ora_fmt_rfc3339nano(char *dst, int64_t sec, int32_t nsec) {
  size_t w = ora_fmt_date(dst, sec);
  dst[w++] = 'T';
  w += fmt_clock(dst + w, sec);
  w += fmt_frac9_trim(dst + w, nsec);
  dst[w++] = 'Z';
  return w;
}

size_t ora_fmt_time_string(char *dst, int64_t sec, int32_t nsec) {
  /* Time.String(): "2006-01-02 15:04:05.999999999 -0700 MST" in UTC */
  size_t w = ora_fmt_date(dst, sec);
  dst[w++] = ' ';
  w += fmt_clock(dst + w, sec);
  w += fmt_frac9_trim(dst + w, nsec);
  memcpy(dst + w, " +0000 UTC", 10);
  return w + 10;
}

/* ---------------- Duration.String (time/time.go) ---------------- */
size_t ora_fmt_duration(char *dst, int64_t ns) {
  char buf[40];
  int w = (int)sizeof buf;
  uint64_t u = ns < 0 ? (uint64_t)(-(ns + 1)) + 1u : (uint64_t)ns;
  int neg = ns < 0;
#define FMTFRAC(prec)                                          \
  do {                                                         \
    int print = 0;                                             \
    for (int i = 0; i < (prec); i++) {                         \
      int digit = (int)(u % 10);                               \
      print = print || digit != 0;                             \
      if (print) buf[--w] = (char)('0' + digit);               \
      u /= 10;                                                 \
    }                                                          \
    if (print) buf[--w] = '.';                                 \
  } while (0)
#define FMTINT(v)                                              \
  do {                                                         \
    uint64_t vv = (v);                                         \
    if (vv == 0) buf[--w] = '0';                               \
    else while (vv > 0) { buf[--w] = (char)('0' + vv % 10); vv /= 10; } \
  } while (0)
  if (u < 1000000000ull) {
    int prec;
    buf[--w] = 's';
    if (u == 0) { memcpy(dst, "0s", 2); return 2; }
    else if (u < 1000ull) { prec = 0; buf[--w] = 'n'; }
    else if (u < 1000000ull) { prec = 3; buf[--w] = (char)0xB5; buf[--w] = (char)0xC2; }
    else { prec = 6; buf[--w] = 'm'; }
    FMTFRAC(prec);
    FMTINT(u);
  } else {
    buf[--w] = 's';
    FMTFRAC(9);
    FMTINT(u % 60);
    u /= 60;
    if (u > 0) {
      buf[--w] = 'm';
      FMTINT(u % 60);
      u /= 60;
      if (u > 0) { buf[--w] = 'h'; FMTINT(u); }
    }
  }
  if (neg) buf[--w] = '-';
  size_t n = sizeof buf - (size_t)w;
  memcpy(dst, buf + w, n);
  return n;
}

/* ---------------- time.Parse(layout, value) ---------------- */
/* Restates time/format.go's chunked layout parsing for the layout elements
 * that occur in the layouts used on the path (filter grammar, stringToTime's
 * 22 layouts filter_rows/util.go:16-39, spf13/cast timeFormats, user CSV
 * TimestampParsers).  Result is normalised to UTC. */
static const char *MONTHS[] = {"January", "February", "March", "April", "May", "June", "July", "August", "September", "October", "November", "December"};
static const char *DAYS[] = {"Sunday", "Monday", "Tuesday", "Wednesday", "Thursday", "Friday", "Saturday"};

enum { C_NONE, C_LONGMONTH, C_MONTH, C_NUMMONTH, C_ZEROMONTH, C_LONGWEEKDAY, C_WEEKDAY, C_DAY, C_UNDERDAY, C_ZERODAY,
       C_HOUR, C_HOUR12, C_ZEROHOUR12, C_MINUTE, C_ZEROMINUTE, C_SECOND, C_ZEROSECOND, C_LONGYEAR, C_YEAR, C_PM, C_pm,
       C_TZ, C_ISOTZ, C_ISOTZCOLON, C_ISOTZSHORT, C_NUMTZ, C_NUMTZCOLON, C_NUMTZSHORT, C_FRAC0, C_FRAC9, C_ZEROYEARDAY, C_UNDERYEARDAY };

static int lower_start(const char *s, size_t n) { return n > 0 && s[0] >= 'a' && s[0] <= 'z'; } /* startsWithLowerCase */

static int starts(const char *s, size_t n, const char *p) { size_t l = strlen(p); return n >= l && memcmp(s, p, l) == 0; }

/* time.nextStdChunk: returns chunk code, sets *plen = prefix literal length, *clen = chunk len, *fracdigits */
static int next_chunk(const char *l, size_t n, size_t *plen, size_t *clen, int *fd) {
  for (size_t i = 0; i < n; i++) {
    const char *s = l + i; size_t r = n - i;
    *plen = i; *fd = 0;
    switch (s[0]) {
      case 'J':
        if (starts(s, r, "January")) { *clen = 7; return C_LONGMONTH; }
        if (starts(s, r, "Jan") && !lower_start(s + 3, r - 3)) { *clen = 3; return C_MONTH; }
        break;
      case 'M':
        if (starts(s, r, "Monday")) { *clen = 6; return C_LONGWEEKDAY; }
        if (starts(s, r, "Mon") && !lower_start(s + 3, r - 3)) { *clen = 3; return C_WEEKDAY; }
        if (starts(s, r, "MST")) { *clen = 3; return C_TZ; }
        break;
      case '0':
        if (r >= 2 && s[1] >= '1' && s[1] <= '6') {
          static const int m[] = {0, C_ZEROMONTH, C_ZERODAY, C_ZEROHOUR12, C_ZEROMINUTE, C_ZEROSECOND, C_YEAR};
          *clen = 2; return m[s[1] - '0'];
        }
        if (r >= 3 && s[1] == '0' && s[2] == '2') { *clen = 3; return C_ZEROYEARDAY; }
        break;
      case '1':
        if (r >= 2 && s[1] == '5') { *clen = 2; return C_HOUR; }
        *clen = 1; return C_NUMMONTH;
      case '2':
        if (starts(s, r, "2006")) { *clen = 4; return C_LONGYEAR; }
        *clen = 1; return C_DAY;
      case '_':
        if (r >= 2 && s[1] == '2') {
          if (starts(s + 1, r - 1, "2006")) { *plen = i + 1; *clen = 4; return C_LONGYEAR; }
          *clen = 2; return C_UNDERDAY;
        }
        if (r >= 3 && s[1] == '_' && s[2] == '2') { *clen = 3; return C_UNDERYEARDAY; }
        break;
      case '3': *clen = 1; return C_HOUR12;
      case '4': *clen = 1; return C_MINUTE;
      case '5': *clen = 1; return C_SECOND;
      case 'P': if (r >= 2 && s[1] == 'M') { *clen = 2; return C_PM; } break;
      case 'p': if (r >= 2 && s[1] == 'm') { *clen = 2; return C_pm; } break;
      case '-':
        if (starts(s, r, "-070000")) break; /* seconds tz: unsupported */
        if (starts(s, r, "-07:00:00")) break;
        if (starts(s, r, "-0700")) { *clen = 5; return C_NUMTZ; }
        if (starts(s, r, "-07:00")) { *clen = 6; return C_NUMTZCOLON; }
        if (starts(s, r, "-07")) { *clen = 3; return C_NUMTZSHORT; }
        break;
      case 'Z':
        if (starts(s, r, "Z0700")) { *clen = 5; return C_ISOTZ; }
        if (starts(s, r, "Z07:00")) { *clen = 6; return C_ISOTZCOLON; }
        if (starts(s, r, "Z07")) { *clen = 3; return C_ISOTZSHORT; }
        break;
      case '.': case ',':
        if (r >= 2 && (s[1] == '0' || s[1] == '9')) {
          char ch = s[1]; size_t j = 1;
          while (j < r && s[j] == ch) j++;
          if (!(j < r && s[j] >= '0' && s[j] <= '9')) {
            *clen = j; *fd = (int)j - 1; return ch == '0' ? C_FRAC0 : C_FRAC9;
          }
        }
        break;
    }
  }
  *plen = n; *clen = 0;
  return C_NONE;
}

static int getnum(const char **s, size_t *n, int fixed, int *out) {
  /* time.getnum: 1 or 2 digits (2 required if fixed) */
  if (*n < 1 || !isdigit((unsigned char)(*s)[0])) return 1;
  if (*n < 2 || !isdigit((unsigned char)(*s)[1])) {
    if (fixed) return 1;
    *out = (*s)[0] - '0'; (*s)++; (*n)--; return 0;
  }
  *out = ((*s)[0] - '0') * 10 + ((*s)[1] - '0'); (*s) += 2; (*n) -= 2; return 0;
}

static int lookup_name(const char **tab, int ntab, int shortform, const char **s, size_t *n, int *out) {
  for (int i = 0; i < ntab; i++) {
    size_t l = shortform ? 3 : strlen(tab[i]);
    if (*n >= l && strncasecmp(*s, tab[i], l) == 0) { *out = i; *s += l; *n -= l; return 0; }
  }
  return 1;
}

static int days_in(int m, int64_t y) {
  static const int dm[] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  if (m == 2 && (y % 4 == 0 && (y % 100 != 0 || y % 400 == 0))) return 29;
  return dm[m - 1];
}

/* time.skip: runs of spaces are equivalent; a space of the layout also takes no space at the value's end */
static int skip_prefix(const char **s, size_t *n, const char *pre, size_t plen) {
  while (plen > 0) {
    if (pre[0] == ' ') {
      if (*n > 0 && (*s)[0] != ' ') return 1;
      while (plen > 0 && pre[0] == ' ') { pre++; plen--; }
      while (*n > 0 && (*s)[0] == ' ') { (*s)++; (*n)--; }
      continue;
    }
    if (*n == 0 || (*s)[0] != pre[0]) return 1;
    pre++; plen--; (*s)++; (*n)--;
  }
  return 0;
}
/* time.parseSignedOffset: length of [+-]digits with a value <= 23, else 0 */
static size_t signed_offset(const char *s, size_t n) {
  if (n == 0 || (s[0] != '+' && s[0] != '-')) return 0;
  size_t k = 1; unsigned long long x = 0;
  while (k < n && isdigit((unsigned char)s[k])) { if (x < (1ull << 62)) x = x * 10 + (unsigned)(s[k] - '0'); k++; }
  return (k > 1 && x <= 23) ? k : 0;
}

int ora_time_parse(const char *layout, const char *val, size_t n, int64_t *osec, int32_t *onsec) {
  const char *l = layout; size_t ln = strlen(layout);
  const char *s = val;
  int64_t year = 0; int month = -1, day = -1, hour = 0, min = 0, sec = 0, nsec = 0, yday = -1;
  int pm_set = 0, pm = 0, have_year = 0;
  int zoff = 0; int have_zoff = 0; /* -1 = named zone (offset unknown → treated as UTC unless "UTC") */
  for (;;) {
    size_t plen, clen; int fd;
    int c = next_chunk(l, ln, &plen, &clen, &fd);
    /* literal prefix must match */
    if (skip_prefix(&s, &n, l, plen)) return 1;
    if (c == C_NONE) { if (n != 0) return 1; break; }
    l += plen + clen; ln -= plen + clen;
    int v;
    switch (c) {
      case C_YEAR: {
        if (n < 2 || !isdigit((unsigned char)s[0]) || !isdigit((unsigned char)s[1])) return 1;
        year = (s[0] - '0') * 10 + (s[1] - '0'); s += 2; n -= 2;
        year += year >= 69 ? 1900 : 2000; have_year = 1; break;
      }
      case C_LONGYEAR: {
        if (n < 4 || !isdigit((unsigned char)s[0])) return 1;
        for (int i = 0; i < 4; i++) if (!isdigit((unsigned char)s[i])) return 1;
        year = (s[0] - '0') * 1000 + (s[1] - '0') * 100 + (s[2] - '0') * 10 + (s[3] - '0');
        s += 4; n -= 4; have_year = 1; break;
      }
      case C_MONTH: if (lookup_name(MONTHS, 12, 1, &s, &n, &v)) return 1; month = v + 1; break;
      case C_LONGMONTH: if (lookup_name(MONTHS, 12, 0, &s, &n, &v)) return 1; month = v + 1; break;
      case C_NUMMONTH: case C_ZEROMONTH:
        if (getnum(&s, &n, c == C_ZEROMONTH, &v) || v <= 0 || v > 12) return 1; month = v; break;
      case C_WEEKDAY: if (lookup_name(DAYS, 7, 1, &s, &n, &v)) return 1; break;
      case C_LONGWEEKDAY: if (lookup_name(DAYS, 7, 0, &s, &n, &v)) return 1; break;
      case C_DAY: case C_UNDERDAY: case C_ZERODAY:
        if (c == C_UNDERDAY && n > 0 && s[0] == ' ') { s++; n--; }
        if (getnum(&s, &n, c == C_ZERODAY, &v)) return 1; day = v; break;
      case C_ZEROYEARDAY: case C_UNDERYEARDAY: { /* getnum3 */
        if (c == C_UNDERYEARDAY) for (int i = 0; i < 2; i++) if (n > 0 && s[0] == ' ') { s++; n--; }
        int nd = 0; yday = 0;
        while (nd < 3 && (size_t)nd < n && isdigit((unsigned char)s[nd])) { yday = yday * 10 + (s[nd] - '0'); nd++; }
        if (nd == 0 || (c == C_ZEROYEARDAY && nd != 3)) return 1;
        s += nd; n -= (size_t)nd; break;
      }
      case C_HOUR: if (getnum(&s, &n, 0, &v) || v < 0 || v >= 24) return 1; hour = v; break;
      case C_HOUR12: case C_ZEROHOUR12:
        if (getnum(&s, &n, c == C_ZEROHOUR12, &v) || v < 0 || v > 12) return 1; hour = v; break;
      case C_MINUTE: case C_ZEROMINUTE:
        if (getnum(&s, &n, c == C_ZEROMINUTE, &v) || v < 0 || v >= 60) return 1; min = v; break;
      case C_SECOND: case C_ZEROSECOND: {
        if (getnum(&s, &n, c == C_ZEROSECOND, &v) || v < 0 || v >= 60) return 1; sec = v;
        /* "fractional second in the input even if the layout lacks it": only
         * if the next layout chunk is not itself a fractional-second chunk */
        if (n >= 2 && (s[0] == '.' || s[0] == ',') && isdigit((unsigned char)s[1])) {
          size_t p2, c2; int f2;
          int nc = next_chunk(l, ln, &p2, &c2, &f2);
          if (!((nc == C_FRAC0 || nc == C_FRAC9) && p2 == 0)) {
            size_t j = 1; while (j < n && isdigit((unsigned char)s[j])) j++;
            int64_t ns = 0; int digits = 0;
            for (size_t k = 1; k < j; k++) if (digits < 9) { ns = ns * 10 + (s[k] - '0'); digits++; }
            while (digits < 9) { ns *= 10; digits++; }
            nsec = (int)ns; s += j; n -= j;
          }
        }
        break;
      }
      case C_PM: if (n < 2) return 1; if (!memcmp(s, "PM", 2)) pm = 1; else if (!memcmp(s, "AM", 2)) pm = 0; else return 1; pm_set = 1; s += 2; n -= 2; break;
      case C_pm: if (n < 2) return 1; if (!memcmp(s, "pm", 2)) pm = 1; else if (!memcmp(s, "am", 2)) pm = 0; else return 1; pm_set = 1; s += 2; n -= 2; break;
      case C_ISOTZ: case C_ISOTZCOLON: case C_ISOTZSHORT: case C_NUMTZ: case C_NUMTZCOLON: case C_NUMTZSHORT: {
        if ((c == C_ISOTZ || c == C_ISOTZCOLON || c == C_ISOTZSHORT) && n >= 1 && s[0] == 'Z') { s++; n--; zoff = 0; have_zoff = 1; break; }
        int sign, hh, mm = 0;
        if (c == C_ISOTZCOLON || c == C_NUMTZCOLON) {
          if (n < 6 || s[3] != ':') return 1;
          if (!isdigit((unsigned char)s[1]) || !isdigit((unsigned char)s[2]) || !isdigit((unsigned char)s[4]) || !isdigit((unsigned char)s[5])) return 1;
          hh = (s[1] - '0') * 10 + s[2] - '0'; mm = (s[4] - '0') * 10 + s[5] - '0';
          sign = s[0]; s += 6; n -= 6;
        } else if (c == C_ISOTZSHORT || c == C_NUMTZSHORT) {
          if (n < 3 || !isdigit((unsigned char)s[1]) || !isdigit((unsigned char)s[2])) return 1;
          hh = (s[1] - '0') * 10 + s[2] - '0'; sign = s[0]; s += 3; n -= 3;
        } else {
          if (n < 5) return 1;
          for (int i = 1; i < 5; i++) if (!isdigit((unsigned char)s[i])) return 1;
          hh = (s[1] - '0') * 10 + s[2] - '0'; mm = (s[3] - '0') * 10 + s[4] - '0';
          sign = s[0]; s += 5; n -= 5;
        }
        if (hh > 24 || mm > 60) return 1;
        if (sign == '+') zoff = hh * 3600 + mm * 60; else if (sign == '-') zoff = -(hh * 3600 + mm * 60); else return 1;
        have_zoff = 1; break;
      }
      case C_TZ: {
        /* time.parseTimeZone: "UTC"; ChST / MeST; GMT with an optional signed hour; a bare signed hour; 3-5 capitals.
         * A name never moves the instant (Go keeps the wall time under a fabricated location). */
        if (n >= 3 && !memcmp(s, "UTC", 3)) { s += 3; n -= 3; break; }
        if (n < 3) return 1;
        if (n >= 4 && (!memcmp(s, "ChST", 4) || !memcmp(s, "MeST", 4))) { s += 4; n -= 4; break; }
        if (!memcmp(s, "GMT", 3)) { size_t k = 3 + signed_offset(s + 3, n - 3); s += k; n -= k; break; }
        if (s[0] == '+' || s[0] == '-') { size_t k = signed_offset(s, n); if (!k) return 1; s += k; n -= k; break; }
        size_t j = 0; while (j < n && j < 6 && s[j] >= 'A' && s[j] <= 'Z') j++;
        if (j < 3 || j > 5) return 1;
        if (j == 5 && s[4] != 'T') return 1;
        if (j == 4 && !(s[3] == 'T' || !memcmp(s, "WITA", 4))) return 1;
        s += j; n -= j;
        break;
      }
      case C_FRAC0: {
        if (n < (size_t)fd + 1 || (s[0] != '.' && s[0] != ',')) return 1;
        int64_t ns = 0;
        for (int i = 1; i <= fd; i++) { if (!isdigit((unsigned char)s[i])) return 1; ns = ns * 10 + (s[i] - '0'); }
        if (n > (size_t)fd + 1 && isdigit((unsigned char)s[fd + 1])) return 1;
        for (int i = fd; i < 9; i++) ns *= 10;
        nsec = (int)ns; s += fd + 1; n -= (size_t)fd + 1; break;
      }
      case C_FRAC9: {
        if (n < 2 || (s[0] != '.' && s[0] != ',') || !isdigit((unsigned char)s[1])) break; /* fraction optional */
        size_t j = 1; while (j < n && isdigit((unsigned char)s[j])) j++;
        int64_t ns = 0; int digits = 0;
        for (size_t k = 1; k < j; k++) if (digits < 9) { ns = ns * 10 + (s[k] - '0'); digits++; }
        while (digits < 9) { ns *= 10; digits++; }
        nsec = (int)ns; s += j; n -= j; break;
      }
      default: return 1;
    }
  }
  if (pm_set) { if (pm && hour < 12) hour += 12; else if (!pm && hour == 12) hour = 0; }
  if (!have_year) year = 0;
  if (yday >= 0) { /* the day of the year decides; a month / day the value also gave must agree (format.go "day-of-year does not match") */
    int m = 0, d = 0;
    if (days_in(2, year) == 29) { if (yday == 31 + 29) { m = 2; d = 29; } else if (yday > 31 + 29) yday--; }
    if (yday < 1 || yday > 365) return 1;
    if (m == 0) { m = 1; d = yday; while (d > days_in(m, 1)) { d -= days_in(m, 1); m++; } }
    if (month >= 0 && month != m) return 1;
    month = m;
    if (day >= 0 && day != d) return 1;
    day = d;
  } else {
    if (month < 0) month = 1;
    if (day < 0) day = 1;
  }
  if (day < 1 || day > days_in(month, year)) return 1;
  int64_t t = ora_days_from_civil(year, month, day) * 86400 + hour * 3600 + min * 60 + sec;
  if (have_zoff) t -= zoff;
  *osec = t; *onsec = nsec;
  return 0;
}

/* time.ParseDuration (Go time/format.go), as the reference reaches it through spf13/cast v1.7.1 (go.mod:64; not vendored under
 * /root/reference — restated from the published source): [-+]?([0-9]*(\.[0-9]*)?[a-z]+)+, units ns us µs μs ms s m h, leadingInt /
 * leadingFraction with their overflow rules, the fraction added through float64.  0 = ok, 1 = error. */
static int leading_int(const char **ps, const char *e, uint64_t *x) {
  const char *s = *ps; *x = 0;
  for (; s < e && *s >= '0' && *s <= '9'; s++) {
    if (*x > (1ull << 63) / 10) return 1;
    *x = *x * 10 + (uint64_t)(*s - '0');
    if (*x > 1ull << 63) return 1;
  }
  *ps = s;
  return 0;
}
int ora_parse_duration(const char *s, size_t n, int64_t *out) {
  const char *e = s + n;
  uint64_t d = 0; int neg = 0;
  *out = 0;
  if (s < e && (*s == '-' || *s == '+')) { neg = *s == '-'; s++; }
  if (e - s == 1 && *s == '0') return 0;
  if (s == e) return 1;
  while (s < e) {
    uint64_t v = 0, f = 0; double scale = 1;
    if (!(*s == '.' || (*s >= '0' && *s <= '9'))) return 1;
    const char *pl = s;
    if (leading_int(&s, e, &v)) return 1;
    int pre = s != pl, post = 0;
    if (s < e && *s == '.') {
      s++;
      const char *pf = s; int over = 0;
      for (; s < e && *s >= '0' && *s <= '9'; s++) {
        if (over) continue;
        if (f > ((1ull << 63) - 1) / 10) { over = 1; continue; }
        uint64_t y = f * 10 + (uint64_t)(*s - '0');
        if (y > 1ull << 63) { over = 1; continue; }
        f = y; scale *= 10;
      }
      post = s != pf;
    }
    if (!pre && !post) return 1;
    const char *u = s;
    while (s < e && !(*s == '.' || (*s >= '0' && *s <= '9'))) s++;
    size_t ul = (size_t)(s - u);
    if (ul == 0) return 1; /* missing unit */
    uint64_t unit;
    if (ul == 2 && !memcmp(u, "ns", 2)) unit = 1;
    else if ((ul == 2 && !memcmp(u, "us", 2)) || (ul == 3 && !memcmp(u, "\xC2\xB5s", 3)) || (ul == 3 && !memcmp(u, "\xCE\xBCs", 3))) unit = 1000;
    else if (ul == 2 && !memcmp(u, "ms", 2)) unit = 1000000;
    else if (ul == 1 && *u == 's') unit = 1000000000ull;
    else if (ul == 1 && *u == 'm') unit = 60000000000ull;
    else if (ul == 1 && *u == 'h') unit = 3600000000000ull;
    else return 1; /* unknown unit */
    if (v > (1ull << 63) / unit) return 1;
    v *= unit;
    if (f > 0) {
      v += (uint64_t)((double)f * ((double)unit / scale));
      if (v > 1ull << 63) return 1;
    }
    d += v;
    if (d > 1ull << 63) return 1;
  }
  if (neg) { *out = (int64_t)(0 - d); return 0; }
  if (d > (1ull << 63) - 1) return 1;
  *out = (int64_t)d;
  return 0;
}
/* cast.ToDurationE(string) (spf13/cast v1.7.1 caste.go): strings.ContainsAny(s, "nsuµmh") ? ParseDuration(s) : ParseDuration(s + "ns") */
int ora_cast_string_to_duration(const char *s, size_t n, int64_t *out) {
  int has = 0;
  for (size_t i = 0; i < n && !has; i++) {
    unsigned char c = (unsigned char)s[i];
    has = c == 'n' || c == 's' || c == 'u' || c == 'm' || c == 'h' || (c == 0xC2 && i + 1 < n && (unsigned char)s[i + 1] == 0xB5);
  }
  if (has) return ora_parse_duration(s, n, out);
  char *t = (char *)malloc(n + 3);
  memcpy(t, s, n); memcpy(t + n, "ns", 2);
  int rc = ora_parse_duration(t, n + 2, out);
  free(t);
  return rc;
}
