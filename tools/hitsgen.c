/*
 * tools/hitsgen.c — synthetic ClickBench-`hits`-shaped CSV generator
 * (SURVEY.md §8d "Synthetic inputs").  Workload tooling for bench.py and the
 * size-independent tests; not part of the product and not the oracle.
 *
 * Every row is generated from splitmix64(seed, row) alone, so any row range
 * can be produced independently — rank g of an N-GPU run generates exactly
 * rows [g*n/N, (g+1)*n/N) of the same logical table.
 *
 * Column roles (chosen by the caller from the reference schema's names/types):
 *   0 int16 0/1 flag       1 int16 small (0..4999)   2 int32 uniform (full range)
 *   3 int32 zipf-ish 1e4   4 int64 uniform           5 int64 small
 *   6 timestamp text       7 date text (2013-07-01..31)
 *   8 title (UTF-8, Cyrillic, commas/quotes → quoted) 9 url (ASCII)
 *  10 sparse text (empty w.p. 0.9)   11 short text (1-2 chars)   12 int32 small
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef struct { uint64_t s; } rng_t;
static inline uint64_t next64(rng_t *r) {
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline double unif(rng_t *r) { return (double)(next64(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline double gauss(rng_t *r) {
  double u1 = unif(r), u2 = unif(r);
  if (u1 < 1e-300) u1 = 1e-300;
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

static inline char *put_i64(char *p, int64_t v) {
  char tmp[24]; int n = 0;
  uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
  if (v < 0) *p++ = '-';
  do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
  while (n) *p++ = tmp[--n];
  return p;
}
static inline char *put2(char *p, int v) { *p++ = (char)('0' + v / 10); *p++ = (char)('0' + v % 10); return p; }

static char *put_text(char *p, rng_t *r, int len, int cyr, int quote_prob_pct) {
  /* a field that contains a delimiter, a quote or edge spaces must be quoted for
   * the reference reader (it TrimSpace()s every field, reader.go:275) */
  static const char ascii[] = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789   -_.:/?&=%";
  char buf[2100]; int n = 0; int need_quote = 0;
  while (n < len) {
    uint64_t x = next64(r);
    int pct = (int)(x % 100);
    if (cyr && pct < 55 && n + 2 <= len) { buf[n++] = (char)0xD0; buf[n++] = (char)(0x90 + (x >> 8) % 0x30); }  /* U+0410..U+043F */
    else if (pct >= 97 && quote_prob_pct) { buf[n++] = ','; need_quote = 1; }
    else if (pct == 96 && quote_prob_pct) { buf[n++] = '"'; need_quote = 1; }
    else buf[n++] = ascii[(x >> 16) % (sizeof ascii - 1)];
  }
  if (n && (buf[0] == ' ' || buf[n - 1] == ' ')) need_quote = 1;
  if (n == 1 && buf[0] == '"') need_quote = 1;
  if (!need_quote) { memcpy(p, buf, (size_t)n); return p + n; }
  *p++ = '"';
  for (int i = 0; i < n; i++) { if (buf[i] == '"') *p++ = '"'; *p++ = buf[i]; }
  *p++ = '"';
  return p;
}

static char *put_url(char *p, rng_t *r, int len) {
  static const char ascii[] = "abcdefghijklmnopqrstuvwxyz0123456789-_./?&=%";
  static const char pre[] = "http://";
  int n = 0;
  for (; n < 7 && n < len; n++) *p++ = pre[n];
  for (; n < len; n++) *p++ = ascii[next64(r) % (sizeof ascii - 1)];
  return p;
}

static int lognormal_len(rng_t *r, double mu, double sigma, int cap) {
  double v = exp(mu + sigma * gauss(r));
  int n = (int)v;
  if (n < 0) n = 0;
  if (n > cap) n = cap;
  return n;
}

/* worst-case bytes per row for the given roles */
uint64_t hits_row_cap(const int32_t *roles, int32_t ncols) {
  uint64_t cap = 2;
  for (int i = 0; i < ncols; i++) {
    switch (roles[i]) {
      case 8: cap += 2 * 512 + 3; break;
      case 9: cap += 1024 + 1; break;
      case 10: cap += 2 * 24 + 3; break;
      default: cap += 24;
    }
  }
  return cap;
}

uint64_t hits_header(const char *const *names, int32_t ncols, char *out) {
  char *p = out;
  for (int i = 0; i < ncols; i++) { if (i) *p++ = ','; size_t n = strlen(names[i]); memcpy(p, names[i], n); p += n; }
  *p++ = '\n';
  return (uint64_t)(p - out);
}

/* Writes rows [row0, row0+nrows) as CSV lines; returns bytes written (0 if cap too small). */
uint64_t hits_csv(uint64_t seed, int64_t row0, int64_t nrows, const int32_t *roles, int32_t ncols, char *out, uint64_t cap) {
  char *p = out;
  uint64_t rowcap = hits_row_cap(roles, ncols);
  for (int64_t r = row0; r < row0 + nrows; r++) {
    if ((uint64_t)(p - out) + rowcap > cap) return 0;
    rng_t g = {seed ^ (uint64_t)r * 0xD1342543DE82EF95ull};
    next64(&g);
    int day = 1 + (int)(next64(&g) % 31);          /* EventDate: 2013-07-01 … 2013-07-31 */
    int sod = (int)(next64(&g) % 86400);
    for (int c = 0; c < ncols; c++) {
      if (c) *p++ = ',';
      switch (roles[c]) {
        case 0: *p++ = (char)('0' + (next64(&g) % 100 < 30)); break;
        case 1: p = put_i64(p, (int64_t)(next64(&g) % 5000)); break;
        case 2: p = put_i64(p, (int64_t)(int32_t)(uint32_t)next64(&g)); break;
        case 3: { double u = unif(&g); p = put_i64(p, (int64_t)(10000.0 * pow(u, 4.0))); break; }
        case 4: p = put_i64(p, (int64_t)next64(&g)); break;
        case 5: p = put_i64(p, (int64_t)(next64(&g) % 1000)); break;
        case 6: {  /* "2013-07-DD HH:MM:SS" around the event time */
          int s = (sod + (int)(next64(&g) % 600)) % 86400;
          memcpy(p, "2013-07-", 8); p += 8; p = put2(p, day); *p++ = ' ';
          p = put2(p, s / 3600); *p++ = ':'; p = put2(p, (s / 60) % 60); *p++ = ':'; p = put2(p, s % 60);
          break;
        }
        case 7: memcpy(p, "2013-07-", 8); p += 8; p = put2(p, day); break;
        case 8: p = put_text(p, &g, lognormal_len(&g, 3.6, 0.8, 512), 1, 1); break;
        case 9: p = put_url(p, &g, lognormal_len(&g, 4.0, 0.7, 1024)); break;
        case 10: if (next64(&g) % 10 == 0) p = put_text(p, &g, 1 + (int)(next64(&g) % 24), 0, 1); break;
        case 11: { int n = 1 + (int)(next64(&g) % 2); for (int i = 0; i < n; i++) *p++ = (char)('A' + next64(&g) % 26); break; }
        default: p = put_i64(p, (int64_t)(next64(&g) % 100000)); break;
      }
    }
    *p++ = '\n';
  }
  return (uint64_t)(p - out);
}
