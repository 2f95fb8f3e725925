// tf_gotime.hpp — Go's time.Parse(layout, value) on device (time/format.go): the layout is cut into its standard
// chunks once on the host (nextStdChunk), the device walks the compiled ops over a field's bytes.  Used by the CSV
// ingest for the user's TimestampParsers (reader_csv.go:405-415) and for spf13/cast's StringToDate layout list
// (cast v1.7.1 caste.go timeFormats), which strictify.Strictify applies to date / datetime / timestamp cells that are
// still strings (strictify.go:118-143).  Results are UTC instants (seconds, nanoseconds); a named zone other than UTC
// has no offset here (Go fabricates a zero-offset location for abbreviations it does not know).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace tf {

enum GtCode : uint8_t {
  GT_END = 0, GT_LONGMONTH, GT_MONTH, GT_NUMMONTH, GT_ZEROMONTH, GT_LONGWEEKDAY, GT_WEEKDAY, GT_DAY, GT_UNDERDAY, GT_ZERODAY,
  GT_HOUR, GT_HOUR12, GT_ZEROHOUR12, GT_MINUTE, GT_ZEROMINUTE, GT_SECOND, GT_ZEROSECOND, GT_LONGYEAR, GT_YEAR, GT_PM, GT_pm,
  GT_TZ, GT_ISOTZ, GT_ISOTZCOLON, GT_ISOTZSHORT, GT_NUMTZ, GT_NUMTZCOLON, GT_NUMTZSHORT, GT_FRAC0, GT_FRAC9, GT_ZEROYEARDAY, GT_UNDERYEARDAY
};
// one op = the literal text in front of a chunk + the chunk; the last op of a layout is GT_END with the trailing literal
struct GtOp { uint8_t code, fd /* fraction digits of GT_FRAC0/9 */, next_frac /* the op behind is a fraction glued to this one */, pad; uint16_t lit_off, lit_len; };

// ---- host: time.nextStdChunk over the whole layout -----------------------------------------------------------------
inline void gotime_compile(const std::string &layout, std::vector<GtOp> &ops, std::string &lits) {
  auto starts = [&](size_t i, const char *p) { return layout.compare(i, std::char_traits<char>::length(p), p) == 0; };
  auto lower_at = [&](size_t i) { return i < layout.size() && layout[i] >= 'a' && layout[i] <= 'z'; };  // startsWithLowerCase: "Month" is not "Mon" + "th"
  size_t lit0 = 0, i = 0;
  const size_t n = layout.size();
  const size_t first = ops.size();
  auto emit = [&](uint8_t code, size_t at, size_t clen, uint8_t fd) {
    GtOp o{};
    o.code = code; o.fd = fd; o.lit_off = (uint16_t)lits.size(); o.lit_len = (uint16_t)(at - lit0);
    lits.append(layout, lit0, at - lit0);
    ops.push_back(o);
    i = at + clen; lit0 = i;
  };
  while (i < n) {
    const char c = layout[i];
    const size_t r = n - i;
    bool hit = true;
    switch (c) {
      case 'J': if (starts(i, "January")) emit(GT_LONGMONTH, i, 7, 0); else if (starts(i, "Jan") && !lower_at(i + 3)) emit(GT_MONTH, i, 3, 0); else hit = false; break;
      case 'M': if (starts(i, "Monday")) emit(GT_LONGWEEKDAY, i, 6, 0); else if (starts(i, "Mon") && !lower_at(i + 3)) emit(GT_WEEKDAY, i, 3, 0); else if (starts(i, "MST")) emit(GT_TZ, i, 3, 0); else hit = false; break;
      case '0':
        if (r >= 2 && layout[i + 1] >= '1' && layout[i + 1] <= '6') {
          static const uint8_t m[] = {0, GT_ZEROMONTH, GT_ZERODAY, GT_ZEROHOUR12, GT_ZEROMINUTE, GT_ZEROSECOND, GT_YEAR};
          emit(m[layout[i + 1] - '0'], i, 2, 0);
        } else if (r >= 3 && layout[i + 1] == '0' && layout[i + 2] == '2') emit(GT_ZEROYEARDAY, i, 3, 0);
        else hit = false;
        break;
      case '1': if (r >= 2 && layout[i + 1] == '5') emit(GT_HOUR, i, 2, 0); else emit(GT_NUMMONTH, i, 1, 0); break;
      case '2': if (starts(i, "2006")) emit(GT_LONGYEAR, i, 4, 0); else emit(GT_DAY, i, 1, 0); break;
      case '_':
        if (r >= 2 && layout[i + 1] == '2') {
          if (starts(i + 1, "2006")) { emit(GT_LONGYEAR, i + 1, 4, 0); }  // "_2006" is a literal '_' followed by the year
          else emit(GT_UNDERDAY, i, 2, 0);
        } else if (r >= 3 && layout[i + 1] == '_' && layout[i + 2] == '2') emit(GT_UNDERYEARDAY, i, 3, 0);
        else hit = false;
        break;
      case '3': emit(GT_HOUR12, i, 1, 0); break;
      case '4': emit(GT_MINUTE, i, 1, 0); break;
      case '5': emit(GT_SECOND, i, 1, 0); break;
      case 'P': if (r >= 2 && layout[i + 1] == 'M') emit(GT_PM, i, 2, 0); else hit = false; break;
      case 'p': if (r >= 2 && layout[i + 1] == 'm') emit(GT_pm, i, 2, 0); else hit = false; break;
      case '-':
        if (starts(i, "-070000") || starts(i, "-07:00:00")) hit = false;  // zone offsets with seconds: not on this path (the oracle leaves them literal too)
        else if (starts(i, "-0700")) emit(GT_NUMTZ, i, 5, 0);
        else if (starts(i, "-07:00")) emit(GT_NUMTZCOLON, i, 6, 0);
        else if (starts(i, "-07")) emit(GT_NUMTZSHORT, i, 3, 0);
        else hit = false;
        break;
      case 'Z':
        if (starts(i, "Z0700")) emit(GT_ISOTZ, i, 5, 0);
        else if (starts(i, "Z07:00")) emit(GT_ISOTZCOLON, i, 6, 0);
        else if (starts(i, "Z07")) emit(GT_ISOTZSHORT, i, 3, 0);
        else hit = false;
        break;
      case '.': case ',':
        hit = false;
        if (r >= 2 && (layout[i + 1] == '0' || layout[i + 1] == '9')) {
          const char ch = layout[i + 1];
          size_t j = 1;
          while (j < r && layout[i + j] == ch) j++;
          if (!(j < r && layout[i + j] >= '0' && layout[i + j] <= '9')) { emit(ch == '0' ? GT_FRAC0 : GT_FRAC9, i, j, (uint8_t)(j - 1)); hit = true; }
        }
        break;
      default: hit = false;
    }
    if (!hit) i++;
  }
  GtOp e{};
  e.code = GT_END; e.lit_off = (uint16_t)lits.size(); e.lit_len = (uint16_t)(n - lit0);
  lits.append(layout, lit0, n - lit0);
  ops.push_back(e);
  for (size_t k = first; k + 1 < ops.size(); k++)
    ops[k].next_frac = ((ops[k + 1].code == GT_FRAC0 || ops[k + 1].code == GT_FRAC9) && ops[k + 1].lit_len == 0) ? 1 : 0;
}

// a set of compiled layouts: layout L is ops[start[L], start[L + 1])
struct GtSet { const GtOp *ops; const uint8_t *lits; const uint16_t *start; int32_t n; };

// ---- device: the walk ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool gt_digit(uint32_t c) { return c - '0' <= 9u; }
__device__ __forceinline__ uint32_t gt_lower(uint32_t c) { return (c >= 'A' && c <= 'Z') ? c + 32 : c; }
template <class F> __device__ __forceinline__ bool gt_getnum(const F &f, uint32_t &p, uint32_t b, bool fixed, int *out) {  // time.getnum
  if (p >= b || !gt_digit(f[p])) return false;
  if (p + 1 >= b || !gt_digit(f[p + 1])) { if (fixed) return false; *out = (int)(f[p] - '0'); p++; return true; }
  *out = (int)((f[p] - '0') * 10 + (f[p + 1] - '0')); p += 2;
  return true;
}
// month / weekday names, matched without regard to case; `shortform`: the first three letters
template <class F> __device__ bool gt_lookup(const char *names, int count, int stride, bool shortform, const F &f, uint32_t &p, uint32_t b, int *out) {
  for (int i = 0; i < count; i++) {
    const char *nm = names + i * stride;
    uint32_t l = 0;
    while (nm[l]) l++;
    if (shortform) l = 3;
    if (b - p < l) continue;
    bool ok = true;
    for (uint32_t k = 0; k < l && ok; k++) ok = gt_lower(f[p + k]) == gt_lower((uint8_t)nm[k]);
    if (ok) { *out = i; p += l; return true; }
  }
  return false;
}
__device__ __forceinline__ int gt_days_in(int m, int64_t y) {
  const bool leap = (y % 4 == 0) && (y % 100 != 0 || y % 400 == 0);
  return m == 2 ? (leap ? 29 : 28) : 30 + ((0x15AA >> m) & 1);
}
__device__ __forceinline__ int64_t gt_days_from_civil(int64_t y, int m, int d) {
  y -= m <= 2;
  const int64_t era = (y >= 0 ? y : y - 399) / 400, yoe = y - era * 400;
  const int64_t doy = (153 * (m > 2 ? m - 3 : m + 9) + 2) / 5 + d - 1;
  return era * 146097 + yoe * 365 + yoe / 4 - yoe / 100 + doy - 719468;
}
// nanoseconds of the digits f[p + 1, e) (at most nine count)
template <class F> __device__ __forceinline__ int32_t gt_nanos(const F &f, uint32_t p, uint32_t e) {
  int64_t ns = 0; int digits = 0;
  for (uint32_t k = p + 1; k < e; k++) if (digits < 9) { ns = ns * 10 + (f[k] - '0'); digits++; }
  for (; digits < 9; digits++) ns *= 10;
  return (int32_t)ns;
}

static __constant__ char GT_MONTHS[12][10] = {"January", "February", "March", "April", "May", "June", "July", "August", "September", "October", "November", "December"};
static __constant__ char GT_DAYS[7][10] = {"Sunday", "Monday", "Tuesday", "Wednesday", "Thursday", "Friday", "Saturday"};

// time.skip: the literal text in front of a chunk, runs of spaces treated as equivalent (a space of the layout takes any
// run of spaces of the value — also none at the value's end)
template <class F> __device__ __forceinline__ bool gt_skip(const F &f, uint32_t &p, uint32_t b, const uint8_t *lit, uint32_t n) {
  uint32_t i = 0;
  while (i < n) {
    if (lit[i] == ' ') {
      if (p < b && f[p] != ' ') return false;
      while (i < n && lit[i] == ' ') i++;
      while (p < b && f[p] == ' ') p++;
      continue;
    }
    if (p >= b || f[p] != lit[i]) return false;
    i++; p++;
  }
  return true;
}
// time.parseSignedOffset: [+-]digits with a value of at most 23; the length taken, 0 if it is not one
template <class F> __device__ __forceinline__ uint32_t gt_signed_offset(const F &f, uint32_t p, uint32_t b) {
  if (p >= b || (f[p] != '+' && f[p] != '-')) return 0;
  uint32_t k = 1; uint64_t x = 0;
  while (p + k < b && gt_digit(f[p + k])) { if (x < (1ull << 62)) x = x * 10 + (f[p + k] - '0'); k++; }  // (leadingInt fails on overflow: still > 23 here)
  return (k > 1 && x <= 23) ? k : 0;
}

// time.Parse(layout, f[a, b)): true on success
template <class F> __device__ bool gotime_parse(const GtOp *ops, int nops, const uint8_t *lits, const F &f, uint32_t a, uint32_t b, int64_t *osec, int32_t *onsec) {
  uint32_t p = a;
  int64_t year = 0; int month = -1, day = -1, hour = 0, mi = 0, sec = 0, yday = -1, v = 0;
  int32_t nsec = 0;
  bool pm_set = false, pm = false, have_year = false, have_zoff = false;
  int zoff = 0;
  for (int oi = 0; oi < nops; oi++) {
    const GtOp op = ops[oi];
    if (!gt_skip(f, p, b, lits + op.lit_off, op.lit_len)) return false;  // the literal text in front of the chunk
    switch (op.code) {
      case GT_END: if (p != b) return false; break;  // extra text
      case GT_YEAR:
        if (b - p < 2 || !gt_digit(f[p]) || !gt_digit(f[p + 1])) return false;
        year = (f[p] - '0') * 10 + (f[p + 1] - '0'); p += 2;
        year += year >= 69 ? 1900 : 2000; have_year = true;
        break;
      case GT_LONGYEAR:
        if (b - p < 4) return false;
        for (int k = 0; k < 4; k++) if (!gt_digit(f[p + k])) return false;
        year = (f[p] - '0') * 1000 + (f[p + 1] - '0') * 100 + (f[p + 2] - '0') * 10 + (f[p + 3] - '0'); p += 4; have_year = true;
        break;
      case GT_MONTH: if (!gt_lookup(&GT_MONTHS[0][0], 12, 10, true, f, p, b, &v)) return false; month = v + 1; break;
      case GT_LONGMONTH: if (!gt_lookup(&GT_MONTHS[0][0], 12, 10, false, f, p, b, &v)) return false; month = v + 1; break;
      case GT_NUMMONTH: case GT_ZEROMONTH: if (!gt_getnum(f, p, b, op.code == GT_ZEROMONTH, &v) || v <= 0 || v > 12) return false; month = v; break;
      case GT_WEEKDAY: if (!gt_lookup(&GT_DAYS[0][0], 7, 10, true, f, p, b, &v)) return false; break;
      case GT_LONGWEEKDAY: if (!gt_lookup(&GT_DAYS[0][0], 7, 10, false, f, p, b, &v)) return false; break;
      case GT_DAY: case GT_UNDERDAY: case GT_ZERODAY:
        if (op.code == GT_UNDERDAY && p < b && f[p] == ' ') p++;
        if (!gt_getnum(f, p, b, op.code == GT_ZERODAY, &v)) return false;
        day = v;  // checked against the month once everything is known
        break;
      case GT_ZEROYEARDAY: case GT_UNDERYEARDAY: {  // getnum3: three digits, or up to three behind up to two spaces
        if (op.code == GT_UNDERYEARDAY) for (int k = 0; k < 2; k++) if (p < b && f[p] == ' ') p++;
        int nd = 0; yday = 0;
        while (nd < 3 && p + nd < b && gt_digit(f[p + nd])) { yday = yday * 10 + (int)(f[p + nd] - '0'); nd++; }
        if (nd == 0 || (op.code == GT_ZEROYEARDAY && nd != 3)) return false;
        p += nd;
        break;
      }
      case GT_HOUR: if (!gt_getnum(f, p, b, false, &v) || v < 0 || v >= 24) return false; hour = v; break;
      case GT_HOUR12: case GT_ZEROHOUR12: if (!gt_getnum(f, p, b, op.code == GT_ZEROHOUR12, &v) || v < 0 || v > 12) return false; hour = v; break;
      case GT_MINUTE: case GT_ZEROMINUTE: if (!gt_getnum(f, p, b, op.code == GT_ZEROMINUTE, &v) || v < 0 || v >= 60) return false; mi = v; break;
      case GT_SECOND: case GT_ZEROSECOND:
        if (!gt_getnum(f, p, b, op.code == GT_ZEROSECOND, &v) || v < 0 || v >= 60) return false;
        sec = v;
        // a fractional second in the input is taken even if the layout has none — unless the layout's next chunk is one
        if (b - p >= 2 && (f[p] == '.' || f[p] == ',') && gt_digit(f[p + 1]) && !op.next_frac) {
          uint32_t e = p + 1;
          while (e < b && gt_digit(f[e])) e++;
          nsec = gt_nanos(f, p, e); p = e;
        }
        break;
      case GT_PM: case GT_pm: {
        if (b - p < 2) return false;
        const uint32_t c0 = f[p], c1 = f[p + 1], up = op.code == GT_PM ? 0u : 32u;
        if (c1 != 'M' + up) return false;
        if (c0 == 'P' + up) pm = true; else if (c0 == 'A' + up) pm = false; else return false;
        pm_set = true; p += 2;
        break;
      }
      case GT_ISOTZ: case GT_ISOTZCOLON: case GT_ISOTZSHORT: case GT_NUMTZ: case GT_NUMTZCOLON: case GT_NUMTZSHORT: {
        const bool iso = op.code == GT_ISOTZ || op.code == GT_ISOTZCOLON || op.code == GT_ISOTZSHORT;
        if (iso && p < b && f[p] == 'Z') { p++; zoff = 0; have_zoff = true; break; }
        int hh, mm = 0; uint32_t sign;
        if (op.code == GT_ISOTZCOLON || op.code == GT_NUMTZCOLON) {
          if (b - p < 6 || f[p + 3] != ':' || !gt_digit(f[p + 1]) || !gt_digit(f[p + 2]) || !gt_digit(f[p + 4]) || !gt_digit(f[p + 5])) return false;
          hh = (f[p + 1] - '0') * 10 + (f[p + 2] - '0'); mm = (f[p + 4] - '0') * 10 + (f[p + 5] - '0'); sign = f[p]; p += 6;
        } else if (op.code == GT_ISOTZSHORT || op.code == GT_NUMTZSHORT) {
          if (b - p < 3 || !gt_digit(f[p + 1]) || !gt_digit(f[p + 2])) return false;
          hh = (f[p + 1] - '0') * 10 + (f[p + 2] - '0'); sign = f[p]; p += 3;
        } else {
          if (b - p < 5) return false;
          for (int k = 1; k < 5; k++) if (!gt_digit(f[p + k])) return false;
          hh = (f[p + 1] - '0') * 10 + (f[p + 2] - '0'); mm = (f[p + 3] - '0') * 10 + (f[p + 4] - '0'); sign = f[p]; p += 5;
        }
        if (hh > 24 || mm > 60) return false;
        if (sign == '+') zoff = hh * 3600 + mm * 60; else if (sign == '-') zoff = -(hh * 3600 + mm * 60); else return false;
        have_zoff = true;
        break;
      }
      case GT_TZ: {  // time.parseTimeZone.  A name never moves the instant (Go fabricates a location for it and keeps the wall time)
        if (b - p >= 3 && f[p] == 'U' && f[p + 1] == 'T' && f[p + 2] == 'C') { p += 3; break; }
        if (b - p < 3) return false;
        if (b - p >= 4 && (f[p] == 'C' || f[p] == 'M') && f[p + 2] == 'S' && f[p + 3] == 'T' && ((f[p] == 'C' && f[p + 1] == 'h') || (f[p] == 'M' && f[p + 1] == 'e'))) { p += 4; break; }  // ChST, MeST
        if (f[p] == 'G' && f[p + 1] == 'M' && f[p + 2] == 'T') { p += 3; p += gt_signed_offset(f, p, b); break; }  // GMT, GMT+3 (anything else behind it is the next chunk's business)
        if (f[p] == '+' || f[p] == '-') { const uint32_t k = gt_signed_offset(f, p, b); if (!k) return false; p += k; break; }  // zones that only have "+03"
        uint32_t j = 0;
        while (p + j < b && j < 6 && f[p + j] >= 'A' && f[p + j] <= 'Z') j++;
        if (j < 3 || j > 5) return false;
        if (j == 5 && f[p + 4] != 'T') return false;
        if (j == 4 && !(f[p + 3] == 'T' || (f[p] == 'W' && f[p + 1] == 'I' && f[p + 2] == 'T' && f[p + 3] == 'A'))) return false;
        p += j;
        break;
      }
      case GT_FRAC0: {
        const uint32_t fd = op.fd;
        if (b - p < fd + 1 || (f[p] != '.' && f[p] != ',')) return false;
        for (uint32_t k = 1; k <= fd; k++) if (!gt_digit(f[p + k])) return false;
        if (b - p > fd + 1 && gt_digit(f[p + fd + 1])) return false;
        nsec = gt_nanos(f, p, p + fd + 1); p += fd + 1;
        break;
      }
      case GT_FRAC9: {
        if (b - p < 2 || (f[p] != '.' && f[p] != ',') || !gt_digit(f[p + 1])) break;  // the fraction is optional
        uint32_t e = p + 1;
        while (e < b && gt_digit(f[e])) e++;
        nsec = gt_nanos(f, p, e); p = e;
        break;
      }
      default: return false;
    }
  }
  if (pm_set) { if (pm && hour < 12) hour += 12; else if (!pm && hour == 12) hour = 0; }
  if (!have_year) year = 0;
  if (yday >= 0) {  // the day of the year decides month and day; what the value also gave must agree
    int m = 0, d = 0;
    if (gt_days_in(2, year) == 29) { if (yday == 31 + 29) { m = 2; d = 29; } else if (yday > 31 + 29) yday--; }
    if (yday < 1 || yday > 365) return false;
    if (m == 0) { m = 1; d = yday; while (d > gt_days_in(m, 1)) { d -= gt_days_in(m, 1); m++; } }  // (a common year's months: the leap day is out already)
    if (month >= 0 && month != m) return false;
    month = m;
    if (day >= 0 && day != d) return false;
    day = d;
  } else {
    if (month < 0) month = 1;
    if (day < 0) day = 1;
  }
  if (day < 1 || day > gt_days_in(month, year)) return false;
  int64_t t = gt_days_from_civil(year, month, day) * 86400 + hour * 3600 + mi * 60 + sec;
  if (have_zoff) t -= zoff;
  *osec = t; *onsec = nsec;
  return true;
}
// the first layout of the set that takes the text
template <class F> __device__ bool gotime_parse_any(const GtSet &s, const F &f, uint32_t a, uint32_t b, int64_t *sec, int32_t *nsec) {
  for (int l = 0; l < s.n; l++)
    if (gotime_parse(s.ops + s.start[l], (int)(s.start[l + 1] - s.start[l]), s.lits, f, a, b, sec, nsec)) return true;
  return false;
}

}  // namespace tf
