// tf_csv.hip — CSV ingest on device: pkg/csv.Reader (reader.go:89-324) +
// s3 CSVReader.doParse (reader_csv.go:233-341, 345-452) + strictify.Strictify
// (strictify.go:18-181), producing typed Arrow-style columns in HBM without
// ever materialising ChangeItems.
//
// Pipeline (all on the library stream):
//   1. csv_count_newlines   : '\n' per 16 KiB tile                (reads B_csv)
//   2. scan + csv_line_index: row_start[] of every complete line  (reads B_csv)
//   3. csv_parse_rows       : one lane per row, wave-uniform loop over field
//                             index; splitString state machine over an 8-byte
//                             register window, typed parse of the fields the
//                             schema maps, coalesced column-major stores;
//                             string columns record (start,len)   (reads B_csv, writes B_fixed)
//   4. segmented scan of string lengths → Arrow offsets
//   5. csv_copy_strings     : unquote / ""-collapse copy           (reads ~B_str, writes B_str)
// Rows that fail (missing cell, cast, range, quote errors) are compacted out
// and reported as tfgpu_row_error, mirroring how parseCSVRows drops them.
//
// HBM-bound byte kernel: algorithmic traffic B_csv + B_bin per row.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <type_traits>

#include "tf_devfmt.hpp"
#include "tf_devparse.hpp"
#include "tf_devfloat.hpp"
#include "tf_segcopy.hpp"
#include "tf_textview.hpp"
#include "tf_wave.hpp"
#include "tf_swar.hpp"
#include "tf_gotime.hpp"
#include "tf_f64range.hpp"
#include "tf_plan.hpp"

// TFGPU_CSV_ABLATE=n (leave a kernel after phase n; results are NOT valid) is a profiling build only: the product's kernels carry none of
// its branches (`if (CSV_ABL(t) == 31)` sat in every integer cell of the one kernel that is instruction-bound).
// tools/build_variant.sh ablate tf_csv.hip -DTF_CSV_ABLATE_BUILD=1 builds the variant tools/ablate_csv.py and gpu_visit.sh ablate load.
#ifdef TF_CSV_ABLATE_BUILD
#define CSV_ABL(x) ((x).ablate)
#else
#define CSV_ABL(x) 0
#endif
namespace tf {

std::unique_ptr<tfgpu_dbatch> compact_rows(const tfgpu_dbatch &in, Buf keep);  // tf_transform.hip

// ---------------------------------------------------------------------------
// line index
// ---------------------------------------------------------------------------
static constexpr int NL_THREADS = 256;
#ifndef TF_NL_ITERS
#define TF_NL_ITERS 7
#endif
static constexpr int NL_ITERS = TF_NL_ITERS;  // (TF_NL_ITERS / TF_CT_THREADS / TF_CR_FCAP: tile-size A/B builds, tools/build_variant.sh)
static constexpr int NL_TILE = NL_THREADS * 16 * NL_ITERS;  // 28 KiB per workgroup
static constexpr int CSV_GRAN = 64 * 16;                  // 1 KiB (what one wave reads per iteration): the unit newline counts are kept in
static constexpr int NL_GPT = NL_TILE / CSV_GRAN;           // granules per workgroup of the line-index kernels
static_assert(NL_TILE == NL_GPT * CSV_GRAN && NL_THREADS * 16 == 4 * CSV_GRAN, "a wave iteration is a granule");

__device__ __forceinline__ uint32_t nl_mask16(uint4 v) {
  // bit i set iff byte i of the 16-byte chunk is '\n'
  uint32_t m = 0;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t x = w[k] ^ 0x0A0A0A0Au;
    uint32_t b0 = ((x & 0xFFu) == 0), b1 = ((x & 0xFF00u) == 0), b2 = ((x & 0xFF0000u) == 0), b3 = ((x & 0xFF000000u) == 0);
    m |= (b0 | b1 << 1 | b2 << 2 | b3 << 3) << (4 * k);
  }
  return m;
}
// number of '\n' among the 16 bytes: exact zero-byte flags by SWAR (no borrow between bytes), summed with one v_dot4 per word
__device__ __forceinline__ uint32_t nl_count16(uint4 v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t x = w[k] ^ 0x0A0A0A0Au;
    const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;  // 0x80 where the byte is '\n'
    acc = __builtin_amdgcn_udot4(z >> 7, 0x01010101u, acc, false);
  }
  return acc;
}

__device__ __forceinline__ uint4 load16_guard(const uint8_t *base, uint64_t pos, uint64_t len) {
  // buffers handed to the kernels are padded to a multiple of 16 bytes past `len`
  uint4 v = *reinterpret_cast<const uint4 *>(base + pos);
  if (pos + 16 > len) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
    for (int i = 0; i < 16; i++) if (pos + i >= len) w[i >> 2] &= ~(0xFFu << (8 * (i & 3)));
    v = make_uint4(w[0], w[1], w[2], w[3]);
  }
  return v;
}

// '\n' per 1 KiB granule (CSV_GRAN = what one wave reads per iteration: 64 lanes x 16 B); tiles of any whole number of
// granules take their first line and their line count from the exclusive scan of these
__global__ void __launch_bounds__(NL_THREADS) csv_count_newlines(const uint8_t *__restrict__ data, uint64_t len, uint32_t *__restrict__ gran_counts, int64_t ngran) {
  const uint64_t tile = (uint64_t)blockIdx.x * NL_TILE;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int it = 0; it < NL_ITERS; it++) {
    const uint64_t pos = tile + ((uint64_t)it * NL_THREADS + threadIdx.x) * 16;
    uint32_t cnt = 0;
    if (pos < len) cnt = nl_count16(load16_guard(data, pos, len));
    const uint32_t inc = wave_scan_add(cnt);
    const int64_t g = (int64_t)blockIdx.x * NL_GPT + it * 4 + wv;
    if (lane == 63 && g < ngran) gran_counts[g] = inc;
  }
}

__global__ void __launch_bounds__(NL_THREADS) csv_line_index(const uint8_t *__restrict__ data, uint64_t len, const uint32_t *__restrict__ tile_base,
                                                            uint32_t *__restrict__ row_start /* [nlines+1], row_start[0] preset */) {
  __shared__ uint32_t wsum[NL_THREADS / 64];
  uint64_t tile = (uint64_t)blockIdx.x * NL_TILE;
  uint32_t base = tile_base[(int64_t)blockIdx.x * NL_GPT];  // exclusive scan of the granule counts
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int it = 0; it < NL_ITERS; it++) {
    uint64_t pos = tile + ((uint64_t)it * NL_THREADS + threadIdx.x) * 16;
    uint32_t m = pos < len ? nl_mask16(load16_guard(data, pos, len)) : 0;
    uint32_t c = __popc(m), inc = c;
    for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
    for (int i = 0; i < NL_THREADS / 64; i++) { uint32_t s = wsum[i]; if (i < wv) wbase += s; tot += s; }
    __syncthreads();
    uint32_t k = base + wbase + inc - c;
    while (m) { int b = __ffs(m) - 1; m &= m - 1; row_start[++k] = (uint32_t)(pos + b + 1); }
    base += tot;
  }
}


// unicode.IsSpace over UTF-8 at s[i..): width of the space rune or 0
template <class F> __device__ __forceinline__ uint32_t space_prefix(const F &f, uint32_t i, uint32_t end) {
  uint32_t c = f[i];
  if (c == ' ' || (c >= 9 && c <= 13)) return 1;
  if (c < 0xC2) return 0;
  uint32_t r = end - i;
  if (c == 0xC2 && r >= 2) { uint32_t d = f[i + 1]; return (d == 0x85 || d == 0xA0) ? 2 : 0; }
  if (r >= 3 && (c == 0xE1 || c == 0xE2 || c == 0xE3)) {
    uint32_t d = f[i + 1], e = f[i + 2];
    if (c == 0xE1) return (d == 0x9A && e == 0x80) ? 3 : 0;
    if (c == 0xE3) return (d == 0x80 && e == 0x80) ? 3 : 0;
    if (d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) return 3;
    if (d == 0x81 && e == 0x9F) return 3;
  }
  return 0;
}
template <class F> __device__ __forceinline__ uint32_t space_suffix(const F &f, uint32_t a, uint32_t b) {
  uint32_t c = f[b - 1];
  if (c == ' ' || (c >= 9 && c <= 13)) return 1;
  if (c < 0x80) return 0;
  if (b - a >= 2 && f[b - 2] == 0xC2 && (c == 0x85 || c == 0xA0)) return 2;
  if (b - a >= 3 && space_prefix(f, b - 3, b) == 3) return 3;
  return 0;
}

struct CsvOpts {
  uint8_t delim, quote, escape, double_quote, include_missing, strings_can_be_null, quoted_strings_can_be_null, pad;
  int32_t n_null, n_true, n_false;
  const uint32_t *list_off;  // offsets of null|true|false values, concatenated, into list_data
  const uint8_t *list_data;
  GtSet user_tp;             // AdditionalReaderOptions.TimestampParsers (Go layouts), compiled
  GtSet cast_tp;             // spf13/cast StringToDate's layout list, compiled
  const uint8_t *dp;         // AdditionalReaderOptions.DecimalPoint
  uint32_t dp_len;
  uint32_t multiline;        // NewlinesInValue: a row may span physical lines
  const uint64_t *p128;      // Eisel-Lemire's 128-bit powers of ten when the schema holds a float32 column, else null
};

// sanitizeElement reader.go:273-324: TrimSpace, unquote, count ""-pairs.
// Returns tfgpu_rowerr (0 ok).  Content = [a, b) within the field; npairs = doubled quotes inside.
// any quote-character bit set in tile positions [lo, hi)?
__device__ __forceinline__ bool any_quote(const uint32_t *qmask, uint32_t lo, uint32_t hi) {
  if (lo >= hi) return false;
  const uint32_t w0 = lo >> 5, w1 = (hi - 1) >> 5;
  for (uint32_t w = w0; w <= w1; w++) {
    uint32_t m = qmask[w];
    if (w == w0) m &= ~0u << (lo & 31);
    if (w == w1) m &= ~0u >> (31 - ((hi - 1) & 31));
    if (m) return true;
  }
  return false;
}

// `qmask` (tile path) = bitmap of quote characters of the staged tile, `qbase` = tile position of f[0]:
// a field without inner quote characters cannot hold doubled quotes, so the byte loop is skipped.
template <class F> __device__ __forceinline__ int sanitize(const CsvOpts &o, const F &f, uint32_t &a, uint32_t &b, uint32_t &npairs,
                                                           const uint32_t *qmask = nullptr, uint32_t qbase = 0) {
  a = 0; b = f.n; npairs = 0;
  uint32_t k;
  while (a < b && (k = space_prefix(f, a, b)) > 0) a += k;
  while (a < b && (k = space_suffix(f, a, b)) > 0) b -= k;
  if (o.quote == 0) return 0;
  if (b - a >= 1) {
    if (b - a == 1 && f[a] == o.quote) return TFGPU_ROW_QUOTE;
    if (f[a] == o.quote && f[b - 1] == o.quote) { a++; b--; }
  }
  if (qmask && !any_quote(qmask, qbase + a, qbase + b)) return 0;
  for (uint32_t i = a; i + 1 < b;) {
    if (f[i] == o.quote && f[i + 1] == o.quote) { npairs++; i += 2; } else i++;
  }
  if (!o.double_quote && npairs) return TFGPU_ROW_DOUBLE_QUOTE;
  return 0;
}

// ---------------------------------------------------------------------------
// typed parsers over a Field range [a,b)
// ---------------------------------------------------------------------------
// spf13/cast trimZeroDecimal: "12.00" → "12"
template <class F> __device__ __forceinline__ uint32_t trim_zero_decimal(const F &f, uint32_t a, uint32_t b) {
  bool found_zero = false;
  for (uint32_t i = b; i > a; i--) {
    uint32_t c = f[i - 1];
    if (c == '.') { if (found_zero) return i - 1; }
    else if (c == '0') found_zero = true;
    else return b;
  }
  return b;
}

template <class F> __device__ __forceinline__ bool field_equals(const F &f, uint32_t a, uint32_t b, const uint8_t *s, uint32_t n) {
  if (b - a != n) return false;
  for (uint32_t i = 0; i < n; i++) if (f[a + i] != s[i]) return false;
  return true;
}
template <class F> __device__ __forceinline__ bool in_list(const CsvOpts &o, int first, int count, const F &f, uint32_t a, uint32_t b) {
  for (int i = 0; i < count; i++) {
    uint32_t s = o.list_off[first + i], e = o.list_off[first + i + 1];
    if (field_equals(f, a, b, o.list_data + s, e - s)) return true;
  }
  return false;
}

// cast.StringToDate (spf13/cast v1.7.1 timeFormats) for the fixed numeric shapes:
//   2006-01-02 | 2006-01-02T15:04:05[.frac][Z07:00] | 2006-01-02 15:04:05[.frac]
// returns 0 ok, TFGPU_ROW_CAST if the shape is one of these but the value is
// invalid, TFGPU_ROW_HOST_FALLBACK if the text has another shape.
template <class F> __device__ int parse_datetime(const F &f, uint32_t a, uint32_t b, int64_t *sec, int32_t *nsec) {
  uint32_t n = b - a;
  if (n < 10) return TFGPU_ROW_HOST_FALLBACK;
  for (int i = 0; i < 10; i++) {
    uint32_t c = f[a + i];
    if (i == 4 || i == 7) { if (c != '-') return TFGPU_ROW_HOST_FALLBACK; } else if (!dg(c)) return TFGPU_ROW_HOST_FALLBACK;
  }
  int64_t y = (f[a] - '0') * 1000 + (f[a + 1] - '0') * 100 + (f[a + 2] - '0') * 10 + (f[a + 3] - '0');
  int mo = (f[a + 5] - '0') * 10 + (f[a + 6] - '0'), d = (f[a + 8] - '0') * 10 + (f[a + 9] - '0');
  int h = 0, mi = 0, se = 0; int64_t ns = 0; int off = 0;
  if (n > 10) {
    uint32_t sep = f[a + 10];
    if (sep != 'T' && sep != ' ') return TFGPU_ROW_HOST_FALLBACK;
    uint32_t k = a + 11;
    // stdHour "15" takes one or two digits; minutes/seconds are fixed two digits
    if (k >= b || !dg(f[k])) return TFGPU_ROW_HOST_FALLBACK;
    h = f[k] - '0'; k++;
    if (k < b && dg(f[k])) { h = h * 10 + (f[k] - '0'); k++; }
    if (k + 6 > b || f[k] != ':' || !dg(f[k + 1]) || !dg(f[k + 2]) || f[k + 3] != ':' || !dg(f[k + 4]) || !dg(f[k + 5])) return TFGPU_ROW_HOST_FALLBACK;
    mi = (f[k + 1] - '0') * 10 + (f[k + 2] - '0'); se = (f[k + 4] - '0') * 10 + (f[k + 5] - '0');
    k += 6;
    if (k + 1 < b && (f[k] == '.' || f[k] == ',') && dg(f[k + 1])) {
      k++; int nd = 0;
      while (k < b && dg(f[k])) { if (nd < 9) { ns = ns * 10 + (f[k] - '0'); nd++; } k++; }
      while (nd < 9) { ns *= 10; nd++; }
    }
    if (k < b) {
      if (sep != 'T') return TFGPU_ROW_HOST_FALLBACK;  // "… 15:04:05 -0700", "…Z07:00" after a space: host path
      if (f[k] == 'Z' && k + 1 == b) k++;
      else if ((f[k] == '+' || f[k] == '-') && k + 6 == b && dg(f[k + 1]) && dg(f[k + 2]) && f[k + 3] == ':' && dg(f[k + 4]) && dg(f[k + 5])) {
        int hh = (f[k + 1] - '0') * 10 + (f[k + 2] - '0'), mm = (f[k + 4] - '0') * 10 + (f[k + 5] - '0');
        if (hh > 24 || mm > 60) return TFGPU_ROW_CAST;
        off = (f[k] == '-' ? -1 : 1) * (hh * 3600 + mm * 60);
        k += 6;
      } else return TFGPU_ROW_HOST_FALLBACK;
    }
  }
  if (mo < 1 || mo > 12 || d < 1 || d > dev::days_in_month(mo, y) || h > 23 || mi > 59 || se > 59) return TFGPU_ROW_CAST;
  *sec = dev::days_from_civil(y, mo, d) * 86400 + h * 3600 + mi * 60 + se - off;
  *nsec = (int32_t)ns;
  return 0;
}

// castx.ToJSONNumberE acceptance (fastfloat.Parse grammar or ParseInt base 10)
template <class F> __device__ bool json_number_ok(const F &f, uint32_t a, uint32_t b) {
  if (a >= b) return false;
  uint32_t p = a;
  if (f[p] == '-' || f[p] == '+') p++;
  uint32_t d0 = p; while (p < b && dg(f[p])) p++;
  uint32_t nd = p - d0; bool ok = true;
  if (p < b && f[p] == '.') { p++; uint32_t f0 = p; while (p < b && dg(f[p])) p++; if (p == f0) nd = 0; else nd += p - f0; }
  if (nd > 0 && p < b && (f[p] == 'e' || f[p] == 'E')) { p++; if (p < b && (f[p] == '-' || f[p] == '+')) p++; uint32_t x0 = p; while (p < b && dg(f[p])) p++; if (p == x0) ok = false; }
  if (ok && nd > 0 && p == b) return true;
  // inf / infinity / nan, case-insensitive
  uint32_t q = a; if (f[q] == '-' || f[q] == '+') q++;
  uint32_t n = b - q;
  auto ci = [&](const char *s, uint32_t sl) { if (n != sl) return false; for (uint32_t i = 0; i < sl; i++) if (lower_(f[q + i]) != (uint32_t)s[i]) return false; return true; };
  return ci("inf", 3) || ci("infinity", 8) || ci("nan", 3);
}

// ---------------------------------------------------------------------------
// cell parse: constructCI / getCorrespondingValue / Strictify for ONE mapped column
// ---------------------------------------------------------------------------
enum CsvKind : int32_t { CK_INT, CK_UINT, CK_STR, CK_JSONNUM, CK_DATE, CK_TIMESTAMP, CK_BOOL, CK_F32, CK_DEFAULT, CK_INTERVAL, CK_SYS /* __file_name / __row_index: filled after the parse */ };

struct CsvCol {
  int32_t field;      // CSV field index (ColSchema.Path); <0 → DefaultValue
  int32_t kind;       // CsvKind
  int32_t width;      // bytes of the fixed-width output element
  int32_t next;       // next column reading the same field, or -1
  int64_t lo; uint64_t hi;  // strictify range limits
  void *values;       // fixed-width output
  int32_t *nanos;
  uint32_t *lens;     // string columns: lens[r] = content length (Arrow offsets after the scan)
  uint32_t *fstart;   // string columns: absolute start of the content | bit31 = has doubled quotes; fstart[-1] = the column has such cells (or DefaultValue text)
  uint32_t *patch;    // float64 columns under DecimalPoint: offset in the cell where the decimal point string becomes '.', or PATCH_NONE
};
static constexpr uint32_t PATCH_NONE = 0xFFFFFFFFu;

// Compact column descriptor of the tile path, staged in LDS in (kind, width) order.
struct TCol {
  void *p0;        // fixed-width: values        | text: lens
  void *p1;        // time: nanos                | text: fstart
  int16_t field;   // CSV field index, <0 = DefaultValue
  uint8_t kind;    // CsvKind
  uint8_t width;
  int32_t ci;      // index in the schema (error reporting, precedence)
};
static_assert(sizeof(TCol) == 24, "TCol is staged as three 8-byte words");

struct CsvParams {
  const uint8_t *data;
  uint64_t len;
  const uint32_t *row_start;  // slow path: already shifted by skip_rows
  const uint32_t *row_end;    // NewlinesInValue: one past the '\n' that completes row r (null: row_start[r + 1])
  int64_t nrows;
  CsvOpts o;
  const CsvCol *cols;
  int32_t ncols;
  const int32_t *field_first;  // first column index per CSV field, -1 = unused
  int32_t nfields_used;        // max mapped field index + 1
  int32_t has_unmapped;        // some field index < nfields_used feeds no column
  const TCol *tcols;           // columns sorted by (kind, width): neighbouring lanes run one code path
  unsigned long long *dbg_phase;  // TFGPU_CSV_PHASES=1 (profiling only): shader cycles per phase of csv_parse_regular, summed over its workgroups
  int32_t null_checks;         // strings_can_be_null || quoted_strings_can_be_null
  uint8_t *err;                // per row tfgpu_rowerr
  int32_t *err_col;
  uint32_t *nerr;
  // tile path
  const uint32_t *gran_pre;    // exclusive scan of the newlines per CSV_GRAN-byte granule, [ngran + 1]
  int64_t ngran;
  uint32_t tile_bytes;         // csv_parse_regular's tile: whole granules, picked per chunk so that a tile holds just under 64 lines
  int64_t ntiles;              // … and how many there are
  int64_t skip_rows;
  uint32_t *slow_n;            // rows the tile path hands to the per-row path
  uint32_t *slow_row;          //   line index (before skip_rows)
  uint32_t *slow_end;          //   position of the terminating '\n'
  uint32_t *last_end;          // max over lines of (position of '\n') + 1
  int32_t ablate;              // TFGPU_CSV_ABLATE=n (profiling only): leave the kernel after phase n; results are NOT valid
  uint32_t *gen_n;             // pieces csv_parse_regular hands to the general tile kernel
  uint32_t *gen_tile;          //   (first granule, granules <= CT_T / CSV_GRAN) pairs
  int32_t force_general;       // TFGPU_CSV_GENERAL=1: every tile takes the general kernel (parity cross-check of the two)
  const struct CsvRun *runs;   // tcols grouped into runs of one (kind, width)
  int32_t nruns;
  // column-lane cell phase (csv_parse_regular): runs cut into tasks of at most 64 columns, each given to one wave
  const struct CsvTask *tasks; // sorted by wave
  const int32_t *wave_task;    // [9]: tasks of wave w are [wave_task[w], wave_task[w + 1])
  int32_t col_lanes;           // 1: the cells run column-lane; 0: the item form (slots of 64 (column, line) items)
  // csv_parse_lanes (lane = line): the column buffers as 4-byte offsets into one arena, dealt to the 16 waves in blocks
  uint8_t *arena;
  const uint32_t *lcols;       // [16 waves][64]: word 4 s + w of wave v = word w of its s-th column {p0 offset, p1 offset, field:16 | kind:8 | width:8, -}
  const uint32_t *lwave_ncols; // [16]: columns per wave
  uint32_t tile_cpt;           // 16-byte chunks per thread: the tile is 15, 30 or 45 KiB
  // single-pass form (csv_parse_regular counts its own lines): [0] ticket counter, [1] overflow flag, [2 + t] state of tile t,
  // [2 + ntiles + t] one past the last '\n' tile t owns (0: none)
  uint32_t *spec;              // null: the line counts come from csv_count_newlines (gran_pre)
  uint32_t cap_lines;          // lines the column buffers were sized for
  // sized-ahead form (two passes, no read-back between them): the host sized the buffers for cap_lines before csv_count_newlines
  // had finished; a tile whose lines end beyond that sets this word and writes nothing, and the host parses the chunk again
  uint32_t *ovf;               // null: the buffers were sized from the true count
};

// ---- one pass instead of two: the tile kernel finds its first line's index itself ----------------------------------------
// csv_count_newlines reads the whole chunk only to tell every tile how many lines precede it.  With `spec` set, tiles take
// tickets in arrival order, each publishes the number of '\n' it owns as soon as it has classified its bytes, and one wave
// looks back over its predecessors' words (Merrill & Garland's decoupled look-back: a word is empty, a tile's own count, or
// the inclusive count up to and including the tile).  Tickets, not blockIdx: the tile with the lowest unfinished ticket is
// always resident, so the wait cannot deadlock whatever order the dispatcher picks.  The host sizes the column buffers from
// the previous chunk's bytes per line; a chunk with more lines than that sets the overflow word and is parsed again the
// two-pass way.
static constexpr uint32_t SP_AGG = 1u << 30, SP_INC = 2u << 30, SP_VAL = (1u << 30) - 1u;
// all 64 lanes of ONE wave; returns the number of lines in front of tile `t` and leaves the tile's inclusive count published
__device__ __forceinline__ uint32_t spec_lookback(uint32_t *state, int64_t t, uint32_t own, int lane) {
  if (t == 0) { if (lane == 0) __atomic_store_n(&state[0], SP_INC | own, __ATOMIC_RELAXED); return 0u; }
  if (lane == 0) __atomic_store_n(&state[t], SP_AGG | own, __ATOMIC_RELAXED);
  uint32_t excl = 0;
  int64_t base = t - 1;
  for (;;) {
    const int64_t idx = base - lane;
    const uint32_t w = idx >= 0 ? __atomic_load_n(&state[idx], __ATOMIC_RELAXED) : SP_INC;
    const uint64_t inc = __ballot((w >> 30) == 2u), empty = __ballot((w >> 30) == 0u);
    const int fi = inc ? __ffsll((long long)inc) - 1 : 63;     // the nearest predecessor that already knows its inclusive count
    const uint64_t need = fi == 63 ? ~0ull : ((2ull << fi) - 1ull);
    if (empty & need) { __builtin_amdgcn_s_sleep(1); continue; }  // someone in between has not classified its bytes yet
    uint32_t v = lane <= fi ? (w & SP_VAL) : 0u;
    v = wave_scan_add(v);
    excl += (uint32_t)__shfl((int)v, 63, 64);
    if (inc) break;
    base -= 64;
  }
  if (lane == 0) __atomic_store_n(&state[t], SP_INC | ((excl + own) & SP_VAL), __ATOMIC_RELAXED);
  return excl;
}

__device__ __forceinline__ void store_int(const CsvCol &c, int64_t r, int64_t v) {
  switch (c.width) {
    case 1: ((int8_t *)c.values)[r] = (int8_t)v; break;
    case 2: ((int16_t *)c.values)[r] = (int16_t)v; break;
    case 4: ((int32_t *)c.values)[r] = (int32_t)v; break;
    default: ((int64_t *)c.values)[r] = v;
  }
}

// DefaultValue(col) (pkg/abstract/change_item_builders.go:88-109) after Strictify
__device__ __forceinline__ void store_default(const CsvCol &c, int64_t r) {
  switch (c.kind) {
    case CK_SYS: break;
    case CK_STR: case CK_JSONNUM: c.lens[r] = (c.kind == CK_JSONNUM) ? 1u : 0u; c.fstart[r] = 0x7FFFFFFFu; if (c.kind == CK_JSONNUM) c.fstart[-1] = 1u; break;  // "" / json "0" (a cell that is no byte range: flag word of the column)
    case CK_DATE: case CK_TIMESTAMP: ((int64_t *)c.values)[r] = 0; if (c.nanos) c.nanos[r] = 0; break;
    case CK_BOOL: ((uint8_t *)c.values)[r] = 0; break;
    case CK_F32: ((float *)c.values)[r] = 0.f; break;
    default: store_int(c, r, 0);
  }
}

template <class F>
__device__ __forceinline__ bool cell_is_null(const CsvOpts &o, const F &fv, uint32_t a, uint32_t b, uint32_t npairs) {
  // parseNullValues (reader_csv.go:384-401)
  if (o.quoted_strings_can_be_null) {
    uint32_t ta = a, tb = b;
    if (tb > ta && ((fv[ta] == '"' && fv[tb - 1] == '"') || (fv[ta] == '\'' && fv[tb - 1] == '\''))) { ta++; tb = tb > ta ? tb - 1 : ta; }
    return npairs == 0 && in_list(o, 0, o.n_null, fv, ta, tb);
  }
  if (o.strings_can_be_null) return npairs == 0 && in_list(o, 0, o.n_null, fv, a, b);
  return false;
}

// strings.Replace(text, DecimalPoint, ".", 1) as a view: the first `k` bytes, a '.', the bytes behind the decimal point string
template <class F> struct Patched {
  const F &f; uint32_t a, k, skip;
  __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return i < k ? f[a + i] : i == k ? (uint32_t)'.' : f[a + i + skip - 1]; }
};
// strconv.ParseFloat(t[0, n), 64): 0 = err == nil, 1 = an error (syntax or range), 2 = a form left to the host (hexadecimal
// mantissa, '_' digit separators).  Decimal grammar of readFloat: [+-] digits [. digits] [e [+-] digits], at least one
// mantissa digit; "inf" / "infinity" (signed) and "nan" (unsigned), any case.
template <class T> __device__ int parsefloat_err(const T &t, uint32_t n) {
  if (n == 0) return 1;
  uint32_t p = 0;
  if (t[0] == '+' || t[0] == '-') p = 1;
  const uint32_t w = n - p;
  auto word = [&](const char *s, uint32_t l) { if (w != l) return false; for (uint32_t i = 0; i < l; i++) if (lower_(t[p + i]) != (uint32_t)s[i]) return false; return true; };
  if (word("inf", 3) || word("infinity", 8)) return 0;
  if (p == 0 && word("nan", 3)) return 0;
  for (uint32_t i = p; i < n; i++) if (t[i] == '_') return 2;
  if (w >= 2 && t[p] == '0' && lower_(t[p + 1]) == 'x') return 2;
  uint32_t nd = 0;
  while (p < n && dg(t[p])) { nd++; p++; }
  if (p < n && t[p] == '.') { p++; while (p < n && dg(t[p])) { nd++; p++; } }
  if (!nd) return 1;
  if (p < n && lower_(t[p]) == 'e') {
    p++;
    if (p < n && (t[p] == '+' || t[p] == '-')) p++;
    if (p >= n || !dg(t[p])) return 1;
    while (p < n && dg(t[p])) p++;
  }
  if (p != n) return 1;
  return jsonnum_overflow(t, n) ? 1 : 0;  // ±Inf: ErrRange
}

// One sanitized field [a,b) of `fv` (npairs doubled quotes inside) → column c, row r.
// `abs_start` is the absolute byte offset of fv[0] in the CSV buffer.  Returns tfgpu_rowerr.
template <class F>
__device__ int parse_cell(const CsvOpts &o, const CsvCol &c, int64_t r, const F &fv, uint32_t a, uint32_t b, uint32_t npairs, uint64_t abs_start) {
  switch (c.kind) {
    case CK_INT: case CK_UINT: {
      // parseNullValues then cast.ToIntNE(string) + range check
      if (cell_is_null(o, fv, a, b, npairs)) { store_default(c, r); return 0; }
      if (npairs) return TFGPU_ROW_CAST;  // a '"' survives in the text: not a number
      uint32_t tb = trim_zero_decimal(fv, a, b);
      if (c.kind == CK_INT || c.hi != ~0ull) {
        int64_t v; int rc = parse_int64(fv, a, tb, true, &v);
        if (rc) return TFGPU_ROW_CAST;
        if (c.kind == CK_UINT) {
          if (v < 0) return TFGPU_ROW_CAST;  // errNegativeNotAllowed
          if ((uint64_t)v > c.hi) return TFGPU_ROW_RANGE;
        } else if (v < c.lo || v > (int64_t)c.hi) return TFGPU_ROW_RANGE;
        store_int(c, r, v);
      } else {  // uint64: cast.ToUint64E parses with ParseUint
        uint64_t v; int rc = parse_uint64(fv, a, tb, true, &v);
        if (rc) return TFGPU_ROW_CAST;
        ((uint64_t *)c.values)[r] = v;
      }
      return 0;
    }
    case CK_STR: case CK_JSONNUM: {
      if (c.kind == CK_STR && cell_is_null(o, fv, a, b, npairs)) { c.lens[r] = 0; c.fstart[r] = 0x7FFFFFFFu; return 0; }
      if (c.kind == CK_JSONNUM && o.dp_len && !npairs && b - a >= o.dp_len) {
        // parseFloatValue (reader_csv.go:363-378): the first DecimalPoint becomes '.', kept only if ParseFloat takes the result
        uint32_t k = PATCH_NONE;
        for (uint32_t i = a; i + o.dp_len <= b && k == PATCH_NONE; i++) {
          bool eq = true;
          for (uint32_t j = 0; j < o.dp_len && eq; j++) eq = fv[i + j] == o.dp[j];
          if (eq) k = i - a;
        }
        if (k != PATCH_NONE) {
          const Patched<F> pt{fv, a, k, o.dp_len};
          const uint32_t pn = (b - a) - o.dp_len + 1;
          const int pe = parsefloat_err(pt, pn);
          if (pe == 2) return TFGPU_ROW_HOST_FALLBACK;
          if (pe == 0) {
            struct View { const Patched<F> &t; __device__ uint32_t operator[](uint32_t i) const { return t[i]; } } v{pt};
            if (!json_number_ok(v, 0, pn)) return TFGPU_ROW_CAST;
            c.lens[r] = pn; c.fstart[r] = (uint32_t)(abs_start + a); c.patch[r] = k;
            return 0;
          }
        }
      }
      if (c.kind == CK_JSONNUM && (npairs || !json_number_ok(fv, a, b))) return TFGPU_ROW_CAST;
      c.lens[r] = (b - a) - npairs;
      c.fstart[r] = (uint32_t)(abs_start + a) | (npairs ? 0x80000000u : 0u);
      if (npairs) c.fstart[-1] = 1u;
      return 0;
    }
    case CK_DATE: case CK_TIMESTAMP: {
      if (npairs) return TFGPU_ROW_CAST;
      int64_t sec = 0; int32_t ns = 0; int rc;
      if (c.kind == CK_TIMESTAMP && parse_int64(fv, a, b, false, &sec) == 0) rc = 0;  // parseTimestampValue :419-426
      else if (c.kind == CK_DATE && o.user_tp.n && gotime_parse_any(o.user_tp, fv, a, b, &sec, &ns)) rc = 0;  // parseDateValue :405-415
      else {
        // still a string: cast.ToTimeE → StringToDate, the first of its layouts that parses (strictify.go:118-143)
        rc = parse_datetime(fv, a, b, &sec, &ns);
        if (rc) { ns = 0; rc = gotime_parse_any(o.cast_tp, fv, a, b, &sec, &ns) ? 0 : TFGPU_ROW_CAST; }
      }
      if (rc) return rc;
      ((int64_t *)c.values)[r] = sec;
      c.nanos[r] = ns;
      return 0;
    }
    case CK_BOOL: {  // parseBooleanValue :431-452 then cast.ToBoolE
      int v = 0;
      if (npairs) return TFGPU_ROW_CAST;
      if (o.strings_can_be_null && in_list(o, 0, o.n_null, fv, a, b)) v = 0;
      else if (in_list(o, o.n_null, o.n_true, fv, a, b)) v = 1;
      else if (in_list(o, o.n_null + o.n_true, o.n_false, fv, a, b)) v = 0;
      else if (parse_bool(fv, a, b, &v)) return TFGPU_ROW_CAST;
      ((uint8_t *)c.values)[r] = (uint8_t)v;
      return 0;
    }
    case CK_F32: {  // parseFloatValue :363-378 (DecimalPoint), then strictify: cast.ToFloat32E(string) = strconv.ParseFloat(s, 32), any error fails the row
      if (npairs) return TFGPU_ROW_CAST;  // a '"' survives in the text: not a number
      float v = 0; int rc = -1;
      if (o.dp_len && b - a >= o.dp_len) {
        uint32_t k = PATCH_NONE;
        for (uint32_t i = a; i + o.dp_len <= b && k == PATCH_NONE; i++) {
          bool eq = true;
          for (uint32_t j = 0; j < o.dp_len && eq; j++) eq = fv[i + j] == o.dp[j];
          if (eq) k = i - a;
        }
        if (k != PATCH_NONE) {  // the replaced text is the value only if ParseFloat(…, 64) takes it
          const Patched<F> pt{fv, a, k, o.dp_len};
          const uint32_t pn = (b - a) - o.dp_len + 1;
          const int pe = parsefloat_err(pt, pn);
          if (pe == 2) return TFGPU_ROW_HOST_FALLBACK;
          if (pe == 0) rc = parse_float32_go(pt, 0, pn, o.p128, &v);
        }
      }
      if (rc < 0) rc = parse_float32_go(fv, a, b, o.p128, &v);
      if (rc == 3) return TFGPU_ROW_HOST_FALLBACK;  // Go's decimal slow path (half-way cases, subnormals, the overflow edge), hex floats, '_'
      if (rc) return TFGPU_ROW_CAST;                // syntax or range: ToFloat32E returns the error
      ((float *)c.values)[r] = v;
      return 0;
    }
    case CK_INTERVAL: {  // parseNullValues, then strictify: cast.ToDurationE(string) (strictify.go:142-147)
      if (cell_is_null(o, fv, a, b, npairs)) { store_default(c, r); return 0; }
      if (npairs) return TFGPU_ROW_CAST;  // a '"' in the text reads as part of a unit: unknown unit
      int64_t d;
      if (parse_duration_go(fv, a, b, &d)) return TFGPU_ROW_CAST;
      ((int64_t *)c.values)[r] = d;
      return 0;
    }
    default:
      return TFGPU_ROW_HOST_FALLBACK;
  }
}

// ---------------------------------------------------------------------------
// per-row path: one lane per line, bytes straight from HBM.  Used for the lines
// the tile path cannot stage (longer than its look-behind window) and, with
// TFGPU_CSV_ROWPATH=1, for whole chunks (parity cross-check of the two paths).
// ---------------------------------------------------------------------------
__device__ void parse_line_hbm(const CsvParams &p, const bool have_row, const int64_t r, uint64_t pos, const uint64_t end /* one past '\n' */) {
  const uint64_t row0 = pos;
  MemBytes scan(p.data), fld(p.data);
  const CsvOpts &o = p.o;
  // Error precedence of the reference for one line: ReadLine errors (first field in
  // line order, reader.go:273-324) > constructCI missing cell (first column in schema
  // order, reader_csv.go:303-316) > Strictify cast/range (first column in schema order).
  int rerr = 0, rerr_col = -1, miss_col = 0x7FFFFFFF, cerr = 0, cerr_col = 0x7FFFFFFF;
  // a line that is just "\n" makes ReadLine return (nil, nil): zero fields (reader.go:146-150)
  const bool nil_line = have_row && (end - pos) <= 1;
  if (have_row && o.quote == 0) {  // readAndDecodeLine: QuoteChar == 0 and a '"' anywhere in the line → errQuotingDisabled (reader.go:185-187)
    for (uint64_t q = pos; q < end; q++) if (scan.at(q) == '"') { rerr = TFGPU_ROW_QUOTING_DISABLED; rerr_col = -1; break; }
  }
  if (have_row && o.multiline) {
    // readMultiline drops empty physical lines even inside a quoted value (reader.go:119-124): such a value is not a byte
    // range of the text any more — the host path rebuilds it
    for (uint64_t q = pos; q + 2 < end; q++) if (scan.at(q) == '\n' && scan.at(q + 1) == '\n') { rerr = TFGPU_ROW_HOST_FALLBACK; rerr_col = -1; break; }
  }
  bool active = have_row && !nil_line;
  bool in_quotes = false; uint32_t prev = 0xFFFFFFFFu;
  int32_t nfields = 0;          // fields produced so far by this lane
  bool seen_delim = false;
  for (int32_t f = 0;; f++) {
    // ---- splitString (reader.go:220-271): advance to the end of field f ----
    uint64_t fs = pos, fe = pos; bool last = false;
    if (active) {
      while (true) {
        if (pos >= end) { last = true; fe = end; break; }
        uint32_t ch = scan.at(pos);
        if (o.escape != 0 && prev == o.escape && in_quotes) { prev = ch; pos++; continue; }
        if (o.quote != 0 && ch == o.quote) { in_quotes = !in_quotes; prev = ch; pos++; continue; }
        if (ch == o.delim && !in_quotes) { fe = pos; pos++; seen_delim = true; prev = ch; break; }
        prev = ch; pos++;
      }
      // lastElement := line[lastDelimPosition+1:] with lastDelimPosition == 0 when the
      // line holds no delimiter: the first byte of the line is dropped (reader.go:263)
      if (last && !seen_delim) fs = row0 + 1 <= end ? row0 + 1 : end;
    }
    const bool has_field = active;
    if (!__any(has_field) && f >= p.nfields_used) break;
    // ---- sanitizeElement on every field, mapped or not (errors abort the line) ----
    Field fv{&fld, fs, (uint32_t)(fe - fs)};
    uint32_t a = 0, b = 0, npairs = 0;
    const int32_t first_col = f < p.nfields_used ? p.field_first[f] : -1;
    if (has_field) {
      int e = sanitize(o, fv, a, b, npairs);
      if (e && !rerr) { rerr = e; rerr_col = first_col; }
      nfields = f + 1;
    }
    // ---- constructCI + getCorrespondingValue + Strictify for the mapped columns ----
    for (int32_t ci = first_col; ci >= 0; ci = p.cols[ci].next) {
      const CsvCol &c = p.cols[ci];
      if (!have_row || rerr) continue;
      if (!has_field) {  // index >= len(row) (reader_csv.go:303-316)
        if (o.include_missing) store_default(c, r); else if (ci < miss_col) miss_col = ci;
        continue;
      }
      int err = parse_cell(o, c, r, fv, a, b, npairs, fs);
      if (err && ci < cerr_col) { cerr = err; cerr_col = ci; }
    }
    if (last) active = false;
  }
  if (!have_row) return;
  // columns whose field index lies beyond every field this line produced, and nil lines
  if (!rerr) {
    for (int32_t ci = 0; ci < p.ncols; ci++) {
      const CsvCol &c = p.cols[ci];
      if (c.field < 0) { store_default(c, r); continue; }
      if (c.field >= nfields) {
        if (c.field < p.nfields_used && !nil_line) continue;  // handled in the loop above
        if (o.include_missing) store_default(c, r); else { if (ci < miss_col) miss_col = ci; break; }
      }
    }
  }
  int err = rerr, err_col = rerr_col;
  if (!err && miss_col != 0x7FFFFFFF) { err = TFGPU_ROW_MISSING_CELL; err_col = miss_col; }
  if (!err && cerr) { err = cerr; err_col = cerr_col; }
  p.err[r] = (uint8_t)err;
  if (err) { p.err_col[r] = err_col; atomicAdd(p.nerr, 1u); }
}

__global__ void __launch_bounds__(256) csv_parse_rows(CsvParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool have_row = r < p.nrows;
  parse_line_hbm(p, have_row, r, have_row ? p.row_start[r] : 0, have_row ? (p.row_end ? p.row_end[r] : p.row_start[r + 1]) : 0);
}

// lines listed by the tile path: only the '\n' position is known; walk back to the line start
__global__ void __launch_bounds__(64) csv_parse_listed(CsvParams p) {
  const uint32_t n = *p.slow_n;
  for (uint64_t base = (uint64_t)blockIdx.x * 64; base < n; base += (uint64_t)gridDim.x * 64) {  // wave-uniform loop
    const uint64_t i = base + threadIdx.x;
    const int64_t line = i < n ? (int64_t)p.slow_row[i] : -1;
    const bool have_row = i < n && line >= p.skip_rows;
    uint64_t end = 0, pos = 0;
    if (have_row) {
      end = (uint64_t)p.slow_end[i] + 1;
      pos = end - 1;
      MemBytes back(p.data);
      while (pos > 0 && back.at(pos - 1) != '\n') pos--;
    }
    parse_line_hbm(p, have_row, line - p.skip_rows, pos, end);
  }
}

// ---------------------------------------------------------------------------
// tile path.  One 256-thread workgroup owns the lines whose '\n' falls into one
// CT_T-byte tile of the input.  It stages the tile plus a CT_SPILL-byte
// look-behind window in LDS with coalesced 16-byte loads, classifies every byte
// once (quote state = a prefix scan over per-byte {identity, toggle, set, clear}
// functions — splitString's state machine is a composition of those), turns the
// field ends into an LDS index, and then parses (column, line) cells with one
// lane per cell, lines fastest, so every column store is coalesced and every
// wave runs one column type.
// ---------------------------------------------------------------------------
#ifndef TF_CT_THREADS
#define TF_CT_THREADS 512
#endif
static constexpr int CT_THREADS = TF_CT_THREADS;
static constexpr uint32_t CT_WAVES = CT_THREADS / 64;  // slots of 64 items are dealt round-robin to them
static_assert((CT_WAVES & (CT_WAVES - 1)) == 0, "wave count must be a power of two");
static constexpr int CT_T = NL_TILE;            // bytes whose '\n' this workgroup owns
static constexpr int CT_SPILL = 4096;           // look-behind: longest line staged in LDS
static constexpr int CT_BYTES = CT_T + CT_SPILL;
static constexpr int CT_CPT = CT_BYTES / 16 / CT_THREADS;  // 16-byte chunks per thread (blocked)
static constexpr int CT_FCAP = 4096;            // field ends indexed per pass
static constexpr int CT_RCAP = 256;             // lines per pass
static constexpr int CT_LCOLS = 192;            // column descriptors staged in LDS (more: read from HBM)
static_assert(CT_CPT * 16 * CT_THREADS == CT_BYTES, "tile must divide evenly");
static_assert(CT_BYTES <= 32768, "positions are 15-bit");

// ---- fast-path helpers of the tile kernel -----------------------------------
// Is the rune that STARTS at tile position `a` (first byte c) possibly a unicode.IsSpace rune / the quote?
__device__ __forceinline__ bool starts_plain(const uint8_t *sb, uint32_t a, uint32_t end, uint32_t c, uint32_t quote) {
  if (c < 0x80) return c > 0x20 && c != 0x7F && c != quote;
  if (c != 0xC2 && c != 0xE1 && c != 0xE2 && c != 0xE3) return true;
  const uint32_t d = a + 1 < end ? sb[a + 1] : 0u, e = a + 2 < end ? sb[a + 2] : 0u;
  if (c == 0xC2) return !(d == 0x85 || d == 0xA0);
  if (c == 0xE1) return !(d == 0x9A && e == 0x80);
  if (c == 0xE3) return !(d == 0x80 && e == 0x80);
  return !((d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) || (d == 0x81 && e == 0x9F));
}
// Is the rune that ENDS at tile position b-1 (last byte c) possibly a space / the quote?
__device__ __forceinline__ bool ends_plain(const uint8_t *sb, uint32_t a, uint32_t b, uint32_t c, uint32_t quote) {
  if (c < 0x80) return c > 0x20 && c != 0x7F && c != quote;
  if (c >= 0xC0) return true;  // a lead byte at the end: malformed, never a space
  const uint32_t d = b - a >= 2 ? sb[b - 2] : 0u, e = b - a >= 3 ? sb[b - 3] : 0u;
  if (d == 0xC2) return !(c == 0x85 || c == 0xA0);
  if (e == 0xE1) return !(d == 0x9A && c == 0x80);
  if (e == 0xE3) return !(d == 0x80 && c == 0x80);
  if (e == 0xE2) return !((d == 0x80 && ((c >= 0x80 && c <= 0x8A) || c == 0xA8 || c == 0xA9 || c == 0xAF)) || (d == 0x81 && c == 0x9F));
  return true;
}
// swapToSingleQuotes (reader.go:307-320) replaces non-overlapping "" left to right: per run of L quote
// characters that is floor(L/2) pairs.  Walks only the set bits of the tile's quote bitmap in [lo, hi).
__device__ __forceinline__ uint32_t count_quote_pairs(const uint32_t *qmask, uint32_t lo, uint32_t hi) {
  if (lo >= hi) return 0;
  uint32_t pairs = 0, prev = 0xFFFFFFF0u;
  bool open = false;
  const uint32_t w0 = lo >> 5, w1 = (hi - 1) >> 5;
  for (uint32_t w = w0; w <= w1; w++) {
    uint32_t m = qmask[w];
    if (w == w0) m &= ~0u << (lo & 31);
    if (w == w1) m &= ~0u >> (31 - ((hi - 1) & 31));
    while (m) {
      const uint32_t pos = (w << 5) + (uint32_t)(__ffs((int)m) - 1);
      m &= m - 1;
      if (open && pos == prev + 1) { pairs++; open = false; } else open = true;
      prev = pos;
    }
  }
  return pairs;
}
__device__ __forceinline__ uint32_t two_digits(uint64_t w, int byte, bool *ok) {
  const uint32_t a = ((uint32_t)(w >> (8 * byte)) & 0xFFu) - '0', b = ((uint32_t)(w >> (8 * byte + 8)) & 0xFFu) - '0';
  *ok = *ok && a <= 9u && b <= 9u;
  return a * 10 + b;
}

// 0x80 in every byte of w equal to the byte replicated in pat4 (exact: no borrow between bytes)
__device__ __forceinline__ uint32_t eq80(uint32_t w, uint32_t pat4) {
  const uint32_t x = w ^ pat4;
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
// the 0x80 flags of 16 bytes (four words) as a 16-bit mask: v_dot4_u32_u8 with weights 1,2,4,8 sums each word's
// flags to nibble << 7
__device__ __forceinline__ uint32_t dense16(uint32_t z0, uint32_t z1, uint32_t z2, uint32_t z3) {
  const uint32_t W = 0x08040201u;
  const uint32_t d0 = __builtin_amdgcn_udot4(z0, W, 0u, false), d1 = __builtin_amdgcn_udot4(z1, W, 0u, false);
  const uint32_t d2 = __builtin_amdgcn_udot4(z2, W, 0u, false), d3 = __builtin_amdgcn_udot4(z3, W, 0u, false);
  return (d0 >> 7) | (d1 >> 3) | (d2 << 1) | (d3 << 5);
}

// quote-state functions on one bit: bit0 = f(0), bit1 = f(1); identity = 0b10
__device__ __forceinline__ uint32_t qf_compose(uint32_t first, uint32_t then) {
  return ((then >> (first & 1)) & 1u) | (((then >> ((first >> 1) & 1)) & 1u) << 1);
}

// One piece: the lines whose '\n' lies in granules [gstart, gstart + gcount), gcount * CSV_GRAN <= CT_T.
__device__ __forceinline__ void parse_tile_general(const CsvParams &p, const int64_t gstart, const int64_t gcount, const uint32_t spec_line0, const uint32_t spec_nlines) {
  __shared__ __attribute__((aligned(16))) uint8_t sb[CT_BYTES + 48];
  __shared__ uint16_t fpos[CT_FCAP];   // field-end position | bit15 = it is the line's '\n'
  __shared__ uint16_t rowend[CT_RCAP]; // ordinal of each line's last field
  __shared__ uint32_t slowf[CT_RCAP];  // line needs the per-row path (anything but a plain cell in it)
  __shared__ uint32_t qmask[CT_BYTES / 32];  // bitmap of quote characters
  __shared__ uint64_t rinfo[CT_RCAP];        // per line: first ordinal - kb | fields << 16 | first byte << 32
  __shared__ uint64_t tcol_lds[CT_LCOLS * 3];
  __shared__ uint32_t wtmp[CT_THREADS / 64];
  __shared__ uint32_t wmax[CT_THREADS / 64];
  __shared__ uint32_t wcnt[CT_THREADS / 64];

  const int64_t gend = min(gstart + gcount, p.ngran);
  const uint32_t line0 = p.spec ? spec_line0 : p.gran_pre[gstart];
  const uint32_t nlines = p.spec ? spec_nlines : p.gran_pre[gend] - line0;
  if (nlines == 0) return;  // no line ends here (inside a very long line)

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const CsvOpts &o = p.o;
  const int64_t g0 = gstart * CSV_GRAN - CT_SPILL;  // absolute offset of sb[0]
  const uint64_t limit = min((uint64_t)gend * CSV_GRAN, p.len);  // what follows belongs to the next piece: staged as zeros, like the end of the buffer

  // ---- stage: coalesced 16 B/lane ----
#pragma unroll
  for (int it = 0; it < CT_CPT; it++) {
    int chunk = it * CT_THREADS + tid;
    int64_t gp = g0 + (int64_t)chunk * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gp >= 0 && (uint64_t)gp < limit) v = *reinterpret_cast<const uint4 *>(p.data + gp);  // buffer is zero-padded past len; limit is a multiple of 16 or len
    *reinterpret_cast<uint4 *>(sb + chunk * 16) = v;
  }
  const bool cols_in_lds = p.ncols <= CT_LCOLS;
  if (cols_in_lds) for (int i = tid; i < p.ncols * 3; i += CT_THREADS) tcol_lds[i] = reinterpret_cast<const uint64_t *>(p.tcols)[i];
  __syncthreads();
  if (CSV_ABL(p) == 1) return;

  // ---- classify: thread t walks bytes [96t, 96t+96) under both entry states ----
  const int base_chunk = tid * CT_CPT;
  uint32_t fe01[CT_CPT];      // field-end delimiters if the thread starts outside (lo16) / inside (hi16) quotes
  uint32_t nlm[CT_CPT];       // '\n' mask
  uint32_t st = 2u;           // function of this thread's bytes (starts as identity)
  int last_lb_nl = -1;        // last '\n' inside the look-behind window
  {
    // 16 bytes at a time: per-byte equality flags by SWAR, packed to 16-bit masks with v_dot4, quote parity by
    // a prefix xor.  Only a chunk where an escape character directly precedes a quote walks its bytes.
    const uint32_t q4 = o.quote * 0x01010101u, d4 = o.delim * 0x01010101u, e4 = o.escape * 0x01010101u;
    bool prev_esc = false;
    if (o.escape != 0 && base_chunk > 0) prev_esc = sb[base_chunk * 16 - 1] == o.escape;
#pragma unroll
    for (int q = 0; q < CT_CPT; q++) {
      const uint4 v = *reinterpret_cast<const uint4 *>(sb + (base_chunk + q) * 16);
      const uint32_t qm = dense16(eq80(v.x, q4), eq80(v.y, q4), eq80(v.z, q4), eq80(v.w, q4));
      const uint32_t dm = dense16(eq80(v.x, d4), eq80(v.y, d4), eq80(v.z, d4), eq80(v.w, d4));
      const uint32_t nl = dense16(eq80(v.x, 0x0A0A0A0Au), eq80(v.y, 0x0A0A0A0Au), eq80(v.z, 0x0A0A0A0Au), eq80(v.w, 0x0A0A0A0Au));
      const uint32_t em = o.escape != 0 ? dense16(eq80(v.x, e4), eq80(v.y, e4), eq80(v.z, e4), eq80(v.w, e4)) : 0u;
      uint32_t f0 = 0, f1 = 0;
      if ((((em << 1) | (prev_esc ? 1u : 0u)) & qm) != 0) {  // `\"`: splitString's escape rule (reader.go:233-240), byte by byte
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        bool pe = prev_esc;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const uint32_t c = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
          if (c == '\n') st = 0u;
          else if (c == o.quote) st = pe ? 3u : (st ^ 3u);
          else if (c == o.delim) { f0 |= ((~st) & 1u) << i; f1 |= ((~st >> 1) & 1u) << i; }
          pe = c == o.escape;
        }
      } else {
        uint32_t px = qm;  // px bit i = parity of the quotes in bytes [0, i]
        px ^= px << 1; px ^= px << 2; px ^= px << 4; px ^= px << 8;
        px &= 0xFFFFu;
        uint32_t s0 = (st & 1u) ? 0xFFFFu : 0u, s1 = (st & 2u) ? 0xFFFFu : 0u;  // in-quotes entering the chunk, per hypothesis
        if (nl == 0) {
          f0 = dm & ~(px ^ s0); f1 = dm & ~(px ^ s1);
          if (__popc(qm) & 1) st ^= 3u;
        } else {  // every '\n' ends the line and the quote state with it
          uint32_t m = nl, start = 0, base = 0;
          while (m) {
            const uint32_t b = (uint32_t)__ffs((int)m) - 1; m &= m - 1;
            const uint32_t seg = ((1u << b) - 1u) & ~((1u << start) - 1u);
            f0 |= dm & ~(px ^ base ^ s0) & seg; f1 |= dm & ~(px ^ base ^ s1) & seg;
            base = ((px >> b) & 1u) ? 0xFFFFu : 0u; s0 = 0; s1 = 0; start = b + 1;
          }
          const uint32_t seg = 0xFFFFu & ~((1u << start) - 1u);
          const uint32_t t = dm & ~(px ^ base) & seg;
          f0 |= t; f1 |= t;
          st = (((px >> 15) ^ base) & 1u) ? 3u : 0u;
        }
      }
      prev_esc = (em >> 15) & 1u;
      fe01[q] = f0 | (f1 << 16);
      nlm[q] = nl;
      reinterpret_cast<uint16_t *>(qmask)[base_chunk + q] = (uint16_t)qm;
      const int cpos = (base_chunk + q) * 16;
      if (nl && cpos < CT_SPILL) last_lb_nl = cpos + 31 - __clz((int)nl);
    }
  }
  if (CSV_ABL(p) == 2) { if (st == 77u + fe01[0] + nlm[1]) p.err[0] = 1; return; }
  // ---- block scan of the quote functions + max of the look-behind '\n' ----
  uint32_t inc = st;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc = qf_compose(t, inc);
  }
  int mx = last_lb_nl;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
  if (lane == 63) wtmp[wv] = inc;
  if (lane == 0) wmax[wv] = (uint32_t)(mx + 1);
  __syncthreads();
  uint32_t pre = 2u;
  for (int i = 0; i < wv; i++) pre = qf_compose(pre, wtmp[i]);
  uint32_t lane_ex = __shfl_up(inc, 1, 64);
  if (lane == 0) lane_ex = 2u;
  const uint32_t s_in = qf_compose(pre, lane_ex) & 1u;  // quote state entering this thread's bytes
  int frs = 0;                                          // first byte of the first line ending in this tile
  for (int i = 0; i < CT_THREADS / 64; i++) frs = max(frs, (int)wmax[i]);
  bool first_long = false;
  if (frs == 0) {
    if (g0 <= 0) frs = (int)(-g0);  // the buffer starts inside the window: line 0 starts at absolute 0
    else first_long = true;         // the first line started before the window: per-row path
  }

  // ---- select the hypothesis, drop what precedes the first line, count ----
  uint32_t fe[CT_CPT];  // field ends: unquoted delimiters and '\n'
  uint32_t cnt = 0;
#pragma unroll
  for (int q = 0; q < CT_CPT; q++) {
    const int cpos = (base_chunk + q) * 16;
    uint32_t m = ((s_in ? (fe01[q] >> 16) : fe01[q]) & 0xFFFFu) | nlm[q];
    uint32_t keep = 0xFFFFu;
    if (cpos + 16 <= frs) keep = 0; else if (cpos < frs) keep = 0xFFFFu & ~((1u << (frs - cpos)) - 1u);
    fe[q] = m & keep; nlm[q] &= keep;  // look-behind '\n's all precede frs
    cnt += __popc(fe[q]) | (__popc(nlm[q]) << 16);
  }
  uint32_t cinc = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(cinc, d, 64);
    if (lane >= d) cinc += t;
  }
  if (lane == 63) wcnt[wv] = cinc;
  __syncthreads();
  uint32_t cpre = 0, ctot = 0;
  for (int i = 0; i < CT_THREADS / 64; i++) { uint32_t x = wcnt[i]; if (i < wv) cpre += x; ctot += x; }
  const uint32_t cex = cpre + cinc - cnt;
  const uint32_t k_thread = cex & 0xFFFFu, j_thread = cex >> 16;  // ordinal / line of this thread's first field end
  const uint32_t nl_tot = ctot >> 16;                             // == nlines by construction
  // the usual tile: every field end fits the index and every line the window → one indexing sweep, no search
  const bool onepass = !first_long && (ctot & 0xFFFFu) <= (uint32_t)CT_FCAP && nl_tot <= (uint32_t)CT_RCAP;

  // ---- passes over the lines of this tile ----
  uint32_t jb = 0, kb = 0;
  uint32_t bstart = (uint32_t)frs;  // first byte of line jb
  while (jb < nl_tot) {
    uint32_t nr = 0, kend = 0;
    bool skip_first = false;
    if (onepass) {
      uint32_t k = k_thread, j = j_thread;
#pragma unroll
      for (int q = 0; q < CT_CPT; q++) {
        uint32_t m = fe[q];
        const int cpos = (base_chunk + q) * 16;
        while (m) {
          const int b = __ffs((int)m) - 1; m &= m - 1;
          const uint32_t isnl = (nlm[q] >> b) & 1u;
          fpos[k] = (uint16_t)((cpos + b) | (isnl << 15));
          if (isnl) rowend[j++] = (uint16_t)k;
          k++;
        }
      }
      nr = nl_tot;
      for (uint32_t i = tid; i < nr; i += CT_THREADS) slowf[i] = 0;
      __syncthreads();
      kend = rowend[nr - 1];  // field ends past the last '\n' belong to the next tile's first line
    } else {
    // pass 1: ordinal of each line's '\n' for lines [jb, jb+CT_RCAP)
    {
      uint32_t k = k_thread, j = j_thread;
#pragma unroll
      for (int q = 0; q < CT_CPT; q++) {
        uint32_t m = nlm[q];
        while (m) {
          const int b = __ffs((int)m) - 1; m &= m - 1;
          const uint32_t kk = k + __popc(fe[q] & ((1u << b) - 1u));
          if (j >= jb && j < jb + CT_RCAP) rowend[j - jb] = (uint16_t)kk;
          j++;
        }
        k += __popc(fe[q]);
      }
    }
    __syncthreads();
    const uint32_t nwin = min((uint32_t)CT_RCAP, nl_tot - jb);
    // lines that fit the field index: rowend is increasing, count entries below the cap
    {
      uint32_t lo = 0, hi = nwin;  // first index whose last field does not fit
      while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if ((uint32_t)rowend[mid] - kb < (uint32_t)CT_FCAP) lo = mid + 1; else hi = mid; }
      nr = lo;
    }
    skip_first = (jb == 0 && first_long) || nr == 0;  // line handled by the per-row path
    if (skip_first) nr = 1;
    kend = rowend[nr - 1];
    // pass 2: positions of the field ends [kb, kend]
    if (!skip_first) {
      uint32_t k = k_thread;
#pragma unroll
      for (int q = 0; q < CT_CPT; q++) {
        uint32_t m = fe[q];
        const int cpos = (base_chunk + q) * 16;
        while (m) {
          const int b = __ffs((int)m) - 1; m &= m - 1;
          if (k >= kb && k <= kend) fpos[k - kb] = (uint16_t)((cpos + b) | (((nlm[q] >> b) & 1u) << 15));
          k++;
        }
      }
    }
    for (uint32_t i = tid; i < nr; i += CT_THREADS) slowf[i] = 0;
    __syncthreads();
    }
    if (CSV_ABL(p) == 3) { if (fpos[tid] == 0xFFFF && rowend[0] == 0xFFFF) p.err[0] = 1; return; }

    uint32_t next_bstart;
    if (skip_first) {
      // find the '\n' of this one line: it is the field end kend; only its owner knows the position
      uint32_t k = k_thread;
#pragma unroll
      for (int q = 0; q < CT_CPT; q++) {
        uint32_t m = fe[q];
        const int cpos = (base_chunk + q) * 16;
        while (m) {
          const int b = __ffs((int)m) - 1; m &= m - 1;
          if (k == kend) {
            fpos[0] = (uint16_t)(cpos + b);
            uint32_t slot = atomicAdd(p.slow_n, 1u);
            p.slow_row[slot] = line0 + jb;
            p.slow_end[slot] = (uint32_t)(g0 + cpos + b);
          }
          k++;
        }
      }
      __syncthreads();
      next_bstart = (uint32_t)fpos[0] + 1;
    } else {
      // ---- per-line info, once per pass ----
      for (uint32_t jj = tid; jj < nr; jj += CT_THREADS) {
        const uint32_t k0 = jj ? (uint32_t)rowend[jj - 1] + 1 : kb, ke = rowend[jj];
        const uint32_t row_start = jj ? (uint32_t)(fpos[k0 - 1 - kb] & 0x7FFFu) + 1 : bstart;
        const uint32_t nl_pos = fpos[ke - kb] & 0x7FFFu;
        const uint32_t nf = (nl_pos == row_start) ? 0u : ke - k0 + 1;  // "\n" alone: ReadLine returns (nil, nil)
        rinfo[jj] = (uint64_t)(k0 - kb) | ((uint64_t)nf << 16) | ((uint64_t)row_start << 32);
      }
      __syncthreads();
      // ---- cells: item = column * nr + line, lines fastest.  Only PLAIN cells are decided here:
      //      nothing to trim, no quote characters inside, canonical decimal / date shapes.  Any other
      //      cell flags its line, and flagged lines are re-read whole by the per-row path, which carries
      //      the complete reference logic (and every error case). ----
      const uint32_t total = (uint32_t)p.ncols * nr;
      const uint32_t inv_nr = nr > 1 ? 0xFFFFFFFFu / nr + 1 : 0;  // it / nr == umulhi(it, inv_nr) while it * nr < 2^32 (it < ncols * 256)
      for (uint32_t it = tid; it < total; it += CT_THREADS) {
        uint32_t oi = nr > 1 ? __umulhi(it, inv_nr) : it;
        uint32_t jj = it - oi * nr;
        if (jj >= nr) { oi++; jj -= nr; }  // guard the rounding of the reciprocal
        const int64_t r = (int64_t)line0 + jb + jj - p.skip_rows;
        if (r < 0) continue;  // header lines
        TCol tc;
        if (cols_in_lds) { const uint64_t *q = tcol_lds + oi * 3; tc.p0 = (void *)q[0]; tc.p1 = (void *)q[1]; uint64_t m = q[2]; tc.field = (int16_t)m; tc.kind = (uint8_t)(m >> 16); tc.width = (uint8_t)(m >> 24); tc.ci = (int32_t)(m >> 32); }
        else tc = p.tcols[oi];
        const uint64_t ri = rinfo[jj];
        const uint32_t k0r = (uint32_t)ri & 0xFFFFu, nf = (uint32_t)(ri >> 16) & 0xFFFFu, row_start = (uint32_t)(ri >> 32);
        if (tc.field < 0 || (uint32_t)tc.field >= nf) {
          if (tc.field < 0 || o.include_missing) {
            CsvCol c{}; c.kind = tc.kind; c.width = tc.width; c.values = tc.p0; c.nanos = (int32_t *)tc.p1; c.lens = (uint32_t *)tc.p0; c.fstart = (uint32_t *)tc.p1;
            store_default(c, r);
          } else slowf[jj] = 1;  // missing cell: an error of this line
          continue;
        }
        const uint32_t kr = k0r + (uint32_t)tc.field;
        uint32_t fs = tc.field ? (uint32_t)(fpos[kr - 1] & 0x7FFFu) + 1 : row_start;
        const uint32_t fend = fpos[kr] & 0x7FFFu;
        if (nf == 1) fs = row_start + 1;  // no delimiter in the line: line[lastDelimPosition+1:] drops byte 0 (reader.go:263)
        const uint32_t n = fend - fs;
        if (CSV_ABL(p) == 6) { if (n == 0x7FFF) p.err[0] = 1; continue; }
        bool done = false;
        if (n == 0) {
          if (tc.kind == CK_STR) { ((uint32_t *)tc.p0)[r] = 0; ((uint32_t *)tc.p1)[r] = (uint32_t)(g0 + fs); done = true; }
        } else {
          // bytes [fs, fs+24) little-endian; the tile is padded, so the aligned words past the field exist
          const uint32_t sh = (fs & 7u) * 8;
          const uint64_t *w = reinterpret_cast<const uint64_t *>(sb + (fs & ~7u));
          const uint64_t w0 = w[0], w1 = w[1];
          const uint64_t b0 = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
          const uint32_t c_first = (uint32_t)b0 & 0xFFu, c_last = sb[fend - 1];
          if (tc.kind == CK_STR) {
            // nothing to trim: either enclosed in quotes ("…": unquote) or plain at both ends
            uint32_t a = fs, b = fend;
            bool ok = false;
            if (c_first == o.quote && c_last == o.quote && n >= 2) { a++; b--; ok = true; }
            else ok = c_first != o.quote && c_last != o.quote && starts_plain(sb, fs, fend, c_first, o.quote) && ends_plain(sb, fs, fend, c_last, o.quote);
            if (ok) {
              const uint32_t npairs = any_quote(qmask, a, b) ? count_quote_pairs(qmask, a, b) : 0u;
              if (!(npairs && !o.double_quote)) {  // errDoubleQuotesDisabled: per-row path
                ((uint32_t *)tc.p0)[r] = (b - a) - npairs;
                ((uint32_t *)tc.p1)[r] = (uint32_t)(g0 + a) | (npairs ? 0x80000000u : 0u);
                if (npairs) ((uint32_t *)tc.p1)[-1] = 1u;  // the column holds cells that are not a plain byte range
                done = true;
              }
            }
          } else if (tc.kind == CK_INT || tc.kind == CK_UINT) {
            // [+-]?(0|[1-9][0-9]*) — base prefixes, '_', ".00", spaces, quotes: per-row path
            if (n <= 20) {
              const bool neg = c_first == '-';
              const uint32_t i0 = (c_first == '-' || c_first == '+') ? 1u : 0u;
              const uint32_t nd = n - i0;
              bool ok = nd >= 1 && nd <= 19 && !(nd > 1 && ((uint32_t)(b0 >> (8 * i0)) & 0xFFu) == '0');
              if (tc.kind == CK_UINT && tc.width == 8 && i0) ok = false;  // ParseUint takes no sign
              uint64_t b1 = 0, b2 = 0;
              if (__any(n > 8)) { const uint64_t w2 = w[2]; b1 = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1; if (__any(n > 16)) { const uint64_t w3 = w[3]; b2 = sh ? (w2 >> sh) | (w3 << (64 - sh)) : w2; } }
              uint64_t v;
              ok = digits_u64(b0, b1, b2, i0, n, &v) && ok;
              const int bits = tc.width * 8;
              if (tc.kind == CK_INT) ok = ok && v <= (neg ? (1ull << (bits - 1)) : (1ull << (bits - 1)) - 1);
              else ok = ok && !(neg && v != 0) && (tc.width == 8 || v <= (1ull << bits) - 1);
              if (ok && CSV_ABL(p) != 5) {
                const int64_t sv = neg ? (int64_t)(0 - v) : (int64_t)v;
                switch (tc.width) {
                  case 1: ((int8_t *)tc.p0)[r] = (int8_t)sv; break;
                  case 2: ((int16_t *)tc.p0)[r] = (int16_t)sv; break;
                  case 4: ((int32_t *)tc.p0)[r] = (int32_t)sv; break;
                  default: ((int64_t *)tc.p0)[r] = sv;
                }
              }
              done = ok;
            }
          } else if (tc.kind == CK_DATE || tc.kind == CK_TIMESTAMP) {
            // 2006-01-02 | 2006-01-02[ T]15:04:05 (cast.StringToDate layouts); a timestamp column also takes
            // plain decimal seconds (parseTimestampValue, reader_csv.go:419-426)
            const uint64_t w2 = w[2], w3 = w[3];
            const uint64_t b1 = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1, b2 = sh ? (w2 >> sh) | (w3 << (64 - sh)) : w2;
            const uint32_t i0 = (c_first == '-' || c_first == '+') ? 1u : 0u;
            uint64_t v = 0;
            if (tc.kind == CK_TIMESTAMP && n - i0 >= 1 && n - i0 <= 18 && digits_u64(b0, b1, b2, i0, n, &v)) {
              // ParseInt(s, 10, 64): leading zeros are fine in base 10
              ((int64_t *)tc.p0)[r] = c_first == '-' ? -(int64_t)v : (int64_t)v; ((int32_t *)tc.p1)[r] = 0; done = true;
            } else if (n == 10 || n == 19) {
              bool ok = true;
              const uint32_t y = two_digits(b0, 0, &ok) * 100 + two_digits(b0, 2, &ok);
              const uint32_t mo = two_digits(b0, 5, &ok), d = two_digits(b1, 0, &ok);
              ok = ok && (((uint32_t)(b0 >> 32) & 0xFFu) == '-') && (((uint32_t)(b0 >> 56) & 0xFFu) == '-');
              uint32_t h = 0, mi = 0, se = 0;
              if (n == 19) {
                const uint32_t sep = (uint32_t)(b1 >> 16) & 0xFFu;
                ok = ok && (sep == ' ' || sep == 'T') && (((uint32_t)(b1 >> 40) & 0xFFu) == ':') && (((uint32_t)b2 & 0xFFu) == ':');
                h = two_digits(b1, 3, &ok); mi = two_digits(b1, 6, &ok); se = two_digits(b2, 1, &ok);
              }
              ok = ok && mo >= 1 && mo <= 12 && d >= 1 && d <= (uint32_t)dev::days_in_month((int)mo, (int64_t)y) && h <= 23 && mi <= 59 && se <= 59;
              if (ok) {
                ((int64_t *)tc.p0)[r] = dev::days_from_civil((int64_t)y, (int)mo, (int)d) * 86400 + h * 3600 + mi * 60 + se;
                ((int32_t *)tc.p1)[r] = 0;
                done = true;
              }
            }
          }
        }
        if (!done) slowf[jj] = 1;
      }
      // ---- sanitizeElement also runs on fields no column reads: anything but a plain field flags the line ----
      if (p.has_unmapped || (kend - kb + 1) != nr * (uint32_t)p.nfields_used) {
        for (uint32_t kk = tid; kk <= kend - kb; kk += CT_THREADS) {
          uint32_t lo = 0, hi = nr - 1;  // line of ordinal kb+kk
          while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if ((uint32_t)rowend[mid] < kb + kk) lo = mid + 1; else hi = mid; }
          const uint32_t jj = lo;
          if ((int64_t)line0 + jb + jj - p.skip_rows < 0) continue;
          const uint64_t ri = rinfo[jj];
          const uint32_t k0r = (uint32_t)ri & 0xFFFFu, nf = (uint32_t)(ri >> 16) & 0xFFFFu, row_start = (uint32_t)(ri >> 32);
          const uint32_t f = kk - k0r;
          if (nf == 0 || (f < (uint32_t)p.nfields_used && p.field_first[f] >= 0)) continue;  // nil line / judged by its column
          uint32_t fs = f ? (uint32_t)(fpos[kk - 1] & 0x7FFFu) + 1 : row_start;
          if (nf == 1) fs = row_start + 1;
          const uint32_t fend = fpos[kk] & 0x7FFFu;
          if (fend > fs && !(starts_plain(sb, fs, fend, sb[fs], o.quote) && ends_plain(sb, fs, fend, sb[fend - 1], o.quote) && !any_quote(qmask, fs, fend))) slowf[jj] = 1;
        }
      }
      __syncthreads();
      // ---- per line: clean, or handed to the per-row path ----
      for (uint32_t jj = tid; jj < nr; jj += CT_THREADS) {
        const int64_t r = (int64_t)line0 + jb + jj - p.skip_rows;
        if (r < 0) continue;
        if (slowf[jj] || p.null_checks) {
          const uint32_t slot = atomicAdd(p.slow_n, 1u);
          p.slow_row[slot] = line0 + jb + jj;
          p.slow_end[slot] = (uint32_t)(g0 + (fpos[rowend[jj] - kb] & 0x7FFFu));
        } else p.err[r] = 0;
      }
      next_bstart = (uint32_t)(fpos[kend - kb] & 0x7FFFu) + 1;
    }
    __syncthreads();
    bstart = next_bstart;
    jb += nr; kb = kend + 1;
  }
  if (tid == 0) {  // one past the last '\n' this tile owns
    atomicMax(p.last_end, (uint32_t)(g0 + bstart));
    if (p.spec) p.spec[2 + p.ntiles + gstart / (CT_T / CSV_GRAN)] = (uint32_t)(g0 + bstart);
  }
}

// The tiles csv_parse_regular could not take (anything but the plain shape it is specialised for), one after another
// per workgroup; the LDS arrays are reused, hence the barrier between tiles.
__global__ void __launch_bounds__(CT_THREADS, 6) csv_parse_tiles_general(CsvParams p) {
  const uint32_t n = *p.gen_n;
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const bool sp = p.spec != nullptr;
    const uint32_t *e = p.gen_tile + (size_t)(sp ? 4 : 2) * i;
    parse_tile_general(p, (int64_t)e[0], (int64_t)e[1], sp ? e[2] : 0u, sp ? e[3] : 0u);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// regular tiles.  What almost every tile of almost every file looks like: no escape character in front of a
// quote, every line's quotes balanced (the quote state a '\n' resets is already "outside"), every line the same
// number NF >= 2 of fields, everything fits the LDS index.  Then splitString's state machine is a plain prefix
// parity of the quote characters, field k of line j is ordinal j * NF + k of the index, and nothing per line has to
// be looked up.  A tile that is anything else is handed, whole, to csv_parse_tiles_general (above), which carries
// the complete state machine; a CELL that is anything but plain still flags its line for csv_parse_listed.
//   barriers: stage | quote parity per wave | field-end counts per wave | index | cells | epilogue
//
// Round 3.  The kernel issues VALU instructions 96 % of its time, so the work of the round went into what it issues
// (profiles/r05*): (1) the cell loops pick, per slot of 64 cells and with one ballot, a straight-line tier for the shapes
// almost every cell has — an unsigned canonical decimal of at most five (narrow types) or eight digits, a signed one of
// up to nineteen through three windows of eight without a branch, an empty / plain / "quoted" text cell — and keep the
// general code for the rest; (2) the byte classes chain their v_dot4 in pairs through the accumulator; (3) the escape
// character is looked for where the tile's few quote characters are walked, not in every chunk; (4) the tile's line
// counts arrive by scalar loads when the kernel starts — the two vector loads the epilogue used to issue waited for
// every store of the cell phase (≈ 10 000 cycles of a workgroup's 46 000).  A lane-per-line / wave-per-column variant
// with 44 KiB tiles was built and measured first (commit c60a587): a third fewer VALU instructions in the cells, but two
// workgroups per CU instead of three, and 0.96 ms against this kernel's 0.75 before the changes above.
// ---------------------------------------------------------------------------
// days since 1970-01-01 of a civil date with 0 <= y <= 9999, in 32-bit arithmetic (constant divisors only)
__device__ __forceinline__ int32_t days_from_civil32(uint32_t y, uint32_t m, uint32_t d) {
  const uint32_t yy = y + 400u - (m <= 2 ? 1u : 0u);  // shifted by one era so that the year stays non-negative
  const uint32_t era = yy / 400u, yoe = yy - era * 400u;
  const uint32_t doy = (153u * (m > 2 ? m - 3 : m + 9) + 2u) / 5u + d - 1u;
  const uint32_t doe = yoe * 365u + yoe / 4u - yoe / 100u + doy;
  return (int32_t)(era * 146097u + doe) - 719468 - 146097;
}
__device__ __forceinline__ uint32_t days_in_month32(uint32_t m, uint32_t y) {
  const uint32_t leap = ((y & 3u) == 0 && (y % 100u != 0 || y % 400u == 0)) ? 1u : 0u;
  return m == 2 ? 28u + leap : 30u + ((0x15AAu >> m) & 1u);  // bit m set for the 31-day months 1,3,5,7,8,10,12
}

#ifndef TF_CR_FCAP
#define TF_CR_FCAP 4608
#endif
static constexpr int CR_FCAP = TF_CR_FCAP;   // field ends the regular kernel indexes per tile (28 KiB of single-digit fields would be 14 000; hits averages 3 800)
static constexpr int CR_LCOLS = 128;   // columns whose descriptors it stages in LDS
static constexpr uint32_t CR_KDUMMY = CR_FCAP + 2;  // index slot of the empty field idle lanes read
struct CsvRun { int32_t kind, width, first, ncols; };  // consecutive tcols of one (kind, width); kind -1: DefaultValue columns (no field)
struct CsvTask { int32_t kind, width, first, ncols; };  // a piece of a run of the fast kinds (integers, text, date / timestamp): ncols <= 64

// what the cell loops of csv_parse_regular share
struct RegTile {
  const uint8_t *sb; const uint16_t *fposx; const uint32_t *qmask; const uint16_t *qpre; const uint2 *keeptab; uint8_t *slowf; const uint64_t *colp0, *colp1; const uint16_t *colfield;
  uint32_t nr, NF, step_q, step_r; float inv_nr; int32_t row0; uint32_t g0; uint32_t quote; bool double_quote; int32_t ablate; int32_t col_mode;
};
// one cell of a run: item → (column, line), its output row and its byte range [fs, fend) in the tile
struct RegCell { bool on; uint32_t col, jj, fs, fend; int32_t r; };
// A wave's items of one run are it0, it0 + 512, it0 + 1024, … (slots of 64 items dealt round-robin to 8 waves): the first
// (column offset, line) pair comes from one division, the following ones from adding 512 / nr and 512 % nr with one carry.
struct ItemIter { uint32_t it, oi, jj; };
__device__ __forceinline__ ItemIter item_first(const RegTile &t, uint32_t it) {
  // it / nr through the float reciprocal (rounded down, so the quotient is never too large; it < 2^24): all full-rate ops
  uint32_t oi = (uint32_t)(__uint2float_rz(it) * t.inv_nr);
  uint32_t jj = it - __umul24(oi, t.nr);
  if (jj >= t.nr) { oi++; jj -= t.nr; }
  return ItemIter{it, oi, jj};
}
__device__ __forceinline__ void item_next(const RegTile &t, ItemIter &x) {
  x.it += 64u * CT_WAVES; x.oi += t.step_q; x.jj += t.step_r;
  if (x.jj >= t.nr) { x.jj -= t.nr; x.oi++; }
}
__device__ __forceinline__ RegCell reg_cell(const RegTile &t, uint32_t first_col, const ItemIter &x, uint32_t items) {
  RegCell c;
  c.col = first_col + x.oi; c.jj = x.jj;
  c.r = t.row0 + (int32_t)x.jj;
  c.on = x.it < items && c.r >= 0;  // header lines produce nothing
  // (idle lanes read the index's empty dummy field: the straight-line tiers run without a per-lane branch)
  const uint32_t kr = c.on ? __umul24(x.jj, t.NF) + (uint32_t)t.colfield[c.col] : CR_KDUMMY;  // < nr * NF <= the index's size (checked before the cells)
  // two ds_read_u16 on purpose (the second index is hidden from the optimizer): merged into one ds_read_b32 at a 2-byte
  // aligned address the pair would cost 64 LDS cycles per wave — byte-addressed LDS reads run at 1/13 of the aligned rate
  // on gfx950 (tools/microbench/valu_rate.hip)
  uint32_t kr1 = kr + 1;
  TF_OPAQUE(kr1);
  c.fs = (uint32_t)t.fposx[kr] + 1; c.fend = t.fposx[kr1];
  return c;
}
// A wave's slots of one run.  TF_CSV_PREFETCH=1 (A/B builds) reads the NEXT slot's index entries before the current slot's body runs,
// so that the two ds_read_u16 of slot s + 8 ride under slot s's arithmetic instead of heading its dependent chain (index → window →
// store).  Measured (gpurun r08g): 1.04-1.09 ms against 0.74 — the kernel sits at the 80 VGPRs six waves per SIMD allow, the second
// cell in flight spills 76 bytes per lane to scratch.  Off.
#ifndef TF_CSV_PREFETCH
#define TF_CSV_PREFETCH 0
#endif
template <class Body> __device__ __forceinline__ void for_slots(const RegTile &t, const CsvRun &run, uint32_t sbase, int wv, int lane, Body body) {
  const uint32_t items = (uint32_t)run.ncols * t.nr, nslots = (items + 63) >> 6, end = sbase + nslots;
  const uint32_t s0 = sbase + (((uint32_t)wv - sbase) & (CT_WAVES - 1u));
  if (s0 >= end) return;
  ItemIter x = item_first(t, (s0 - sbase) * 64 + (uint32_t)lane);
#if TF_CSV_PREFETCH
  RegCell c = reg_cell(t, (uint32_t)run.first, x, items);
  for (uint32_t s = s0; s < end; s += CT_WAVES) {
    RegCell cn = c;
    if (s + CT_WAVES < end) { item_next(t, x); cn = reg_cell(t, (uint32_t)run.first, x, items); }  // (wave-uniform)
    body(c);
    c = cn;
  }
#else
  for (uint32_t s = s0; s < end; s += CT_WAVES, item_next(t, x)) body(reg_cell(t, (uint32_t)run.first, x, items));
#endif
}
// the column buffers are device allocations: say so, or the pointer read back from LDS makes every store a flat_store
template <class T> __device__ __forceinline__ T *global_ptr(uint64_t p) { return TF_GLOBAL_PTR(T, p); }
// number of quote characters in tile positions [a, b), a <= b: prefix counts per 32-byte word + the bitmap
__device__ __forceinline__ uint32_t quotes_in(const RegTile &t, uint32_t a, uint32_t b) {
  const uint32_t ca = (uint32_t)t.qpre[a >> 5] + (uint32_t)__popc(t.qmask[a >> 5] & ((1u << (a & 31)) - 1u));
  const uint32_t cb = (uint32_t)t.qpre[b >> 5] + (uint32_t)__popc(t.qmask[b >> 5] & ((1u << (b & 31)) - 1u));
  return cb - ca;
}

// one integer cell (c: its line, byte range, output row) of a column whose values live at `out`; every lane of the wave calls it
template <int KIND, int W, class T> __device__ __forceinline__ void reg_int_body(const RegTile &t, const RegCell &c, T *const out) {
  constexpr uint32_t FASTD = W < 4 ? 5u : 8u;  // digits the first tier takes (more cannot fit / need a second window)
  const uint32_t n = c.fend - c.fs;
  if (CSV_ABL(t) != 31 && wave_all(n - 1u < 4u || !c.on)) {
    // tier A: one to four characters, an unsigned canonical decimal — the common cell of a flag / small-integer column.  One
    // aligned LDS word pair holds the field's bytes [fend - 4, fend); masks come from shifts (an LDS table read costs the CU
    // as much as six VALU instructions), the leading-zero rule from the first character in byte 0.
    const uint32_t *w4 = reinterpret_cast<const uint32_t *>(t.sb + (int)((c.fend & ~3u) - 4u));
    const uint32_t xr = __builtin_amdgcn_alignbyte(w4[1], w4[0], c.fend) ^ 0x30303030u;
    const uint32_t sh = 32u - 8u * n;             // (an idle lane's n = 0 shifts by nothing: its result is not stored)
    const uint32_t f0 = xr >> sh;                 // first character in byte 0
    const uint32_t d = f0 << sh;                  // units in byte 3, …, thousands in byte 0; what precedes the field is zero
    bool badA = ((((d + 0x76767676u) | d) & 0x80808080u) != 0) | (((f0 & 0xFFu) == 0) & (n > 1u));  // a byte above 9; a leading zero in front of more digits
    const uint32_t v = __umul24(d & 0xFFu, 1000u) + __builtin_amdgcn_udot4(d, 0x010A6400u, 0u, false);
    if constexpr (W == 1) badA = badA | (v > ((KIND == CK_INT) ? 127u : 255u));
    if (c.on && !badA) out[c.r] = (T)v;
    if (c.on && badA) t.slowf[c.jj] = 1;
    return;
  }
  const uint32_t c0 = t.sb[c.fs];
  uint32_t wlo, whi;
  window8(t.sb, c.fend, &wlo, &whi);
  bool bad;
  // first tier: no sign, no leading zero (a lone "0" is fine; a first byte above '9' fails the digit check), 1..FASTD digits
  const bool fast = (c0 >= '1' || n == 1u) && n - 1u < FASTD;
  if (CSV_ABL(t) == 31) { if (c.on && fast && wlo == 0x12345678u && whi == 77u) out[c.r] = 0; return; }  // (profiling: the slot's skeleton and reads only)
  if (wave_all(fast || !c.on)) {
    const uint2 keep = t.keeptab[c.on ? n : 0u];  // the top n bytes of the window
    const uint32_t tlo = (wlo ^ 0x30303030u) & keep.x, thi = (whi ^ 0x30303030u) & keep.y;
    bad = (((tlo + 0x76767676u) | tlo | (thi + 0x76767676u) | thi) & 0x80808080u) != 0;
    uint32_t v;
    if constexpr (W < 4) {
      v = __umul24(tlo >> 24, 10000u) + four_dot(thi);
      if constexpr (KIND == CK_INT) bad = bad || v > (1u << (W * 8 - 1)) - 1u; else bad = bad || v > (1u << (W * 8)) - 1u;
    } else v = __umul24(four_dot(tlo), 10000u) + four_dot(thi);  // < 10^8: in range for 32 and 64 bits
    if (c.on && !bad) out[c.r] = (T)v;
  } else if constexpr (W >= 4) {
    // second tier, straight-line as well: [+-]?(0|[1-9][0-9]*) of up to 19 digits.  The 24 bytes in front of the field's end
    // are three windows of eight; a window's digits are kept by the same table (none of them: the window counts as zero).
    const bool neg = c0 == '-', sgn = neg || c0 == '+';
    const uint32_t nd = n - (sgn ? 1u : 0u);  // digits
    const uint32_t fd = sgn ? (uint32_t)t.sb[c.fs + 1] : c0;  // the first of them
    bool ok = nd - 1u < 19u && !(nd > 1 && fd == '0');
    if constexpr (KIND == CK_UINT && W == 8) ok = ok && !sgn;  // ParseUint takes no sign
    const uint32_t y = c.fend;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(t.sb + (int)((y - 24u) & ~3u));
    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4];
    const uint32_t w2lo = __builtin_amdgcn_alignbyte(d1, d0, y), w2hi = __builtin_amdgcn_alignbyte(d2, d1, y);  // [y - 24, y - 16)
    const uint32_t w1lo = __builtin_amdgcn_alignbyte(d3, d2, y), w1hi = __builtin_amdgcn_alignbyte(d4, d3, y);  // [y - 16, y - 8); [y - 8, y) is wlo, whi
    const uint32_t ndc = min(nd, 24u), n0 = min(ndc, 8u), n1d = min(ndc - n0, 8u), n2d = min(ndc - n0 - n1d, 8u);
    const uint2 k0 = t.keeptab[n0], k1 = t.keeptab[n1d], k2 = t.keeptab[n2d];
    const uint32_t a0 = (wlo ^ 0x30303030u) & k0.x, b0 = (whi ^ 0x30303030u) & k0.y;
    const uint32_t a1 = (w1lo ^ 0x30303030u) & k1.x, b1 = (w1hi ^ 0x30303030u) & k1.y;
    const uint32_t a2 = (w2lo ^ 0x30303030u) & k2.x, b2 = (w2hi ^ 0x30303030u) & k2.y;
    const uint32_t nondigit = ((a0 + 0x76767676u) | a0 | (b0 + 0x76767676u) | b0 | (a1 + 0x76767676u) | a1 | (b1 + 0x76767676u) | b1 |
                               (a2 + 0x76767676u) | a2 | (b2 + 0x76767676u) | b2) & 0x80808080u;
    ok = ok && nondigit == 0;
    const uint32_t g0v = __umul24(four_dot(a0), 10000u) + four_dot(b0), g1v = __umul24(four_dot(a1), 10000u) + four_dot(b1);
    const uint32_t g2v = __umul24(four_dot(a2), 10000u) + four_dot(b2);  // < 10^8 each; nd <= 19: g2v < 1000
    const uint64_t v = ((uint64_t)g2v * 100000000ull + g1v) * 100000000ull + g0v;  // < 10^19 < 2^64
    if constexpr (KIND == CK_INT) ok = ok && v <= (neg ? (1ull << (W * 8 - 1)) : (1ull << (W * 8 - 1)) - 1);
    else ok = ok && !(neg && v != 0) && (W == 8 || v <= (1ull << (W * 8 % 64)) - 1);
    if (c.on && ok) out[c.r] = (T)(neg ? (int64_t)(0 - v) : (int64_t)v);
    bad = !ok;
  } else {
    // the narrow types' rare shapes (a sign, a leading zero, too many digits)
    const bool neg = c0 == '-', sgn = neg || c0 == '+';
    const uint32_t nd = n - (sgn ? 1u : 0u);  // digits
    bool ok = nd - 1u < 19u && !(nd > 1 && t.sb[c.fend - min(nd, 24u)] == '0');
    uint32_t g0v = 0;
    ok = digits8_window<true>(wlo, whi, min(nd, 8u), &g0v) && ok;
    // six digits and more are out of range whatever they are (no leading zeros): the narrow digit sum reads five
    if constexpr (KIND == CK_INT) ok = ok && nd <= 5 && g0v <= (1u << (W * 8 - 1)) - (neg ? 0u : 1u);
    else ok = ok && nd <= 5 && g0v <= (1u << (W * 8)) - 1u && !(neg && g0v != 0);
    if (c.on && ok) out[c.r] = (T)(neg ? -(int32_t)g0v : (int32_t)g0v);
    bad = !ok;
  }
  if (c.on && bad) t.slowf[c.jj] = 1;
}
template <int KIND, int W> __device__ __forceinline__ void reg_cells_int(const RegTile &t, const CsvRun &run, uint32_t sbase, int wv, int lane) {
  using T = std::conditional_t<W == 1, int8_t, std::conditional_t<W == 2, int16_t, std::conditional_t<W == 4, int32_t, int64_t>>>;
  for_slots(t, run, sbase, wv, lane, [&](const RegCell &c) { reg_int_body<KIND, W, T>(t, c, global_ptr<T>(t.colp0[c.on ? c.col : (uint32_t)run.first])); });
}

__device__ __forceinline__ void reg_str_body(const RegTile &t, const RegCell &c, uint32_t *const lens, uint32_t *const fstart) {
  const uint32_t n = c.fend - c.fs;
  const uint32_t c_first = t.sb[c.fs], c_last = t.sb[c.fend - 1];
  const uint32_t qn = quotes_in(t, c.fs, c.fend);  // quote characters in the field
  // straight path: empty; no quote inside and printable ASCII (not a space, not DEL) at both ends; or enclosed in the
  // only two quotes it holds ("…": unquote, nothing to collapse)
  const bool plain = qn == 0 && c_first - 0x21u < 0x5Eu && c_last - 0x21u < 0x5Eu;
  const bool quoted = qn == 2 && c_first == t.quote && c_last == t.quote;  // (n >= 2 then: two quote characters are in it)
  if (wave_all(n == 0 || plain || quoted || !c.on)) {
    const uint32_t q1 = quoted ? 1u : 0u;
    if (c.on) { lens[c.r] = n - 2u * q1; fstart[c.r] = t.g0 + c.fs + q1; }
    return;
  }
  if (!c.on) return;
  if (n == 0) { lens[c.r] = 0; fstart[c.r] = t.g0 + c.fs; return; }
  bool done = false;
  if (n <= 0x7FFFu) {
    // nothing to trim: either enclosed in quotes ("…": unquote) or plain at both ends
    uint32_t a = c.fs, b = c.fend;
    bool ok = false;
    if (c_first == t.quote && c_last == t.quote && n >= 2) { a++; b--; ok = true; }
    else ok = c_first != t.quote && c_last != t.quote && starts_plain(t.sb, c.fs, c.fend, c_first, t.quote) && ends_plain(t.sb, c.fs, c.fend, c_last, t.quote);
    if (ok) {
      const uint32_t npairs = quotes_in(t, a, b) ? count_quote_pairs(t.qmask, a, b) : 0u;
      if (!(npairs && !t.double_quote)) {  // errDoubleQuotesDisabled: per-row path
        lens[c.r] = (b - a) - npairs;
        fstart[c.r] = (t.g0 + a) | (npairs ? 0x80000000u : 0u);
        if (npairs) fstart[-1] = 1u;  // the column holds cells that are not a plain byte range
        done = true;
      }
    }
  }
  if (!done) t.slowf[c.jj] = 1;
}
__device__ __forceinline__ void reg_cells_str(const RegTile &t, const CsvRun &run, uint32_t sbase, int wv, int lane) {
  for_slots(t, run, sbase, wv, lane, [&](const RegCell &c) {
    const uint32_t col = c.on ? c.col : (uint32_t)run.first;
    reg_str_body(t, c, global_ptr<uint32_t>(t.colp0[col]), global_ptr<uint32_t>(t.colp1[col]));
  });
}

// 2006-01-02 | 2006-01-02[ T]15:04:05 (cast.StringToDate layouts); a timestamp column also takes plain decimal
// seconds (parseTimestampValue, reader_csv.go:419-426)
__device__ __forceinline__ uint32_t dig2(uint32_t w, int byte, uint32_t *bad) {  // two ASCII digits at bytes byte, byte + 1 of w
  const uint32_t a = ((w >> (8 * byte)) & 0xFFu) - '0', b = ((w >> (8 * byte + 8)) & 0xFFu) - '0';
  *bad |= (a > 9u) | (b > 9u);
  return a * 10 + b;
}
template <int KIND> __device__ __forceinline__ void reg_time_body(const RegTile &t, const RegCell &c, int64_t *const sec, int32_t *const nanos) {
  if (!c.on) return;
  const uint32_t n = c.fend - c.fs;
  bool done = false;
  if ((n == 10 || n == 19) && t.sb[c.fs + 4] == '-') {
    // the 20 bytes from the field's start as five words: "2006" "-01-" "02 1" "5:04" ":05."
    const uint32_t *w = reinterpret_cast<const uint32_t *>(t.sb + (c.fs & ~3u));
    const uint32_t x0 = w[0], x1 = w[1], x2 = w[2], x3 = w[3], x4 = w[4], x5 = w[5];
    const uint32_t w0 = __builtin_amdgcn_alignbyte(x1, x0, c.fs), w1 = __builtin_amdgcn_alignbyte(x2, x1, c.fs), w2 = __builtin_amdgcn_alignbyte(x3, x2, c.fs);
    uint32_t bad = 0;
    const uint32_t y = dig2(w0, 0, &bad) * 100 + dig2(w0, 2, &bad);
    const uint32_t mo = dig2(w1, 1, &bad), d = dig2(w2, 0, &bad);
    bad |= (w1 & 0xFF0000FFu) != 0x2D00002Du;  // '-' .. '-'
    uint32_t h = 0, mi = 0, se = 0;
    if (n == 19) {
      const uint32_t w3 = __builtin_amdgcn_alignbyte(x4, x3, c.fs), w4 = __builtin_amdgcn_alignbyte(x5, x4, c.fs);
      const uint32_t sep = (w2 >> 16) & 0xFFu;
      bad |= !(sep == ' ' || sep == 'T') | (((w3 >> 8) & 0xFFu) != ':') | ((w4 & 0xFFu) != ':');
      h = (((w2 >> 24) & 0xFFu) - '0') * 10 + ((w3 & 0xFFu) - '0');
      bad |= (((w2 >> 24) & 0xFFu) - '0' > 9u) | ((w3 & 0xFFu) - '0' > 9u);
      mi = dig2(w3, 2, &bad); se = dig2(w4, 1, &bad);
    }
    const bool ok = !bad && mo >= 1 && mo <= 12 && d >= 1 && d <= days_in_month32(mo, y) && h <= 23 && mi <= 59 && se <= 59;
    if (ok) {
      sec[c.r] = (int64_t)days_from_civil32(y, mo, d) * 86400 + (int64_t)(h * 3600 + mi * 60 + se);
      nanos[c.r] = 0;
      done = true;
    }
  } else if (KIND == CK_TIMESTAMP && n - 1u < 19u) {
    // ParseInt(s, 10, 64): [+-]?digits, leading zeros are fine in base 10
    const uint32_t c0 = t.sb[c.fs];
    const bool neg = c0 == '-', sgn = neg || c0 == '+';
    const uint32_t nd = n - (sgn ? 1u : 0u);
    uint32_t g0v = 0, g1v = 0, g2v = 0;
    bool ok = nd - 1u < 18u;
    ok = digits8_end(t.sb, c.fend, min(nd, 8u), &g0v) && ok;
    if (nd > 8) ok = digits8_end(t.sb, c.fend - 8, min(nd - 8, 8u), &g1v) && ok;
    if (nd > 16) ok = digits8_end(t.sb, c.fend - 16, nd - 16, &g2v) && ok;
    if (ok) {
      const uint64_t v = ((uint64_t)g2v * 100000000ull + g1v) * 100000000ull + g0v;
      sec[c.r] = neg ? -(int64_t)v : (int64_t)v; nanos[c.r] = 0; done = true;
    }
  }
  if (!done) t.slowf[c.jj] = 1;
}
template <int KIND> __device__ __forceinline__ void reg_cells_time(const RegTile &t, const CsvRun &run, uint32_t sbase, int wv, int lane) {
  for_slots(t, run, sbase, wv, lane, [&](const RegCell &c) {
    if (!c.on) return;
    reg_time_body<KIND>(t, c, global_ptr<int64_t>(t.colp0[c.col]), global_ptr<int32_t>(t.colp1[c.col]));
  });
}

// ---- the column-lane form of the cell phase -----------------------------------------------------------------------------------
// A wave takes a TASK: G <= 64 columns of one (kind, width).  Lane l owns column first + l % G for the whole task and the lines
// jj = it * P + l / G (P = 64 / G lines per step), so everything a cell's column fixes — its field ordinal, its output pointers —
// is read ONCE per task into the lane's registers, and a step's skeleton is: the line (one add), the index entry (one mad), two
// index reads, the row.  The item form pays per cell for the (column, line) of a slot item (a carry chain), three descriptor reads
// from LDS and the slot bookkeeping of every run in every wave.  A lane's stores go to its own column: a wave store touches G
// cache lines with P consecutive values each (the tile's 35 lines of a column are 70-280 contiguous bytes: they meet in L2).
struct LaneCol { uint32_t g, ph, P; bool live; };  // P is wave-uniform (a scalar): the step loops below are scalar loops
__device__ __forceinline__ LaneCol lane_col(const CsvTask &k, int lane, int mode) {
  const uint32_t G = (uint32_t)__builtin_amdgcn_readfirstlane(k.ncols), P = 64u / G;
  if (mode == 2) {  // a column's P lines on NEIGHBOURING lanes: a wave store is G runs of P consecutive values
    const uint32_t g = (uint32_t)((uint32_t)lane * (65536u / P + 1u)) >> 16;  // lane / P
    return LaneCol{g, (uint32_t)lane - g * P, P, g < G};
  }
  const uint32_t ph = (uint32_t)((uint32_t)lane * (65536u / G + 1u)) >> 16;  // lane / G for lane < 64 (G <= 64)
  return LaneCol{(uint32_t)lane - ph * G, ph, P, ph < P};
}
__device__ __forceinline__ RegCell lane_cell(const RegTile &t, const LaneCol &lc, uint32_t jj0, uint32_t colfield) {
  RegCell c;
  c.jj = jj0 + lc.ph; c.col = 0;
  c.r = t.row0 + (int32_t)c.jj;
  c.on = lc.live && c.jj < t.nr && c.r >= 0;  // header lines produce nothing
  const uint32_t kr = c.on ? __umul24(c.jj, t.NF) + colfield : CR_KDUMMY;
  uint32_t kr1 = kr + 1;
  TF_OPAQUE(kr1);  // (two ds_read_u16: see reg_cell)
  c.fs = (uint32_t)t.fposx[kr] + 1; c.fend = t.fposx[kr1];
  return c;
}
template <int KIND, int W> __device__ __forceinline__ void task_cells_int(const RegTile &t, const CsvTask &k, int lane) {
  using T = std::conditional_t<W == 1, int8_t, std::conditional_t<W == 2, int16_t, std::conditional_t<W == 4, int32_t, int64_t>>>;
  const LaneCol lc = lane_col(k, lane, t.col_mode);
  const uint32_t col = (uint32_t)k.first + (lc.live ? lc.g : 0u);
  const uint32_t colfield = t.colfield[col];
  T *const out = global_ptr<T>(t.colp0[col]);
  for (uint32_t jj0 = 0; jj0 < t.nr; jj0 += lc.P) reg_int_body<KIND, W, T>(t, lane_cell(t, lc, jj0, colfield), out);
}
__device__ __forceinline__ void task_cells_str(const RegTile &t, const CsvTask &k, int lane) {
  const LaneCol lc = lane_col(k, lane, t.col_mode);
  const uint32_t col = (uint32_t)k.first + (lc.live ? lc.g : 0u);
  const uint32_t colfield = t.colfield[col];
  uint32_t *const lens = global_ptr<uint32_t>(t.colp0[col]), *const fstart = global_ptr<uint32_t>(t.colp1[col]);
  for (uint32_t jj0 = 0; jj0 < t.nr; jj0 += lc.P) reg_str_body(t, lane_cell(t, lc, jj0, colfield), lens, fstart);
}
template <int KIND> __device__ __forceinline__ void task_cells_time(const RegTile &t, const CsvTask &k, int lane) {
  const LaneCol lc = lane_col(k, lane, t.col_mode);
  const uint32_t col = (uint32_t)k.first + (lc.live ? lc.g : 0u);
  const uint32_t colfield = t.colfield[col];
  int64_t *const sec = global_ptr<int64_t>(t.colp0[col]); int32_t *const nanos = global_ptr<int32_t>(t.colp1[col]);
  for (uint32_t jj0 = 0; jj0 < t.nr; jj0 += lc.P) reg_time_body<KIND>(t, lane_cell(t, lc, jj0, colfield), sec, nanos);
}
__device__ __forceinline__ bool task_kind(int kind) { return kind == CK_INT || kind == CK_UINT || kind == CK_STR || kind == CK_DATE || kind == CK_TIMESTAMP; }

template <bool COLS> __device__ __forceinline__ void csv_parse_regular_body(const CsvParams &p) {
  static_assert(CT_SPILL == 64 * CT_CPT * 16, "the look-behind window is exactly wave 0's bytes");
  __shared__ __attribute__((aligned(16))) uint8_t sbuf[32 + CT_BYTES + 48];  // 32 bytes in front: the 24-byte window of a cell that ends within the tile's first bytes
  __shared__ uint16_t fposx[CR_FCAP + 4];    // [k + 1] = end of field k; [0] = first line's start - 1; [CR_KDUMMY], [CR_KDUMMY + 1]: an empty field for idle lanes
  __shared__ uint2 keeptab[9];               // [n]: mask of the top n bytes of an 8-byte window
  __shared__ uint32_t qmask[CT_BYTES / 32];  // bitmap of quote characters
  __shared__ uint32_t nlbits[(CR_FCAP + 32) / 32];  // bit k + 1: field k ends its line
  __shared__ uint16_t qpre[CT_BYTES / 32 + 2];      // quote characters in front of each 32-byte word
  __shared__ uint32_t wqc[CT_THREADS / 64];
  __shared__ uint8_t slowf[CT_RCAP];         // line needs the per-row path
  __shared__ uint64_t colp0[CR_LCOLS], colp1[CR_LCOLS];
  __shared__ uint16_t colfield[CR_LCOLS];
  __shared__ uint32_t wpar[CT_THREADS / 64];
  __shared__ uint32_t wcnt[CT_THREADS / 64];
  __shared__ uint32_t misc[4];               // 0: first line's start, 1: tile is not regular, 2: NF
  __shared__ uint32_t spec_sh[2];            // single-pass form: 0 = this workgroup's ticket, 1 = lines in front of its tile
  uint8_t *const sb = sbuf + 32;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool spec = p.spec != nullptr;

  int64_t tile;
  if (spec) {  // tiles in arrival order (see spec_lookback)
    if (tid == 0) spec_sh[0] = atomicAdd(p.spec, 1u);
    __syncthreads();
    tile = (int64_t)spec_sh[0];
  } else {  // XCD-aware tile order: consecutive tiles (which share their look-behind bytes) run on one XCD's L2
    const int64_t per_xcd = (p.ntiles + 7) / 8;
    tile = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  }
  if (tile >= p.ntiles) return;
  // the tile's lines from the exclusive scan of the newline counts per 4 KiB granule: wave-uniform addresses, read by the
  // scalar unit now — a vector load at the end of the kernel would wait for every store of the cell phase
  constexpr int64_t GPT = CT_T / CSV_GRAN;
  const int64_t gr0 = tile * GPT, gr1 = min(gr0 + GPT, p.ngran);
  const auto *gpre = TF_CONST_PTR(uint32_t, p.gran_pre);
  uint32_t line0 = 0, line1 = 0, lines_all = 0, nlines = 0;
  if (!spec) {
    line0 = gpre[gr0]; line1 = gpre[gr1]; lines_all = gpre[p.ngran];
    nlines = line1 - line0;
    if (nlines == 0) return;  // no line ends here (inside a very long line)
    if (p.ovf && line1 > p.cap_lines) { if (tid == 0) *p.ovf = 1u; return; }  // sized-ahead form: more lines than the buffers hold (uniform)
  }
  const CsvOpts &o = p.o;
  const int64_t g0 = tile * CT_T - CT_SPILL;  // absolute offset of sb[0]
  // a piece for the general kernel: (first granule, granules) and, in the single-pass form, (first line, lines)
  auto give_up = [&]() {
    if (tid == 0) {
      const uint32_t slot = atomicAdd(p.gen_n, 1u);
      if (spec) { p.gen_tile[4 * slot] = (uint32_t)gr0; p.gen_tile[4 * slot + 1] = (uint32_t)(gr1 - gr0); p.gen_tile[4 * slot + 2] = line0; p.gen_tile[4 * slot + 3] = nlines; }
      else { p.gen_tile[2 * slot] = (uint32_t)gr0; p.gen_tile[2 * slot + 1] = (uint32_t)(gr1 - gr0); }
    }
  };
  unsigned long long tstamp = p.dbg_phase ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull;
  auto phase = [&](int k) {  // TFGPU_CSV_PHASES=1, profiling only: thread 0 adds the cycles since the previous stamp to slot k
    if (p.dbg_phase && tid == 0) { const unsigned long long now = (unsigned long long)__builtin_amdgcn_s_memtime(); atomicAdd(p.dbg_phase + k, now - tstamp); tstamp = now; }
  };
  if (!spec && (p.force_general || nlines > (uint32_t)CT_RCAP || p.ncols > CR_LCOLS)) { give_up(); return; }  // (the host does not pick the single-pass form with force_general or that many columns)

  // ---- stage: coalesced 16 B/lane ----
  if (g0 >= 0 && (uint64_t)(g0 + CT_BYTES) <= p.len) {
    const uint8_t *src = p.data + g0;
#pragma unroll
    for (int it = 0; it < CT_CPT; it++) {
      const int chunk = it * CT_THREADS + tid;
      *reinterpret_cast<uint4 *>(sb + chunk * 16) = *reinterpret_cast<const uint4 *>(src + chunk * 16);
    }
  } else {
#pragma unroll
    for (int it = 0; it < CT_CPT; it++) {
      const int chunk = it * CT_THREADS + tid;
      const int64_t gp = g0 + (int64_t)chunk * 16;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gp >= 0 && (uint64_t)gp < p.len) v = *reinterpret_cast<const uint4 *>(p.data + gp);  // buffer is zero-padded past len
      *reinterpret_cast<uint4 *>(sb + chunk * 16) = v;
    }
  }
  for (int i = tid; i < p.ncols; i += CT_THREADS) { const TCol tc = p.tcols[i]; colp0[i] = (uint64_t)tc.p0; colp1[i] = (uint64_t)tc.p1; colfield[i] = (uint16_t)tc.field; }
  if (tid < 4) misc[tid] = 0;
  if (tid < 12) reinterpret_cast<uint32_t *>(sb + CT_BYTES)[tid] = 0;  // the bytes past the tile that cell windows may touch
  if (tid >= 64 && tid < 72) reinterpret_cast<uint32_t *>(sbuf)[tid - 64] = 0;  // … and those in front of it
  if (tid == 80) { fposx[CR_KDUMMY] = 8; fposx[CR_KDUMMY + 1] = 9; }
  if (tid >= 96 && tid < 105) { const int n8 = tid - 96; const uint64_t k = n8 ? ~0ull << (8 * (8 - n8)) : 0ull; keeptab[n8] = make_uint2((uint32_t)k, (uint32_t)(k >> 32)); }
  for (int i = tid; i < (CR_FCAP + 32) / 32; i += CT_THREADS) nlbits[i] = 0;
  __syncthreads();
  phase(0);
  if (CSV_ABL(p) == 1) return;  // (TFGPU_CSV_ABLATE, profiling only: leave after phase n; results are not valid)

  // ---- pass 1: per-byte classes of this thread's 64 bytes, as 16-bit masks per 16-byte chunk ----
  const int base_chunk = tid * CT_CPT;
  uint32_t qm[CT_CPT], dm[CT_CPT], nl[CT_CPT];
  uint32_t bad = 0;
  {
    // (An escape character in front of a quote — the one thing the plain parity cannot express — is looked for where the
    //  tile's quote characters are walked, in the index sweep.)
    const uint32_t q4 = o.quote * 0x01010101u, d4 = o.delim * 0x01010101u;
#pragma unroll
    for (int q = 0; q < CT_CPT; q++) {
      const uint4 v = *reinterpret_cast<const uint4 *>(sb + (base_chunk + q) * 16);
      const Chunk16 ck = chunk16(v);
      qm[q] = class16c(ck, q4); dm[q] = class16c(ck, d4); nl[q] = class16c(ck, 0x0A0A0A0Au);
      reinterpret_cast<uint16_t *>(qmask)[base_chunk + q] = (uint16_t)qm[q];
    }
  }
  if (wv == 0) {
    // the first line this tile owns starts behind the last '\n' of the look-behind window; what precedes it belongs to
    // the tile before (the highest lane holding a '\n' holds the last one)
    int mx = -1;
#pragma unroll
    for (int q = 0; q < CT_CPT; q++) if (nl[q]) mx = (base_chunk + q) * 16 + 31 - __clz((int)nl[q]);
    const uint64_t holders = __ballot(mx >= 0);
    int frs = 0;
    if (holders) frs = __shfl(mx, 63 - __clzll((long long)holders), 64) + 1;
    else if (g0 <= 0) frs = (int)(-g0);  // the buffer starts inside the window: line 0 starts at absolute 0
    else bad |= 1u;                      // the first line started before the window: general path
#pragma unroll
    for (int q = 0; q < CT_CPT; q++) {
      const int cpos = (base_chunk + q) * 16;
      uint32_t keep = 0xFFFFu;
      if (cpos + 16 <= frs) keep = 0; else if (cpos < frs) keep = 0xFFFFu & ~((1u << (frs - cpos)) - 1u);
      qm[q] &= keep; dm[q] &= keep; nl[q] &= keep;
    }
    if (lane == 0) misc[0] = (uint32_t)frs;
  }
  uint32_t par = 0;
#pragma unroll
  for (int q = 0; q < CT_CPT; q++) par ^= (uint32_t)__popc(qm[q]);
  par &= 1u;
  const uint64_t pb = __ballot(par != 0);
  const uint32_t par_in = lanes_below(pb) & 1u;  // parity of the quotes of this wave's bytes before this thread's
  if (lane == 0) wpar[wv] = (uint32_t)__popcll(pb) & 1u;
  if (__any(bad != 0) && lane == 0) misc[1] = 1u;
  __syncthreads();
  phase(1);
  if (CSV_ABL(p) == 2) { if (par_in == 77u + qm[0] + dm[1] + nl[2] + qm[3] + dm[0] + nl[0]) p.err[0] = 1; return; }

  // ---- pass 2: field ends = delimiters outside quotes, and every '\n' ----
  uint32_t s_in = par_in;
  for (int i = 0; i < wv; i++) s_in ^= wpar[i];
  uint32_t fe[CT_CPT];
  uint32_t cnt = 0;
  bad = 0;
#pragma unroll
  for (int q = 0; q < CT_CPT; q++) {
    uint32_t px = qm[q];  // bit i = parity of the quotes in bytes [0, i]
    px ^= px << 1; px ^= px << 2; px ^= px << 4; px ^= px << 8;
    const uint32_t inq = (px ^ (s_in ? 0xFFFFu : 0u)) & 0xFFFFu;
    bad |= nl[q] & inq;  // a '\n' inside quotes cuts the line and resets the state: not the plain parity any more
    fe[q] = (dm[q] & ~inq) | nl[q];
    s_in = inq >> 15;
    cnt += (uint32_t)__popc(fe[q]) | ((uint32_t)__popc(nl[q]) << 16);
  }
  const uint32_t cinc = wave_scan_add(cnt);
  const uint32_t qlo = (uint32_t)__popc(qmask[tid * 2]), qc = qlo + (uint32_t)__popc(qmask[tid * 2 + 1]);  // (all of the tile's quotes, also those in front of the first line)
  const uint32_t qinc = wave_scan_add(qc);
  if (lane == 63) { wcnt[wv] = cinc; wqc[wv] = qinc; }
  if (__any(bad != 0) && lane == 0) misc[1] = 1u;
  __syncthreads();
  phase(2);

  uint32_t cpre = 0, ctot = 0;
  for (int i = 0; i < CT_THREADS / 64; i++) { const uint32_t x = wcnt[i]; if (i < wv) cpre += x; ctot += x; }
  const uint32_t nfe = ctot & 0xFFFFu, nr = ctot >> 16;
  if (spec) {
    // the tile's own count is final here, whatever becomes of the tile: publish it and find the lines in front.  The last wave
    // does it (its value reaches the others at the next barrier); on the early exits below it is the one that stays.
    nlines = nr;
    const bool quit = misc[1] || nfe > (uint32_t)CR_FCAP || nr > (uint32_t)CT_RCAP || nr == 0;
    if (wv == CT_THREADS / 64 - 1) {
      line0 = spec_lookback(p.spec + 2, tile, nr, lane);
      if (lane == 0) spec_sh[1] = line0;
      if (quit && nr != 0 && lane == 0) {
        if (line0 + nr > p.cap_lines) p.spec[1] = 1u;
        else { const uint32_t slot = atomicAdd(p.gen_n, 1u); p.gen_tile[4 * slot] = (uint32_t)gr0; p.gen_tile[4 * slot + 1] = (uint32_t)(gr1 - gr0); p.gen_tile[4 * slot + 2] = line0; p.gen_tile[4 * slot + 3] = nr; }
      }
    }
    if (quit) return;
  } else if (misc[1] || nfe > (uint32_t)CR_FCAP || nr != nlines) { give_up(); return; }  // uniform: LDS flags were written before the barrier
  const uint32_t frs = misc[0];
  if (CSV_ABL(p) == 3) { if (fe[0] + fe[1] + fe[2] + fe[3] == 0xFFFFFFFFu) p.err[0] = 1; return; }

  // ---- index: one sweep over this thread's field ends, two chunks (32 bytes) per loop ----
  {
    const uint32_t cex = cpre + cinc - cnt;
    uint32_t k = cex & 0xFFFFu;
#pragma unroll
    for (int h = 0; h < CT_CPT; h += 2) {
      uint32_t m = fe[h] | (fe[h + 1] << 16);
      const uint32_t nlh = nl[h] | (nl[h + 1] << 16);
      const uint32_t cpos = (uint32_t)(base_chunk + h) * 16;
      if (nlh) {  // the rare '\n's: which ordinals end a line; the first of the tile gives NF
        uint32_t mm = nlh, j = (cex >> 16) + (h ? (uint32_t)__popc(nl[0] | (nl[1] << 16)) : 0u);
        while (mm) {
          const uint32_t b = (uint32_t)__ffs((int)mm) - 1; mm &= mm - 1;
          const uint32_t kk = k + (uint32_t)__popc(m & ((2u << b) - 1u));  // = ordinal + 1
          atomicOr(&nlbits[kk >> 5], 1u << (kk & 31));
          if (j == 0) misc[2] = kk;
          j++;
        }
      }
      while (m) {
        const uint32_t b = (uint32_t)__ffs((int)m) - 1; m &= m - 1;
        fposx[++k] = (uint16_t)(cpos + b);
      }
      if (o.escape != 0) {  // an escape character directly in front of a quote (reader.go:233-240) needs the byte-wise state machine: general path
        uint32_t mm = qm[h] | (qm[h + 1] << 16);
        while (mm) {
          const uint32_t b = (uint32_t)__ffs((int)mm) - 1; mm &= mm - 1;
          if (sb[(int)(cpos + b) - 1] == o.escape) misc[1] = 1u;  // (read after the cells; until then they are computed optimistically)
        }
      }
    }
    if (tid == 0) fposx[0] = (uint16_t)(frs - 1);
    for (uint32_t i = tid; i < nr; i += CT_THREADS) slowf[i] = 0;
    uint32_t qb = qinc - qc;
    for (int i = 0; i < wv; i++) qb += wqc[i];
    qpre[tid * 2] = (uint16_t)qb; qpre[tid * 2 + 1] = (uint16_t)(qb + qlo);
    if (tid == CT_THREADS - 1) qpre[CT_BYTES / 32] = (uint16_t)(qb + qc);
  }
  __syncthreads();
  phase(3);
  if (CSV_ABL(p) == 4) return;
  if (spec) {
    line0 = spec_sh[1]; line1 = line0 + nr;
    if (line1 > p.cap_lines) { if (tid == 0) p.spec[1] = 1u; return; }  // more lines than the buffers were sized for: the host parses the chunk again
  }
  const uint32_t NF = misc[2];  // fields of the first line; every line must have as many
  if (NF < 2 || NF < (uint32_t)p.nfields_used || nr * NF > nfe) { give_up(); return; }  // uniform.  (A lone "\n" is a line of no fields, reader.go:146-150.)
  for (uint32_t jj = tid; jj < nr; jj += CT_THREADS) {
    const uint32_t idx = (jj + 1) * NF;
    if (idx > nfe || !((nlbits[idx >> 5] >> (idx & 31)) & 1u)) misc[1] = 1u;  // read after the next barrier; until then the cells are computed optimistically
  }

  // ---- cells, run by run: a run is a set of columns of one kind and width, its items (column, line) with lines
  //      fastest, so that column stores are coalesced; 64 items are one slot, and the slots of all runs are dealt
  //      round-robin to the waves ----
  {
    RegTile t;
    t.sb = sb; t.fposx = fposx; t.qmask = qmask; t.qpre = qpre; t.keeptab = keeptab; t.slowf = slowf; t.colp0 = colp0; t.colp1 = colp1; t.colfield = colfield;
    t.nr = nr; t.NF = NF; t.step_q = 64u * CT_WAVES / nr; t.step_r = 64u * CT_WAVES - t.step_q * nr; t.inv_nr = __uint_as_float(__float_as_uint(1.0f / (float)nr) - 2u);  // a hair below 1 / nr: the quotient estimate never overshoots
    t.row0 = (int32_t)((int64_t)line0 - p.skip_rows); t.g0 = (uint32_t)g0; t.quote = o.quote; t.double_quote = o.double_quote != 0; t.ablate = CSV_ABL(p); t.col_mode = p.col_lanes;
    if constexpr (COLS) {
      for (int ti = p.wave_task[wv]; ti < p.wave_task[wv + 1]; ti++) {
        const CsvTask k = p.tasks[ti];
        if (CSV_ABL(p) >= 10) {  // 10: no cells; 11: integer tasks only; 12: text; 13: date / timestamp
          const int grp = (k.kind == CK_INT || k.kind == CK_UINT) ? 11 : k.kind == CK_STR ? 12 : 13;
          if (CSV_ABL(p) != grp && !(CSV_ABL(p) == 31 && grp == 11)) continue;
        }
        switch (k.kind) {
          case CK_INT:
            if (k.width == 1) task_cells_int<CK_INT, 1>(t, k, lane); else if (k.width == 2) task_cells_int<CK_INT, 2>(t, k, lane);
            else if (k.width == 4) task_cells_int<CK_INT, 4>(t, k, lane); else task_cells_int<CK_INT, 8>(t, k, lane);
            break;
          case CK_UINT:
            if (k.width == 1) task_cells_int<CK_UINT, 1>(t, k, lane); else if (k.width == 2) task_cells_int<CK_UINT, 2>(t, k, lane);
            else if (k.width == 4) task_cells_int<CK_UINT, 4>(t, k, lane); else task_cells_int<CK_UINT, 8>(t, k, lane);
            break;
          case CK_STR: task_cells_str(t, k, lane); break;
          case CK_DATE: task_cells_time<CK_DATE>(t, k, lane); break;
          default: task_cells_time<CK_TIMESTAMP>(t, k, lane);
        }
      }
    }
    uint32_t sbase = 0;
    for (int ri = 0; ri < p.nruns; ri++) {
      const CsvRun run = p.runs[ri];
      if (COLS && task_kind(run.kind)) continue;  // its cells ran as tasks
      const uint32_t items = (uint32_t)run.ncols * nr, nslots = (items + 63) >> 6;
      if (CSV_ABL(p) >= 10) {  // 10: no cells; 11: integer runs only; 12: text runs only; 13: date / timestamp runs only
        const int grp = (run.kind == CK_INT || run.kind == CK_UINT) ? 11 : run.kind == CK_STR ? 12 : 13;
        if (CSV_ABL(p) != grp && !(CSV_ABL(p) == 31 && grp == 11)) { sbase += nslots; continue; }
      }
      switch (run.kind) {
        case CK_INT:
          if (run.width == 1) reg_cells_int<CK_INT, 1>(t, run, sbase, wv, lane); else if (run.width == 2) reg_cells_int<CK_INT, 2>(t, run, sbase, wv, lane);
          else if (run.width == 4) reg_cells_int<CK_INT, 4>(t, run, sbase, wv, lane); else reg_cells_int<CK_INT, 8>(t, run, sbase, wv, lane);
          break;
        case CK_UINT:
          if (run.width == 1) reg_cells_int<CK_UINT, 1>(t, run, sbase, wv, lane); else if (run.width == 2) reg_cells_int<CK_UINT, 2>(t, run, sbase, wv, lane);
          else if (run.width == 4) reg_cells_int<CK_UINT, 4>(t, run, sbase, wv, lane); else reg_cells_int<CK_UINT, 8>(t, run, sbase, wv, lane);
          break;
        case CK_STR: reg_cells_str(t, run, sbase, wv, lane); break;
        case CK_DATE: reg_cells_time<CK_DATE>(t, run, sbase, wv, lane); break;
        case CK_TIMESTAMP: reg_cells_time<CK_TIMESTAMP>(t, run, sbase, wv, lane); break;
        default:
          for (uint32_t s = sbase + (((uint32_t)wv - sbase) & (CT_WAVES - 1u)); s < sbase + nslots; s += CT_WAVES) {
            const uint32_t it = (s - sbase) * 64 + (uint32_t)lane;
            if (it >= items) continue;
            const uint32_t oi = it / nr, jj = it - oi * nr;
            const int32_t r = t.row0 + (int32_t)jj;
            if (r < 0) continue;
            if (run.kind < 0) {  // DefaultValue columns (ColSchema.Path < 0)
              const TCol tc = p.tcols[run.first + oi];
              CsvCol c{}; c.kind = tc.kind; c.width = tc.width; c.values = tc.p0; c.nanos = (int32_t *)tc.p1; c.lens = (uint32_t *)tc.p0; c.fstart = (uint32_t *)tc.p1;
              store_default(c, r);
            } else slowf[jj] = 1;  // bool, float32, json.Number, interval cells: the per-row path carries their rules
          }
      }
      sbase += nslots;
    }
    // ---- sanitizeElement also runs on fields no column reads: anything but a plain field flags the line ----
    if (p.has_unmapped || NF != (uint32_t)p.nfields_used) {
      const uint32_t tot2 = nr * NF;
      const uint32_t inv_nf = 0xFFFFFFFFu / NF + 1;
      for (uint32_t it = tid; it < tot2; it += CT_THREADS) {
        uint32_t jj = __umulhi(it, inv_nf);
        uint32_t f = it - jj * NF;
        if (f >= NF) { jj++; f -= NF; }
        if (t.row0 + (int32_t)jj < 0) continue;
        if (f < (uint32_t)p.nfields_used && p.field_first[f] >= 0) continue;  // judged by its column
        const uint32_t fs = (uint32_t)fposx[it] + 1, fend = fposx[it + 1];
        if (fend > fs && !(starts_plain(sb, fs, fend, sb[fs], o.quote) && ends_plain(sb, fs, fend, sb[fend - 1], o.quote) && !any_quote(qmask, fs, fend))) slowf[jj] = 1;
      }
    }
  }
  phase(4);  // (thread 0's own cells)
  __syncthreads();
  phase(5);
  if (misc[1]) { give_up(); return; }  // some line has another number of fields, or a quote is escaped: nothing of the above counts
  // ---- per line: clean, or handed to the per-row path ----
  for (uint32_t jj = tid; jj < nr; jj += CT_THREADS) {
    const int64_t r = (int64_t)line0 + jj - p.skip_rows;
    if (r < 0) continue;
    if (slowf[jj] || p.null_checks) {
      const uint32_t slot = atomicAdd(p.slow_n, 1u);
      p.slow_row[slot] = line0 + jj;
      p.slow_end[slot] = (uint32_t)(g0 + fposx[(jj + 1) * NF]);
    } else p.err[r] = 0;
  }
  // one past the last '\n' of the chunk: only the tile that owns the last line has it (30 000 atomics on one word are not free)
  // (single-pass form: nobody knows the total yet; every tile leaves its own last '\n', csv_collect takes the last tile's that has one)
  if (tid == 0) { if (spec) p.spec[2 + p.ntiles + tile] = (uint32_t)(g0 + fposx[nr * NF] + 1); else if (line1 == lines_all) atomicMax(p.last_end, (uint32_t)(g0 + fposx[nr * NF] + 1)); }
  phase(6);
}

// the kernel of the bench line: the item form of the cell phase
__global__ void __launch_bounds__(CT_THREADS, 6) csv_parse_regular(CsvParams p) { csv_parse_regular_body<false>(p); }
// … and with the column-lane form (TFGPU_CSV_COL_LANES=1 | 2): built, measured, slower — see the host code
__global__ void __launch_bounds__(CT_THREADS, 6) csv_parse_cols(CsvParams p) { csv_parse_regular_body<true>(p); }

#include "tf_csv_lanes.inc"

// rows that failed contribute no string bytes
// (sized-ahead form: `nrows` is what the buffers were sized for, err[] beyond the true count — lines_total - skip — was written by nobody)
__global__ void csv_zero_err_lens(const uint8_t *err, int64_t nrows, const CsvCol *cols, int32_t ncols, const uint32_t *lines_total, int64_t skip) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (lines_total) nrows = min(nrows, max((int64_t)*lines_total - skip, (int64_t)0));
  if (r >= nrows || !err[r]) return;
  for (int32_t ci = 0; ci < ncols; ci++) if (cols[ci].lens) cols[ci].lens[r] = 0;
}

static constexpr int FS_HDR = 4;  // words in front of a column's fstart array; the last one is its flag word
struct CopyCol { const uint32_t *offsets; const uint32_t *fstart; uint8_t *out; int32_t is_jsonnum; const uint32_t *patch; uint32_t dp_len; };
struct CopyParams { const uint8_t *data; int64_t nrows; const CopyCol *cols; int32_t ncols; uint8_t quote; };

// lane = row, wave-uniform loop over string columns.  8-byte words where the
// destination is aligned, byte stores for head/tail and for ""-collapse.
__global__ void __launch_bounds__(256) csv_copy_strings(CopyParams p) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  MemBytes src(p.data);
  for (int32_t ci = 0; ci < p.ncols; ci++) {
    const CopyCol &c = p.cols[ci];
    uint32_t o0 = c.offsets[r], n = c.offsets[r + 1] - o0;
    if (!n) continue;
    uint32_t fsv = c.fstart[r];
    uint8_t *dst = c.out + o0;
    if ((fsv & 0x7FFFFFFFu) == 0x7FFFFFFFu) { if (c.is_jsonnum) dst[0] = '0'; continue; }  // DefaultValue of a double is json.Number("0")
    uint64_t s = fsv & 0x7FFFFFFFu;
    if (c.patch && c.patch[r] != PATCH_NONE) {  // DecimalPoint: the k bytes in front of it, a '.', the rest
      const uint32_t k = c.patch[r];
      for (uint32_t w = 0; w < n; w++) dst[w] = w < k ? (uint8_t)src.at(s + w) : w == k ? (uint8_t)'.' : (uint8_t)src.at(s + w + c.dp_len - 1);
      continue;
    }
    if (fsv & 0x80000000u) {  // collapse doubled quotes: "" → "
      uint32_t w = 0;
      while (w < n) {
        uint32_t ch = src.at(s);
        if (ch == p.quote && src.at(s + 1) == p.quote) { dst[w++] = '"'; s += 2; }
        else { dst[w++] = (uint8_t)ch; s++; }
      }
      continue;
    }
    uint32_t i = 0;
    while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 7)) { dst[i] = (uint8_t)src.at(s + i); i++; }
    for (; i + 8 <= n; i += 8) {
      uint64_t w = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) w |= (uint64_t)src.at(s + i + k) << (8 * k);
      *reinterpret_cast<uint64_t *>(dst + i) = w;
    }
    for (; i < n; i++) dst[i] = (uint8_t)src.at(s + i);
  }
}

// String payloads, destination-centric (tf_segcopy.hpp): blockIdx.y = string column, blockIdx.x = a run of
// 256 * RPT rows whose cells are pulled straight out of the CSV in HBM at fstart[row].  Cells that are not a plain
// byte range (doubled quotes, DefaultValue) are zero-filled here and written by csv_copy_special afterwards.
template <int RPT> __global__ void __launch_bounds__(256) csv_copy_words(CopyParams p) {
  __shared__ uint32_t doff[256 * RPT + 1];
  __shared__ uint32_t soff[256 * RPT];
  const CopyCol c = p.cols[blockIdx.y];
  auto so = [&](int64_t g) { const uint32_t f = c.fstart[g]; return cell_plain(f) ? f : SEG_NONE; };
  segcopy_run<RPT>(c.offsets, p.nrows, (int64_t)blockIdx.x * 256 * RPT, p.data, c.out, so, doff, soff);
}
// Short, mostly empty cells (the ~24 sparse text columns of `hits`: 1.3 B per row on average): cell-centric, lane = row.
// Nine lanes in ten have nothing to move; the tenth moves its few bytes as (unaligned) 8-byte words.
static constexpr int CC_GROUP = 8;  // columns per thread: their offsets are fetched together, one memory round trip
__global__ void __launch_bounds__(256) csv_copy_cells(CopyParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  const int c0 = blockIdx.y * CC_GROUP;
  uint32_t o0[CC_GROUP], n[CC_GROUP];
#pragma unroll
  for (int j = 0; j < CC_GROUP; j++) {
    n[j] = 0; o0[j] = 0;
    if (c0 + j < p.ncols) { const uint32_t *off = p.cols[c0 + j].offsets; o0[j] = off[r]; n[j] = off[r + 1] - o0[j]; }
  }
  struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
#pragma unroll
  for (int j = 0; j < CC_GROUP; j++) {
    if (!n[j]) continue;
    const CopyCol c = p.cols[c0 + j];
    const uint32_t fsv = c.fstart[r];
    if (!cell_plain(fsv)) continue;  // csv_copy_special
    const uint8_t *src = p.data + fsv;
    uint8_t *dst = c.out + o0[j];
    uint32_t i = 0;
    for (; i + 8 <= n[j]; i += 8) reinterpret_cast<U64 *>(dst + i)->v = reinterpret_cast<const U64 *>(src + i)->v;
    if (i < n[j]) { uint64_t x = reinterpret_cast<const U64 *>(src + i)->v; for (; i < n[j]; i++) { dst[i] = (uint8_t)x; x >>= 8; } }
  }
}

// Cells csv_copy_words leaves zero-filled: ""-collapse (swapToSingleQuotes, reader.go:307-320) and the
// DefaultValue of a double (json.Number "0") — tf_textview.hpp.  lane = row to find them.
__global__ void __launch_bounds__(256) csv_copy_special(CopyParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const CopyCol c = p.cols[blockIdx.y];
  uint32_t fsv = 0, o0 = 0, n = 0;
  if (r < p.nrows) {
    fsv = c.fstart[r];
    if (!cell_plain(fsv)) { o0 = c.offsets[r]; n = c.offsets[r + 1] - o0; }
  }
  text_copy_special_wave(p.data, p.quote, c.out, c.is_jsonnum != 0, fsv, o0, n, threadIdx.x & 63);
}

// one contiguous summary for a single read-back: [nerr, consumed, total bytes of each string column]
// (single-pass form: `nrows` is what the buffers were sized for; the true count is the last tile's inclusive line count)
__global__ void csv_collect(const uint32_t *nerr, const uint32_t *last, const uint32_t *lens_all, int64_t seg_stride, int64_t nrows, int nstr, const uint32_t *fstart_all,
                            int64_t fstride, uint32_t *out, const uint32_t *spec, int64_t ntiles, int64_t skip, const uint32_t *lines_total, const uint32_t *ovf) {
  int i = threadIdx.x;
  if (ovf) {  // sized-ahead form: `nrows` is what the buffers were sized for; the true count is csv_count_newlines' total
    const uint32_t nl = *lines_total;
    if (i == 2) { out[2 + 2 * nstr] = nl; out[3 + 2 * nstr] = *ovf; }
    nrows = min(nrows, max((int64_t)nl - skip, (int64_t)0));
    if (i == 1) out[1] = *last;
  } else if (spec) {
    const uint32_t nl = ntiles ? (spec[2 + ntiles - 1] & SP_VAL) : 0u;
    if (i == 2) { out[2 + 2 * nstr] = nl; out[3 + 2 * nstr] = spec[1]; }
    nrows = min(nrows, max((int64_t)nl - skip, (int64_t)0));
    // the chunk's last '\n': the highest tile that owns one (all 64 lanes)
    uint32_t le = 0;
    for (int64_t hi = ntiles - 1; hi >= 0 && le == 0; hi -= 64) {
      const int64_t t = hi - i;
      const uint32_t v = t >= 0 ? spec[2 + ntiles + t] : 0u;
      const uint64_t has = __ballot(v != 0);
      if (has) le = (uint32_t)__shfl((int)v, __ffsll((long long)has) - 1, 64);
    }
    if (i == 1) out[1] = le;
  } else if (i == 1) out[1] = *last;
  if (i == 0) out[0] = *nerr;
  for (int s = i; s < nstr; s += blockDim.x) { out[2 + s] = lens_all[(int64_t)s * seg_stride + nrows]; out[2 + nstr + s] = fstart_all[(int64_t)s * fstride + FS_HDR - 1]; }
}
// everything the parse kernels count into, in one launch: the text columns' flag words, the two hand-over lists' counters (and the
// consumed offset behind the first), the error count, the sized-ahead form's overflow word
__global__ void csv_zero_flags(uint32_t *fstart_all, int64_t fstride, int nstr, uint32_t *slow, uint32_t *gen, uint32_t *nerr, uint32_t *ovf) {
  for (int s = threadIdx.x; s < nstr; s += blockDim.x) fstart_all[(int64_t)s * fstride + FS_HDR - 1] = 0;
  if (threadIdx.x < 2) { slow[threadIdx.x] = 0; gen[threadIdx.x] = 0; }
  if (threadIdx.x == 2) *nerr = 0;
  if (threadIdx.x == 3 && ovf) *ovf = 0;
}
__global__ void csv_keep_from_err(const uint8_t *err, int64_t n, uint32_t *keep) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) keep[r] = err[r] ? 0u : 1u;
}

__global__ void csv_shift_rows(const uint32_t *row_start, int64_t n, uint32_t *out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) out[i] = row_start[i];
}

// ---- NewlinesInValue (readMultiline, reader.go:110-137): a row is the physical lines from one where no quote is open to the
//      first after which checkCompleteQuotes holds; empty physical lines are skipped.  Per physical line the quote state
//      machine is a function {outside, inside} → {outside, inside} (the escape rule only acts inside quotes, so both entry
//      states are walked at once); composing those functions in line order gives every line's entry state. ----
__global__ void __launch_bounds__(256) csv_ml_line_fn(const uint8_t *data, const uint32_t *row_start, int64_t nlines, uint8_t quote, uint8_t escape, uint8_t *fn) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlines) return;
  const uint64_t a = row_start[i], b = row_start[i + 1];
  if (b - a <= 1) { fn[i] = 2u | 4u; return; }  // identity, blank
  MemBytes m(data);
  uint32_t q0 = 0, q1 = 1, p0 = '\n', p1 = '\n';  // checkCompleteQuotes :190-218 under both entry states; prev = the previous line's '\n'
  for (uint64_t k = a; k < b; k++) {
    const uint32_t ch = m.at(k);
    if (!(escape != 0 && p0 == escape && q0)) { if (quote != 0 && ch == quote) q0 ^= 1u; }
    if (!(escape != 0 && p1 == escape && q1)) { if (quote != 0 && ch == quote) q1 ^= 1u; }
    p0 = ch; p1 = ch;
  }
  fn[i] = (uint8_t)(q0 | (q1 << 1));
}
// one workgroup walks all lines, 1024 at a time: entry state of every line, then row begins / ends
__global__ void __launch_bounds__(1024) csv_ml_rows(const uint8_t *fn, const uint32_t *row_start, int64_t nlines, uint32_t *row_begin, uint32_t *row_end, uint32_t *nrows_out) {
  __shared__ uint32_t wf[16], wb[16], we[16];
  __shared__ uint32_t carry[3];  // state, rows begun, rows ended
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) { carry[0] = 0; carry[1] = 0; carry[2] = 0; }
  __syncthreads();
  for (int64_t base = 0; base < nlines; base += 1024) {
    const int64_t i = base + tid;
    const uint32_t f = i < nlines ? fn[i] : 2u;  // identity past the end
    // inclusive scan of function composition within the wave (log steps), then across the 16 waves
    uint32_t inc = f & 3u;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc = qf_compose(t, inc); }
    if (lane == 63) wf[wv] = inc;
    __syncthreads();
    uint32_t pre = 2u;
    for (int k = 0; k < wv; k++) pre = qf_compose(pre, wf[k]);
    uint32_t ex = __shfl_up(inc, 1, 64);
    if (lane == 0) ex = 2u;
    const uint32_t before = qf_compose(pre, ex);            // everything of this chunk in front of line i
    const uint32_t s_in = (before >> carry[0]) & 1u;         // applied to the state the chunk was entered with
    const uint32_t s_out = ((f & 3u) >> s_in) & 1u;
    const bool blank = (f & 4u) != 0 || i >= nlines;
    const uint32_t isb = (!blank && s_in == 0) ? 1u : 0u, ise = (!blank && s_out == 0) ? 1u : 0u;
    const uint32_t packed = wave_scan_add(isb | (ise << 16));
    if (lane == 63) { wb[wv] = packed & 0xFFFFu; we[wv] = packed >> 16; }
    __syncthreads();
    uint32_t pb = carry[1], pe = carry[2];
    for (int k = 0; k < wv; k++) { pb += wb[k]; pe += we[k]; }
    if (isb) row_begin[pb + (packed & 0xFFFFu) - 1] = row_start[i];
    if (ise) row_end[pe + (packed >> 16) - 1] = row_start[i + 1];
    __syncthreads();
    if (tid == 1023) {
      uint32_t tb = 0, te = 0;
      for (int k = 0; k < 16; k++) { tb += wb[k]; te += we[k]; }
      carry[0] = s_out; carry[1] += tb; carry[2] += te;
    }
    __syncthreads();
  }
  if (tid == 0) *nrows_out = carry[2];  // complete rows (a row still open at the end of the chunk is not one)
}

// constructCI's system columns (reader_csv.go:275-290)
__global__ void csv_fill_row_index(uint64_t *out, int64_t n, uint64_t base) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) out[r] = base + (uint64_t)r;  // rowsCounter advances for every line read, failed or not (:217-220)
}
__global__ void csv_fill_const_text(uint32_t *off, uint8_t *data, int64_t n, const uint8_t *text, uint32_t len) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  off[r] = (uint32_t)r * len;
  if (r < n) for (uint32_t i = 0; i < len; i++) data[(uint64_t)r * len + i] = text[i];
}
// csv.Reader.Encoding (reader.go:171-179): every byte through the charmap decoder's table, as UTF-8.  Two passes over
// 4 KiB tiles: bytes out per tile, then the text.
static constexpr int ENC_TILE = 4096;
__device__ __forceinline__ uint32_t utf8_len(uint32_t r) { return r < 0x80 ? 1u : r < 0x800 ? 2u : 3u; }
__global__ void __launch_bounds__(256) csv_decode_count(const uint8_t *in, uint64_t len, const uint32_t *table, uint32_t *tile_bytes) {
  __shared__ uint32_t tab[256];
  __shared__ uint32_t tot;
  tab[threadIdx.x] = table[threadIdx.x];
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  uint32_t n = 0;
  for (int i = 0; i < ENC_TILE / 256; i++) {
    const uint64_t p = (uint64_t)blockIdx.x * ENC_TILE + (uint64_t)i * 256 + threadIdx.x;
    if (p < len) n += utf8_len(tab[in[p]]);
  }
  atomicAdd(&tot, n);
  __syncthreads();
  if (threadIdx.x == 0) tile_bytes[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(64) csv_decode_write(const uint8_t *in, uint64_t len, const uint32_t *table, const uint32_t *tile_off, uint8_t *out) {
  // one wave per tile, 64 bytes per step; positions by a ballot-free wave scan of the lengths
  uint64_t o = tile_off[blockIdx.x];
  for (int i = 0; i < ENC_TILE / 64; i++) {
    const uint64_t p = (uint64_t)blockIdx.x * ENC_TILE + (uint64_t)i * 64 + threadIdx.x;
    const uint32_t r = p < len ? table[in[p]] : 0u;
    const uint32_t l = p < len ? utf8_len(r) : 0u;
    const uint32_t inc = wave_scan_add(l);
    if (l) {
      uint8_t *d = out + o + inc - l;
      if (l == 1) d[0] = (uint8_t)r;
      else if (l == 2) { d[0] = (uint8_t)(0xC0 | (r >> 6)); d[1] = (uint8_t)(0x80 | (r & 0x3F)); }
      else { const uint32_t q = r > 0xFFFF ? 0xFFFDu : r; d[0] = (uint8_t)(0xE0 | (q >> 12)); d[1] = (uint8_t)(0x80 | ((q >> 6) & 0x3F)); d[2] = (uint8_t)(0x80 | (q & 0x3F)); }
    }
    o += __shfl(inc, 63, 64);
  }
}
__global__ void __launch_bounds__(256) csv_last_newline(const uint8_t *in, uint64_t len, uint32_t *last_plus1) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < len && in[p] == '\n') atomicMax(last_plus1, (uint32_t)p + 1u);
}

// Shared with the JSON ingest (tf_json.hip): out[0] = 0 and out[k] = offset one past the k-th '\n' of
// data[0, len) (k = 1..n); returns n.  `data` must be 16-byte aligned and zero-padded past len.
uint32_t newline_starts(const uint8_t *data, uint64_t len, Buf *out) {
  hipStream_t st = ctx().stream;
  const int64_t ntiles = (int64_t)((len + NL_TILE - 1) / NL_TILE), ngran = (int64_t)((len + CSV_GRAN - 1) / CSV_GRAN);
  uint32_t n = 0;
  Buf tile_counts = dalloc((size_t)(ngran + 1) * 4);
  if (ntiles) {
    { KernelTimer t("csv_count_newlines"); csv_count_newlines<<<(unsigned)ntiles, NL_THREADS, 0, st>>>(data, len, ptr<uint32_t>(tile_counts), ngran); }
    exclusive_scan_u32(ptr<uint32_t>(tile_counts), ptr<uint32_t>(tile_counts), ngran, true);
    const uint32_t *h = d2h_u32(ptr<uint32_t>(tile_counts) + ngran);
    tf::sync();
    n = *h;
  }
  *out = dalloc_zero((size_t)(n + 2) * 4);
  if (n) { KernelTimer t("csv_line_index"); csv_line_index<<<(unsigned)ntiles, NL_THREADS, 0, st>>>(data, len, ptr<uint32_t>(tile_counts), ptr<uint32_t>(*out)); }
  return n;
}

// Pack the lazy text columns of a batch.  `text` overrides the source pointer (a caller-owned buffer that is only
// guaranteed to live for the duration of the parse call).
static void materialize_cols(const tfgpu_dbatch &b, const std::vector<const DColumn *> *only, const uint8_t *text) {
  hipStream_t st = ctx().stream;
  const int64_t nrows = b.nrows;
  struct Grp { const uint8_t *src; uint8_t quote; std::vector<CopyCol> all, lng, sht; };
  std::vector<Grp> groups;  // by source text (one, in practice)
  auto visit = [&](const DColumn &d) {
    if (!d.lazy()) return;
    if (only && std::find(only->begin(), only->end(), &d) == only->end()) return;
    TextView &v = *d.view;
    const uint8_t *src = text ? text : ptr<uint8_t>(v.src);
    if (!src) throw Error(TFGPU_ERR_INVALID, "internal: text column " + d.name + " lost its source text");
    v.packed = dalloc(d.data_len + 8);
    CopyCol c{ptr<uint32_t>(d.offsets), ptr<uint32_t>(v.fstart), ptr<uint8_t>(v.packed), v.jsonnum ? 1 : 0, nullptr, 0};
    Grp *g = nullptr;
    for (auto &x : groups) if (x.src == src && x.quote == v.quote) g = &x;
    if (!g) { groups.push_back(Grp{src, v.quote, {}, {}, {}}); g = &groups.back(); }
    if (v.has_special) g->all.push_back(c);
    // long cells: short runs of rows, several words per lane; short (mostly empty) cells: long runs, so the fixed
    // latency of a run is spread over enough bytes
    ((d.data_len >= (uint64_t)nrows * 8) ? g->lng : g->sht).push_back(c);
  };
  for (auto &d : b.cols) visit(d);
  for (auto &d : b.old_keys) visit(d);
  if (!nrows) return;
  for (auto &g : groups) {
    Buf ball = upload_small(g.all.data(), g.all.size() * sizeof(CopyCol));
    Buf blng = upload_small(g.lng.data(), g.lng.size() * sizeof(CopyCol)), bsht = upload_small(g.sht.data(), g.sht.size() * sizeof(CopyCol));
    KernelTimer t("csv_copy_words");
    if (!g.lng.empty()) { CopyParams cp{g.src, nrows, ptr<CopyCol>(blng), (int32_t)g.lng.size(), g.quote}; csv_copy_words<1><<<dim3((unsigned)((nrows + 255) / 256), (unsigned)g.lng.size()), 256, 0, st>>>(cp); }
    if (!g.sht.empty()) { CopyParams cp{g.src, nrows, ptr<CopyCol>(bsht), (int32_t)g.sht.size(), g.quote}; csv_copy_cells<<<dim3((unsigned)((nrows + 255) / 256), (unsigned)((g.sht.size() + CC_GROUP - 1) / CC_GROUP)), 256, 0, st>>>(cp); }
    if (!g.all.empty()) {  // only the columns whose cells are not all plain byte ranges
      CopyParams cp{g.src, nrows, ptr<CopyCol>(ball), (int32_t)g.all.size(), g.quote};
      csv_copy_special<<<dim3((unsigned)((nrows + 255) / 256), (unsigned)g.all.size()), 256, 0, st>>>(cp);
    }
  }
}
void materialize(const tfgpu_dbatch &b, const std::vector<const DColumn *> *only) { materialize_cols(b, only, nullptr); }
static void materialize_from(const tfgpu_dbatch &b, const uint8_t *text) { materialize_cols(b, nullptr, text); }

}  // namespace tf

using namespace tf;

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

static inline unsigned blocks_for(int64_t n, int t) { return (unsigned)std::max<int64_t>(1, (n + t - 1) / t); }

// ---- csv.Splitter (pkg/csv/splitter.go:37-85): entries end at the '\n's that are outside double quotes ----
namespace tf {
__global__ void __launch_bounds__(256) csv_split_count(const uint8_t *data, const uint32_t *row_start, int64_t nlines, uint32_t *quotes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlines) return;
  MemBytes m(data);
  uint32_t c = 0;
  for (uint64_t k = row_start[i]; k < row_start[i + 1]; k++) c += m.at(k) == '"';
  quotes[i] = c;
}
// updateState: outsideQuote / closingQuote ⇔ an even number of '"' so far, quoteOpen ⇔ odd; ConsumeRow returns at a '\n'
// read in a state that the '\n' itself leaves outside, i.e. after an even count
__global__ void __launch_bounds__(256) csv_split_keep(const uint32_t *quotes_excl, const uint32_t *quotes, int64_t nlines, uint32_t *keep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nlines) keep[i] = ((quotes_excl[i] + quotes[i]) & 1u) ? 0u : 1u;
}
__global__ void __launch_bounds__(256) csv_split_scatter(const uint32_t *keep_scan, const uint32_t *row_start, int64_t nlines, uint32_t *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nlines && keep_scan[i] != keep_scan[i + 1]) out[keep_scan[i]] = row_start[i + 1];
}
}  // namespace tf

extern "C" int tfgpu_csv_split_rows(const void *bytes, uint64_t len, int mem, tfgpu_dbuf **row_ends, int64_t *nrows) {
  TF_API_BEGIN
  if (!row_ends || !nrows) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_csv_split_rows: null argument");
  if (len >= 0x7FFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_csv_split_rows: chunk must be < 2 GiB");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  hipStream_t st = cx.stream;
  Buf staged;
  const uint8_t *data;
  if (mem == TFGPU_MEM_HOST) {
    staged = dalloc(len + 64);
    h2d(staged->p, bytes, len);
    TF_HIP(hipMemsetAsync((char *)staged->p + len, 0, 64, st));
    data = ptr<uint8_t>(staged);
  } else {
    data = (const uint8_t *)bytes;
    if (reinterpret_cast<uintptr_t>(data) & 15) return tf::fail(TFGPU_ERR_INVALID, "csv: device buffer must be 16-byte aligned");
  }
  Buf rs;
  const uint32_t nl = newline_starts(data, len, &rs);
  auto res = std::make_unique<tfgpu_dbuf>();
  uint32_t m = 0;
  res->mem = dalloc((size_t)(nl + 1) * 4 + 64);
  if (nl) {
    Buf q = dalloc((size_t)(nl + 1) * 4), qs = dalloc((size_t)(nl + 1) * 4), keep = dalloc((size_t)(nl + 1) * 4);
    csv_split_count<<<blocks_for(nl, 256), 256, 0, st>>>(data, ptr<uint32_t>(rs), nl, ptr<uint32_t>(q));
    exclusive_scan_u32(ptr<uint32_t>(q), ptr<uint32_t>(qs), nl, false);
    csv_split_keep<<<blocks_for(nl, 256), 256, 0, st>>>(ptr<uint32_t>(qs), ptr<uint32_t>(q), nl, ptr<uint32_t>(keep));
    exclusive_scan_u32(ptr<uint32_t>(keep), ptr<uint32_t>(keep), nl, true);
    const uint32_t *h = d2h_u32(ptr<uint32_t>(keep) + nl);
    csv_split_scatter<<<blocks_for(nl, 256), 256, 0, st>>>(ptr<uint32_t>(keep), ptr<uint32_t>(rs), nl, ptr<uint32_t>(res->mem));
    tf::sync();
    m = *h;
  }
  res->size = (uint64_t)m * 4;
  *nrows = m;
  *row_ends = res.release();
  return TFGPU_OK;
  TF_API_END
}

extern "C" void tfgpu_csv_options_default(tfgpu_csv_options *o) {  // csv.NewReader reader.go:337-350
  std::memset(o, 0, sizeof *o);
  o->delimiter = ','; o->quote_char = '"'; o->escape_char = '\\'; o->double_quote = 1;
}


// `allow_spec`: the single-pass form may be used (csv_parse_regular counts its own lines, the buffers sized from the lane's
// previous chunk).  *retry = the chunk held more lines than that (or ended oddly): nothing was returned, call again without it.
static int csv_parse_body(const tfgpu_csv_options *opts, const tfgpu_schema *schema, const void *bytes, uint64_t len, int mem,
                          tfgpu_dbatch **out, uint64_t *consumed, tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs, bool allow_spec, bool *retry) {
  TF_API_BEGIN
  if (!opts || !schema || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_csv_parse: null argument");
  if (len >= 0x7FFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_csv_parse: chunk must be < 2 GiB (the reference reads 20 MiB chunks, chunk_reader.go:12)");
  // validDelimiter reader.go:326-328
  if (opts->delimiter == 0 || opts->delimiter == '\r' || opts->delimiter == '\n' || opts->delimiter >= 0x80)
    return tf::fail(TFGPU_ERR_CONFIG, "csv: invalid delimiter");
  const bool multiline = opts->newlines_in_value && opts->quote_char;  // ReadLine :93-99
  if (multiline && opts->escape_char == '\n') return tf::fail(TFGPU_ERR_UNSUPPORTED, "csv: NewlinesInValue with '\\n' as the escape character");
  const std::string dp = (opts->decimal_point && opts->decimal_point[0]) ? opts->decimal_point : "";
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  PinScope pin_scope;
  hipStream_t st = cx.stream;

  // ---- input in HBM, padded so 16-byte loads never run off the allocation ----
  Buf staged;
  const uint8_t *data;
  if (mem == TFGPU_MEM_HOST) {
    staged = dalloc(len + 64);
    h2d(staged->p, bytes, len);
    TF_HIP(hipMemsetAsync((char *)staged->p + len, 0, 64, st));
    data = ptr<uint8_t>(staged);
  } else {
    data = (const uint8_t *)bytes;  // tfgpu_dbuf_upload pads its buffers the same way
    if (reinterpret_cast<uintptr_t>(data) & 15) return tf::fail(TFGPU_ERR_INVALID, "csv: device buffer must be 16-byte aligned");
  }

  // ---- csv.Reader.Encoding: the whole chunk through the charmap table (lines are cut at the raw '\n' first, which is the
  //      same thing as long as '\n' decodes to itself and nothing else decodes to it) ----
  const uint8_t *raw_data = data; const uint64_t raw_len = len;
  Buf decoded, raw_keep = staged;  // (the raw bytes are read once more for the consumed offset)
  if (opts->encoding_table) {
    const uint32_t *tab = opts->encoding_table;
    for (int b = 0; b < 256; b++) if ((tab[b] == 0x0A) != (b == 0x0A)) return tf::fail(TFGPU_ERR_UNSUPPORTED, "csv: an Encoding that moves the line feed (EBCDIC code pages) is not device-resident");
    Buf btab = upload_small(tab, 256 * 4);
    const int64_t et = (int64_t)((len + ENC_TILE - 1) / ENC_TILE);
    Buf tb = dalloc((size_t)(et + 1) * 4);
    uint32_t dlen = 0;
    if (et) {
      csv_decode_count<<<(unsigned)et, 256, 0, st>>>(data, len, ptr<uint32_t>(btab), ptr<uint32_t>(tb));
      exclusive_scan_u32(ptr<uint32_t>(tb), ptr<uint32_t>(tb), et, true);
      const uint32_t *h = d2h_u32(ptr<uint32_t>(tb) + et);
      tf::sync();
      dlen = *h;
    }
    if ((uint64_t)len * 3 >= 0x7FFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "csv: chunk too large to decode in one piece");
    decoded = dalloc((size_t)dlen + 64);
    if (et) csv_decode_write<<<(unsigned)et, 64, 0, st>>>(data, len, ptr<uint32_t>(btab), ptr<uint32_t>(tb), ptr<uint8_t>(decoded));
    TF_HIP(hipMemsetAsync((char *)decoded->p + dlen, 0, 64, st));
    data = ptr<uint8_t>(decoded); len = dlen;
    staged = decoded;  // the text the columns keep alive
  }

  // ---- 1/2: line index ----
  const int64_t ntiles = (int64_t)((len + NL_TILE - 1) / NL_TILE);   // workgroups of the line-index kernels (28 KiB each)
  const int64_t ngran = (int64_t)((len + CSV_GRAN - 1) / CSV_GRAN);  // newline counts are kept per 4 KiB granule
  // TFGPU_CSV_ROWPATH=1 forces the per-row path for whole chunks (the tile path's parity cross-check)
  static const bool force_rowpath = [] { const char *e = std::getenv("TFGPU_CSV_ROWPATH"); return e && e[0] == '1'; }();
  // the tile kernels are specialised for the plain shape of the options; the rest takes the per-row path, which carries every rule
  const bool rowpath = force_rowpath || opts->escape_char == '\n' || opts->escape_char >= 0x80 || opts->quote_char >= 0x80 || opts->quote_char == 0 ||
                       opts->n_timestamp_parsers > 0 || !dp.empty() || multiline;
  // ---- the single-pass form (TFGPU_CSV_SPEC=1): a lane that has parsed a chunk of this shape before sizes the buffers from that
  //      chunk's bytes per line (+ 1/16) and lets the tile kernel count; more lines than that → *retry.
  //      OFF by default.  Measured on the MI355X (profiles/r06s_ab_csv_*.json): the count pass and its read-back go (0.163 + 0.014 ms)
  //      but csv_parse_regular grows from 0.728 to 1.014 ms — a ticket and a look-back are three to five dependent round trips
  //      to memory-side coherent words (the eight XCDs share no L2), ~8 us per workgroup of a 20 us lifetime, and with three
  //      workgroups per CU there is nobody to hide them behind.  The step: 1.77 ms two-pass, 1.89 ms single-pass. ----
  static const bool spec_off = [] { const char *e = std::getenv("TFGPU_CSV_SPEC"); return !(e && e[0] == '1'); }();
  static const bool spec_blockers = [] { for (const char *n : {"TFGPU_CSV_ABLATE", "TFGPU_CSV_GENERAL", "TFGPU_CSV_PHASES", "TFGPU_CSV_LANES"}) { const char *e = std::getenv(n); if (e && e[0] && e[0] != '0') return true; } return false; }();
  const bool spec = allow_spec && !spec_off && !spec_blockers && !rowpath && schema->ncols <= CR_LCOLS && cx.csv_hint_ncols == schema->ncols && cx.csv_hint_bpl >= 1.0 &&
                    len >= 2 * (uint64_t)CT_T && len < (1ull << 30);
  // ---- the sized-ahead form (default; TFGPU_CSV_AHEAD=0 turns it off): the two passes as they are, but a lane that has parsed a chunk
  //      of this shape before sizes the buffers from that chunk's bytes per line (+ 1/16) and does NOT wait for the count — the column plan
  //      below is made while csv_count_newlines runs, the tile kernel is queued behind it, and the true count arrives with the one
  //      read-back at the end.  A tile whose lines end beyond the buffers sets a flag and writes nothing → *retry. ----
  static const bool ahead_off = [] { const char *e = std::getenv("TFGPU_CSV_AHEAD"); return e && e[0] == '0'; }();
  const bool ahead = allow_spec && !ahead_off && !spec && !spec_blockers && !rowpath && cx.csv_hint_ncols == schema->ncols && cx.csv_hint_bpl >= 1.0 &&
                     len >= 2 * (uint64_t)CT_T && len < (1ull << 30);
  uint32_t cap_lines = 0;
  Buf tile_counts = dalloc((size_t)((spec ? 0 : ngran) + 1) * 4);
  uint32_t nlines = 0;
  if (spec || ahead) {
    const double est = (double)len / cx.csv_hint_bpl;
    cap_lines = (uint32_t)std::min<double>((double)len, est + est / 16.0 + 64.0);
    nlines = cap_lines;  // everything below is sized for this many; the true count arrives with the summary
  }
  if (!spec && ntiles) {
    { KernelTimer t("csv_count_newlines"); csv_count_newlines<<<(unsigned)ntiles, NL_THREADS, 0, st>>>(data, len, ptr<uint32_t>(tile_counts), ngran); }
    exclusive_scan_u32(ptr<uint32_t>(tile_counts), ptr<uint32_t>(tile_counts), ngran, true);
    if (!ahead) {
      const uint32_t *h = d2h_u32(ptr<uint32_t>(tile_counts) + ngran);
      tf::sync();
      nlines = *h;
    }
  }
  Buf row_start;
  if (rowpath) {
    row_start = dalloc_zero((size_t)(nlines + 2) * 4);
    if (nlines) { KernelTimer t("csv_line_index"); csv_line_index<<<(unsigned)ntiles, NL_THREADS, 0, st>>>(data, len, ptr<uint32_t>(tile_counts), ptr<uint32_t>(row_start)); }
  }
  int64_t nlogical = nlines;  // rows ReadLine yields
  Buf ml_begin, ml_end;
  if (multiline) {
    Buf fn = dalloc((size_t)nlines + 16), cnt = dalloc_zero(4);
    ml_begin = dalloc((size_t)(nlines + 1) * 4); ml_end = dalloc((size_t)(nlines + 1) * 4);
    if (nlines) {
      KernelTimer t("csv_line_index");
      csv_ml_line_fn<<<blocks_for(nlines, 256), 256, 0, st>>>(data, ptr<uint32_t>(row_start), nlines, opts->quote_char, opts->escape_char, ptr<uint8_t>(fn));
      csv_ml_rows<<<1, 1024, 0, st>>>(ptr<uint8_t>(fn), ptr<uint32_t>(row_start), nlines, ptr<uint32_t>(ml_begin), ptr<uint32_t>(ml_end), ptr<uint32_t>(cnt));
    }
    const uint32_t *h = d2h_u32(cnt->p);
    tf::sync();
    nlogical = *h;
  }
  int64_t skip = std::min<int64_t>(std::max<int64_t>(opts->skip_rows, 0), nlogical);
  int64_t nrows = nlogical - skip;
  const uint32_t *rs = multiline ? ptr<uint32_t>(ml_begin) + skip : rowpath ? ptr<uint32_t>(row_start) + skip : nullptr;

  // ---- column plan ----
  int ncols = schema->ncols;
  std::vector<CsvCol> cols((size_t)ncols);
  auto db = std::make_unique<tfgpu_dbatch>();
  db->nrows = nrows;
  int nstr = 0;
  for (int i = 0; i < ncols; i++) {
    int k = schema->cols[i].dtype;
    const std::string nm = schema->cols[i].name ? schema->cols[i].name : "";
    if (nm == "__file_name" || nm == "__row_index") continue;
    if (k == TFGPU_T_UTF8 || k == TFGPU_T_BYTES || k == TFGPU_T_ANY || k == TFGPU_T_FLOAT64) nstr++;
  }
  int64_t seg_stride = ((nrows + 1 + 3) / 4) * 4;
  // One arena for everything the parse kernels write per column (values, nanos, text lengths and positions): the column
  // buffers are views into it — one block from the lane's cache per chunk instead of two hundred.
  auto a256 = [](size_t n) { return (n + 255) & ~(size_t)255; };
  const size_t lens_bytes = a256((size_t)std::max(nstr, 1) * (size_t)seg_stride * 4 + 16 + 64);
  size_t arena_bytes = lens_bytes;
  {
    const int64_t fstride0 = ((std::max<int64_t>(nrows, 1) + 3) / 4) * 4 + FS_HDR;
    arena_bytes += a256((size_t)std::max(nstr, 1) * (size_t)fstride0 * 4 + 64);
    for (int i = 0; i < ncols; i++) {
      const int k = schema->cols[i].dtype;
      const std::string nm = schema->cols[i].name ? schema->cols[i].name : "";
      if (nm == "__file_name" || nm == "__row_index" || k == TFGPU_T_UTF8 || k == TFGPU_T_BYTES || k == TFGPU_T_ANY || k == TFGPU_T_FLOAT64) continue;
      arena_bytes += a256((size_t)std::max<int64_t>(nrows, 1) * 8 + 64) + a256((size_t)std::max<int64_t>(nrows, 1) * 4 + 64);  // at most 8-byte values + nanos
    }
  }
  if (arena_bytes >= (16ull << 30)) return tf::fail(TFGPU_ERR_UNSUPPORTED, "csv: the columns of one chunk exceed 16 GiB");
  Buf arena = dalloc(arena_bytes);
  size_t arena_off = 0;
  auto carve = [&](size_t bytes) { Buf b = subbuf(arena, arena_off, bytes); arena_off += a256(bytes + 64); return b; };  // (64 bytes of slack: kernels read whole aligned words around payloads)
  Buf lens_all = carve((size_t)std::max(nstr, 1) * (size_t)seg_stride * 4 + 16);
  const int64_t fstride = ((std::max<int64_t>(nrows, 1) + 3) / 4) * 4 + FS_HDR;
  Buf fstart_all = carve((size_t)std::max(nstr, 1) * (size_t)fstride * 4);
  int max_field = -1;
  int si = 0;
  std::vector<int> str_col_index;
  std::vector<std::pair<int, int>> sys_cols;  // (schema index, 1 = __file_name | 2 = __row_index)
  for (int i = 0; i < ncols; i++) {
    const tfgpu_colschema &sc = schema->cols[i];
    CsvCol &c = cols[(size_t)i];
    std::memset(&c, 0, sizeof c);
    const std::string cname = sc.name ? sc.name : "";
    const int sys = cname == "__file_name" ? 1 : cname == "__row_index" ? 2 : 0;  // s3_reader.SystemColumnNames: no CSV field, no Path
    if (sys) {
      c.field = -1; c.next = -1; c.kind = CK_SYS;
      DColumn d;
      d.name = cname; d.dtype = sc.dtype; d.repr = sys == 1 ? TFGPU_R_STRING : TFGPU_R_UINT64;
      sys_cols.push_back({i, sys});
      db->cols.push_back(std::move(d));
      continue;
    }
    const char *path = sc.path ? sc.path : "";
    char *endp = nullptr;
    long idx = std::strtol(path, &endp, 10);
    if (endp == path || *endp) return tf::fail(TFGPU_ERR_CONFIG, std::string("csv: column ") + (sc.name ? sc.name : "") + ": ColSchema.Path is not an integer (strconv.Atoi)");
    c.field = (int32_t)idx; c.next = -1;
    DColumn d;
    d.name = sc.name ? sc.name : ""; d.dtype = sc.dtype;
    switch (sc.dtype) {
      case TFGPU_T_INT8: c.kind = CK_INT; c.width = 1; c.lo = INT8_MIN; c.hi = INT8_MAX; d.repr = TFGPU_R_INT8; break;
      case TFGPU_T_INT16: c.kind = CK_INT; c.width = 2; c.lo = INT16_MIN; c.hi = INT16_MAX; d.repr = TFGPU_R_INT16; break;
      case TFGPU_T_INT32: c.kind = CK_INT; c.width = 4; c.lo = INT32_MIN; c.hi = INT32_MAX; d.repr = TFGPU_R_INT32; break;
      case TFGPU_T_INT64: c.kind = CK_INT; c.width = 8; c.lo = INT64_MIN; c.hi = INT64_MAX; d.repr = TFGPU_R_INT64; break;
      case TFGPU_T_UINT8: c.kind = CK_UINT; c.width = 1; c.hi = UINT8_MAX; d.repr = TFGPU_R_UINT8; break;
      case TFGPU_T_UINT16: c.kind = CK_UINT; c.width = 2; c.hi = UINT16_MAX; d.repr = TFGPU_R_UINT16; break;
      case TFGPU_T_UINT32: c.kind = CK_UINT; c.width = 4; c.hi = UINT32_MAX; d.repr = TFGPU_R_UINT32; break;
      case TFGPU_T_UINT64: c.kind = CK_UINT; c.width = 8; c.hi = ~0ull; d.repr = TFGPU_R_UINT64; break;
      case TFGPU_T_BOOLEAN: c.kind = CK_BOOL; c.width = 1; d.repr = TFGPU_R_BOOL; break;
      case TFGPU_T_DATE: case TFGPU_T_DATETIME: c.kind = CK_DATE; c.width = 8; d.repr = TFGPU_R_TIME; break;
      case TFGPU_T_TIMESTAMP: c.kind = CK_TIMESTAMP; c.width = 8; d.repr = TFGPU_R_TIME; break;
      case TFGPU_T_FLOAT32: c.kind = CK_F32; c.width = 4; d.repr = TFGPU_R_FLOAT32; break;
      case TFGPU_T_FLOAT64: c.kind = CK_JSONNUM; d.repr = TFGPU_R_JSONNUM; break;  // strictify: float64 → json.Number text
      case TFGPU_T_UTF8: case TFGPU_T_ANY: c.kind = CK_STR; d.repr = TFGPU_R_STRING; break;
      case TFGPU_T_BYTES: c.kind = CK_STR; d.repr = TFGPU_R_BYTES; break;
      case TFGPU_T_INTERVAL: c.kind = CK_INTERVAL; c.width = 8; d.repr = TFGPU_R_DURATION; break;
      default: return tf::fail(TFGPU_ERR_CONFIG, "csv: cannot strictify value of unknown type");
    }
    if (c.kind == CK_STR || c.kind == CK_JSONNUM) {
      c.lens = ptr<uint32_t>(lens_all) + (int64_t)si * seg_stride;
      c.fstart = ptr<uint32_t>(fstart_all) + (int64_t)si * fstride + FS_HDR;
      str_col_index.push_back(i);
      si++;
    } else {
      d.values = carve((size_t)std::max<int64_t>(nrows, 1) * (size_t)c.width);
      c.values = d.values->p;
      if (d.repr == TFGPU_R_TIME) { d.nanos = carve((size_t)std::max<int64_t>(nrows, 1) * 4); c.nanos = ptr<int32_t>(d.nanos); }
    }
    if (c.field > max_field) max_field = c.field;
    db->cols.push_back(std::move(d));
  }
  std::vector<int32_t> field_first((size_t)std::max(max_field + 1, 1), -1);
  for (int i = ncols - 1; i >= 0; i--) {
    int f = cols[(size_t)i].field;
    if (f < 0) continue;
    cols[(size_t)i].next = field_first[(size_t)f];
    field_first[(size_t)f] = i;
  }
  // null / true / false value lists
  std::vector<uint32_t> loff{0}; std::string ldata;
  auto add_list = [&](int n, const char *const *v) { for (int i = 0; i < n; i++) { ldata += v[i] ? v[i] : ""; loff.push_back((uint32_t)ldata.size()); } };
  add_list(opts->n_null_values, opts->null_values); add_list(opts->n_true_values, opts->true_values); add_list(opts->n_false_values, opts->false_values);
  auto up = [&](const void *src, size_t bytes) { return upload_const(src, bytes); };  // tables the kernels only read
  Buf bcols = up(cols.data(), cols.size() * sizeof(CsvCol)), bff = up(field_first.data(), field_first.size() * 4);
  Buf bloff = up(loff.data(), loff.size() * 4), bldata = up(ldata.data(), ldata.size());
  // time layouts: the user's TimestampParsers, and spf13/cast v1.7.1 StringToDate's list (caste.go timeFormats) in its order
  static const char *const CAST_LAYOUTS[] = {
      "2006-01-02", "2006-01-02T15:04:05Z07:00", "2006-01-02T15:04:05", "Mon, 02 Jan 2006 15:04:05 -0700", "Mon, 02 Jan 2006 15:04:05 MST",
      "02 Jan 06 15:04 -0700", "02 Jan 06 15:04 MST", "Monday, 02-Jan-06 15:04:05 MST", "2006-01-02 15:04:05.999999999 -0700 MST",
      "2006-01-02T15:04:05-0700", "2006-01-02 15:04:05Z0700", "2006-01-02 15:04:05", "Mon Jan _2 15:04:05 2006", "Mon Jan _2 15:04:05 MST 2006",
      "Mon Jan 02 15:04:05 -0700 2006", "2006-01-02 15:04:05Z07:00", "02 Jan 2006", "2006-01-02 15:04:05 -07:00", "2006-01-02 15:04:05 -0700",
      "3:04PM", "Jan _2 15:04:05", "Jan _2 15:04:05.000", "Jan _2 15:04:05.000000", "Jan _2 15:04:05.000000000"};
  std::vector<GtOp> gops; std::string glits; std::vector<uint16_t> gstart_user{0}, gstart_cast;
  for (int i = 0; i < opts->n_timestamp_parsers; i++) { gotime_compile(opts->timestamp_parsers[i] ? opts->timestamp_parsers[i] : "", gops, glits); gstart_user.push_back((uint16_t)gops.size()); }
  gstart_cast.push_back((uint16_t)gops.size());
  for (const char *l : CAST_LAYOUTS) { gotime_compile(l, gops, glits); gstart_cast.push_back((uint16_t)gops.size()); }
  Buf bgops = up(gops.data(), gops.size() * sizeof(GtOp)), bglits = up(glits.data(), glits.size());
  Buf bgsu = up(gstart_user.data(), gstart_user.size() * 2), bgsc = up(gstart_cast.data(), gstart_cast.size() * 2), bdp = up(dp.data(), dp.size());
  Buf patch_all;
  if (!dp.empty()) {  // where a float64 cell's decimal point string sits (reader_csv.go:363-378)
    int nf = 0;
    for (auto &c : cols) if (c.kind == CK_JSONNUM) nf++;
    patch_all = dalloc((size_t)std::max(nf, 1) * (size_t)std::max<int64_t>(nrows, 1) * 4);
    TF_HIP(hipMemsetAsync(patch_all->p, 0xFF, patch_all->bytes, st));
    int k = 0;
    for (auto &c : cols) if (c.kind == CK_JSONNUM) c.patch = ptr<uint32_t>(patch_all) + (size_t)(k++) * (size_t)std::max<int64_t>(nrows, 1);
    bcols = up(cols.data(), cols.size() * sizeof(CsvCol));
  }
  Buf err = dalloc((size_t)nrows + 16), err_col = dalloc((size_t)nrows * 4 + 16), nerr = dalloc(4);

  // lines the tile path hands to the per-row path (at most one per tile per pass) + consumed offset
  const int64_t slow_cap = (int64_t)nlines + 5 * (ngran + 1) + 8;  // every line at most once, plus long / over-wide ones per piece
  Buf slow = dalloc((size_t)(2 * slow_cap + 2) * 4);
  Buf gen = dalloc((size_t)(4 * (ngran + 2) + 2) * 4);  // [0] = count, then (first granule, granules) pairs — with (first line, lines) in the single-pass form
  const uint32_t tile_bytes = CT_T;  // csv_parse_regular's tile: seven granules
  const int64_t rtiles = (int64_t)((len + tile_bytes - 1) / tile_bytes);
  Buf spec_state;
  if (spec) { spec_state = dalloc((size_t)(2 * rtiles + 2) * 4); TF_HIP(hipMemsetAsync(spec_state->p, 0, (size_t)(2 * rtiles + 2) * 4, st)); }
  CsvParams pp;
  std::memset(&pp, 0, sizeof pp);
  pp.data = data; pp.len = len; pp.row_start = rs; pp.nrows = nrows;
  pp.row_end = multiline ? ptr<uint32_t>(ml_end) + skip : nullptr;
  pp.gran_pre = ptr<uint32_t>(tile_counts); pp.ngran = ngran; pp.tile_bytes = tile_bytes; pp.ntiles = rtiles; pp.skip_rows = skip;
  pp.slow_n = ptr<uint32_t>(slow); pp.last_end = ptr<uint32_t>(slow) + 1;
  pp.slow_row = ptr<uint32_t>(slow) + 2; pp.slow_end = ptr<uint32_t>(slow) + 2 + slow_cap;
  pp.o.delim = opts->delimiter; pp.o.quote = opts->quote_char; pp.o.escape = opts->escape_char; pp.o.double_quote = opts->double_quote;
  pp.o.include_missing = opts->include_missing_columns; pp.o.strings_can_be_null = opts->strings_can_be_null;
  pp.o.quoted_strings_can_be_null = opts->quoted_strings_can_be_null; pp.o.pad = 0;
  pp.o.n_null = opts->n_null_values; pp.o.n_true = opts->n_true_values; pp.o.n_false = opts->n_false_values;
  pp.o.list_off = ptr<uint32_t>(bloff); pp.o.list_data = ptr<uint8_t>(bldata);
  pp.o.user_tp = GtSet{ptr<GtOp>(bgops), ptr<uint8_t>(bglits), ptr<uint16_t>(bgsu), opts->n_timestamp_parsers};
  pp.o.cast_tp = GtSet{ptr<GtOp>(bgops), ptr<uint8_t>(bglits), ptr<uint16_t>(bgsc), (int32_t)(sizeof CAST_LAYOUTS / sizeof *CAST_LAYOUTS)};
  pp.o.dp = ptr<uint8_t>(bdp); pp.o.dp_len = (uint32_t)dp.size(); pp.o.multiline = multiline ? 1u : 0u;
  pp.o.p128 = nullptr;
  for (auto &c : cols) if (c.kind == CK_F32) pp.o.p128 = reinterpret_cast<const uint64_t *>(pow10_table() + 632);
  pp.cols = ptr<CsvCol>(bcols); pp.ncols = ncols; pp.field_first = ptr<int32_t>(bff); pp.nfields_used = max_field + 1;
  pp.err = ptr<uint8_t>(err); pp.err_col = ptr<int32_t>(err_col); pp.nerr = ptr<uint32_t>(nerr);
#ifdef TF_CSV_ABLATE_BUILD
  static const int ablate = [] { const char *e = std::getenv("TFGPU_CSV_ABLATE"); return e ? std::atoi(e) : 0; }();
#else
  static const int ablate = [] { const char *e = std::getenv("TFGPU_CSV_ABLATE"); if (e && std::atoi(e)) std::fprintf(stderr, "tfgpu: TFGPU_CSV_ABLATE needs the ablate build of tf_csv.hip (tools/build_variant.sh ablate tf_csv.hip -DTF_CSV_ABLATE_BUILD=1); ignored\n"); return 0; }();
#endif
  pp.ablate = ablate;
  static const bool force_general = [] { const char *e = std::getenv("TFGPU_CSV_GENERAL"); return e && e[0] == '1'; }();
  pp.gen_n = ptr<uint32_t>(gen); pp.gen_tile = ptr<uint32_t>(gen) + 1; pp.force_general = force_general ? 1 : 0;
  pp.spec = spec ? ptr<uint32_t>(spec_state) : nullptr; pp.cap_lines = cap_lines;
  Buf ahead_ovf;
  if (ahead) { ahead_ovf = dalloc(4); pp.ovf = ptr<uint32_t>(ahead_ovf); }
  csv_zero_flags<<<1, 64, 0, st>>>(ptr<uint32_t>(fstart_all), fstride, nstr, ptr<uint32_t>(slow), ptr<uint32_t>(gen), ptr<uint32_t>(nerr), pp.ovf);
  pp.has_unmapped = 0;
  for (int f = 0; f <= max_field; f++) if (field_first[(size_t)f] < 0) pp.has_unmapped = 1;
  std::vector<int32_t> order((size_t)ncols);
  for (int i = 0; i < ncols; i++) order[(size_t)i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
    const CsvCol &a = cols[(size_t)x], &b = cols[(size_t)y];
    return a.kind != b.kind ? a.kind < b.kind : a.width != b.width ? a.width < b.width : a.field < b.field;
  });
  std::vector<TCol> tcols((size_t)ncols);
  for (int i = 0; i < ncols; i++) {
    const CsvCol &c = cols[(size_t)order[(size_t)i]];
    TCol &t = tcols[(size_t)i];
    const bool text = c.kind == CK_STR || c.kind == CK_JSONNUM;
    t.p0 = text ? (void *)c.lens : c.values; t.p1 = text ? (void *)c.fstart : (void *)c.nanos;
    if (c.field > 32767) return tf::fail(TFGPU_ERR_UNSUPPORTED, "csv: ColSchema.Path above 32767");
    t.field = (int16_t)c.field; t.kind = (uint8_t)c.kind; t.width = (uint8_t)c.width; t.ci = order[(size_t)i];
  }
  Buf btcols = up(tcols.data(), tcols.size() * sizeof(TCol));
  pp.tcols = ptr<TCol>(btcols);
  static const bool phases = [] { const char *e = std::getenv("TFGPU_CSV_PHASES"); return e && e[0] == '1'; }();
  Buf bphase;
  if (phases) { bphase = dalloc_zero(8 * 8); pp.dbg_phase = ptr<unsigned long long>(bphase); }
  std::vector<CsvRun> runs;
  for (int i = 0; i < ncols; i++) {
    const TCol &t = tcols[(size_t)i];
    const int kind = t.field < 0 ? -1 : (int)t.kind, width = t.field < 0 ? 0 : (int)t.width;
    if (!runs.empty() && runs.back().kind == kind && runs.back().width == width && runs.back().first + runs.back().ncols == i) runs.back().ncols++;
    else runs.push_back(CsvRun{kind, width, i, 1});
  }
  Buf bruns = up(runs.data(), runs.size() * sizeof(CsvRun));
  pp.runs = ptr<CsvRun>(bruns); pp.nruns = (int32_t)runs.size();
  // ---- the column-lane cell phase: runs of the fast kinds cut into tasks of <= 64 columns (pieces of about 16: four lines per step),
  //      dealt to the eight waves longest first (TFGPU_CSV_COL_LANES=0: the item form, for A/B runs and as the cross-check) ----
  // OFF by default.  Measured on the MI355X (gpurun r07s / r07t, 2^20 hits rows, one box): 0.946 ms (pieces of 16 columns), 0.816 (8), 0.825 (4),
  // 1.23 (32) against the item form's 0.741 — although a step issues a third fewer VALU instructions (≈ 30 against ≈ 48 in the one-to-four-digit
  // tier).  Group by group (TFGPU_CSV_ABLATE): integers +0.30 ms against +0.17, text +0.48 against +0.15, times +0.10 against +0.065.  A tile holds
  // 35 lines: a task is a handful of DEPENDENT steps (index reads → window → store) in ONE wave, seven other waves of the workgroup wait at the
  // barrier behind the longest task, and lines-per-step rounding wastes a quarter of the lanes; the item form deals 58 independent slots round-robin
  // to all eight waves.  Which lanes hold a column's lines (1 | 2) makes no difference: the stores are not what it waits for.
  static const int col_lanes = [] { const char *e = std::getenv("TFGPU_CSV_COL_LANES"); return e ? std::atoi(e) : 0; }();  // 0: item form; 1: columns on neighbouring lanes; 2: a column's lines on neighbouring lanes
  std::vector<CsvTask> tasks; std::vector<int32_t> wave_task(CT_THREADS / 64 + 1, 0);
  if (col_lanes) {
    struct Pending { CsvTask k; int cost; };
    std::vector<Pending> pend;
    for (auto &r : runs) {
      if (!(r.kind == CK_INT || r.kind == CK_UINT || r.kind == CK_STR || r.kind == CK_DATE || r.kind == CK_TIMESTAMP)) continue;
      const int unit = (r.kind == CK_INT || r.kind == CK_UINT) ? (r.width == 8 ? 5 : r.width == 4 ? 3 : 2) : r.kind == CK_STR ? 4 : 6;  // relative cost of a cell
      static const int piece = [] { const char *e = std::getenv("TFGPU_CSV_COL_PIECE"); return e ? std::max(1, std::min(64, std::atoi(e))) : 16; }();
      const int pieces = r.ncols <= piece + piece / 3 ? 1 : (r.ncols + piece - 1) / piece;
      for (int q = 0, at = 0; q < pieces; q++) {
        const int n = (r.ncols - at + (pieces - q) - 1) / (pieces - q);
        pend.push_back({CsvTask{r.kind, r.width, r.first + at, n}, n * unit});
        at += n;
      }
    }
    std::stable_sort(pend.begin(), pend.end(), [](const Pending &a, const Pending &b) { return a.cost > b.cost; });
    const int nw = CT_THREADS / 64;
    std::vector<int> load((size_t)nw, 0); std::vector<std::vector<CsvTask>> per((size_t)nw);
    for (auto &q : pend) { int w = 0; for (int i = 1; i < nw; i++) if (load[(size_t)i] < load[(size_t)w]) w = i; load[(size_t)w] += q.cost; per[(size_t)w].push_back(q.k); }
    for (int w = 0; w < nw; w++) { wave_task[(size_t)w] = (int32_t)tasks.size(); for (auto &k : per[(size_t)w]) tasks.push_back(k); }
    wave_task[(size_t)nw] = (int32_t)tasks.size();
  }
  if (tasks.empty()) tasks.push_back(CsvTask{0, 0, 0, 1});
  Buf btasks = up(tasks.data(), std::max<size_t>(tasks.size(), 1) * sizeof(CsvTask)), bwt = up(wave_task.data(), wave_task.size() * 4);
  pp.tasks = ptr<CsvTask>(btasks); pp.wave_task = ptr<int32_t>(bwt); pp.col_lanes = col_lanes;
  pp.null_checks = (opts->strings_can_be_null || opts->quoted_strings_can_be_null) ? 1 : 0;

  // ---- csv_parse_lanes: the sorted columns cut into 16 contiguous blocks of about equal cost, one per wave; the tile sized so
  //      that it holds just under 64 lines (or a multiple), within the index's capacity ----
  // (Measured on the MI355X, profiles/r06c-e: 0.765 ms per 2^20 hits rows against csv_parse_regular's 0.728 — the cells issue a third
  //  fewer VALU instructions, the byte phases of a 1024-thread workgroup more; both kernels end at 4.2e8 VALU wave-instructions and
  //  issue VALU 91 % of the time.  It stays selectable, TFGPU_CSV_LANES=1, as the base for the next step)
  static const bool lanes_on = [] { const char *e = std::getenv("TFGPU_CSV_LANES"); return e && e[0] == '1'; }();
  bool use_lanes = lanes_on && !rowpath && ncols <= CL_MAXCOLS && nlines > 0;
  int64_t ltiles = 0;
  Buf blcols, blwn;
  if (use_lanes) {
    auto cost = [](const TCol &t) { return t.field < 0 ? 1 : (t.kind == CK_INT || t.kind == CK_UINT) ? (t.width == 8 ? 3 : 2) : t.kind == CK_STR ? 4 : 8; };
    int total = 0;
    for (auto &t : tcols) total += cost(t);
    std::vector<uint32_t> lcols((size_t)CL_WAVES * 64, 0u), lwn((size_t)CL_WAVES, 0u);
    int ci = 0, spent = 0;
    for (int w = 0; w < CL_WAVES && ci < ncols; w++) {
      // columns for this wave: until its share of the remaining cost is used up (at least one, at most CL_WCOLS, and never
      // fewer than what the remaining waves can still take)
      const int share = (total - spent + (CL_WAVES - w) - 1) / (CL_WAVES - w);
      int got = 0, s = 0;
      while (ci < ncols && s < CL_WCOLS && (s == 0 || got + cost(tcols[(size_t)ci]) <= share || ncols - ci > (CL_WAVES - w - 1) * CL_WCOLS)) {
        const TCol &t = tcols[(size_t)ci];
        const uint64_t o0 = t.p0 ? (uint64_t)((uint8_t *)t.p0 - (uint8_t *)arena->p) : 0, o1 = t.p1 ? (uint64_t)((uint8_t *)t.p1 - (uint8_t *)arena->p) : 0;
        if ((o0 | o1) & 3) { use_lanes = false; break; }
        lcols[(size_t)w * 64 + 4 * s] = (uint32_t)(o0 >> 2); lcols[(size_t)w * 64 + 4 * s + 1] = (uint32_t)(o1 >> 2);
        lcols[(size_t)w * 64 + 4 * s + 2] = (uint32_t)(uint16_t)t.field | ((uint32_t)t.kind << 16) | ((uint32_t)t.width << 24);
        got += cost(t); s++; ci++;
      }
      lwn[(size_t)w] = (uint32_t)s; spent += got;
    }
    if (ci < ncols) use_lanes = false;
    if (use_lanes) {
      // lines per tile: the largest tile whose expected line count stays under 64 m - margin for some m, and whose expected
      // field count fits the index; rows longer than ~700 bytes get the largest tile whatever its count
      const double avg_row = (double)len / (double)nlines, nf_est = std::max<double>(max_field + 1, 1);
      uint32_t cpt = 1;
      for (uint32_t c = CL_CPT_MAX; c >= 1; c--) {
        const double lines = cl_tile(c) / avg_row;
        if (lines * nf_est <= 0.88 * CL_FCAP && lines <= 0.88 * CL_RCAP) { cpt = c; break; }
      }
      pp.tile_cpt = cpt;
      ltiles = (int64_t)((len + cl_tile(cpt) - 1) / cl_tile(cpt));
      pp.ntiles = ltiles;
      pp.arena = (uint8_t *)arena->p;
      blcols = up(lcols.data(), lcols.size() * 4); blwn = up(lwn.data(), lwn.size() * 4);
      pp.lcols = ptr<uint32_t>(blcols); pp.lwave_ncols = ptr<uint32_t>(blwn);
    }
  }

  if (ablate) {  // profiling only: what the skipped phases would have written must not be garbage
    TF_HIP(hipMemsetAsync(lens_all->p, 0, (size_t)std::max(nstr, 1) * (size_t)seg_stride * 4, st));
    TF_HIP(hipMemsetAsync(fstart_all->p, 0, (size_t)std::max(nstr, 1) * (size_t)fstride * 4, st));
    TF_HIP(hipMemsetAsync(err->p, 0, (size_t)nrows + 16, st));
    for (auto &d : db->cols) { if (d.values) TF_HIP(hipMemsetAsync(d.values->p, 0, d.values->bytes, st)); if (d.nanos) TF_HIP(hipMemsetAsync(d.nanos->p, 0, d.nanos->bytes, st)); }
  }
  if (nrows && rowpath) {
    KernelTimer t("csv_parse_rows");
    csv_parse_rows<<<blocks_for(nrows, 256), 256, 0, st>>>(pp);
  } else if (nlines) {
    // header lines are parsed by nobody: err[] of the data rows is written by exactly one of the two kernels
    if (use_lanes) {
      const int64_t per_xcd = (ltiles + 7) / 8;
      KernelTimer t("csv_parse_lanes");
      csv_parse_lanes<<<(unsigned)(per_xcd * 8), CL_THREADS, 0, st>>>(pp);
    } else {
      const int64_t per_xcd = (rtiles + 7) / 8;
      KernelTimer t("csv_parse_regular");
      if (pp.col_lanes) csv_parse_cols<<<(unsigned)(spec ? rtiles : per_xcd * 8), CT_THREADS, 0, st>>>(pp);
      else csv_parse_regular<<<(unsigned)(spec ? rtiles : per_xcd * 8), CT_THREADS, 0, st>>>(pp);
    }
    { KernelTimer t("csv_parse_tiles_general"); csv_parse_tiles_general<<<(unsigned)std::min<int64_t>(ntiles + rtiles, 2048), CT_THREADS, 0, st>>>(pp); }
    { KernelTimer t("csv_parse_listed"); csv_parse_listed<<<(unsigned)std::min<int64_t>(blocks_for(slow_cap, 64), 1024), 64, 0, st>>>(pp); }
  }
  if (nrows && nstr) {
    csv_zero_err_lens<<<blocks_for(nrows, 256), 256, 0, st>>>(ptr<uint8_t>(err), nrows, ptr<CsvCol>(bcols), ncols, ahead ? ptr<uint32_t>(tile_counts) + ngran : nullptr, skip);
    exclusive_scan_u32_segments(ptr<uint32_t>(lens_all), nrows, nstr, seg_stride);
  } else if (nstr) {
    TF_HIP(hipMemsetAsync(lens_all->p, 0, (size_t)nstr * (size_t)seg_stride * 4, st));
  }
  // ---- one read-back: error count, consumed offset, string totals ----
  Buf summary = dalloc((size_t)(2 * nstr + 4) * 4);
  csv_collect<<<1, 64, 0, st>>>(ptr<uint32_t>(nerr), rowpath ? ptr<uint32_t>(row_start) + nlines : pp.last_end, ptr<uint32_t>(lens_all),
                                seg_stride, nrows, nstr, ptr<uint32_t>(fstart_all), fstride, ptr<uint32_t>(summary), pp.spec, rtiles, skip,
                                ptr<uint32_t>(tile_counts) + (spec ? 0 : ngran), pp.ovf);
  const uint32_t *hsum = d2h_u32(summary->p, (size_t)(2 * nstr) + 4);
  tf::sync();
  const uint32_t hnerr = hsum[0], last = hsum[1];
  if (spec || ahead) {
    const uint32_t nl_true = hsum[2 + 2 * nstr], overflow = hsum[3 + 2 * nstr];
    // more lines than the buffers hold, fewer than the header skips, or the last '\n' outside the last two tiles: the two-pass way
    if (overflow || nl_true > cap_lines || (int64_t)nl_true < skip || (nl_true && !last)) { *retry = true; return TFGPU_OK; }
    nlines = nl_true; nlogical = nl_true; nrows = (int64_t)nl_true - skip;
    db->nrows = nrows;
  }
  if (!rowpath && nlines >= 16 && last) { cx.csv_hint_bpl = (double)last / (double)nlines; cx.csv_hint_ncols = schema->ncols; }
  if (phases && !rowpath) {
    unsigned long long ph[8]; d2h(ph, bphase->p, sizeof ph); tf::sync();
    std::fprintf(stderr, "tfgpu csv phases (s_memtime ticks per workgroup, %lld tiles): stage %.0f classify %.0f fields %.0f index %.0f cells(wave 0) %.0f cells-barrier %.0f epilogue %.0f\n", (long long)rtiles,
                 (double)ph[0] / rtiles, (double)ph[1] / rtiles, (double)ph[2] / rtiles, (double)ph[3] / rtiles, (double)ph[4] / rtiles, (double)ph[5] / rtiles, (double)ph[6] / rtiles);
  }
  static const bool debug = [] { const char *e = std::getenv("TFGPU_CSV_DEBUG"); return e && e[0] == '1'; }();
  if (debug && !rowpath) {  // how the tiles and lines were routed
    uint32_t g = 0, sl = 0;
    d2h(&g, gen->p, 4); d2h(&sl, slow->p, 4);
    tf::sync();
    std::fprintf(stderr, "tfgpu csv: %s, %lld tiles of %u KiB, %u general pieces; %u lines, %u per-row\n", use_lanes ? "lanes" : "regular", (long long)(use_lanes ? ltiles : rtiles), (use_lanes ? cl_tile(pp.tile_cpt) : tile_bytes) / 1024u, g, nlines, sl);
  }

  // ---- string payloads: offsets are views into the scanned lens array.  The tile path leaves the cells where they
  //      are: a text column is (offsets, where each cell sits in the source text) until someone needs it packed
  //      (materialize(), below) — row compaction packs the kept cells straight from the text. ----
  Buf src_block = staged ? staged : find_device_block(data);  // a foreign device pointer cannot be kept alive: pack now
  static const bool force_eager = [] { const char *e = std::getenv("TFGPU_CSV_EAGER"); return e && e[0] == '1'; }();
  for (int s = 0; s < nstr; s++) {
    DColumn &d = db->cols[(size_t)str_col_index[(size_t)s]];
    d.data_len = hsum[2 + s];
    d.offsets = subbuf(lens_all, (size_t)s * (size_t)seg_stride * 4, (size_t)(nrows + 1) * 4);
    if (rowpath) continue;
    auto v = std::make_shared<TextView>();
    v->src = src_block ? src_block : nullptr;
    v->fstart = subbuf(fstart_all, ((size_t)s * (size_t)fstride + FS_HDR) * 4, (size_t)std::max<int64_t>(nrows, 1) * 4);
    v->has_special = hsum[2 + nstr + s] != 0;
    v->quote = opts->quote_char; v->jsonnum = d.repr == TFGPU_R_JSONNUM;
    d.view = std::move(v);
  }
  if (nstr && rowpath) {
    std::vector<CopyCol> cc;
    for (int s = 0; s < nstr; s++) {
      DColumn &d = db->cols[(size_t)str_col_index[(size_t)s]];
      d.data = dalloc(d.data_len + 8);
      const CsvCol &hc = cols[(size_t)str_col_index[(size_t)s]];
      cc.push_back(CopyCol{ptr<uint32_t>(d.offsets), hc.fstart, ptr<uint8_t>(d.data), d.repr == TFGPU_R_JSONNUM ? 1 : 0, hc.patch, dp.empty() ? 0u : (uint32_t)dp.size()});
    }
    Buf bcc = up(cc.data(), cc.size() * sizeof(CopyCol));
    if (nrows) {
      CopyParams cp{data, nrows, ptr<CopyCol>(bcc), nstr, opts->quote_char};
      KernelTimer t("csv_copy_strings");
      csv_copy_strings<<<blocks_for(nrows, 256), 256, 0, st>>>(cp);
    }
  } else if (nstr && (!src_block || force_eager)) {
    materialize_from(*db, data);
  }

  for (auto &sy : sys_cols) {  // constructCI :275-290
    DColumn &d = db->cols[(size_t)sy.first];
    const int64_t n1 = std::max<int64_t>(nrows, 1);
    if (sy.second == 2) {
      d.values = dalloc((size_t)n1 * 8);
      if (nrows) csv_fill_row_index<<<blocks_for(nrows, 256), 256, 0, st>>>(ptr<uint64_t>(d.values), nrows, opts->row_number_base);
    } else {
      const std::string fn = opts->file_name ? opts->file_name : "";
      Buf btext = upload_small(fn.data(), fn.size());
      d.offsets = dalloc((size_t)(nrows + 1) * 4 + 16);
      d.data_len = (uint64_t)fn.size() * (uint64_t)nrows;
      if (d.data_len >> 32) return tf::fail(TFGPU_ERR_UNSUPPORTED, "csv: __file_name column exceeds 4 GiB");
      d.data = dalloc(d.data_len + 8);
      csv_fill_const_text<<<blocks_for(nrows + 1, 256), 256, 0, st>>>(ptr<uint32_t>(d.offsets), ptr<uint8_t>(d.data), nrows, ptr<uint8_t>(btext), (uint32_t)fn.size());
    }
    if (opts->hide_system_cols) {  // config.hideSystemCols: both are nil
      d.validity = dalloc_zero((size_t)(nrows + 7) / 8 + 8);
      if (sy.second == 1) { d.data_len = 0; TF_HIP(hipMemsetAsync(d.offsets->p, 0, (size_t)(nrows + 1) * 4, st)); }
    }
  }

  std::unique_ptr<tfgpu_dbatch> result;
  int64_t ne = 0;
  if (hnerr) {  // failed lines are dropped from the batch and reported (parseCSVRows :205-218)
    std::vector<uint8_t> he((size_t)nrows); std::vector<int32_t> hc((size_t)nrows);
    d2h(he.data(), err->p, (size_t)nrows); d2h(hc.data(), err_col->p, (size_t)nrows * 4);
    Buf keep = dalloc((size_t)(nrows + 1) * 4);
    csv_keep_from_err<<<blocks_for(nrows, 256), 256, 0, st>>>(ptr<uint8_t>(err), nrows, ptr<uint32_t>(keep));
    tf::sync();
    for (int64_t r = 0; r < nrows; r++)
      if (he[(size_t)r]) { if (errs && ne < errs_cap) errs[ne] = tfgpu_row_error{r, he[(size_t)r], 0, hc[(size_t)r]}; ne++; }
    result = compact_rows(*db, keep);
  } else {
    result = std::move(db);
  }
  uint64_t consumed_raw = last;
  if (decoded) {  // offsets count the bytes read from the stream (reader.go:169): one past the last raw '\n'
    Buf lp = dalloc_zero(4);
    if (raw_len) csv_last_newline<<<blocks_for((int64_t)raw_len, 256), 256, 0, st>>>(raw_data, raw_len, ptr<uint32_t>(lp));
    const uint32_t *h = d2h_u32(lp->p);
    tf::sync();
    consumed_raw = *h;
  }
  if (consumed) *consumed = consumed_raw;
  if (nerrs) *nerrs = ne;
  *out = result.release();
  return TFGPU_OK;
  TF_API_END
}

extern "C" int tfgpu_csv_parse(const tfgpu_csv_options *opts, const tfgpu_schema *schema, const void *bytes, uint64_t len, int mem,
                               tfgpu_dbatch **out, uint64_t *consumed, tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs) {
  bool retry = false;
  int rc = csv_parse_body(opts, schema, bytes, len, mem, out, consumed, errs, errs_cap, nerrs, true, &retry);
  if (rc == TFGPU_OK && retry) { retry = false; rc = csv_parse_body(opts, schema, bytes, len, mem, out, consumed, errs, errs_cap, nerrs, false, &retry); }
  return rc;
}

// ======================================================================================================
// tfgpu_strictify — strictify.Strictify (pkg/abstract/changeitem/strictify/strictify.go:17-157) over a device batch: what the
// strictifying serializers run before they serialize (pkg/serializer/strictify.go:24-36).  Every column named by the
// TableSchema is brought to the strict Go type of its DataType.  The text → typed conversions ARE the CSV ingest's cell
// conversions (parse_cell with reader options that do nothing); the integer family converts with Go's range rules.
// ======================================================================================================
namespace tf {

enum StrictMode : int32_t { SM_TEXT = 1, SM_TEXT_JSONNUM_OUT = 2, SM_INTS = 3, SM_JSONNUM_TO_TIME = 4, SM_FAIL = 5 /* no conversion exists: every value fails */,
                            SM_FLOATS = 6 /* Go float64 / float32 values under integer / bool / float DataTypes */ };
DColumn column_to_text(const DColumn &c, int64_t n, bool to_bytes);  // tf_transform.hip: fmt's %v of integers, bools, time.Time, time.Duration = their strconv / String() forms
struct StrictCol {
  int32_t mode, src_repr;
  const void *values; const uint32_t *offsets; const uint8_t *data; const uint8_t *validity;
  CsvCol out;
};
__device__ __forceinline__ bool strict_load_int(const StrictCol &c, int64_t r, int64_t *v, uint64_t *u, bool *is_unsigned) {
  *is_unsigned = false;
  switch (c.src_repr) {
    case TFGPU_R_INT8: *v = ((const int8_t *)c.values)[r]; return true;
    case TFGPU_R_INT16: *v = ((const int16_t *)c.values)[r]; return true;
    case TFGPU_R_INT32: *v = ((const int32_t *)c.values)[r]; return true;
    case TFGPU_R_INT64: *v = ((const int64_t *)c.values)[r]; return true;
    case TFGPU_R_UINT8: case TFGPU_R_BOOL: *u = ((const uint8_t *)c.values)[r]; break;
    case TFGPU_R_UINT16: *u = ((const uint16_t *)c.values)[r]; break;
    case TFGPU_R_UINT32: *u = ((const uint32_t *)c.values)[r]; break;
    case TFGPU_R_UINT64: *u = ((const uint64_t *)c.values)[r]; break;
    default: return false;
  }
  *is_unsigned = true; *v = (int64_t)*u;
  return true;
}
// item = column * nrows + row; first_bad[column] = min over failing rows of (row << 8 | tfgpu_rowerr)
__global__ void __launch_bounds__(256) strictify_cells(CsvOpts o, const StrictCol *cols, int32_t ncols, int64_t nrows, unsigned long long *first_bad) {
  const int32_t j = (int32_t)blockIdx.y; const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (the column is the grid's y: a scalar)
  if (r >= nrows || j >= ncols) return;
  const StrictCol &c = cols[j];
  if (c.validity && !((c.validity[r >> 3] >> (r & 7)) & 1)) { if (c.mode != SM_TEXT_JSONNUM_OUT && c.mode != SM_FAIL) store_default(c.out, r); return; }  // nil stays nil
  int rc = 0;
  if (c.mode == SM_FAIL) rc = TFGPU_ROW_CAST;  // castx.ToByteSliceE of anything but []byte / string: "no known conversion"
  else if (c.mode == SM_FLOATS) {
    // cast.ToInt64E / ToUint64E / ToBoolE / ToFloat32E of a float: Go conversions (caste.go of spf13/cast); what a conversion of a NaN or of a
    // magnitude beyond the integer range yields is the machine's business, so those go back to the host
    const double f = c.src_repr == TFGPU_R_FLOAT32 ? (double)((const float *)c.values)[r] : ((const double *)c.values)[r];
    switch (c.out.kind) {
      case CK_INT:
        if (!(f > -9223372036854775808.0 && f < 9223372036854775808.0)) rc = TFGPU_ROW_HOST_FALLBACK;
        else { const int64_t x = (int64_t)f; if (x < c.out.lo || x > (int64_t)c.out.hi) rc = TFGPU_ROW_RANGE; else store_int(c.out, r, x); }
        break;
      case CK_UINT:
        if (f != f || f >= 18446744073709551616.0) rc = TFGPU_ROW_HOST_FALLBACK;
        else if (f < 0) rc = TFGPU_ROW_CAST;  // errNegativeNotAllowed
        else { const uint64_t x = (uint64_t)f; if (x > c.out.hi) rc = TFGPU_ROW_RANGE; else if (c.out.width == 8) ((uint64_t *)c.out.values)[r] = x; else store_int(c.out, r, (int64_t)x); }
        break;
      case CK_BOOL: ((uint8_t *)c.out.values)[r] = f != 0 ? 1 : 0; break;
      default: ((float *)c.out.values)[r] = (float)f;  // CK_F32
    }
  }
  else if (c.mode == SM_INTS) {
    int64_t v = 0; uint64_t u = 0; bool uns = false;
    strict_load_int(c, r, &v, &u, &uns);
    switch (c.out.kind) {
      case CK_INT:  // cast.ToInt64E of an integer kind is a Go conversion (a uint64 wraps), then toSignedInt's limits (strictify.go:159-169)
        if (v < c.out.lo || v > (int64_t)c.out.hi) rc = TFGPU_ROW_RANGE; else store_int(c.out, r, v);
        break;
      case CK_UINT:  // errNegativeNotAllowed, then toUnsignedInt's limit (:171-181)
        if (!uns && v < 0) rc = TFGPU_ROW_CAST;
        else { const uint64_t x = uns ? u : (uint64_t)v; if (x > c.out.hi) rc = TFGPU_ROW_RANGE; else if (c.out.width == 8) ((uint64_t *)c.out.values)[r] = x; else store_int(c.out, r, (int64_t)x); }
        break;
      case CK_BOOL: ((uint8_t *)c.out.values)[r] = (uns ? u != 0 : v != 0) ? 1 : 0; break;
      case CK_F32: ((float *)c.out.values)[r] = uns ? (float)(double)u : (float)(double)v; break;
      case CK_DATE: case CK_TIMESTAMP: ((int64_t *)c.out.values)[r] = v; c.out.nanos[r] = 0; break;
      default: ((int64_t *)c.out.values)[r] = v;  // CK_INTERVAL: time.Duration(v)
    }
  } else {
    const uint32_t a = c.offsets[r], n = c.offsets[r + 1] - a;
    MemBytes rd(c.data);
    const Field fv{&rd, a, n};
    if (c.mode == SM_TEXT_JSONNUM_OUT) rc = json_number_ok(fv, 0, n) ? 0 : TFGPU_ROW_CAST;  // castx.ToJSONNumberE: the text itself is the json.Number
    else if (c.mode == SM_JSONNUM_TO_TIME) {  // cast.ToTimeE(json.Number): its Int64 as Unix seconds
      int64_t sec;
      if (parse_int64(fv, 0, n, false, &sec)) rc = TFGPU_ROW_CAST; else { ((int64_t *)c.out.values)[r] = sec; c.out.nanos[r] = 0; }
    } else rc = parse_cell(o, c.out, r, fv, 0, n, 0, 0);
  }
  if (rc) atomicMin(&first_bad[j], ((unsigned long long)r << 8) | (unsigned long long)rc);
}

// castx.ToStringE of a Go float (caste.go:64-67): strconv.FormatFloat(f, 'f', -1, bits) — the text of a "utf8" column, and (through
// castx.ToJSONNumberE) the json.Number of a "double" one
__global__ void __launch_bounds__(256) strict_float_text(const void *values, int is32, const uint8_t *validity, int64_t n, uint32_t *off, uint8_t *data) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const bool nil = validity && !((validity[r >> 3] >> (r & 7)) & 1);
  const double f = is32 ? (double)((const float *)values)[r] : ((const double *)values)[r];
  if (!data) { dev::CountOut c; if (!nil) dev::fmt_float(c, f, 'f', is32 ? 32 : 64); off[r] = c.n; return; }
  if (nil) return;
  dev::StoreOut o{data + off[r]};
  dev::fmt_float(o, f, 'f', is32 ? 32 : 64);
}
static DColumn float_column_text(const DColumn &c, int64_t n) {
  hipStream_t st = ctx().stream;
  DColumn o;
  o.offsets = dalloc((size_t)(n + 1) * 4 + 16);
  const int is32 = c.repr == TFGPU_R_FLOAT32;
  if (n) strict_float_text<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(c.values->p, is32, ptr<uint8_t>(c.validity), n, ptr<uint32_t>(o.offsets), nullptr);
  exclusive_scan_u32(ptr<uint32_t>(o.offsets), ptr<uint32_t>(o.offsets), n, true);
  const uint32_t *tot = d2h_u32(ptr<uint32_t>(o.offsets) + n);
  tf::sync();
  o.data_len = *tot;
  o.data = dalloc((size_t)o.data_len + 16);
  if (n) strict_float_text<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(c.values->p, is32, ptr<uint8_t>(c.validity), n, ptr<uint32_t>(o.offsets), ptr<uint8_t>(o.data));
  o.validity = c.validity;
  return o;
}

}  // namespace tf

extern "C" int tfgpu_strictify(const tfgpu_dbatch *in, const tfgpu_schema *schema, tfgpu_dbatch **out, int64_t *bad_row, int32_t *bad_col) {
  TF_API_BEGIN
  tf::dense(in);  // its rows may still be a selection (tfgpu_dbatch::pending)
  if (!in || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_strictify: null argument");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  hipStream_t st = cx.stream;
  materialize(*in);
  const int64_t n = in->nrows;
  if (bad_row) *bad_row = -1;
  if (bad_col) *bad_col = -1;
  auto dtype_of = [&](const DColumn &c) -> int {  // tableSchema[columnName]; a column the schema does not name is left alone
    if (schema) { for (int i = 0; i < schema->ncols; i++) if (schema->cols[i].name && c.name == schema->cols[i].name) return schema->cols[i].dtype; return -1; }
    if (!in->schema.empty()) { for (auto &p : in->schema) if (p.first == c.name) return p.second; return -1; }
    return c.dtype;
  };
  auto r = std::make_unique<tfgpu_dbatch>(*in);  // shares every buffer; converted columns are replaced below
  std::vector<StrictCol> sc; std::vector<int> which;
  std::vector<size_t> float_text;  // Go floats under "utf8" / "double": FormatFloat 'f'
  std::vector<size_t> to_text;  // columns whose strict form is their text (castx.ToStringE of a number / bool / time): made after the checks, they cannot fail
  bool need_p128 = false;
  for (size_t i = 0; i < in->cols.size(); i++) {
    const DColumn &c = in->cols[i];
    const int dt = dtype_of(c);
    if (dt < 0 || dt == TFGPU_T_ANY) continue;
    DColumn &d = r->cols[i];
    d.dtype = dt;
    CsvCol oc{}; int strict = TFGPU_R_INVALID;
    switch (dt) {
      case TFGPU_T_INT8: oc.kind = CK_INT; oc.width = 1; oc.lo = INT8_MIN; oc.hi = INT8_MAX; strict = TFGPU_R_INT8; break;
      case TFGPU_T_INT16: oc.kind = CK_INT; oc.width = 2; oc.lo = INT16_MIN; oc.hi = INT16_MAX; strict = TFGPU_R_INT16; break;
      case TFGPU_T_INT32: oc.kind = CK_INT; oc.width = 4; oc.lo = INT32_MIN; oc.hi = INT32_MAX; strict = TFGPU_R_INT32; break;
      case TFGPU_T_INT64: oc.kind = CK_INT; oc.width = 8; oc.lo = INT64_MIN; oc.hi = INT64_MAX; strict = TFGPU_R_INT64; break;
      case TFGPU_T_UINT8: oc.kind = CK_UINT; oc.width = 1; oc.hi = UINT8_MAX; strict = TFGPU_R_UINT8; break;
      case TFGPU_T_UINT16: oc.kind = CK_UINT; oc.width = 2; oc.hi = UINT16_MAX; strict = TFGPU_R_UINT16; break;
      case TFGPU_T_UINT32: oc.kind = CK_UINT; oc.width = 4; oc.hi = UINT32_MAX; strict = TFGPU_R_UINT32; break;
      case TFGPU_T_UINT64: oc.kind = CK_UINT; oc.width = 8; oc.hi = ~0ull; strict = TFGPU_R_UINT64; break;
      case TFGPU_T_BOOLEAN: oc.kind = CK_BOOL; oc.width = 1; strict = TFGPU_R_BOOL; break;
      case TFGPU_T_DATE: case TFGPU_T_DATETIME: case TFGPU_T_TIMESTAMP: oc.kind = CK_DATE; oc.width = 8; strict = TFGPU_R_TIME; break;  // cast.ToTimeE: StringToDate only (no reader step)
      case TFGPU_T_FLOAT32: oc.kind = CK_F32; oc.width = 4; strict = TFGPU_R_FLOAT32; break;
      case TFGPU_T_FLOAT64: oc.kind = CK_JSONNUM; strict = TFGPU_R_JSONNUM; break;
      case TFGPU_T_UTF8: oc.kind = CK_STR; strict = TFGPU_R_STRING; break;
      case TFGPU_T_BYTES: oc.kind = CK_STR; strict = TFGPU_R_BYTES; break;
      case TFGPU_T_INTERVAL: oc.kind = CK_INTERVAL; oc.width = 8; strict = TFGPU_R_DURATION; break;
      default: return tf::fail(TFGPU_ERR_CONFIG, "tfgpu_strictify: cannot strictify value of unknown type (column " + c.name + ")");
    }
    if (c.repr == strict) continue;  // already the strict Go type
    const bool text = c.repr == TFGPU_R_STRING || c.repr == TFGPU_R_JSONNUM;
    const bool ints = (c.repr >= TFGPU_R_INT8 && c.repr <= TFGPU_R_UINT64) || c.repr == TFGPU_R_BOOL;
    auto unsupported = [&]() { return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_strictify: column " + c.name + ": a " + std::string(type_name(dt)) + " column holding Go values of representation " + std::to_string(c.repr) + " is converted on the host (cast." "To…E of that kind is not device-resident)"); };
    StrictCol s{};
    s.src_repr = c.repr; s.values = c.values ? c.values->p : nullptr; s.offsets = ptr<uint32_t>(c.offsets); s.data = ptr<uint8_t>(c.payload()); s.validity = ptr<uint8_t>(c.validity);
    if (text && (oc.kind == CK_STR)) {  // castx.ToStringE / ToByteSliceE of a string (or of a json.Number's text): the same bytes
      if (c.repr == TFGPU_R_JSONNUM && dt == TFGPU_T_BYTES) return unsupported();
      d.repr = strict;
      continue;
    }
    if (c.repr == TFGPU_R_BYTES && dt == TFGPU_T_UTF8) { d.repr = TFGPU_R_STRING; continue; }  // ToStringE([]byte) = string(b)
    if (text && oc.kind == CK_JSONNUM) { if (c.repr != TFGPU_R_STRING) continue; s.mode = SM_TEXT_JSONNUM_OUT; d.repr = TFGPU_R_JSONNUM; }
    else if (text) {
      if (c.repr == TFGPU_R_JSONNUM && oc.kind == CK_INTERVAL) return unsupported();
      s.mode = (c.repr == TFGPU_R_JSONNUM && oc.kind == CK_DATE) ? SM_JSONNUM_TO_TIME : SM_TEXT;
      if (oc.kind == CK_F32) need_p128 = true;
    } else if ((ints || c.repr == TFGPU_R_TIME || c.repr == TFGPU_R_DURATION) && dt == TFGPU_T_UTF8) {
      to_text.push_back(i);  // castx.ToStringE (caste.go:58-106): FormatInt / FormatUint / FormatBool, Time.String(), Duration.String()
      continue;
    } else if ((ints || c.repr == TFGPU_R_TIME || c.repr == TFGPU_R_DURATION || c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) && dt == TFGPU_T_BYTES) {
      s.mode = SM_FAIL;        // castx.ToByteSliceE (caste.go:16-28) takes []byte and string only: the first value fails the call
    } else if ((c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) && (dt == TFGPU_T_UTF8 || dt == TFGPU_T_FLOAT64)) {
      float_text.push_back(i);  // FormatFloat(f, 'f', -1, bits): the string, or — every such text parses (fastfloat takes "NaN" and "+Inf" too) — the json.Number
      continue;
    } else if ((c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) && (oc.kind == CK_INT || oc.kind == CK_UINT || oc.kind == CK_BOOL || oc.kind == CK_F32)) {
      s.mode = SM_FLOATS;
    } else if (ints) {
      if (oc.kind == CK_STR || oc.kind == CK_JSONNUM) return unsupported();                 // an integer under "double": castx.ToJSONNumberE of its text — host
      if (oc.kind == CK_INTERVAL && (c.repr >= TFGPU_R_UINT8 && c.repr <= TFGPU_R_UINT64)) return unsupported();
      if (c.repr == TFGPU_R_BOOL && (oc.kind == CK_DATE || oc.kind == CK_INTERVAL || oc.kind == CK_F32)) return unsupported();
      s.mode = SM_INTS;
    } else return unsupported();
    if (s.mode == SM_FAIL) {  // reached only when every row is nil: an all-nil []byte column
      d.values = nullptr; d.nanos = nullptr; d.view = nullptr; d.data_len = 0;
      d.offsets = dalloc_zero((size_t)(n + 1) * 4 + 16); d.data = dalloc(16);
      d.repr = strict;
    } else if (s.mode != SM_TEXT_JSONNUM_OUT) {
      d.values = dalloc((size_t)std::max<int64_t>(n, 1) * (size_t)oc.width);
      d.offsets = nullptr; d.data = nullptr; d.view = nullptr; d.data_len = 0; d.nanos = nullptr;
      if (oc.kind == CK_DATE) d.nanos = dalloc((size_t)std::max<int64_t>(n, 1) * 4);
      d.repr = strict;
      oc.values = d.values->p; oc.nanos = ptr<int32_t>(d.nanos);
    }
    s.out = oc;
    sc.push_back(s); which.push_back((int)i);
  }
  if (!sc.empty() && n > 0) {
    // neutral reader options: no null / true / false lists, no DecimalPoint, no user layouts — what is left of parse_cell is strictifyValue
    static const char *const CAST_LAYOUTS[] = {
        "2006-01-02", "2006-01-02T15:04:05Z07:00", "2006-01-02T15:04:05", "Mon, 02 Jan 2006 15:04:05 -0700", "Mon, 02 Jan 2006 15:04:05 MST",
        "02 Jan 06 15:04 -0700", "02 Jan 06 15:04 MST", "Monday, 02-Jan-06 15:04:05 MST", "2006-01-02 15:04:05.999999999 -0700 MST",
        "2006-01-02T15:04:05-0700", "2006-01-02 15:04:05Z0700", "2006-01-02 15:04:05", "Mon Jan _2 15:04:05 2006", "Mon Jan _2 15:04:05 MST 2006",
        "Mon Jan 02 15:04:05 -0700 2006", "2006-01-02 15:04:05Z07:00", "02 Jan 2006", "2006-01-02 15:04:05 -07:00", "2006-01-02 15:04:05 -0700",
        "3:04PM", "Jan _2 15:04:05", "Jan _2 15:04:05.000", "Jan _2 15:04:05.000000", "Jan _2 15:04:05.000000000"};
    std::vector<GtOp> gops; std::string glits; std::vector<uint16_t> gstart{0};
    for (const char *l : CAST_LAYOUTS) { gotime_compile(l, gops, glits); gstart.push_back((uint16_t)gops.size()); }
    Buf bgops = upload_const(gops.data(), gops.size() * sizeof(GtOp)), bglits = upload_const(glits.data(), glits.size()), bgs = upload_const(gstart.data(), gstart.size() * 2);
    CsvOpts o{};
    o.delim = ','; o.quote = '"';
    o.user_tp = GtSet{ptr<GtOp>(bgops), ptr<uint8_t>(bglits), ptr<uint16_t>(bgs), 0};
    o.cast_tp = GtSet{ptr<GtOp>(bgops), ptr<uint8_t>(bglits), ptr<uint16_t>(bgs), (int32_t)(sizeof CAST_LAYOUTS / sizeof *CAST_LAYOUTS)};
    o.p128 = need_p128 ? reinterpret_cast<const uint64_t *>(pow10_table() + 632) : nullptr;
    Buf bsc = upload_small(sc.data(), sc.size() * sizeof(StrictCol));
    Buf bad = dalloc(sc.size() * 8);
    TF_HIP(hipMemsetAsync(bad->p, 0xFF, sc.size() * 8, st));
    {
      KernelTimer t("strictify_cells");
      strictify_cells<<<dim3((unsigned)((n + 255) / 256), (unsigned)sc.size()), 256, 0, st>>>(o, reinterpret_cast<const StrictCol *>(bsc->p), (int32_t)sc.size(), n, reinterpret_cast<unsigned long long *>(bad->p));
    }
    std::vector<uint64_t> hb(sc.size());
    d2h(hb.data(), bad->p, hb.size() * 8);
    tf::sync();
    // the first failing value in the reference's order: rows in order, a row's columns in order
    uint64_t best = ~0ull; int bcol = -1;
    for (size_t k = 0; k < hb.size(); k++) if (hb[k] != ~0ull && ((hb[k] >> 8) < (best >> 8) || best == ~0ull)) { best = hb[k]; bcol = which[k]; }
    if (bcol >= 0) {
      const int64_t row = (int64_t)(best >> 8); const int code = (int)(best & 0xFF);
      if (bad_row) *bad_row = row;
      if (bad_col) *bad_col = bcol;
      if (code == TFGPU_ROW_HOST_FALLBACK)
        return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_strictify: row " + std::to_string(row) + ", column " + in->cols[(size_t)bcol].name + ": a value form the device does not decide (Go's decimal slow path, a free-form date): strictify this batch on the host");
      return tf::fail(TFGPU_ERR_INVALID, "failed to strictify the value of column [" + std::to_string(bcol) + "] \"" + in->cols[(size_t)bcol].name + "\": row " + std::to_string(row) + ": " + (code == TFGPU_ROW_RANGE ? "value is out of the type's range" : "unable to cast the value"));
    }
  }
  for (size_t i : float_text) {
    const int dt = r->cols[i].dtype;
    DColumn t = float_column_text(in->cols[i], n);
    t.name = in->cols[i].name; t.dtype = dt; t.repr = dt == TFGPU_T_FLOAT64 ? TFGPU_R_JSONNUM : TFGPU_R_STRING;
    r->cols[i] = std::move(t);
  }
  for (size_t i : to_text) {
    const int dt = r->cols[i].dtype;
    DColumn t = column_to_text(in->cols[i], n, false);
    t.name = in->cols[i].name; t.dtype = dt;
    r->cols[i] = std::move(t);
  }
  *out = r.release();
  return TFGPU_OK;
  TF_API_END
}
