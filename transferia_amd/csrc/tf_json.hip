// tf_json.hip — NDJSON ingest on device: parsers/generic GenericParser{Format:"json"}
// (pkg/parsers/generic/generic_parser.go: doGenericParser :519-555, Unmarshal :672-731,
// makeChangeItem :297-404, ParseVal :888-1123, addAuxFields :99-154) producing typed
// Arrow-style columns in HBM without materialising map[string]interface{} rows.
//
// The grammar is github.com/valyala/fastjson v1.6.4 (go.mod:71, not vendored): Parse /
// parseValue / parseObject / parseArray / parseRawString / parseRawNumber, skipWS,
// unescapeStringBestEffort and fastfloat.{ParseBestEffort, ParseInt64BestEffort,
// ParseUint64BestEffort}, restated from the published algorithm; the parity tests pin every
// branch to the reference's parser canon through the test-side restatement.
//
// Pipeline (all on the library stream):
//   1. newline_starts        : offsets one past every '\n'                        (reads B_json)
//   2. json_merge_bounds     : line boundaries = '\n' ends ∪ message starts (bufio.ScanLines per message)
//   3. json_segments + scan  : trimmed [start,len) of every segment, ordinal of the non-empty ones
//   4. json_parse_lines      : one lane per line: full fastjson grammar (explicit container stack),
//                              top-level keys → column through an FNV-1a table, typed cell per column
//                              DataType (Unmarshal + ParseVal), coalesced-by-ordinal column stores;
//                              text cells record (source, mode, output length)       (reads B_json, writes B_fixed)
//   5. json_finish           : key / required rules in schema order → row, `_unparsed` code, or host fallback
//   6. segmented scan of text lengths → Arrow offsets; json_copy_cells: copy / unescape / compact
//   7. validity bitmaps, dedupe-key columns, compaction of the dropped lines.
// Value forms the reference handles but this file does not (dateparse strings, nested `any`,
// `_rest` with unknown keys, floats needing more than the exact fast path, …) make the LINE a
// TFGPU_ROW_HOST_FALLBACK: the shim re-parses exactly those lines with the stock Go code.
//
// HBM-bound byte kernel: algorithmic traffic B_json + B_bin per row.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "tf_common.hpp"
#include "tf_devfmt.hpp"
#include "tf_devfloat.hpp"
#include "tf_devparse.hpp"
#include "tf_segcopy.hpp"
#include "tf_wave.hpp"
#include "tf_swar.hpp"
#include "tf_jsontile.hpp"
#include "tf_jsonquick.hpp"
#include "tf_jsonscan.hpp"  // encoding/json's grammar: what lookupComplex's json.Unmarshal accepts

namespace tf {

std::unique_ptr<tfgpu_dbatch> compact_rows(const tfgpu_dbatch &in, Buf keep);  // tf_transform.hip
uint32_t newline_starts(const uint8_t *data, uint64_t len, Buf *out);           // tf_csv.hip

enum JKind : int32_t { JK_INT, JK_UINT, JK_F64, JK_BOOL, JK_TEXT, JK_ANY, JK_DATETIME };
enum JMode : uint32_t { JM_COPY = 0, JM_UNESCAPE = 1, JM_QUOTED = 2, JM_COMPACT = 3, JM_FLOAT = 4, JM_TSKV = 5, JM_ANYCANON = 6, JM_REST = 7, JM_REST_EMPTY = 8, JM_SCRATCH = 9 /* a byte range of JParams::scratch (lookupComplex) */ };
enum JLineSt : uint8_t { JL_ROW = 0, JL_SKIPPED = 1, JL_SYNTAX = 2, JL_FALLBACK = 3 };
enum JCellSt : uint8_t { JC_NIL = 0, JC_OK = 1, JC_ERR = 2 };
enum JVal : int32_t { V_NULL, V_STR, V_NUM, V_TRUE, V_FALSE, V_OBJ, V_ARR };
static constexpr int JCODE_SKIPPED = 255;

struct JCol {
  int32_t kind, width;
  uint32_t flags;   // TFGPU_COL_KEY | TFGPU_COL_REQUIRED
  int32_t next;     // next column reading the same key, or -1
  void *values;     // fixed-width output, indexed by line ordinal
  int32_t *nanos;   // datetime
  uint32_t *lens;   // text: output length (Arrow offsets after the scan)
  uint32_t *fstart; // text: absolute offset of the source bytes
  uint32_t *meta;   // text: source length | JMode << 28
  uint8_t *cellst;  // JCellSt per line
  uint32_t scr_base;            // a nested ColSchema.Path: where this column's scratch copy of the input starts in JParams::scratch
  uint32_t npath_off, npath_n;  // a nested ColSchema.Path (lookupComplex): npath_n field names behind the top-level key, at names + npath_off as (u16 length, bytes) …
};
static constexpr uint32_t JCOL_NESTED = 0x10000u;  // JCol.flags: the column reads through lookupComplex (its ParseVal errors follow generic_parser.go:338-346)

// open-addressing table over the top-level keys the parser knows: ColPath (or ColumnName) → column
struct JSlot { uint32_t hash; int32_t col; uint32_t soff, slen; uint32_t aux; };  // slen == ~0u: empty; aux: the key is an aux column's name
static constexpr int32_t JS_KNOWN = -1;  // a known name no column reads through (p.known, :1203-1210)
static constexpr int32_t JS_UNKNOWN = -3;

struct JParams {
  const uint8_t *data;
  const uint32_t *seg_start, *seg_len, *seg_ord;  // per segment; seg_ord = exclusive scan of the non-empty flags
  int64_t nseg;
  const JCol *cols; int32_t ncols;
  const JSlot *slots; uint32_t slot_mask; const uint8_t *names;
  const double *pow10;   // math.Pow10(n) at [n + 323]
  const uint64_t *pow128; // {lo, hi} of 10^e, e = -348..347 (Eisel-Lemire)
  uint8_t *linest;       // JLineSt per line
  uint32_t *line_pos;    // start offset per line (message lookup)
  uint8_t add_rest, use_numbers;
  uint32_t *rest_lens, *rest_fstart, *rest_meta;  // `_rest` column (AddRest): text cell per line, null without it
  uint8_t format;        // TFGPU_JFMT_*
  uint8_t tskv_unescape; // Format tskv + UnescapeStringValues: tryToUnescapeTSKV on the values
  uint8_t *scratch;      // as large as the input, only with nested ColSchema.Paths: where lookupComplex's rewritten / decoded texts are built
};

__device__ __forceinline__ uint32_t fnv1a(uint32_t h, uint32_t c) { return (h ^ c) * 16777619u; }

// ---------------------------------------------------------------------------
// fastjson pieces
// ---------------------------------------------------------------------------
struct JStr { uint64_t s; uint32_t n; uint32_t hash; bool bs, ctrl, plain; };

// hash of a top-level key for the column table: 8-byte little-endian chunks of the raw key (the last one zero-padded)
// folded with the length; key_hash_host computes the same over the column paths.
__device__ __forceinline__ uint32_t key_mix(uint32_t h, uint64_t w) {
  h = (h ^ (uint32_t)w) * 0x85EBCA6Bu; h = (h ^ (uint32_t)(w >> 32)) * 0xC2B2AE35u;
  return h ^ (h >> 15);
}
__device__ __forceinline__ uint32_t key_finish(uint32_t h, uint32_t n) { return h ^ (n * 0x27D4EB2Fu); }

// parseRawString: `pos` is one past the opening quote; the closing quote is the first '"' preceded by an even
// number of backslashes, i.e. the first quote met when every backslash swallows the byte after it.  Eight bytes
// per step: the first quote / backslash of the word by SWAR zero-byte flags (the lowest flag is exact), control
// and non-ASCII bytes by SWAR range tests over the bytes before it.  false: no closing quote.
__device__ __forceinline__ bool scan_string(MemBytes &rd, uint64_t &pos, const uint64_t end, JStr &o) {
  const uint64_t ONES = 0x0101010101010101ull, HI = 0x8080808080808080ull;
  o.s = pos;
  uint32_t h = 0x9E3779B9u;
  bool bs = false, ctrl = false, hi = false;
  while (pos < end) {
    const uint64_t x = rd.word(pos);
    const uint32_t nb = end - pos < 8 ? (uint32_t)(end - pos) : 8u;
    const uint64_t xq = x ^ 0x2222222222222222ull, xb = x ^ 0x5C5C5C5C5C5C5C5Cull;
    uint64_t sp = (((xq - ONES) & ~xq) | ((xb - ONES) & ~xb)) & HI;
    if (nb < 8) sp &= (1ull << (8 * nb)) - 1;
    const uint32_t k = sp ? (uint32_t)(__ffsll((long long)sp) - 1) >> 3 : nb;  // bytes before the first quote / backslash
    if (k) {
      const uint64_t m = k >= 8 ? ~0ull : (1ull << (8 * k)) - 1;
      const uint64_t y = (x & m) | (0x3030303030303030ull & ~m);
      ctrl = ctrl || (((y - 0x2020202020202020ull) & ~y & HI) != 0);
      hi = hi || (y & HI) != 0;
      h = key_mix(h, x & m);
    }
    pos += k;
    if (!sp) continue;
    if (((x >> (8 * k)) & 0xFFu) == '"') {
      o.n = (uint32_t)(pos - o.s); o.hash = key_finish(h, o.n); o.bs = bs; o.ctrl = ctrl; o.plain = !bs && !ctrl && !hi;
      pos++;
      return true;
    }
    bs = true;  // a backslash: the byte after it can neither end the string nor start another escape
    if (pos + 1 >= end) { pos = end; return false; }
    if (rd.at(pos + 1) < 0x20u) ctrl = true;
    pos += 2;
  }
  return false;
}

__device__ __forceinline__ bool ci3(MemBytes &rd, uint64_t p, uint32_t a, uint32_t b, uint32_t c) {
  return lower_(rd.at(p)) == a && lower_(rd.at(p + 1)) == b && lower_(rd.at(p + 2)) == c;
}

// parseRawNumber: pos at the first byte of the token.  false: "unexpected char".
__device__ __forceinline__ bool scan_number(MemBytes &rd, uint64_t &pos, const uint64_t end) {
  const uint64_t b = pos;
  uint64_t i = pos;
  for (; i < end; i++) {
    const uint32_t ch = rd.at(i);
    // [0-9.+-eE] as one 64-bit table over ch - '+': '+'0 '-'2 '.'3 '0'..'9' 5..14 'E'26 'e'58
    const uint32_t d = ch - 0x2Bu;
    if (d < 64u && ((0x0400000004007FEDull >> d) & 1ull)) continue;
    const uint32_t b0 = rd.at(b);
    if (i == b || (i == b + 1 && (b0 == '-' || b0 == '+'))) {
      if (end - i >= 3 && (ci3(rd, i, 'i', 'n', 'f') || ci3(rd, i, 'n', 'a', 'n'))) { pos = i + 3; return true; }
      return false;
    }
    break;
  }
  pos = i;
  return true;
}

struct CountSink { uint32_t n = 0; __device__ __forceinline__ void put(uint32_t) { n++; } };
// bytes leave eight at a time in one (possibly unaligned) 8-byte store; finish() writes the tail
struct StoreSink {
  uint8_t *d; uint32_t n = 0; uint64_t acc = 0;
  struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
  __device__ __forceinline__ void put(uint32_t c) {
    acc |= (uint64_t)(c & 0xFFu) << (8 * (n & 7));
    if ((++n & 7) == 0) { reinterpret_cast<U64 *>(d + n - 8)->v = acc; acc = 0; }
  }
  __device__ __forceinline__ void finish() { for (uint32_t k = n & ~7u; k < n; k++) { d[k] = (uint8_t)acc; acc >>= 8; } }
};

template <class S> __device__ __forceinline__ void put_utf8(S &o, uint32_t cp) {  // string(rune(x)); invalid → U+FFFD
  if (cp > 0x10FFFF || (cp >= 0xD800 && cp < 0xE000)) cp = 0xFFFD;
  if (cp < 0x80) { o.put(cp); return; }
  if (cp < 0x800) { o.put(0xC0 | cp >> 6); o.put(0x80 | (cp & 0x3F)); return; }
  if (cp < 0x10000) { o.put(0xE0 | cp >> 12); o.put(0x80 | ((cp >> 6) & 0x3F)); o.put(0x80 | (cp & 0x3F)); return; }
  o.put(0xF0 | cp >> 18); o.put(0x80 | ((cp >> 12) & 0x3F)); o.put(0x80 | ((cp >> 6) & 0x3F)); o.put(0x80 | (cp & 0x3F));
}
__device__ __forceinline__ bool hex4(MemBytes &rd, uint64_t p, uint32_t *out) {  // strconv.ParseUint(xs, 16, 16)
  uint32_t v = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t c = rd.at(p + i);
    uint32_t d;
    if (c >= '0' && c <= '9') d = c - '0'; else if (c >= 'a' && c <= 'f') d = c - 'a' + 10; else if (c >= 'A' && c <= 'F') d = c - 'A' + 10; else return false;
    v = v * 16 + d;
  }
  *out = v;
  return true;
}
// unescapeStringBestEffort over the raw contents [s, s+n)
template <class S> __device__ void unescape_walk(MemBytes &rd, const uint64_t s, const uint32_t n, S &o) {
  uint32_t i = 0;
  while (i < n) { const uint32_t c = rd.at(s + i); if (c == '\\') break; o.put(c); i++; }
  if (i >= n) return;
  i++;
  while (i < n) {
    const uint32_t ch = rd.at(s + i); i++;
    switch (ch) {
      case '"': case '\\': case '/': o.put(ch); break;
      case 'b': o.put(8); break; case 'f': o.put(12); break; case 'n': o.put(10); break; case 'r': o.put(13); break; case 't': o.put(9); break;
      case 'u': {
        uint32_t x;
        if (n - i < 4 || !hex4(rd, s + i, &x)) { o.put('\\'); o.put('u'); break; }
        const uint32_t xs = i;
        i += 4;
        if (!(x >= 0xD800 && x < 0xE000)) { put_utf8(o, x); break; }
        uint32_t x1;
        if (n - i < 6 || rd.at(s + i) != '\\' || rd.at(s + i + 1) != 'u' || !hex4(rd, s + i + 2, &x1)) {
          o.put('\\'); o.put('u');
          for (int k = 0; k < 4; k++) o.put(rd.at(s + xs + k));
          break;
        }
        uint32_t r = 0xFFFD;  // utf16.DecodeRune
        if (x < 0xDC00 && x1 >= 0xDC00 && x1 < 0xE000) r = (((x - 0xD800) << 10) | (x1 - 0xDC00)) + 0x10000;
        put_utf8(o, r);
        i += 6;
        break;
      }
      default: o.put('\\'); o.put(ch);
    }
    while (i < n) { const uint32_t c = rd.at(s + i); if (c == '\\') break; o.put(c); i++; }
    if (i >= n) break;
    i++;
  }
}
// Value.MarshalTo of a container whose strings are still raw: the source minus whitespace outside strings
template <class S> __device__ void compact_walk(MemBytes &rd, const uint64_t s, const uint32_t n, S &o) {
  bool in_str = false; uint32_t run = 0;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t c = rd.at(s + i);
    if (in_str) {
      o.put(c);
      if (c == '\\') run++; else { if (c == '"' && !(run & 1u)) in_str = false; run = 0; }
    } else if (!(c == ' ' || c == '\n' || c == '\t' || c == '\r')) {
      o.put(c);
      if (c == '"') { in_str = true; run = 0; }
    }
  }
}

// tryToUnescapeTSKV (generic_parser.go:643-670): \\\\ \\n \\r \\t \\= ; a lone backslash at the end or any other escape
// leaves the WHOLE input as it is.  tskv_unescaped_len: the output length, or ~0u when the input stays unchanged.
__device__ uint32_t tskv_unescaped_len(MemBytes &rd, const uint64_t s, const uint32_t n) {
  uint32_t out = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (rd.at(s + i) != '\\') { out++; continue; }
    if (i == n - 1) return ~0u;
    const uint32_t c = rd.at(s + i + 1);
    if (!(c == '\\' || c == 'n' || c == 'r' || c == 't' || c == '=')) return ~0u;
    out++; i++;
  }
  return out;
}
template <class S> __device__ void tskv_walk(MemBytes &rd, const uint64_t s, const uint32_t n, S &o) {  // input already known to be well-formed
  for (uint32_t i = 0; i < n; i++) {
    uint32_t c = rd.at(s + i);
    if (c == '\\') { c = rd.at(s + ++i); c = c == 'n' ? '\n' : c == 'r' ? '\r' : c == 't' ? '\t' : c; }
    o.put(c);
  }
}

// f, err := strconv.ParseFloat(s, 64); if err != nil && !math.IsInf(f, 0) → 0 (fastfloat's fallback)
template <class F> __device__ int strconv_or0(const F &f, uint32_t n, const double *p10, const uint64_t *p128, double *out) {
  const int rc = parse_float_go(f, 0, n, p10, p128, out);
  if (rc == 1) { *out = 0; return 0; }
  if (rc == 2) return 0;  // ErrRange with ±Inf is kept
  return rc;
}

// fastfloat.ParseBestEffort: 0 ok, 3 = needs a correctly rounded ParseFloat this file does not carry
template <class F> __device__ int ff_best_effort(const F &f, const uint32_t n, const double *p10, const uint64_t *p128, double *out) {
  *out = 0;
  if (n == 0) return 0;
  uint32_t i = 0;
  const bool minus = f[0] == '-';
  if (minus) { i++; if (i >= n) return 0; }
  if (f[i] == '.' && (i + 1 >= n || !dg(f[i + 1]))) return 0;
  uint64_t d = 0;
  uint32_t j = i;
  while (i < n) {
    const uint32_t c = f[i];
    if (!dg(c)) break;
    d = d * 10 + (c - '0');
    i++;
    if (i > 18) return strconv_or0(f, n, p10, p128, out);
  }
  if (i <= j && f[i] != '.') {
    uint32_t t = i, tn = n - i;
    if (tn && f[t] == '+') { t++; tn--; }
    auto eq = [&](const char *w, uint32_t wl) { if (tn != wl) return false; for (uint32_t k = 0; k < wl; k++) if (lower_(f[t + k]) != (uint32_t)w[k]) return false; return true; };
    if (eq("inf", 3) || eq("infinity", 8)) *out = minus ? -INFINITY : INFINITY;
    else if (eq("nan", 3)) *out = NAN;
    return 0;
  }
  double v = (double)d;
  if (i >= n) { *out = minus ? -v : v; return 0; }
  if (f[i] == '.') {
    i++;
    if (i >= n) return 0;
    const uint32_t k = i;
    while (i < n) {
      const uint32_t c = f[i];
      if (!dg(c)) break;
      d = d * 10 + (c - '0');
      i++;
      if (i - j >= 17) return strconv_or0(f, n, p10, p128, out);
    }
    if (i < k) return 0;
    v = (double)d / p10[323 + (i - k)];
    if (i >= n) { *out = minus ? -v : v; return 0; }
  }
  if (f[i] == 'e' || f[i] == 'E') {
    i++;
    if (i >= n) return 0;
    bool exp_minus = false;
    if (f[i] == '+' || f[i] == '-') { exp_minus = f[i] == '-'; i++; if (i >= n) return 0; }
    int exp = 0;
    j = i;
    while (i < n) {
      const uint32_t c = f[i];
      if (!dg(c)) break;
      exp = exp * 10 + (int)(c - '0');
      i++;
      if (exp > 300) return strconv_or0(f, n, p10, p128, out);
    }
    if (i <= j) return 0;
    if (exp_minus) exp = -exp;
    v *= p10[323 + exp];
    if (i >= n) { *out = minus ? -v : v; return 0; }
  }
  return 0;
}
// fastfloat.ParseInt64BestEffort
template <class F> __device__ int64_t ff_int64(const F &f, const uint32_t n) {
  if (n == 0) return 0;
  if (n <= 18) {  // the whole token in at most three words: sign, then digits only, else 0
    const uint64_t b0 = f.m->word(f.start), b1 = n > 8 ? f.m->word(f.start + 8) : 0, b2 = n > 16 ? f.m->word(f.start + 16) : 0;
    const uint32_t i0 = ((uint32_t)b0 & 0xFFu) == '-' ? 1u : 0u;
    uint64_t v;
    if (n == i0 || !digits_u64(b0, b1, b2, i0, n, &v)) return 0;
    return i0 ? -(int64_t)v : (int64_t)v;
  }
  uint32_t i = 0;
  const bool minus = f[0] == '-';
  if (minus) { i++; if (i >= n) return 0; }
  int64_t d = 0;
  const uint32_t j = i;
  while (i < n) {
    const uint32_t c = f[i];
    if (!dg(c)) break;
    d = d * 10 + (int64_t)(c - '0');
    i++;
    if (i > 18) { int64_t dd; if (parse_int64(f, 0, n, false, &dd)) return 0; return dd; }
  }
  if (i <= j || i < n) return 0;
  return minus ? -d : d;
}
// fastfloat.ParseUint64BestEffort
template <class F> __device__ uint64_t ff_uint64(const F &f, const uint32_t n) {
  if (n == 0) return 0;
  if (n <= 18) {
    const uint64_t b0 = f.m->word(f.start), b1 = n > 8 ? f.m->word(f.start + 8) : 0, b2 = n > 16 ? f.m->word(f.start + 16) : 0;
    uint64_t v;
    return digits_u64(b0, b1, b2, 0, n, &v) ? v : 0;
  }
  uint32_t i = 0;
  uint64_t d = 0;
  while (i < n) {
    const uint32_t c = f[i];
    if (!dg(c)) break;
    d = d * 10 + (c - '0');
    i++;
    if (i > 18) { uint64_t dd; if (parse_uint64(f, 0, n, false, &dd)) return 0; return dd; }
  }
  if (i == 0 || i < n) return 0;
  return d;
}
// encoding/json isValidNumber (what json.Number must look like to be marshalled)
template <class F> __device__ bool valid_json_number(const F &f, const uint32_t n) {
  uint32_t i = 0;
  if (n == 0) return false;
  if (f[i] == '-') { i++; if (i == n) return false; }
  if (f[i] == '0') i++;
  else if (f[i] >= '1' && f[i] <= '9') { i++; while (i < n && dg(f[i])) i++; }
  else return false;
  if (i + 1 < n && f[i] == '.' && dg(f[i + 1])) { i += 2; while (i < n && dg(f[i])) i++; }
  if (i + 1 < n && (f[i] == 'e' || f[i] == 'E')) {
    i++;
    if (f[i] == '+' || f[i] == '-') { i++; if (i == n) return false; }
    while (i < n && dg(f[i])) i++;
  }
  return i == n;
}

// ---------------------------------------------------------------------------
// `any` columns holding a container: json.Marshal(wrapIntoEmptyInterface(v, useNumbers)) (generic_parser.go:602-633) —
// maps with their keys ascending (the last duplicate wins), strings through encoding/json's escaping, numbers as
// json.Number text or as the float64 fastfloat gives.  Decided on device when every string and key in the value is plain
// ASCII without escapes, every number is one json.Marshal accepts, and the nesting fits FJ_CANON_DEPTH; else the host.
// The value was validated by parse_json_line before store_cell sees it.
// ---------------------------------------------------------------------------
constexpr int FJ_CANON_DEPTH = 16;
__device__ __forceinline__ bool fj_ws(uint32_t c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }
// past one validated value at pos (white space in front of it allowed)
__device__ void fj_skip(MemBytes &rd, uint64_t &pos, const uint64_t end) {
  while (pos < end && fj_ws(rd.at(pos))) pos++;
  int depth = 0;
  while (pos < end) {
    const uint32_t c = rd.at(pos);
    if (c == '"') {
      pos++;
      while (pos < end) { const uint32_t ch = rd.at(pos); if (ch == '\\') pos += 2; else { pos++; if (ch == '"') break; } }
      if (depth == 0) return;
      continue;
    }
    if (c == '{' || c == '[') { depth++; pos++; continue; }
    if (c == '}' || c == ']') { if (depth == 0) return; depth--; pos++; if (depth == 0) return; continue; }
    if (depth == 0) {  // a number / literal token runs to the next delimiter
      if (c == ',' || fj_ws(c)) return;
      pos++;
      continue;
    }
    pos++;
  }
}
__device__ bool fj_any_ok(const JParams &p, MemBytes &rd, const uint64_t vs, const uint64_t ve) {
  int depth = 0;
  uint64_t pos = vs;
  while (pos < ve) {
    const uint32_t c = rd.at(pos);
    if (fj_ws(c) || c == ':' || c == ',') { pos++; continue; }
    if (c == '{' || c == '[') { if (++depth > FJ_CANON_DEPTH) return false; pos++; continue; }
    if (c == '}' || c == ']') { depth--; pos++; continue; }
    if (c == '"') {
      pos++;
      for (;;) { const uint32_t ch = rd.at(pos); if (ch == '"') break; if (ch == '\\' || ch >= 0x80u) return false; pos++; }
      pos++;
      continue;
    }
    if (c == 't') { pos += 4; continue; }
    if (c == 'f') { pos += 5; continue; }
    if (c == 'n' && pos + 1 < ve && rd.at(pos + 1) == 'u') { pos += 4; continue; }
    uint64_t q = pos;  // a number token (nan / inf included: json.Marshal refuses them)
    if (!scan_number(rd, q, ve)) return false;
    const Field tok{&rd, pos, (uint32_t)(q - pos)};
    if (p.use_numbers) { if (!valid_json_number(tok, tok.n)) return false; }
    else { double v; if (ff_best_effort(tok, tok.n, p.pow10, p.pow128, &v) || v != v || v == INFINITY || v == -INFINITY) return false; }
    pos = q;
  }
  return true;
}
// encoding/json's string escaping of plain ASCII bytes (escapeHTML on): only control bytes and < > & change
template <class S> __device__ void fj_go_ascii(S &o, MemBytes &rd, const uint64_t s, const uint32_t n) {
  const char *H = "0123456789abcdef";
  o.put('"');
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t c = rd.at(s + i);
    if (c >= 0x20 && c != '<' && c != '>' && c != '&') { o.put(c); continue; }
    o.put('\\');
    if (c == '\n') o.put('n'); else if (c == '\r') o.put('r'); else if (c == '\t') o.put('t'); else if (c == '\b') o.put('b'); else if (c == '\f') o.put('f');
    else { o.put('u'); o.put('0'); o.put('0'); o.put(H[c >> 4]); o.put(H[c & 15]); }
  }
  o.put('"');
}
template <class S> __device__ void fj_emit_any(S &o, const JParams &p, MemBytes &rd, const uint64_t vs, const uint64_t ve) {
  struct Frame { uint64_t s, e, cur; uint32_t prev_n; uint8_t obj, first; };  // array: cur = next element; object: [cur, cur + prev_n) = last key written
  Frame st[FJ_CANON_DEPTH];
  int sp = 0;
  auto skip_ws = [&](uint64_t &q, uint64_t e) { while (q < e && fj_ws(rd.at(q))) q++; };
  auto begin_value = [&](uint64_t a, uint64_t b) {
    const uint32_t c = rd.at(a);
    if (c == '{' || c == '[') {
      if (sp == FJ_CANON_DEPTH) return;  // excluded by fj_any_ok
      Frame &f = st[sp++];
      f.s = a; f.e = b; f.cur = a + 1; f.prev_n = 0xFFFFFFFFu; f.obj = c == '{'; f.first = 1;
      o.put(c);
    } else if (c == '"') fj_go_ascii(o, rd, a + 1, (uint32_t)(b - a - 2));
    else if (c == 't' || c == 'f' || (c == 'n' && rd.at(a + 1) == 'u') || p.use_numbers) { for (uint64_t q = a; q < b; q++) o.put(rd.at(q)); }
    else { const Field tok{&rd, a, (uint32_t)(b - a)}; double v = 0; ff_best_effort(tok, tok.n, p.pow10, p.pow128, &v); dev::fmt_json_float(o, v, 64); }
  };
  begin_value(vs, ve);
  while (sp > 0) {
    Frame &f = st[sp - 1];
    if (!f.obj) {
      uint64_t pos = f.cur;
      skip_ws(pos, f.e);
      if (rd.at(pos) == ']') { o.put(']'); sp--; continue; }
      uint64_t q = pos;
      fj_skip(rd, q, f.e);
      uint64_t nx = q;
      skip_ws(nx, f.e);
      if (rd.at(nx) == ',') nx++;
      f.cur = nx;
      if (!f.first) o.put(',');
      f.first = 0;
      begin_value(pos, q);
      continue;
    }
    // the next key: the smallest one greater than the key written last, the last occurrence among equals
    uint64_t bks = 0, bvs = 0, bve = 0; uint32_t bkn = 0xFFFFFFFFu;
    uint64_t pos = f.s + 1;
    auto key_cmp = [&](uint64_t a, uint32_t an, uint64_t b, uint32_t bn) -> int {  // plain ASCII bodies: byte order is Go's string order
      const uint32_t m = an < bn ? an : bn;
      for (uint32_t i = 0; i < m; i++) { const uint32_t x = rd.at(a + i), y = rd.at(b + i); if (x != y) return x < y ? -1 : 1; }
      return an == bn ? 0 : an < bn ? -1 : 1;
    };
    for (;;) {
      skip_ws(pos, f.e);
      if (rd.at(pos) == '}') break;
      const uint64_t ks = pos + 1;
      uint64_t q = pos;
      fj_skip(rd, q, f.e);
      const uint32_t kn = (uint32_t)(q - pos - 2);
      pos = q;
      skip_ws(pos, f.e);
      pos++;  // ':'
      skip_ws(pos, f.e);
      const uint64_t a = pos;
      fj_skip(rd, pos, f.e);
      const uint64_t b = pos;
      skip_ws(pos, f.e);
      if (rd.at(pos) == ',') pos++;
      if (f.prev_n != 0xFFFFFFFFu && key_cmp(f.cur, f.prev_n, ks, kn) >= 0) continue;
      if (bkn == 0xFFFFFFFFu || key_cmp(ks, kn, bks, bkn) <= 0) { bks = ks; bkn = kn; bvs = a; bve = b; }
    }
    if (bkn == 0xFFFFFFFFu) { o.put('}'); sp--; continue; }
    if (!f.first) o.put(',');
    f.first = 0;
    f.cur = bks; f.prev_n = bkn;
    fj_go_ascii(o, rd, bks, bkn);
    o.put(':');
    begin_value(bvs, bve);
  }
}

// ---------------------------------------------------------------------------
// one top-level member → the cell of one column (Unmarshal :672-731 then ParseVal :888-1123)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void jstore_int(const JCol &c, int64_t r, int64_t v) {
  switch (c.width) {
    case 1: ((int8_t *)c.values)[r] = (int8_t)v; break;
    case 2: ((int16_t *)c.values)[r] = (int16_t)v; break;
    case 4: ((int32_t *)c.values)[r] = (int32_t)v; break;
    default: ((int64_t *)c.values)[r] = v;
  }
}
__device__ __forceinline__ void jtext(const JCol &c, int64_t r, uint64_t src, uint32_t srclen, uint32_t mode, uint32_t outlen) {
  c.fstart[r] = (uint32_t)src; c.meta[r] = srclen | (mode << 28); c.lens[r] = outlen;
}

// returns false when the value form is not decided on device (the line goes to the host)
// 1 stored, 0 the value form is the host's, 2 (HEAVY == false only) the cell needs the map emitter: the line is re-parsed
// by json_parse_listed, whose kernel carries it — the kernel every line runs through stays lean.
template <bool HEAVY>
__device__ int store_cell_plain(const JParams &p, const JCol &c, const int64_t r, MemBytes &aux, const int vtype, const uint64_t vstart,
                           const uint64_t vend, const JStr &sv, const uint32_t clen, const bool kbad) {
  uint8_t st = JC_OK;
  const Field tok{&aux, vstart, (uint32_t)(vend - vstart)};  // raw token of numbers / literals / containers
  const Field str{&aux, sv.s, sv.n};                         // raw contents of a string value
  if (vtype == V_NULL) {
    if (c.kind == JK_TEXT || c.kind == JK_ANY) c.lens[r] = 0;
    c.cellst[r] = JC_NIL;
    return true;
  }
  switch (c.kind) {
    case JK_TEXT:
      if (vtype == V_STR) {  // string(v.GetStringBytes()): unescaped
        if (!sv.bs) jtext(c, r, sv.s, sv.n, JM_COPY, sv.n);
        else if (p.format == TFGPU_JFMT_TSKV) {  // tryToUnescapeTSKV: unescaped, or the input itself when an escape is broken
          const uint32_t ul = tskv_unescaped_len(aux, sv.s, sv.n);
          if (ul == ~0u) jtext(c, r, sv.s, sv.n, JM_COPY, sv.n); else jtext(c, r, sv.s, sv.n, JM_TSKV, ul);
        }
        else { CountSink cs; unescape_walk(aux, sv.s, sv.n, cs); jtext(c, r, sv.s, sv.n, JM_UNESCAPE, cs.n); }
      } else if (vtype == V_OBJ || vtype == V_ARR) {  // v.String(): MarshalTo with raw strings
        if (kbad) return false;  // an unescaped key that needs strconv.AppendQuote
        jtext(c, r, vstart, tok.n, JM_COMPACT, clen);
      } else jtext(c, r, vstart, tok.n, JM_COPY, tok.n);  // number / true / false: the raw token
      break;
    case JK_INT: {
      int64_t v = 0;
      if (vtype == V_STR) {  // strconv.ParseInt(vv, 0, bits)
        if (sv.bs) return false;
        const int bits = c.width * 8;
        if (parse_int64(str, 0, sv.n, true, &v)) st = JC_ERR;
        else if (bits < 64 && (v < -(1ll << (bits - 1)) || v > (1ll << (bits - 1)) - 1)) st = JC_ERR;
      } else if (vtype == V_NUM) v = ff_int64(tok, tok.n);  // intN(v.GetInt()): Go's truncating conversion
      if (st == JC_OK) jstore_int(c, r, v);
      break;
    }
    case JK_UINT: {
      uint64_t v = 0;
      if (vtype == V_STR) {  // strconv.ParseUint(vv, 0, bits)
        if (sv.bs) return false;
        const int bits = c.width * 8;
        if (parse_uint64(str, 0, sv.n, true, &v)) st = JC_ERR;
        else if (bits < 64 && v > (1ull << bits) - 1) st = JC_ERR;
      } else if (vtype == V_NUM) v = ff_uint64(tok, tok.n);
      if (st == JC_OK) jstore_int(c, r, (int64_t)v);
      break;
    }
    case JK_F64: {
      double v = 0;
      if (vtype == V_STR) {  // strconv.ParseFloat(vv, 64)
        if (sv.bs) return false;
        const int rc = parse_float_go(str, 0, sv.n, p.pow10, p.pow128, &v);
        if (rc == 3) return false;
        if (rc) st = JC_ERR;
      } else if (vtype == V_NUM) { if (ff_best_effort(tok, tok.n, p.pow10, p.pow128, &v)) return false; }
      if (st == JC_OK) ((double *)c.values)[r] = v;
      break;
    }
    case JK_BOOL: {
      int v = vtype == V_TRUE;
      if (vtype == V_STR) {  // strconv.ParseBool(vv)
        if (sv.bs) return false;
        if (parse_bool(str, 0, sv.n, &v)) st = JC_ERR;
      }
      if (st == JC_OK) ((uint8_t *)c.values)[r] = (uint8_t)v;
      break;
    }
    case JK_ANY:  // stored as json.Marshal(value)
      if (vtype == V_TRUE || vtype == V_FALSE) jtext(c, r, vstart, tok.n, JM_COPY, tok.n);
      else if (vtype == V_NUM) {
        if (p.use_numbers) {  // json.Number: marshalled as its own text, which must be a valid JSON number
          if (!valid_json_number(tok, tok.n)) return false;
          jtext(c, r, vstart, tok.n, JM_COPY, tok.n);
        } else {  // float64 → encoding/json's float text (NaN / ±Inf are a Marshal error: host)
          double v;
          if (ff_best_effort(tok, tok.n, p.pow10, p.pow128, &v)) return false;
          if (v != v || v == INFINITY || v == -INFINITY) return false;
          dev::CountOut co;
          dev::fmt_json_float(co, v, 64);
          jtext(c, r, vstart, tok.n, JM_FLOAT, co.n);
        }
      } else if (vtype == V_STR) {
        // ParseVal: ReplaceAll(`\\`, `\`) then json.Unmarshal into a map — a map only if the text is an object
        if (!sv.plain) return false;  // printable ASCII without escapes so far; encoding/json also escapes < > &
        for (uint32_t q = 0; q < sv.n; q++) { const uint32_t ch = str[q]; if (ch == '<' || ch == '>' || ch == '&') return false; }
        uint32_t i = 0;
        while (i < sv.n && (str[i] == ' ' || str[i] == '\t')) i++;
        if (i < sv.n && (str[i] == '{' || str[i] == 'n')) return false;  // an object, or `null` (Unmarshal succeeds with a nil map)
        jtext(c, r, sv.s, sv.n, JM_QUOTED, sv.n + 2);
      } else {  // map / slice: re-marshalled with sorted keys when every token is one the device decides
        if constexpr (!HEAVY) return 2;
        else {
          if (!fj_any_ok(p, aux, vstart, vend)) return false;
          CountSink cs;
          fj_emit_any(cs, p, aux, vstart, vend);
          jtext(c, r, vstart, tok.n, JM_ANYCANON, cs.n);
        }
      }
      break;
    default:  // JK_DATETIME: extractTimeValue :818-886
      if (vtype == V_NUM) {
        int64_t sec;
        if (p.use_numbers) { if (parse_int64(tok, 0, tok.n, false, &sec)) st = JC_ERR; }  // json.Number → ParseInt(10, 64)
        else {
          double v;
          if (ff_best_effort(tok, tok.n, p.pow10, p.pow128, &v)) return false;
          const double a = fabs(v);
          sec = (a >= 9223372036854775808.0 || a != a) ? INT64_MIN : (int64_t)a;  // int64(math.Abs(f)), amd64 CVTTSD2SQ
        }
        if (st == JC_OK) { ((int64_t *)c.values)[r] = sec; c.nanos[r] = 0; }
      } else if (vtype == V_STR) return false;  // dateparse.ParseAny
      else st = JC_ERR;                         // "unable extract timestamp"
  }
  c.cellst[r] = st;
  return true;
}

// ---------------------------------------------------------------------------
// lookupComplex (pkg/parsers/generic/lookup.go:10-59; generic_parser.go:323-347): a column whose ColSchema.Path has '.' or '/'
// reads item[first name] — which must be a Go STRING holding JSON — json.Unmarshal's it into a map, takes the next name, and so
// on; a string on the way is parsed again, a map is walked.  parseJSON tries the text as it is, then with every `\\"` turned
// into `\"` ("possible double escape": what the metrika logs need), then without any backslash.  The rewritten texts and the
// decoded strings are built in a scratch copy of the input (JParams::scratch, same offsets: none of them is longer than what it
// is made from), so every step reads plain memory through the encoding/json scanner (tf_jsonscan.hpp).
// Decided here: every outcome whose target is a string or null, or an error.  Handed to the host: a member name that is not plain
// ASCII (compared after decoding), ill-formed UTF-8 inside a string that must be decoded (U+FFFD grows it), a number / bool /
// container at the end of the path (ParseVal of encoding/json's float64: not restated by the oracle either), nesting beyond 128.
// 1 stored (a value, nil or a ParseVal error), 0 host.
// ---------------------------------------------------------------------------
struct NText { bool scr; uint32_t s, e; };
// every byte >= 0x80 of [s, e) sits in a well-formed UTF-8 sequence (utf8.DecodeRune's ranges): decoding never grows the text
__device__ bool nested_utf8_ok(MemBytes &rd, uint32_t s, uint32_t e) {
  for (uint32_t i = s; i < e;) {
    const uint32_t c = rd.at(i);
    if (c < 0x80) { i++; continue; }
    uint32_t need = 0, lo = 0x80, hi = 0xBF;
    if (c >= 0xC2 && c <= 0xDF) need = 1;
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
    if (!need || e - i <= need) return false;
    for (uint32_t k = 1; k <= need; k++) { const uint32_t d = rd.at(i + k); if (d < (k == 1 ? lo : 0x80u) || d > (k == 1 ? hi : 0xBFu)) return false; }
    i += need + 1;
  }
  return true;
}
// json.Unmarshal(text, &map[string]interface{}): 1 = the text is one valid value and an object (at *at), 0 = it is not (invalid, or
// another kind of value: no map), -1 = nesting beyond what the scanner walks
__device__ int nested_try(MemBytes &rd, uint32_t s, uint32_t e, uint32_t *at) {
  uint32_t pos = s;
  while (pos < e && sr::is_ws(rd.at(pos))) pos++;
  if (pos >= e) return 0;
  const bool object = rd.at(pos) == '{';
  uint32_t q = pos, vt = 0;
  const int rc = sr::skip_value(rd, q, e, vt);
  if (rc == 2) return -1;
  if (rc != 0) return 0;
  while (q < e && sr::is_ws(rd.at(q))) q++;
  if (q != e || !object) return 0;
  *at = pos;
  return 1;
}
// parseJSON (lookup.go:41-59).  1 = t holds a text whose value is an object at *at, 2 = error (no map), 0 = host
__device__ int nested_parse(const JParams &p, uint8_t *const scr, NText &t, uint32_t *at) {
  {
    MemBytes rd(t.scr ? scr : p.data);
    const int r = nested_try(rd, t.s, t.e, at);
    if (r) return r < 0 ? 0 : 1;
    bool bs = false;
    for (uint32_t i = t.s; i < t.e && !bs; i++) bs = rd.at(i) == '\\';
    if (!bs) return 2;  // the retries would try the same text
  }
  {  // strings.ReplaceAll(s, `\\"`, `\"`): left to right, non-overlapping
    MemBytes rd(t.scr ? scr : p.data);
    sr::ByteSink o{scr + t.s};
    uint32_t n = 0;
    for (uint32_t i = t.s; i < t.e;) {
      if (t.e - i >= 3 && rd.at(i) == '\\' && rd.at(i + 1) == '\\' && rd.at(i + 2) == '"') { o.put('\\'); o.put('"'); n += 2; i += 3; }
      else { o.put(rd.at(i)); n++; i++; }
    }
    o.flush();
    t.scr = true; t.e = t.s + n;
  }
  {
    MemBytes rd(scr);
    const int r = nested_try(rd, t.s, t.e, at);
    if (r) return r < 0 ? 0 : 1;
  }
  {  // strings.ReplaceAll(s, `\`, ``)
    MemBytes rd(scr);
    sr::ByteSink o{scr + t.s};
    uint32_t n = 0;
    for (uint32_t i = t.s; i < t.e; i++) { const uint32_t c = rd.at(i); if (c != '\\') { o.put(c); n++; } }
    o.flush();
    t.e = t.s + n;
  }
  MemBytes rd(scr);
  const int r = nested_try(rd, t.s, t.e, at);
  return r < 0 ? 0 : r ? 1 : 2;
}
// the Go string of the JSON string token [fs, fe) of text t (decode.go unquote) → t2; false: host
__device__ bool nested_decode(const JParams &p, uint8_t *const scr, const NText &t, uint32_t fs, uint32_t fe, bool plain, NText &t2) {
  if (plain && !t.scr) { t2 = NText{false, fs + 1, fe - 1}; return true; }  // its bytes, where they lie
  MemBytes rd(t.scr ? scr : p.data);
  if (!plain && !nested_utf8_ok(rd, fs + 1, fe - 1)) return false;
  sr::ByteSink o{scr + fs + 1};
  sr::CountSink cnt;
  sr::emit_unquoted(cnt, rd, fs, fe - fs);
  MemBytes rd2(t.scr ? scr : p.data);
  sr::emit_unquoted(o, rd2, fs, fe - fs);  // (in place when t is the scratch: every rune leaves at most the bytes it took)
  o.flush();
  t2 = NText{true, fs + 1, fs + 1 + cnt.n};
  return true;
}
__device__ int store_nested(const JParams &p, const JCol &c, const int64_t r, const int vtype, const JStr &sv) {
  auto nil = [&]() { if (c.kind == JK_TEXT || c.kind == JK_ANY) c.lens[r] = 0; c.cellst[r] = JC_NIL; return 1; };  // "lookupComplex error" / nil: json_finish applies the key / required rule
  if (vtype == V_NULL) return nil();  // unexpected value type: <nil>
  if (vtype != V_STR || sv.bs || !p.scratch) return 0;
  uint8_t *const scr = p.scratch + c.scr_base;  // this column's own copy: columns that read one top-level value must not rewrite each other's texts
  if ((sv.s + sv.n) >> 32) return 0;
  NText t{false, (uint32_t)sv.s, (uint32_t)(sv.s + sv.n)};  // the Go string lookupComplex holds: parsed when a name is taken from it
  bool is_obj = false;                                       // … or an object of an already validated text: walked
  uint32_t at = 0;
  const uint8_t *seg = p.names + c.npath_off;
  for (uint32_t k = 0; k < c.npath_n; k++) {
    const uint32_t sl = (uint32_t)seg[0] | ((uint32_t)seg[1] << 8);
    const uint8_t *name = seg + 2;
    seg += 2 + sl;
    if (!is_obj) { const int pr = nested_parse(p, scr, t, &at); if (pr == 0) return 0; if (pr == 2) return nil(); }  // unable to parse json / a nil map: unable to get field
    else at = t.s;
    // members of the object at `at`: the LAST member with this name wins (a Go map)
    MemBytes rd(t.scr ? scr : p.data);
    uint32_t fs = 0, fe = 0, ft = sr::VT_ABSENT;
    uint32_t pos = at + 1;
    for (;;) {
      while (pos < t.e && sr::is_ws(rd.at(pos))) pos++;
      if (pos >= t.e) return 0;  // (cannot happen in validated text)
      if (rd.at(pos) == '}') break;
      if (rd.at(pos) == ',') { pos++; continue; }
      const uint32_t ks = pos + 1;
      bool kplain = false;
      if (rd.at(pos) != '"' || !sr::scan_string(rd, pos, t.e, &kplain)) return 0;
      if (!kplain) return 0;  // compared after decoding: host
      const uint32_t kn = pos - 1 - ks;
      bool same = kn == sl;
      for (uint32_t i = 0; i < kn && same; i++) same = rd.at(ks + i) == name[i];
      while (pos < t.e && sr::is_ws(rd.at(pos))) pos++;
      if (pos >= t.e || rd.at(pos) != ':') return 0;
      pos++;
      while (pos < t.e && sr::is_ws(rd.at(pos))) pos++;
      const uint32_t vs0 = pos;
      uint32_t vt = 0;
      if (sr::skip_value(rd, pos, t.e, vt) != 0) return 0;
      if (same) { fs = vs0; fe = pos; ft = vt; }
    }
    if (ft == sr::VT_ABSENT) return nil();  // unable to get field
    const uint32_t vk = ft & sr::VT_MASK;
    const bool last = k + 1 == c.npath_n;
    if (vk == sr::VT_STR) {
      NText t2;
      if (!nested_decode(p, scr, t, fs, fe, (ft & sr::VT_PLAIN) != 0, t2)) return 0;
      if (!last) { t = t2; is_obj = false; continue; }
      // ParseVal(the string, DataType)
      if (c.kind == JK_TEXT) {
        jtext(c, r, t2.scr ? c.scr_base + t2.s : t2.s, t2.e - t2.s, t2.scr ? JM_SCRATCH : JM_COPY, t2.e - t2.s);
        c.cellst[r] = JC_OK;
        return 1;
      }
      if (c.kind == JK_ANY && t2.scr) return 0;  // (its text cell would be cut from the input)
      MemBytes rv(t2.scr ? scr : p.data);
      JStr v{};
      v.s = t2.s; v.n = t2.e - t2.s; v.hash = 0; v.ctrl = false; v.bs = false; v.plain = !t2.scr;
      return store_cell_plain<false>(p, c, r, rv, V_STR, (uint64_t)t2.s, (uint64_t)t2.e, v, 0, false) == 1 ? 1 : 0;
    }
    if (last) return vk == sr::VT_NULL ? nil() : 0;  // nil; or ParseVal of float64 / bool / map / slice: host
    if (vk == sr::VT_OBJ) { t = NText{t.scr, fs, fe}; is_obj = true; continue; }
    return nil();  // unexpected value type
  }
  return 0;
}
template <bool HEAVY>
__device__ __forceinline__ int store_cell(const JParams &p, const JCol &c, const int64_t r, MemBytes &aux, const int vtype, const uint64_t vstart,
                                          const uint64_t vend, const JStr &sv, const uint32_t clen, const bool kbad) {
  if (c.npath_n) {  // lookupComplex lives in the kernels that carry the map emitter: the ones every line runs through stay lean
    if constexpr (!HEAVY) return 2; else return store_nested(p, c, r, vtype, sv);
  }
  return store_cell_plain<HEAVY>(p, c, r, aux, vtype, vstart, vend, sv, clen, kbad);
}

// ---------------------------------------------------------------------------
// `_rest` (AddRest, generic_parser.go:323-349): json.Marshal of the map of the top-level members no column knows —
// p.known = ColumnName and ColPath of the raw fields — each value as Unmarshal left it.  Up to REST_MAX such members with
// plain ASCII keys and values the `any` emitter decides; else the line goes to the host.  The line was validated.
// ---------------------------------------------------------------------------
constexpr int REST_MAX = 16;
template <class S> __device__ bool rest_emit(S &o, const JParams &p, MemBytes &rd, const uint64_t ls, const uint64_t le) {
  uint64_t ks[REST_MAX], vs[REST_MAX], ve[REST_MAX]; uint32_t kn[REST_MAX];
  int n = 0;
  uint64_t pos = ls;
  while (pos < le && fj_ws(rd.at(pos))) pos++;
  pos++;  // '{'
  for (;;) {
    while (pos < le && fj_ws(rd.at(pos))) pos++;
    if (pos >= le || rd.at(pos) == '}') break;
    pos++;  // '"'
    JStr k;
    scan_string(rd, pos, le, k);
    while (pos < le && fj_ws(rd.at(pos))) pos++;
    pos++;  // ':'
    while (pos < le && fj_ws(rd.at(pos))) pos++;
    const uint64_t a = pos;
    fj_skip(rd, pos, le);
    const uint64_t b = pos;
    while (pos < le && fj_ws(rd.at(pos))) pos++;
    if (pos < le && rd.at(pos) == ',') pos++;
    bool known = false;
    for (uint32_t s = k.hash & p.slot_mask;; s = (s + 1) & p.slot_mask) {
      const JSlot sl = p.slots[s];
      if (sl.slen == ~0u) break;
      if (sl.hash == k.hash && sl.slen == k.n) {
        bool same = true;
        for (uint32_t i = 0; i < k.n && same; i += 8) {
          const uint32_t nb = k.n - i < 8 ? k.n - i : 8u;
          same = (rd.word(k.s + i) & (nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1)) == *reinterpret_cast<const uint64_t *>(p.names + sl.soff + i);
        }
        if (same) { known = true; break; }
      }
    }
    if (known) continue;
    if (!k.plain || n == REST_MAX) return false;
    // the value as Unmarshal stores it: a string stays a Go string (plain ASCII only here), anything else through the `any` rules
    if (rd.at(a) == '"') { for (uint64_t q = a + 1; q + 1 < b; q++) { const uint32_t ch = rd.at(q); if (ch == '\\' || ch >= 0x80u) return false; } }
    else if (!fj_any_ok(p, rd, a, b)) return false;
    ks[n] = k.s; kn[n] = k.n; vs[n] = a; ve[n] = b; n++;
  }
  auto key_cmp = [&](int x, int y) -> int {
    const uint32_t m = kn[x] < kn[y] ? kn[x] : kn[y];
    for (uint32_t i = 0; i < m; i++) { const uint32_t cx = rd.at(ks[x] + i), cy = rd.at(ks[y] + i); if (cx != cy) return cx < cy ? -1 : 1; }
    return kn[x] == kn[y] ? 0 : kn[x] < kn[y] ? -1 : 1;
  };
  // ascending keys by selection (n <= 16); among equal keys the last occurrence is the map's value
  o.put('{');
  int prev = -1; bool first = true;
  for (;;) {
    int best = -1;
    for (int i = 0; i < n; i++) {
      if (prev >= 0 && key_cmp(i, prev) <= 0) continue;
      if (best < 0 || key_cmp(i, best) <= 0) best = i;
    }
    if (best < 0) break;
    if (!first) o.put(',');
    first = false;
    fj_go_ascii(o, rd, ks[best], kn[best]);
    o.put(':');
    fj_emit_any(o, p, rd, vs[best], ve[best]);
    prev = best;
  }
  o.put('}');
  return true;
}

// ---------------------------------------------------------------------------
// one line: fastjson Parser.Parse with an explicit container stack; the members of a top-level
// object are handed to store_cell as they complete
// ---------------------------------------------------------------------------
// HEAVY == false: the kernel every line runs through; returns true when the line needs the map emitter (an `any` container,
// a non-empty `_rest`) and must be re-parsed by the HEAVY form — nothing final has been written for it then.
template <bool HEAVY>
__device__ bool parse_json_line(const JParams &p, const int64_t r, const uint64_t ls, const uint64_t le) {
  MemBytes rd(p.data), aux(p.data);
  uint64_t pos = ls;
  const uint64_t end = le;
#define JSKIPWS() while (pos < end) { const uint32_t w_ = rd.at(pos); if (w_ == ' ' || w_ == '\n' || w_ == '\t' || w_ == '\r') pos++; else break; }
  uint64_t stack = 0;  // bit 0 = innermost container is an object
  int ncont = 0;
  bool root_obj = false, fallback = false, syntax = false, key_next = false, heavy = false;
  uint32_t root_kids = 0;
  int32_t kcol = JS_UNKNOWN;
  uint32_t nunknown = 0;
  uint64_t vstart = 0;
  int vtype = V_NULL;
  uint32_t clen = 0;
  bool kbad = false;
  JStr sval{};
  JSKIPWS();
  for (;;) {
    if (key_next) {  // parseObject: `"key" :` then a value
      JSKIPWS();
      if (pos >= end || rd.at(pos) != '"') { syntax = true; break; }
      pos++;
      JStr k;
      if (!scan_string(rd, pos, end, k)) { syntax = true; break; }
      clen += k.n + 3;
      if (ncont == 1) {  // a key of the root object: which column reads it?
        kcol = JS_UNKNOWN;
        if (k.bs) fallback = true;  // compare after unescaping: host
        else {
          for (uint32_t s = k.hash & p.slot_mask;; s = (s + 1) & p.slot_mask) {
            const JSlot sl = p.slots[s];
            if (sl.slen == ~0u) break;
            if (sl.hash == k.hash && sl.slen == k.n) {
              bool same = true;  // eight bytes per compare: the table's names are zero-padded to whole words
              for (uint32_t i = 0; i < k.n && same; i += 8) {
                const uint32_t nb = k.n - i < 8 ? k.n - i : 8u;
                same = (aux.word(k.s + i) & (nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1)) == *reinterpret_cast<const uint64_t *>(p.names + sl.soff + i);
              }
              if (same) { kcol = sl.col; if (sl.aux) fallback = true; break; }  // colTypeMap types it as the aux column (:1218-1225): host
            }
          }
          if (p.add_rest && kcol == JS_UNKNOWN) nunknown++;  // `_rest` will not be {}
        }
      } else if (ncont == 2 && vtype == V_OBJ && (k.bs || k.ctrl)) kbad = true;
      JSKIPWS();
      if (pos >= end || rd.at(pos) != ':') { syntax = true; break; }
      pos++;
      JSKIPWS();
      key_next = false;
    }
    // ---- parseValue ----
    if (pos >= end) { syntax = true; break; }
    {
      const uint32_t c = rd.at(pos);
      const bool member = ncont == 1 && root_obj;
      int vt;
      if (c == '{' || c == '[') {
        vt = c == '{' ? V_OBJ : V_ARR;
        if (member) { vstart = pos; clen = 0; kbad = false; vtype = vt; }
        if (ncont == 0) root_obj = c == '{';
        pos++; clen++;
        JSKIPWS();
        if (pos >= end) { syntax = true; break; }
        if (rd.at(pos) == (c == '{' ? '}' : ']')) { pos++; clen++; }
        else {
          if (ncont >= 64) { fallback = true; break; }  // fastjson allows 300 levels: deeper than 64 goes to the host
          stack = (stack << 1) | (c == '{' ? 1u : 0u);
          ncont++;
          key_next = c == '{';
          continue;
        }
      } else {
        if (member) { vstart = pos; clen = 0; kbad = false; }
        const uint64_t left = end - pos;
        if (c == '"') {
          pos++;
          if (!scan_string(rd, pos, end, sval)) { syntax = true; break; }
          clen += sval.n + 2;
          vt = V_STR;
        } else if (c == 't') {
          if (left < 4 || rd.at(pos + 1) != 'r' || rd.at(pos + 2) != 'u' || rd.at(pos + 3) != 'e') { syntax = true; break; }
          pos += 4; clen += 4; vt = V_TRUE;
        } else if (c == 'f') {
          if (left < 5 || rd.at(pos + 1) != 'a' || rd.at(pos + 2) != 'l' || rd.at(pos + 3) != 's' || rd.at(pos + 4) != 'e') { syntax = true; break; }
          pos += 5; clen += 5; vt = V_FALSE;
        } else if (c == 'n') {
          if (left >= 4 && rd.at(pos + 1) == 'u' && rd.at(pos + 2) == 'l' && rd.at(pos + 3) == 'l') { pos += 4; clen += 4; vt = V_NULL; }
          else if (left >= 3 && ci3(rd, pos, 'n', 'a', 'n')) { pos += 3; clen += 3; vt = V_NUM; }
          else { syntax = true; break; }
        } else {
          const uint64_t b = pos;
          if (!scan_number(rd, pos, end)) { syntax = true; break; }
          clen += (uint32_t)(pos - b);
          vt = V_NUM;
        }
        if (member) vtype = vt;
      }
    }
    // ---- a value is complete: close containers while they end here ----
    bool done = false;
    for (;;) {
      if (ncont == 0) {
        JSKIPWS();
        if (pos < end) syntax = true;  // "unexpected tail"
        done = true;
        break;
      }
      if (ncont == 1) {
        root_kids++;
        if (root_obj && !fallback && !heavy) {
          for (int32_t ci = kcol; ci >= 0; ci = p.cols[ci].next) {
            const int rc = store_cell<HEAVY>(p, p.cols[ci], r, aux, vtype, vstart, pos, sval, clen, kbad);
            if (rc == 0) { fallback = true; break; }
            if (rc == 2) { heavy = true; break; }
          }
        }
      }
      JSKIPWS();
      if (pos >= end) { syntax = true; done = true; break; }
      const uint32_t c = rd.at(pos);
      const bool is_obj = stack & 1u;
      if (c == ',') {
        pos++; clen++;
        if (is_obj) key_next = true; else { JSKIPWS(); }
        break;
      }
      if (c == (is_obj ? '}' : ']')) { pos++; clen++; stack >>= 1; ncont--; continue; }
      syntax = true; done = true;
      break;
    }
    if (done) break;
  }
#undef JSKIPWS
  uint8_t st = JL_ROW;
  if (syntax) st = JL_SYNTAX;              // Unmarshal error → NewUnparsed (:545-550); decided before any fallback,
  else if (heavy) return true;             //   and before the map emitter is asked for
  else if (fallback) st = JL_FALLBACK;     //   except that a line abandoned for depth is never known to be malformed
  else if (!root_obj || root_kids == 0) st = JL_SKIPPED;  // len(item) == 0 (:536)
  if (st == JL_ROW && p.rest_lens) {
    if (nunknown == 0) { p.rest_fstart[r] = (uint32_t)ls; p.rest_meta[r] = (JM_REST_EMPTY << 28); p.rest_lens[r] = 2; }
    else if constexpr (!HEAVY) return true;
    else {
      CountSink cs;
      if (!rest_emit(cs, p, aux, ls, le)) st = JL_FALLBACK;
      else { p.rest_fstart[r] = (uint32_t)ls; p.rest_meta[r] = (uint32_t)(le - ls) | (JM_REST << 28); p.rest_lens[r] = cs.n; }
    }
  }
  p.linest[r] = st;
  p.line_pos[r] = (uint32_t)ls;
  return false;
}

// GenericParser.Unmarshal, Format "tskv" (generic_parser.go:732-746): strings.Split(line, "\t"), SplitN(field, "=", 2); a field
// without '=' is skipped, every value is a Go string (ParseVal's string branch types it), the last duplicate of a key wins.
// Unmarshal cannot fail here: a line is a row, skipped (no key=value field at all), or handed to the host.
template <bool HEAVY>
__device__ bool parse_tskv_line(const JParams &p, const int64_t r, const uint64_t ls, const uint64_t le) {
  MemBytes rd(p.data), aux(p.data);
  uint64_t pos = ls;
  uint32_t kids = 0;
  bool fallback = false, heavy = false;
  for (;;) {
    uint64_t fe = pos, eq = ~0ull;
    bool bs = false, ctrl = false, hi = false, quote = false;
    while (fe < le) {
      const uint32_t c = rd.at(fe);
      if (c == '\t') break;
      if (eq == ~0ull) { if (c == '=') eq = fe; }
      else { bs = bs || c == '\\'; ctrl = ctrl || c < 0x20u; hi = hi || c >= 0x80u; quote = quote || c == '"'; }
      fe++;
    }
    if (eq != ~0ull) {
      kids++;
      const uint32_t kn = (uint32_t)(eq - pos);
      uint32_t h = 0x9E3779B9u;
      for (uint32_t i = 0; i < kn; i += 8) { const uint32_t nb = kn - i < 8 ? kn - i : 8u; h = key_mix(h, aux.word(pos + i) & (nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1)); }
      h = key_finish(h, kn);
      int32_t kcol = JS_UNKNOWN;
      for (uint32_t s = h & p.slot_mask;; s = (s + 1) & p.slot_mask) {
        const JSlot sl = p.slots[s];
        if (sl.slen == ~0u) break;
        if (sl.hash == h && sl.slen == kn) {
          bool same = true;
          for (uint32_t i = 0; i < kn && same; i += 8) {
            const uint32_t nb = kn - i < 8 ? kn - i : 8u;
            same = (aux.word(pos + i) & (nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1)) == *reinterpret_cast<const uint64_t *>(p.names + sl.soff + i);
          }
          if (same) { kcol = sl.col; if (sl.aux) fallback = true; break; }
        }
      }
      if (p.add_rest && kcol == JS_UNKNOWN) fallback = true;  // `_rest` would not be {}
      if (!fallback && kcol >= 0) {
        JStr sv;
        sv.s = eq + 1; sv.n = (uint32_t)(fe - eq - 1); sv.hash = 0;
        sv.bs = p.tskv_unescape && bs;  // raw bytes ARE the value unless tryToUnescapeTSKV runs
        sv.ctrl = ctrl; sv.plain = !bs && !ctrl && !hi && !quote;
        for (int32_t ci = kcol; ci >= 0; ci = p.cols[ci].next) {
          const int rc = store_cell<HEAVY>(p, p.cols[ci], r, aux, V_STR, eq + 1, fe, sv, 0, false);  // strings never need the map emitter; a nested path needs the kernel that carries it
          if (rc == 2) { heavy = true; break; }
          if (rc != 1) { fallback = true; break; }
        }
        if (heavy) return true;
      }
    }
    if (fe >= le) break;
    pos = fe + 1;
  }
  p.linest[r] = fallback ? JL_FALLBACK : kids == 0 ? JL_SKIPPED : JL_ROW;
  if (!fallback && kids && p.rest_lens) { p.rest_fstart[r] = (uint32_t)ls; p.rest_meta[r] = (JM_REST_EMPTY << 28); p.rest_lens[r] = 2; }
  p.line_pos[r] = (uint32_t)ls;
  return false;
}

__global__ void __launch_bounds__(256) json_parse_lines(JParams p, uint32_t *slow_n, uint32_t *slow_seg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.nseg) return;
  const uint32_t n = p.seg_len[i];
  if (!n) return;
  const uint64_t s = p.seg_start[i];
  const bool again = p.format == TFGPU_JFMT_TSKV ? parse_tskv_line<false>(p, (int64_t)p.seg_ord[i], s, s + n) : parse_json_line<false>(p, (int64_t)p.seg_ord[i], s, s + n);
  if (again) { const uint32_t k = atomicAdd(slow_n, 1u); slow_seg[k] = (uint32_t)i; }  // → json_parse_listed
}
// the lines the wave path hands over (nested values, anything it does not fully understand): one lane per line
__global__ void __launch_bounds__(256) json_parse_listed(JParams p, const uint32_t *slow_n, const uint32_t *slow_seg) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= *slow_n) return;
  const uint32_t i = slow_seg[k];
  const uint64_t s = p.seg_start[i];
  if (p.format == TFGPU_JFMT_TSKV) parse_tskv_line<true>(p, (int64_t)p.seg_ord[i], s, s + p.seg_len[i]);
  else parse_json_line<true>(p, (int64_t)p.seg_ord[i], s, s + p.seg_len[i]);
}

// the lines the tile path hands over, through the lean per-line parser first; what needs the map emitter is listed again
__global__ void __launch_bounds__(256) json_parse_listed_lean(JParams p, const uint32_t *slow_n, const uint32_t *slow_seg, uint32_t *heavy_n, uint32_t *heavy_seg) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= *slow_n) return;
  const uint32_t i = slow_seg[k];
  const uint64_t s = p.seg_start[i];
  if (parse_json_line<false>(p, (int64_t)p.seg_ord[i], s, s + p.seg_len[i])) { const uint32_t q = atomicAdd(heavy_n, 1u); heavy_seg[q] = i; }
}

// ---------------------------------------------------------------------------
// wave path.  One 64-lane wave owns a line.  Stage 1 is simdjson's, on the scalar unit: 64 bytes per step, one
// byte per lane, every character class a ballot (a 64-bit mask in SGPRs), escaped quotes by the add-carry trick,
// the in-string mask by a prefix xor — ~50 scalar ops per 64 bytes, shared by the whole wave.  The masks go to
// LDS; then lane m takes top-level member m: it finds its colon by a rank/select over the colon mask, its key and
// value by next/previous-bit queries, checks that nothing but whitespace sits in between, and runs the same
// store_cell as the per-line path.  Only FLAT objects are taken (scalar values, no duplicate keys, <= 128 members,
// <= JF_MAX bytes); every other line — and every line that does not validate — goes to json_parse_listed, so
// the wave path never has to classify an error.
// ---------------------------------------------------------------------------
static constexpr int JF_MAX = 4096;
static constexpr int JF_WORDS = JF_MAX / 64;
static constexpr int JF_WAVES = 4;
static constexpr int JF_OWN = 512;
static constexpr uint32_t JF_NONE = 0xFFFFFFFFu;
struct JFastLds {
  uint64_t q[JF_WORDS], col[JF_WORDS], com[JF_WORDS], nws[JF_WORDS], bs[JF_WORDS], ctl[JF_WORDS], npl[JF_WORDS], num[JF_WORDS];
  uint16_t ccol[JF_WORDS + 2], ccom[JF_WORDS + 2];
  uint32_t owner[JF_OWN];
};
__device__ __forceinline__ uint64_t jf_find_escaped(uint64_t bs, uint64_t &carry) {  // bytes escaped by an odd backslash run
  bs &= ~carry;
  const uint64_t follows = (bs << 1) | carry;
  const uint64_t even = 0x5555555555555555ull;
  const uint64_t odd_starts = bs & ~even & ~follows;
  const uint64_t sum = odd_starts + bs;
  carry = sum < bs ? 1ull : 0ull;
  return (even ^ (sum << 1)) & follows;
}
__device__ __forceinline__ uint64_t jf_prefix_xor(uint64_t x) { x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32; return x; }
// any set bit of m in [a, b)
__device__ __forceinline__ bool jf_any(const uint64_t *m, uint32_t a, uint32_t b) {
  if (a >= b) return false;
  const uint32_t w0 = a >> 6, w1 = (b - 1) >> 6;
  for (uint32_t w = w0; w <= w1; w++) {
    uint64_t x = m[w];
    if (w == w0) x &= ~0ull << (a & 63);
    if (w == w1) x &= ~0ull >> (63 - ((b - 1) & 63));
    if (x) return true;
  }
  return false;
}
__device__ __forceinline__ bool jf_all(const uint64_t *m, uint32_t a, uint32_t b) {  // every bit of [a, b) set
  if (a >= b) return true;
  const uint32_t w0 = a >> 6, w1 = (b - 1) >> 6;
  for (uint32_t w = w0; w <= w1; w++) {
    uint64_t x = ~m[w];
    if (w == w0) x &= ~0ull << (a & 63);
    if (w == w1) x &= ~0ull >> (63 - ((b - 1) & 63));
    if (x) return false;
  }
  return true;
}
__device__ __forceinline__ uint32_t jf_next(const uint64_t *m, uint32_t p, uint32_t nw) {  // first set bit at or after p
  uint32_t w = p >> 6;
  if (w >= nw) return JF_NONE;
  uint64_t x = m[w] & (~0ull << (p & 63));
  while (!x) { if (++w >= nw) return JF_NONE; x = m[w]; }
  return (w << 6) + (uint32_t)__ffsll((long long)x) - 1;
}
__device__ __forceinline__ uint32_t jf_prev(const uint64_t *m, uint32_t p) {  // last set bit strictly below p
  if (p == 0) return JF_NONE;
  uint32_t w = (p - 1) >> 6;
  uint64_t x = m[w] & (~0ull >> (63 - ((p - 1) & 63)));
  while (!x) { if (w == 0) return JF_NONE; x = m[--w]; }
  return (w << 6) + 63 - (uint32_t)__clzll((long long)x);
}
__device__ __forceinline__ uint32_t jf_select(const uint64_t *m, const uint16_t *c, uint32_t nw, uint32_t j) {  // j-th set bit, c = prefix counts
  uint32_t lo = 0, hi = nw;  // last word with c[w] <= j
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (c[mid] <= j) lo = mid; else hi = mid; }
  uint64_t x = m[lo];
  for (uint32_t k = j - c[lo]; k; k--) x &= x - 1;
  return (lo << 6) + (uint32_t)__ffsll((long long)x) - 1;
}

__global__ void __launch_bounds__(256) json_parse_waves(JParams p, uint32_t *slow_n, uint32_t *slow_seg) {
  __shared__ JFastLds lds_all[JF_WAVES];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  JFastLds &L = lds_all[wv];
  const int64_t nwaves = (int64_t)gridDim.x * JF_WAVES;
  uint32_t iter = 0;
  MemBytes aux(p.data);
  for (int64_t i = (int64_t)blockIdx.x * JF_WAVES + wv; i < p.nseg; i += nwaves) {
    const uint32_t n = p.seg_len[i];
    if (!n) continue;
    iter++;
    const uint64_t base = p.seg_start[i];
    const int64_t r = (int64_t)p.seg_ord[i];
    bool slow = n > JF_MAX;
    uint32_t nm = 0, open_pos = JF_NONE, close_pos = JF_NONE;
    const uint32_t nw = slow ? 0u : (n + 63) >> 6;
    if (!slow) {
      uint64_t esc_carry = 0, str_carry = 0;
      uint32_t ncol = 0, ncom = 0, nopen = 0, nclose = 0, nnws = 0, first_nws = JF_NONE, last_nws = JF_NONE;
      for (uint32_t k = 0; k < nw; k++) {
        const uint32_t pos = k * 64 + lane;
        const uint32_t c = pos < n ? (uint32_t)p.data[base + pos] : 0x20u;
        uint64_t Q = __ballot(c == '"');
        const uint64_t B = __ballot(c == '\\');
        Q &= ~jf_find_escaped(B, esc_carry);
        const uint64_t instr = jf_prefix_xor(Q) ^ str_carry;  // bit i = inside a string after byte i (opening quote 1, closing quote 0)
        str_carry = (uint64_t)((int64_t)instr >> 63);
        const uint64_t outside = ~instr & ~Q;
        const uint64_t col = __ballot(c == ':') & outside, com = __ballot(c == ',') & outside;
        const uint64_t opn = __ballot(c == '{' || c == '[') & outside, cls = __ballot(c == '}' || c == ']') & outside;
        const uint64_t nws = ~__ballot(c == ' ' || c == '\n' || c == '\t' || c == '\r');
        const uint64_t ctl = __ballot(c < 0x20u);
        const uint64_t npl = __ballot(c < 0x20u || c > 0x7Fu || c == '\\' || c == '"' || c == '<' || c == '>' || c == '&');
        const uint64_t num = __ballot((c >= '0' && c <= '9') || c == '.' || c == '-' || c == '+' || c == 'e' || c == 'E');
        if (lane == 0) {
          L.q[k] = Q; L.col[k] = col; L.com[k] = com; L.nws[k] = nws; L.bs[k] = B; L.ctl[k] = ctl; L.npl[k] = npl; L.num[k] = num;
          L.ccol[k] = (uint16_t)ncol; L.ccom[k] = (uint16_t)ncom;
        }
        if (opn && !nopen) open_pos = k * 64 + (uint32_t)__ffsll((long long)opn) - 1;
        if (cls) close_pos = k * 64 + 63 - (uint32_t)__clzll((long long)cls);
        if (nws) { if (first_nws == JF_NONE) first_nws = k * 64 + (uint32_t)__ffsll((long long)nws) - 1; last_nws = k * 64 + 63 - (uint32_t)__clzll((long long)nws); }
        ncol += (uint32_t)__popcll(col); ncom += (uint32_t)__popcll(com); nopen += (uint32_t)__popcll(opn); nclose += (uint32_t)__popcll(cls);
        nnws += (uint32_t)__popcll(nws);
      }
      if (lane == 0) { L.ccol[nw] = (uint16_t)ncol; L.ccom[nw] = (uint16_t)ncom; }
      nm = ncol;
      // ws* { members } ws* with balanced quotes and no nested container
      slow = str_carry != 0 || nopen != 1 || nclose != 1 || open_pos != first_nws || close_pos != last_nws;
      if (!slow) slow = p.data[base + open_pos] != '{' || p.data[base + close_pos] != '}';
      if (!slow) slow = nm == 0 ? (ncom != 0 || nnws != 2) : (ncom != nm - 1 || nm > 128);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- members: validate all, then store ----
    uint32_t ks[2], kn[2], vs[2], ve[2];
    int32_t kcol[2];
    int vt[2];
    bool bad = false;
#pragma unroll
    for (int rd = 0; rd < 2; rd++) {
      const uint32_t m = (uint32_t)rd * 64 + (uint32_t)lane;
      kcol[rd] = JS_UNKNOWN; vt[rd] = -1; ks[rd] = kn[rd] = vs[rd] = ve[rd] = 0;
      if (slow || m >= nm) continue;
      const uint32_t pc = jf_select(L.col, L.ccol, nw, m);
      const uint32_t pprev = m ? jf_select(L.com, L.ccom, nw, m - 1) : open_pos;
      const uint32_t pnext = m + 1 < nm ? jf_select(L.com, L.ccom, nw, m) : close_pos;
      const uint32_t q2 = jf_prev(L.q, pc), q1 = q2 != JF_NONE ? jf_prev(L.q, q2) : JF_NONE;
      if (q1 == JF_NONE || q1 <= pprev || pprev >= pc || pc >= pnext || jf_any(L.nws, pprev + 1, q1) || jf_any(L.nws, q2 + 1, pc)) { bad = true; continue; }
      const uint32_t v0 = jf_next(L.nws, pc + 1, nw);
      if (v0 == JF_NONE || v0 >= pnext) { bad = true; continue; }
      const uint32_t v1 = jf_prev(L.nws, pnext) + 1;  // one past the last byte of the value; > v0
      const uint32_t c0 = p.data[base + v0];
      int t;
      if (c0 == '"') {
        if (v1 - v0 < 2 || jf_next(L.q, v0 + 1, nw) != v1 - 1) { bad = true; continue; }
        t = V_STR;
      } else {
        if (!jf_all(L.nws, v0, v1) || jf_any(L.q, v0, v1)) { bad = true; continue; }
        const uint32_t tn = v1 - v0;
        auto is = [&](const char *w, uint32_t wl) { if (tn != wl) return false; for (uint32_t k = 0; k < wl; k++) if (p.data[base + v0 + k] != (uint8_t)w[k]) return false; return true; };
        if (c0 == 't') { if (!is("true", 4)) { bad = true; continue; } t = V_TRUE; }
        else if (c0 == 'f') { if (!is("false", 5)) { bad = true; continue; } t = V_FALSE; }
        else if (c0 == 'n') { if (!is("null", 4)) { bad = true; continue; } t = V_NULL; }
        else {  // parseRawNumber: a run of [0-9.+-eE]; a lone sign is "unexpected char"; inf / nan spellings: per-line path
          if (!jf_all(L.num, v0, v1) || (tn == 1 && (c0 == '-' || c0 == '+'))) { bad = true; continue; }
          t = V_NUM;
        }
      }
      ks[rd] = q1 + 1; kn[rd] = q2 - q1 - 1; vs[rd] = v0; ve[rd] = v1; vt[rd] = t;
      if (jf_any(L.bs, ks[rd], q2)) { bad = true; continue; }  // key compared after unescaping: per-line path (→ host)
      // which column reads this key?
      uint32_t h = 0x9E3779B9u;
      for (uint32_t k = 0; k < kn[rd]; k += 8) {
        const uint32_t nb = kn[rd] - k < 8 ? kn[rd] - k : 8u;
        h = key_mix(h, aux.word(base + ks[rd] + k) & (nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1));
      }
      h = key_finish(h, kn[rd]);
      int32_t kc = JS_UNKNOWN;
      for (uint32_t sl_ = h & p.slot_mask;; sl_ = (sl_ + 1) & p.slot_mask) {
        const JSlot sl = p.slots[sl_];
        if (sl.slen == ~0u) break;
        if (sl.hash == h && sl.slen == kn[rd]) {
          bool same = true;
          for (uint32_t k = 0; k < kn[rd] && same; k += 8) {
            const uint32_t nb = kn[rd] - k < 8 ? kn[rd] - k : 8u;
            same = (aux.word(base + ks[rd] + k) & (nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1)) == *reinterpret_cast<const uint64_t *>(p.names + sl.soff + k);
          }
          if (same) { kc = sl.col; if (sl.aux) bad = true; break; }
        }
      }
      if (p.add_rest && kc == JS_UNKNOWN) bad = true;  // `_rest` will not be {}: the per-line path writes it
      kcol[rd] = kc;
      if (kc >= 0) { if (kc >= JF_OWN) bad = true; else L.owner[kc] = (iter << 8) | m; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int rd = 0; rd < 2; rd++)  // a key read twice: "the last one wins" needs the members in order
      if (kcol[rd] >= 0 && kcol[rd] < JF_OWN && L.owner[kcol[rd]] != ((iter << 8) | ((uint32_t)rd * 64 + (uint32_t)lane))) bad = true;
    slow = slow || __any(bad);
    if (slow) {
      if (lane == 0) { const uint32_t k = atomicAdd(slow_n, 1u); slow_seg[k] = (uint32_t)i; }
      continue;
    }
    bool fb = false;
#pragma unroll
    for (int rd = 0; rd < 2; rd++) {
      if (vt[rd] < 0) continue;
      JStr sv{};
      uint64_t a0 = base + vs[rd], a1 = base + ve[rd];
      if (vt[rd] == V_STR) {
        sv.s = a0 + 1; sv.n = ve[rd] - vs[rd] - 2;
        sv.bs = jf_any(L.bs, vs[rd] + 1, ve[rd] - 1); sv.ctrl = jf_any(L.ctl, vs[rd] + 1, ve[rd] - 1); sv.plain = !jf_any(L.npl, vs[rd] + 1, ve[rd] - 1);
      }
      for (int32_t ci = kcol[rd]; ci >= 0; ci = p.cols[ci].next)
        if (store_cell<false>(p, p.cols[ci], r, aux, vt[rd], a0, a1, sv, 0, false) != 1) { fb = true; break; }  // scalars only on this path
    }
    fb = __any(fb);
    if (lane == 0) {
      p.linest[r] = fb ? JL_FALLBACK : (nm ? JL_ROW : JL_SKIPPED);
      p.line_pos[r] = (uint32_t)base;
      if (!fb && nm && p.rest_lens) { p.rest_fstart[r] = (uint32_t)base; p.rest_meta[r] = (JM_REST_EMPTY << 28); p.rest_lens[r] = 2; }  // every key was known: {}
    }
  }
}

// ---------------------------------------------------------------------------
// tile path.  The CSV tile parser's shape applied to NDJSON: a workgroup stages a run of consecutive lines in LDS,
// classifies its bytes once with SWAR masks (unescaped quotes by the backslash-run carry trick, the in-string state by a
// prefix xor carried across lanes and waves), indexes the separators { } [ ] , : that lie outside strings, and then
// treats (member, line) pairs as cells dealt to the waves by VALUE KIND — members sorted by the kind of the column that
// reads them, lines fastest — so that a wave runs one typed parse per step and neighbouring lanes store neighbouring
// rows of one column.  The member → column map is looked up once per tile, for its first line; every other line must
// spell the same keys in the same order (bytes compared in LDS), which is what a producer's serializer emits.
// Only what is certain is decided here: compact flat objects (no blank outside strings), values that are strings,
// null / true / false or plain decimal integers.  Every other line — another key sequence, a nested value, a float, a
// blank, anything that does not validate — is listed for json_parse_listed, which re-parses it whole.
// ---------------------------------------------------------------------------
// the rare cells (a float that is not a plain integer, a string with escapes) read their bytes from HBM, where the tile just was
__device__ __forceinline__ bool jt_float_token(const uint8_t *data, uint64_t at, uint32_t n, const double *p10, const uint64_t *p128, double *out) {
  MemBytes aux(data);
  const Field tok{&aux, at, n};
  return ff_best_effort(tok, n, p10, p128, out) == 0;
}
__device__ __forceinline__ uint32_t jt_unescaped_len(const uint8_t *data, uint64_t at, uint32_t n) {
  MemBytes aux(data);
  CountSink cs;
  unescape_walk(aux, at, n, cs);
  return cs.n;
}
enum JtClass : uint32_t { JTC_SKIP = 0, JTC_I8, JTC_I16, JTC_I32, JTC_I64, JTC_U8, JTC_U16, JTC_U32, JTC_U64, JTC_TEXT, JTC_ANY, JTC_BOOL, JTC_TIME, JTC_F64, JTC_COUNT };

__global__ void __launch_bounds__(JT_THREADS, 4) json_parse_tiles(JParams p, int32_t lines_per_tile, uint32_t *slow_n, uint32_t *slow_seg) {
  __shared__ JtLds L;
  uint8_t *const sb = L.sbuf + 16;
  uint16_t *const spos = L.spos, *const qpre = L.qpre, *const bpre = L.bpre, *const lstart = L.lstart, *const lend = L.lend, *const lbase = L.lbase, *const lK = L.lK;
  uint32_t *const qmask = L.qmask, *const misc = L.misc;
  uint8_t *const lslow = L.lslow;
  __shared__ int32_t lrow[JT_LINES];
  __shared__ int16_t mcol[JT_MEM];
  __shared__ uint16_t mks[JT_MEM], mkn[JT_MEM];
  __shared__ uint8_t mcls[JT_MEM], perm[JT_MEM];
  __shared__ uint64_t mp0[JT_MEM], mp1[JT_MEM], mp2[JT_MEM], mp3[JT_MEM];
  __shared__ uint16_t owner[JT_OWN];
  __shared__ uint32_t cfirst[JTC_COUNT + 1];
  // the member map outlives a tile: a workgroup walks many tiles, and a stream keeps its key order
  __shared__ __attribute__((aligned(16))) uint8_t kref[JT_KREF + 32];  // the keys the map was built from, back to back
  __shared__ uint16_t mko[JT_MEM];
  __shared__ uint32_t mapst[4];  // 0: members of the mapped key sequence, 1: the map is valid, 2: lines of this tile that spell other keys
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < 4) mapst[tid] = 0;
  const int64_t ntiles = (p.nseg + lines_per_tile - 1) / lines_per_tile;
  // a workgroup takes a contiguous run of tiles: the short column segments two neighbouring tiles write share cache lines,
  // and written one after the other by the same CU they leave its L2 as whole lines
  const int64_t tiles_each = (ntiles + gridDim.x - 1) / gridDim.x;
  for (int64_t tile = (int64_t)blockIdx.x * tiles_each; tile < min(ntiles, ((int64_t)blockIdx.x + 1) * tiles_each); tile++) {
  __syncthreads();  // the previous tile's LDS is free
  const int64_t i0 = tile * lines_per_tile;
  int nl = (int)min<int64_t>(lines_per_tile, p.nseg - i0);  // lines_per_tile <= JT_LINES <= 64: one lane per line
  auto all_slow = [&]() {  // every line of the tile goes to the per-line kernel
    for (int j = tid; j < nl; j += JT_THREADS) if (p.seg_len[i0 + j]) { const uint32_t k = atomicAdd(slow_n, 1u); slow_seg[k] = (uint32_t)(i0 + j); }
  };
  // the tile's byte range: from its first non-empty line to the end of the last one that still fits (every wave works it
  // out for itself: one vector load per wave instead of a scalar loop over the lines)
  uint32_t first, last, g0;
  {
    const uint32_t ln = lane < nl ? p.seg_len[i0 + lane] : 0u, ls_ = lane < nl ? p.seg_start[i0 + lane] : 0u;
    const uint64_t nonempty = __ballot(ln != 0);
    if (!nonempty) continue;  // nothing but empty lines
    first = (uint32_t)__shfl((int)ls_, __ffsll((long long)nonempty) - 1, 64);
    g0 = first & ~15u;
    const uint64_t fits = __ballot(ln != 0 && ls_ + ln - g0 <= (uint32_t)JT_BYTES);  // lines are in ascending order: a prefix of the non-empty ones
    if (!fits || p.ncols > JT_OWN) { all_slow(); continue; }
    const int hi = 63 - __clzll((long long)fits);
    last = (uint32_t)__shfl((int)(ls_ + ln), hi, 64);
    // the lines behind the last one that fits: per-line kernel
    if (wv == 0 && lane > hi && ln != 0) { const uint32_t k = atomicAdd(slow_n, 1u); slow_seg[k] = (uint32_t)(i0 + lane); }
    nl = hi + 1;
  }

  if (tid < nl) {
    const uint32_t n = p.seg_len[i0 + tid], s = p.seg_start[i0 + tid];
    lstart[tid] = (uint16_t)(n ? s - g0 : 0); lend[tid] = (uint16_t)(n ? s - g0 + n : 0);
    lrow[tid] = n ? (int32_t)p.seg_ord[i0 + tid] : -1;
    lslow[tid] = n ? 0 : 2;  // 2: no line here
    lbase[tid] = 0xFFFFu;
  }
  for (int i = tid; i < JT_OWN; i += JT_THREADS) owner[i] = 0xFFFFu;
  // ---- stage, classify, index the separators, frame the lines (tf_jsontile.hpp) ----
  if (!jt_front(L, p.data, first, last, g0, nl)) { all_slow(); continue; }
  JtTile t;
  t.sb = sb; t.spos = spos; t.qmask = qmask; t.qpre = qpre; t.bpre = bpre; t.g0 = g0;
  auto backslashes_in = [&](uint32_t a, uint32_t b) { return jt_backslashes_in(L, a, b); };
  // the reference line (uniform reads)
  int jref = 0;
  while (jref < nl && lslow[jref] != 0) jref++;  // the first line that is framed: the one whose keys are looked up when there is no map
  const bool build_map = mapst[1] == 0;  // uniform
  if (build_map && (jref >= nl || lslow[jref])) { all_slow(); continue; }
  const uint32_t bref = build_map ? lbase[jref] : 0u;
  const uint32_t K = build_map ? lK[jref] : mapst[0];

  // ---- member map: looked up for the first line of the first tile this workgroup sees, kept while the lines keep spelling it ----
  if (build_map) {
    uint32_t mbad = 0;
    for (uint32_t m = (uint32_t)tid; m < K; m += JT_THREADS) {
      const uint32_t pp = spos[bref + 2 * m], pc = spos[bref + 1 + 2 * m];
      int32_t kc = JS_UNKNOWN;
      uint32_t cls = JTC_SKIP;
      uint64_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
      // sep ws* "key" ws* : — blanks outside strings are skipped the way fastjson's skipWS does
      uint32_t kq = pp + 1, ke = pc;
      while (kq < pc && jt_ws(sb[kq])) kq++;
      while (ke > kq && jt_ws(sb[ke - 1])) ke--;
      const bool framed = ke >= kq + 2 && sb[kq] == '"' && sb[ke - 1] == '"' && sb[pc] == ':' && jt_quotes_in(t, pp + 1, pc) == 2 && backslashes_in(kq, ke) == 0;
      const uint32_t ks = kq + 1, kn = framed ? ke - 1 - ks : 0;
      if (!framed) mbad = 1;
      else {
        uint32_t h = 0x9E3779B9u;
        for (uint32_t k = 0; k < kn; k += 8) {
          const uint32_t nb = kn - k < 8 ? kn - k : 8u;
          h = key_mix(h, jt_word(sb, ks + k) & (nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1));
        }
        h = key_finish(h, kn);
        for (uint32_t sl_ = h & p.slot_mask;; sl_ = (sl_ + 1) & p.slot_mask) {
          const JSlot sl = p.slots[sl_];
          if (sl.slen == ~0u) break;
          if (sl.hash == h && sl.slen == kn) {
            bool same = true;
            for (uint32_t k = 0; k < kn && same; k += 8) {
              const uint32_t nb = kn - k < 8 ? kn - k : 8u;
              same = (jt_word(sb, ks + k) & (nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1)) == *reinterpret_cast<const uint64_t *>(p.names + sl.soff + k);
            }
            if (same) { kc = sl.col; if (sl.aux) mbad = 1; break; }
          }
        }
        if (p.add_rest && kc == JS_UNKNOWN) mbad = 1;  // `_rest` will not be {}: the per-line path writes it
        if (kc >= 0) {
          const JCol c = p.cols[kc];
          if (c.next >= 0) mbad = 1;  // several columns read this key: per-line path
          switch (c.kind) {
            case JK_INT: cls = c.width == 1 ? JTC_I8 : c.width == 2 ? JTC_I16 : c.width == 4 ? JTC_I32 : JTC_I64; q0 = (uint64_t)c.values; break;
            case JK_UINT: cls = c.width == 1 ? JTC_U8 : c.width == 2 ? JTC_U16 : c.width == 4 ? JTC_U32 : JTC_U64; q0 = (uint64_t)c.values; break;
            case JK_F64: cls = JTC_F64; q0 = (uint64_t)c.values; break;
            case JK_BOOL: cls = JTC_BOOL; q0 = (uint64_t)c.values; break;
            case JK_TEXT: cls = JTC_TEXT; q0 = (uint64_t)c.lens; q1 = (uint64_t)c.fstart; q3 = (uint64_t)c.meta; break;
            case JK_ANY: cls = JTC_ANY; q0 = (uint64_t)c.lens; q1 = (uint64_t)c.fstart; q3 = (uint64_t)c.meta; break;
            default: cls = JTC_TIME; q0 = (uint64_t)c.values; q1 = (uint64_t)c.nanos;
          }
          q2 = (uint64_t)c.cellst;
        }
      }
      mcol[m] = (int16_t)kc; mks[m] = (uint16_t)ks; mkn[m] = (uint16_t)kn; mcls[m] = (uint8_t)cls;
      mp0[m] = q0; mp1[m] = q1; mp2[m] = q2; mp3[m] = q3;
    }
    if (mbad) misc[0] = 1u;
  }
  __syncthreads();
  if (build_map) {
  // a key read twice: "the last one wins" needs the members in order → per-line path
  for (uint32_t m = (uint32_t)tid; m < K; m += JT_THREADS) if (mcol[m] >= 0) owner[mcol[m]] = (uint16_t)m;
  if (tid == 0) {  // where each key's text is kept
    uint32_t o = 0;
    for (uint32_t m = 0; m < K; m++) { mko[m] = (uint16_t)o; o += ((uint32_t)mkn[m] + 7u) & ~7u; if (o > (uint32_t)JT_KREF) { misc[0] = 1u; break; } }
  }
  }
  __syncthreads();
  if (build_map && !misc[0]) {
  for (uint32_t m = (uint32_t)tid; m < K; m += JT_THREADS) {
    if (mcol[m] >= 0 && owner[mcol[m]] != (uint16_t)m) misc[0] = 1u;
    for (uint32_t k = 0; k < mkn[m]; k += 8) *reinterpret_cast<uint64_t *>(kref + mko[m] + k) = jt_word(sb, (uint32_t)mks[m] + k);  // (the bytes past the key are never compared)
  }
  for (uint32_t m = (uint32_t)tid; m < K; m += JT_THREADS) if (mcol[m] >= 0) owner[mcol[m]] = 0xFFFFu;  // for the next map
  }
  // members by class (counting sort by wave 0: K <= 192 is three rounds of 64 lanes)
  if (build_map && wv == 0) {
    uint32_t nxt = 0;
    for (uint32_t c = 0; c < JTC_COUNT; c++) {
      if (lane == 0) cfirst[c] = nxt;
      for (uint32_t r0 = 0; r0 < K; r0 += 64) {
        const uint32_t m = r0 + (uint32_t)lane;
        const bool mine = m < K && mcls[m] == c;
        const uint64_t bal = __ballot(mine);
        if (mine) perm[nxt + lanes_below(bal)] = (uint8_t)m;
        nxt += (uint32_t)__popcll(bal);
      }
    }
    if (lane == 0) cfirst[JTC_COUNT] = nxt;
  }
  __syncthreads();
  if (misc[0]) { all_slow(); continue; }  // uniform (a map that could not be built stays invalid)
  if (build_map && tid == 0) { mapst[0] = K; mapst[1] = 1u; }
  if (tid == 0) mapst[2] = 0;
  __syncthreads();

  // ---- cells: class by class, items (member, line) with lines fastest, slots of 64 dealt round-robin to the waves ----
  {
    const uint32_t unl = (uint32_t)nl;
    const float inv_nl = __uint_as_float(__float_as_uint(1.0f / (float)unl) - 2u);
    uint32_t sbase = 0;
    for (uint32_t c = 0; c < JTC_COUNT; c++) {
      const uint32_t f0 = cfirst[c], nm = cfirst[c + 1] - f0;
      if (!nm) continue;
      const uint32_t items = nm * unl, nslots = (items + 63) >> 6;
      for (uint32_t s = sbase + (((uint32_t)wv - sbase) & 7u); s < sbase + nslots; s += 8) {
        const uint32_t it = (s - sbase) * 64 + (uint32_t)lane;
        if (it >= items) continue;
        uint32_t oi = (uint32_t)(__uint2float_rz(it) * inv_nl), j = it - oi * unl;  // it / unl by a reciprocal a hair too small + one correction
        if (j >= unl) { oi++; j -= unl; }
        if (lslow[j]) continue;
        if (lK[j] != K) { if (lslow[j] == 0) { lslow[j] = 1; atomicAdd(&mapst[2], 1u); } continue; }
        const uint32_t m = perm[f0 + oi];
        const uint32_t b = lbase[j];
        const uint32_t pp = spos[b + 2 * m], pc = spos[b + 1 + 2 * m], pn = spos[b + 2 + 2 * m];
        const uint32_t kn = mkn[m];
        // sep ws* "key" ws* : ws* value ws* sep — the key spelled as in the reference line
        uint32_t kq = pp + 1, ke = pc;
        while (kq < pc && jt_ws(sb[kq])) kq++;
        while (ke > kq && jt_ws(sb[ke - 1])) ke--;
        bool ok = ke == kq + 2 + kn && sb[pc] == ':' && sb[kq] == '"' && sb[ke - 1] == '"' && sb[pn] == (m + 1 == K ? '}' : ',');
        if (ok && !jt_same2(sb, kq + 1, kref, mko[m], kn)) { ok = false; if (lslow[j] == 0) atomicAdd(&mapst[2], 1u); }
        uint32_t vs = pc + 1, ve = pn;
        while (vs < pn && jt_ws(sb[vs])) vs++;
        while (ve > vs && jt_ws(sb[ve - 1])) ve--;
        const uint32_t n = ve - vs;
        ok = ok && n > 0;
        const int32_t r = lrow[j];
        if (ok) {
          const uint32_t c0 = sb[vs];
          const bool is_null = jt_lit(sb, vs, n, 0x6C6C756Eu, 4);
          uint8_t *const cellst = TF_GLOBAL_PTR(uint8_t, mp2[m]);
          if (c == JTC_SKIP) {
            // nobody reads the value, but the line must still be one the per-line parser accepts
            if (c0 == '"') ok = n >= 2 && sb[ve - 1] == '"' && jt_quotes_in(t, vs, ve) == 2;
            else { bool ng; uint64_t mag; uint32_t nd; ok = is_null || jt_lit(sb, vs, n, 0x65757274u, 4) || jt_lit(sb, vs, n, 0x736C6166u, 5) || (jt_quotes_in(t, vs, ve) == 0 && (jt_int_token(sb, vs, ve, &ng, &mag, &nd) || jt_number_chars(sb, vs, ve))); }
          } else if (is_null) {
            cellst[r] = JC_NIL;  // (text lengths are zero already)
          } else if (c >= JTC_I8 && c <= JTC_U64) {
            bool ng; uint64_t mag; uint32_t nd;
            ok = c0 != '"' && jt_int_token(sb, vs, ve, &ng, &mag, &nd);
            if (c >= JTC_U8) ok = ok && !ng;  // ParseUint64BestEffort of "-5" is 0: per-line path
            if (ok) {
              // fastfloat.ParseInt64BestEffort / ParseUint64BestEffort: a token of more than 18 characters goes through
              // strconv.ParseInt, whose range error makes the value 0; then intN(v.GetInt()), Go's truncating conversion
              int64_t v = ng ? (int64_t)(0 - mag) : (int64_t)mag;
              if (c <= JTC_I64 && n > 18 && mag > (ng ? (1ull << 63) : (1ull << 63) - 1)) v = 0;
              const uint32_t w = (c - 1u) & 3u;
              if (w == 0) TF_GLOBAL_PTR(int8_t, mp0[m])[r] = (int8_t)v;
              else if (w == 1) TF_GLOBAL_PTR(int16_t, mp0[m])[r] = (int16_t)v;
              else if (w == 2) TF_GLOBAL_PTR(int32_t, mp0[m])[r] = (int32_t)v;
              else TF_GLOBAL_PTR(int64_t, mp0[m])[r] = v;
              cellst[r] = JC_OK;
            }
          } else if (c == JTC_TIME) {
            // extractTimeValue: json.Number → ParseInt; float64 → int64(math.Abs(f)).  Plain digits that a float64 holds exactly
            bool ng; uint64_t mag; uint32_t nd;
            ok = c0 != '"' && jt_int_token(sb, vs, ve, &ng, &mag, &nd) && !ng && nd <= 15;
            if (ok) { TF_GLOBAL_PTR(int64_t, mp0[m])[r] = (int64_t)mag; TF_GLOBAL_PTR(int32_t, mp1[m])[r] = 0; cellst[r] = JC_OK; }
          } else if (c == JTC_F64) {
            bool ng; uint64_t mag; uint32_t nd;
            if (c0 != '"' && jt_int_token(sb, vs, ve, &ng, &mag, &nd) && nd <= 15 && !(ng && mag == 0)) {
              TF_GLOBAL_PTR(double, mp0[m])[r] = ng ? -(double)mag : (double)mag; cellst[r] = JC_OK;  // exact: what ParseBestEffort gives
            } else {
              ok = c0 != '"' && jt_number_chars(sb, vs, ve);
              double v = 0;
              if (ok) ok = jt_float_token(p.data, (uint64_t)g0 + vs, n, p.pow10, p.pow128, &v);  // (bytes from HBM: the tile just read them)
              if (ok) { TF_GLOBAL_PTR(double, mp0[m])[r] = v; cellst[r] = JC_OK; }
            }
          } else if (c == JTC_BOOL) {
            const bool tr = jt_lit(sb, vs, n, 0x65757274u, 4), fl = jt_lit(sb, vs, n, 0x736C6166u, 5);
            ok = tr || fl;
            if (ok) { TF_GLOBAL_PTR(uint8_t, mp0[m])[r] = tr ? 1 : 0; cellst[r] = JC_OK; }
          } else {  // JTC_TEXT, JTC_ANY
            uint32_t *const lens = TF_GLOBAL_PTR(uint32_t, mp0[m]), *const fstart = TF_GLOBAL_PTR(uint32_t, mp1[m]), *const meta = TF_GLOBAL_PTR(uint32_t, mp3[m]);
            if (c0 == '"') {
              ok = n >= 2 && sb[ve - 1] == '"' && jt_quotes_in(t, vs, ve) == 2;
              const uint32_t sn = n - 2, ss = vs + 1;
              const bool bs = ok && backslashes_in(ss, ss + sn) != 0;
              if (ok && c == JTC_TEXT) {
                if (!bs) { fstart[r] = g0 + ss; meta[r] = sn | (JM_COPY << 28); lens[r] = sn; }
                else { fstart[r] = g0 + ss; meta[r] = sn | (JM_UNESCAPE << 28); lens[r] = jt_unescaped_len(p.data, (uint64_t)g0 + ss, sn); }
                cellst[r] = JC_OK;
              } else if (ok) {
                // `any` holding a string: printable ASCII without " \ < > &, not an object or `null` in disguise (ParseVal)
                ok = !bs && sn <= 256;
                for (uint32_t q = 0; q < sn && ok; q++) { const uint32_t ch = sb[ss + q]; ok = ch >= 0x20u && ch <= 0x7Eu && ch != '<' && ch != '>' && ch != '&'; }
                if (ok) {
                  uint32_t i = 0;
                  while (i < sn && (sb[ss + i] == ' ' || sb[ss + i] == '\t')) i++;
                  if (i < sn && (sb[ss + i] == '{' || sb[ss + i] == 'n')) ok = false;
                }
                if (ok) { fstart[r] = g0 + ss; meta[r] = sn | (JM_QUOTED << 28); lens[r] = sn + 2; cellst[r] = JC_OK; }
              }
            } else {
              const bool lit = jt_lit(sb, vs, n, 0x65757274u, 4) || jt_lit(sb, vs, n, 0x736C6166u, 5);
              bool ng = false; uint64_t mag = 0; uint32_t nd = 0;
              const bool num = !lit && jt_quotes_in(t, vs, ve) == 0 && jt_int_token(sb, vs, ve, &ng, &mag, &nd);
              if (c == JTC_TEXT) ok = lit || num || (jt_quotes_in(t, vs, ve) == 0 && jt_number_chars(sb, vs, ve));  // the raw token
              else ok = lit || (num && p.use_numbers && !(nd > 1 && sb[ve - nd] == '0'));  // json.Number text as it stands; float64 text: per-line path
              if (ok) { fstart[r] = g0 + vs; meta[r] = n | (JM_COPY << 28); lens[r] = n; cellst[r] = JC_OK; }
            }
          }
        }
        if (!ok) lslow[j] = 1;
      }
      sbase += nslots;
    }
  }
  __syncthreads();
  // ---- per line: a row, or handed to the per-line kernel ----
  if (tid < nl && lslow[tid] != 2) {
    if (lslow[tid]) { const uint32_t k = atomicAdd(slow_n, 1u); slow_seg[k] = (uint32_t)(i0 + tid); }
    else {
      const int32_t r = lrow[tid];
      p.linest[r] = JL_ROW;
      p.line_pos[r] = g0 + lstart[tid];
      if (p.rest_lens) { p.rest_fstart[r] = g0 + lstart[tid]; p.rest_meta[r] = (JM_REST_EMPTY << 28); p.rest_lens[r] = 2; }  // every key was known: {}
    }
  }
  // most of the tile spelled other keys: the stream changed its key order — the next tile maps its own first line
  if (tid == 0 && mapst[2] * 2 > (uint32_t)nl) mapst[1] = 0;
  }  // tiles
}

#include "tf_jsonquick.inc"

// ---------------------------------------------------------------------------
// line boundaries
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint32_t v) {  // first i: a[i] >= v
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}
__device__ __forceinline__ uint32_t upper_bound_u32(const uint32_t *a, uint32_t n, uint32_t v) {  // first i: a[i] > v
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
  return lo;
}
// merged, sorted boundaries: nl[0..nnl) (offset one past each '\n') and ms[0..nmsg) (message starts);
// equal values keep the message start first.  bounds[nnl + nmsg] = len.
__global__ void json_merge_bounds(const uint32_t *nl, uint32_t nnl, const uint32_t *ms, uint32_t nmsg, uint32_t len, uint32_t *bounds, uint32_t *rank_ms) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nnl) { const uint32_t v = nl[t]; bounds[t + upper_bound_u32(ms, nmsg, v)] = v; }
  else if (t - nnl < nmsg) {
    const uint32_t m = t - nnl, v = ms[m];
    const uint32_t r = m + lower_bound_u32(nl, nnl, v);
    bounds[r] = v; rank_ms[m] = r;
  }
  if (t == 0) bounds[nnl + nmsg] = len;
}
// bufio.ScanLines: the token is the segment minus its '\n' and one trailing '\r'; empty tokens are skipped
__global__ void json_segments(const uint8_t *data, const uint32_t *bounds, int64_t nseg, uint32_t *seg_start, uint32_t *seg_len, uint32_t *flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg) return;
  const uint32_t s = bounds[i];
  uint32_t e = bounds[i + 1];
  if (e > s && data[e - 1] == '\n') e--;
  if (e > s && data[e - 1] == '\r') e--;
  seg_start[i] = s; seg_len[i] = e - s; flag[i] = e > s ? 1u : 0u;
}

// ---------------------------------------------------------------------------
// rows: key / required rules (makeChangeItem :355-372) in schema order, dedupe keys, validity
// ---------------------------------------------------------------------------
struct JFinish {
  const JCol *cols; int32_t ncols; int64_t nlines;
  const uint8_t *linest; uint8_t null_keys_allowed;
  uint8_t *code; int32_t *ecol; uint32_t *keep; uint32_t *nerr;
  // message lookup + dedupe keys
  const uint32_t *line_pos; const uint32_t *ms; uint32_t nmsg; const uint32_t *rank_ms; const uint32_t *seg_ord;
  const uint64_t *msg_offset; const int64_t *msg_wtime;
  uint32_t *part_id;
  int64_t *ts_sec; int32_t *ts_nanos; uint64_t *off_out; uint32_t *idx_out;  // NULL without AddDedupeKeys
};
__global__ void __launch_bounds__(256) json_finish(JFinish p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nlines) return;
  const uint8_t st = p.linest[r];
  int code = 0, ecol = -1;
  if (st == JL_SKIPPED) code = JCODE_SKIPPED;
  else if (st == JL_SYNTAX) code = TFGPU_ROW_JSON_SYNTAX;
  else if (st == JL_FALLBACK) code = TFGPU_ROW_HOST_FALLBACK;
  else {
    for (int32_t c = 0; c < p.ncols; c++) {
      const uint8_t cs = p.cols[c].cellst[r];
      const bool key = p.cols[c].flags & TFGPU_COL_KEY, req = p.cols[c].flags & TFGPU_COL_REQUIRED;
      const bool nested = p.cols[c].flags & JCOL_NESTED;  // a ParseVal error behind lookupComplex: _unparsed whatever the column's flags (generic_parser.go:338-343)
      if (cs == JC_ERR) { if (nested ? !p.null_keys_allowed : ((!p.null_keys_allowed && key) || req)) { code = TFGPU_ROW_PARSE_VAL; ecol = c; break; } }
      else if (cs == JC_NIL && (key || req) && !p.null_keys_allowed) { code = TFGPU_ROW_NIL_KEY; ecol = c; break; }
    }
  }
  p.code[r] = (uint8_t)code; p.ecol[r] = ecol; p.keep[r] = code ? 0u : 1u;
  if (code) {
    for (int32_t c = 0; c < p.ncols; c++) if (p.cols[c].lens) p.cols[c].lens[r] = 0;
    if (code != JCODE_SKIPPED) atomicAdd(p.nerr, 1u);
  }
  const uint32_t m = upper_bound_u32(p.ms, p.nmsg, p.line_pos[r]) - 1;
  p.part_id[r] = m;
  if (p.ts_sec) {
    const int64_t wt = p.msg_wtime ? p.msg_wtime[m] : 0;
    int64_t sec = wt / 1000000000, ns = wt % 1000000000;
    if (ns < 0) { ns += 1000000000; sec--; }
    p.ts_sec[r] = sec; p.ts_nanos[r] = (int32_t)ns;
    p.off_out[r] = p.msg_offset ? p.msg_offset[m] : 0;
    p.idx_out[r] = (uint32_t)r - p.seg_ord[p.rank_ms[m]] + 1;  // 1-based index of the line inside its message
  }
}
// validity bit = the cell holds a value; one thread per output byte, blockIdx.y = column.  has_nil[column] is raised
// when a KEPT line holds a nil there: columns without one travel without a bitmap (NULL = no nil values).
__global__ void __launch_bounds__(256) json_validity(const JCol *cols, int64_t nlines, uint8_t *const *out, const uint32_t *keep, uint32_t *has_nil) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b * 8 >= nlines) return;
  const uint8_t *cs = cols[blockIdx.y].cellst;
  uint32_t v = 0;
  bool nil = false;
  for (int j = 0; j < 8; j++) {
    const int64_t k = b * 8 + j;
    if (k >= nlines) break;
    if (cs[k] == JC_OK) v |= 1u << j; else if (keep[k]) nil = true;
  }
  out[blockIdx.y][b] = (uint8_t)v;
  if (nil) has_nil[blockIdx.y] = 1u;
}
// constant text columns (`_partition` = PartID, `_rest` = {}): offsets[r] = r * n, data = n bytes repeated
__global__ void json_const_text(const uint8_t *text, uint32_t n, int64_t nrows, uint32_t *offsets, uint8_t *data) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > nrows) return;
  offsets[r] = (uint32_t)r * n;
  if (r < nrows) for (uint32_t i = 0; i < n; i++) data[(uint64_t)r * n + i] = text[i];
}

// ---------------------------------------------------------------------------
// text payloads: lane = (column, line) cell
// ---------------------------------------------------------------------------
struct JCopyCol { const uint32_t *offsets, *fstart, *meta; uint8_t *out; };
// HEAVY == false: every cell but those the map emitter writes; HEAVY == true: only those (a second launch over the same grid)
// The cells that are a plain byte range of the input (JM_COPY: strings without escapes, raw number / literal tokens — nearly all of
// them) are packed destination-centrically (tf_segcopy.hpp: a lane owns aligned 8-byte words of the column's payload and pulls
// their bytes from the text); every other cell is zero-filled here and written by json_copy_cells afterwards.
__global__ void __launch_bounds__(256) json_copy_words(const uint8_t *data, const JCopyCol *cols, int64_t nlines) {
  __shared__ uint32_t doff[256 + 1];
  __shared__ uint32_t soff[256];
  const JCopyCol c = cols[blockIdx.y];
  auto so = [&](int64_t g) { return (c.meta[g] >> 28) == JM_COPY ? c.fstart[g] : SEG_NONE; };
  segcopy_run<1>(c.offsets, nlines, (int64_t)blockIdx.x * 256, data, c.out, so, doff, soff);
}
template <bool HEAVY>
__global__ void __launch_bounds__(256) json_copy_cells(JParams jp, const JCopyCol *cols, int64_t nlines, int plain_done) {
  const uint8_t *data = jp.data; const double *pow10 = jp.pow10; const uint64_t *pow128 = jp.pow128;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nlines) return;
  const JCopyCol c = cols[blockIdx.y];
  const uint32_t o0 = c.offsets[r], n = c.offsets[r + 1] - o0;
  if (!n) return;
  const uint32_t meta = c.meta[r], srclen = meta & 0x0FFFFFFFu, mode = meta >> 28;
  const uint64_t s = c.fstart[r];
  uint8_t *dst = c.out + o0;
  MemBytes rd(data);
  if ((mode == JM_ANYCANON || mode == JM_REST) != HEAVY) return;
  if (mode == JM_COPY && plain_done) return;  // json_copy_words moved it
  if constexpr (HEAVY) {
    StoreSink sk{dst};
    if (mode == JM_ANYCANON) fj_emit_any(sk, jp, rd, s, s + srclen); else rest_emit(sk, jp, rd, s, s + srclen);
    sk.finish();
    return;
  }
  if (mode == JM_UNESCAPE) { StoreSink sk{dst}; unescape_walk(rd, s, srclen, sk); sk.finish(); return; }
  if (mode == JM_COMPACT) { StoreSink sk{dst}; compact_walk(rd, s, srclen, sk); sk.finish(); return; }
  if (mode == JM_TSKV) { StoreSink sk{dst}; tskv_walk(rd, s, srclen, sk); sk.finish(); return; }
  if (mode == JM_REST_EMPTY) { dst[0] = '{'; dst[1] = '}'; return; }
  if (mode == JM_SCRATCH) { MemBytes rs(jp.scratch); StoreSink sk{dst}; for (uint32_t i = 0; i < srclen; i++) sk.put(rs.at(s + i)); sk.finish(); return; }
  if (mode == JM_FLOAT) {  // the number token again, as encoding/json prints the float64 it parses to
    const Field tok{&rd, s, srclen};
    double v = 0;
    ff_best_effort(tok, srclen, pow10, pow128, &v);
    dev::StoreOut so{dst};
    dev::fmt_json_float(so, v, 64);
    return;
  }
  StoreSink sk{dst};
  if (mode == JM_QUOTED) sk.put('"');
  for (uint32_t i = 0; i < srclen; i++) sk.put(rd.at(s + i));
  if (mode == JM_QUOTED) sk.put('"');
  sk.finish();
}

// Which text columns hold cells that are not a plain byte range (bit 0: escapes to undo, compaction, quoting, a float to print,
// scratch text; bit 1: cells the map emitter writes)?  A lean kernel at full occupancy: the kernels that carry those walkers are
// launched for the columns that need them, not for every (line, column) pair to find out that nearly all of them have nothing to do
// (two such launches were 0.46 ms of the json step's 3.4).
__global__ void __launch_bounds__(256) json_mark_special(const uint32_t *lens_all, int64_t seg_stride, const uint32_t *const *metas, int64_t nlines, uint32_t *spec) {
  constexpr int RPT = 16;  // rows per lane: a lane that looks at one cell and leaves makes the launch dispatch-bound (34 k workgroups: 0.2 ms)
  const int s = (int)blockIdx.y;
  const uint32_t *off = lens_all + (int64_t)s * seg_stride, *meta = metas[s];
  uint32_t f = 0;
#pragma unroll 4
  for (int k = 0; k < RPT; k++) {
    const int64_t r = ((int64_t)blockIdx.x * RPT + k) * 256 + threadIdx.x;
    if (r < nlines && off[r + 1] != off[r]) { const uint32_t mode = meta[r] >> 28; f |= (mode == JM_ANYCANON || mode == JM_REST) ? 2u : mode != JM_COPY ? 1u : 0u; }
  }
  const uint32_t any = (__any(f & 1u) ? 1u : 0u) | (__any(f & 2u) ? 2u : 0u);
  // (one atomic per column, not per wave: a flag that is already up is seen by a plain L2 read — 134 k atomics on 33 words were 0.5 ms)
  if (any && (threadIdx.x & 63) == 0 && (__atomic_load_n(&spec[s], __ATOMIC_RELAXED) & any) != any) atomicOr(&spec[s], any);
}
__global__ void json_collect(const uint32_t *nerr, const uint32_t *lens_all, int64_t seg_stride, int64_t nrows, int nstr, const uint32_t *has_nil, int ncols, const uint32_t *spec, uint32_t *out) {
  const int i = threadIdx.x;
  if (i == 0) out[0] = *nerr;
  for (int s = i; s < nstr; s += blockDim.x) { out[1 + s] = lens_all[(int64_t)s * seg_stride + nrows]; out[1 + nstr + ncols + s] = spec[s]; }
  for (int c = i; c < ncols; c += blockDim.x) out[1 + nstr + c] = has_nil[c];
}

// math.Pow10(n), n = -323..308, exactly as Go builds it: pow10postab32[n/32] * pow10tab[n%32] and
// pow10negtab32[-n/32] / pow10tab[-n%32], every table entry being the correctly rounded literal.
const double *pow10_table() {
  Context &cx = ctx();
  if (!cx.pow10tab) {
    std::vector<double> t(632);
    for (int n = -323; n <= 308; n++) {
      char a[16], b[16];
      if (n >= 0) { std::snprintf(a, sizeof a, "1e%d", (n / 32) * 32); std::snprintf(b, sizeof b, "1e%d", n % 32); t[(size_t)(n + 323)] = std::strtod(a, nullptr) * std::strtod(b, nullptr); }
      else { std::snprintf(a, sizeof a, "1e-%d", ((-n) / 32) * 32); std::snprintf(b, sizeof b, "1e%d", (-n) % 32); t[(size_t)(n + 323)] = std::strtod(a, nullptr) / std::strtod(b, nullptr); }
    }
    static const uint64_t P128[696][2] = {
#include "tf_pow10_128.inc"
    };
    cx.pow10tab = dalloc(t.size() * 8 + sizeof P128);
    h2d(cx.pow10tab->p, t.data(), t.size() * 8);
    h2d((char *)cx.pow10tab->p + t.size() * 8, P128, sizeof P128);
    tf::sync();
  }
  return ptr<double>(cx.pow10tab);
}

static inline unsigned jblocks(int64_t n, int t) { return (unsigned)std::max<int64_t>(1, (n + t - 1) / t); }

}  // namespace tf

using namespace tf;

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

namespace {
struct AuxCol { std::string name; int dtype; uint32_t flags; };

bool has_name(const tfgpu_schema *f, const std::vector<AuxCol> &aux, const std::string &n) {
  for (int i = 0; i < f->ncols; i++) if (n == (f->cols[i].name ? f->cols[i].name : "")) return true;
  for (auto &a : aux) if (a.name == n) return true;
  return false;
}
// addAuxFields (generic_parser.go:112-154) with dedupColumnName (:89-96)
std::vector<AuxCol> aux_columns(const tfgpu_json_options *o, const tfgpu_schema *f) {
  std::vector<AuxCol> aux;
  auto add = [&](std::string name, int dtype, uint32_t flags) {
    while (has_name(f, aux, name)) name = "_delivery_" + name;
    aux.push_back(AuxCol{name, dtype, flags});
  };
  if (o->add_rest) add("_rest", TFGPU_T_ANY, 0);
  if (o->add_dedupe_keys) {
    bool skip = false;  // :127-136: with MarkDedupeKeysAsSystem the dedupe keys stop being keys once a user field is one
    if (o->mark_dedupe_keys_as_system) for (int i = 0; i < f->ncols; i++) if (f->cols[i].flags & TFGPU_COL_KEY) skip = true;
    const uint32_t kf = skip ? 0u : (uint32_t)(TFGPU_COL_KEY | TFGPU_COL_REQUIRED);
    add("_timestamp", TFGPU_T_TIMESTAMP, kf); add("_partition", TFGPU_T_BYTES, kf); add("_offset", TFGPU_T_UINT64, kf); add("_idx", TFGPU_T_UINT32, kf);
  }
  return aux;
}
char *dup_cstr(const std::string &s) { char *r = (char *)std::malloc(s.size() + 1); std::memcpy(r, s.c_str(), s.size() + 1); return r; }
uint32_t key_hash_host(const std::string &s) {  // = scan_string's hash of the same bytes
  uint32_t h = 0x9E3779B9u;
  for (size_t k = 0; k < s.size(); k += 8) {
    uint64_t w = 0;
    for (size_t i = 0; i < 8 && k + i < s.size(); i++) w |= (uint64_t)(unsigned char)s[k + i] << (8 * i);
    h = (h ^ (uint32_t)w) * 0x85EBCA6Bu; h = (h ^ (uint32_t)(w >> 32)) * 0xC2B2AE35u;
    h ^= h >> 15;
  }
  return h ^ ((uint32_t)s.size() * 0x27D4EB2Fu);
}
}  // namespace

extern "C" int tfgpu_json_result_schema(const tfgpu_json_options *opts, const tfgpu_schema *fields, tfgpu_schema **out) {
  TF_API_BEGIN
  if (!opts || !fields || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_json_result_schema: null argument");
  std::vector<AuxCol> aux = aux_columns(opts, fields);
  auto *s = (tfgpu_schema *)std::calloc(1, sizeof(tfgpu_schema));
  s->cols = (tfgpu_colschema *)std::calloc((size_t)fields->ncols + aux.size() + 1, sizeof(tfgpu_colschema));
  for (int i = 0; i < fields->ncols; i++) {
    const tfgpu_colschema &c = fields->cols[i];
    tfgpu_colschema &o = s->cols[s->ncols++];
    o.name = dup_cstr(c.name ? c.name : ""); o.dtype = c.dtype; o.flags = c.flags;
    o.path = dup_cstr(c.path ? c.path : ""); o.original_type = dup_cstr(c.original_type ? c.original_type : "");
  }
  for (auto &a : aux) {
    tfgpu_colschema &o = s->cols[s->ncols++];
    o.name = dup_cstr(a.name); o.dtype = a.dtype; o.flags = a.flags; o.path = dup_cstr(""); o.original_type = dup_cstr("");
  }
  *out = s;
  return TFGPU_OK;
  TF_API_END
}

extern "C" int tfgpu_json_parse(const tfgpu_json_options *opts, const tfgpu_schema *fields, const void *bytes, uint64_t len, int mem,
                                const tfgpu_messages *msgs, tfgpu_dbatch **out, tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs) {
  TF_API_BEGIN
  if (!opts || !fields || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_json_parse: null argument");
  if (len >= 0xFFFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_json_parse: batch must be < 4 GiB of JSON (32-bit cell offsets; Bufferer-sized batches are far below, bufferer.go:117-249)");
  if (opts->format != TFGPU_JFMT_JSON && opts->format != TFGPU_JFMT_TSKV) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_json_parse: unknown format");
  const bool tskv = opts->format == TFGPU_JFMT_TSKV;
  if (opts->unescape_string_values && !tskv) return tf::fail(TFGPU_ERR_UNSUPPORTED, "json: UnescapeStringValues (tryToUnescapeJSON) is not device-resident");
  if (opts->unpack_bytes_base64) return tf::fail(TFGPU_ERR_UNSUPPORTED, "json: UnpackBytesBase64 is not device-resident");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  hipStream_t st = cx.stream;
  const int nraw = fields->ncols;
  std::vector<AuxCol> aux = aux_columns(opts, fields);

  // ---- column plan ----
  std::vector<JCol> cols((size_t)nraw);
  std::vector<std::string> lookup((size_t)nraw);
  std::vector<std::pair<int, std::string>> nested_blobs;  // (column, its path names behind the first)
  for (int i = 0; i < nraw; i++) {
    const tfgpu_colschema &sc = fields->cols[i];
    const std::string name = sc.name ? sc.name : "", path = (sc.path && sc.path[0]) ? sc.path : name;
    const std::string &key = path;  // makeChangeItem reads item[key.ColPath()] (:351); IgnoreColumnPaths only re-keys colTypeMap / known
    lookup[(size_t)i] = key;
    JCol &c = cols[(size_t)i];
    std::memset(&c, 0, sizeof c);
    c.flags = sc.flags; c.next = -1;
    if (key.find('.') != std::string::npos || key.find('/') != std::string::npos) {
      // IsNestedKey (col_schema.go:95-97) → lookupComplex: strings.Split(path, "."), or by "/" when that gives one part (lookup.go:11-14)
      const char sep = key.find('.') != std::string::npos ? '.' : '/';
      std::vector<std::string> segs;
      for (size_t a = 0;;) { const size_t b = key.find(sep, a); segs.push_back(key.substr(a, b == std::string::npos ? std::string::npos : b - a)); if (b == std::string::npos) break; a = b + 1; }
      if (segs.size() > 16) return tf::fail(TFGPU_ERR_UNSUPPORTED, "json: column " + name + ": a nested ColSchema.Path of more than 16 names");
      if (opts->add_rest) return tf::fail(TFGPU_ERR_UNSUPPORTED, "json: column " + name + ": a nested ColSchema.Path together with AddRest (the top-level member is not a known name: it belongs to _rest)");
      lookup[(size_t)i] = segs[0];
      std::string blob;
      for (size_t k = 1; k < segs.size(); k++) { if (segs[k].size() > 0xFFFF) return tf::fail(TFGPU_ERR_UNSUPPORTED, "json: nested path name too long"); blob += (char)(segs[k].size() & 0xFF); blob += (char)(segs[k].size() >> 8); blob += segs[k]; }
      nested_blobs.push_back({i, blob});
      c.npath_n = (uint32_t)segs.size() - 1;
      c.flags |= JCOL_NESTED;
    }
    switch (sc.dtype) {
      case TFGPU_T_INT8: c.kind = JK_INT; c.width = 1; break;
      case TFGPU_T_INT16: c.kind = JK_INT; c.width = 2; break;
      case TFGPU_T_INT32: c.kind = JK_INT; c.width = 4; break;
      case TFGPU_T_INT64: c.kind = JK_INT; c.width = 8; break;
      case TFGPU_T_UINT8: c.kind = JK_UINT; c.width = 1; break;
      case TFGPU_T_UINT16: c.kind = JK_UINT; c.width = 2; break;
      case TFGPU_T_UINT32: c.kind = JK_UINT; c.width = 4; break;
      case TFGPU_T_UINT64: c.kind = JK_UINT; c.width = 8; break;
      case TFGPU_T_FLOAT64: c.kind = JK_F64; c.width = 8; break;
      case TFGPU_T_BOOLEAN: c.kind = JK_BOOL; c.width = 1; break;
      case TFGPU_T_UTF8: case TFGPU_T_BYTES: c.kind = JK_TEXT; break;
      case TFGPU_T_ANY: c.kind = JK_ANY; break;
      case TFGPU_T_DATETIME: c.kind = JK_DATETIME; c.width = 8; break;
      default:  // float / date / timestamp / interval columns keep mixed Go types in the reference (ParseVal :1019-1122)
        return tf::fail(TFGPU_ERR_UNSUPPORTED, "json: column " + name + ": DataType is not device-resident for the generic parser");
    }
  }
  // Unmarshal types a non-string value by colTypeMap[key] — keyed by ColPath (ColumnName with IgnoreColumnPaths)
  // over the FINAL schema, the last column winning (:1226-1233) — and every column reading that key then runs
  // ParseVal on the typed value.  Device cells fuse the two steps, which is only the same thing when the key
  // is typed with the column's own DataType.
  {
    std::vector<std::pair<std::string, int>> col_type;
    auto put = [&](const std::string &k, int t) { for (auto &e : col_type) if (e.first == k) { e.second = t; return; } col_type.push_back({k, t}); };
    for (int i = 0; i < nraw; i++) {
      const tfgpu_colschema &sc = fields->cols[i];
      const std::string name = sc.name ? sc.name : "";
      put(opts->ignore_column_paths ? name : ((sc.path && sc.path[0]) ? std::string(sc.path) : name), sc.dtype);
    }
    for (auto &a : aux) put(a.name, a.dtype);
    for (int i = 0; i < nraw; i++) {
      if (cols[(size_t)i].npath_n) continue;  // reads a Go string whatever colTypeMap says of its first name (anything else: host)
      int t = TFGPU_T_INVALID;
      for (auto &e : col_type) if (e.first == lookup[(size_t)i]) t = e.second;
      if (t != fields->cols[i].dtype)
        return tf::fail(TFGPU_ERR_UNSUPPORTED, std::string("json: column ") + (fields->cols[i].name ? fields->cols[i].name : "") +
                        ": its key is typed through another column's DataType (shared ColPath, or IgnoreColumnPaths with Path != ColumnName)");
    }
  }
  // key table: ColPath → first column reading it; names / aux names for `_rest`
  std::vector<std::pair<std::string, int32_t>> entries;
  for (int i = nraw - 1; i >= 0; i--) {
    int32_t head = -1;
    for (auto &e : entries) if (e.first == lookup[(size_t)i]) head = e.second;
    cols[(size_t)i].next = head;
    bool found = false;
    for (auto &e : entries) if (e.first == lookup[(size_t)i]) { e.second = i; found = true; }
    if (!found) entries.push_back({lookup[(size_t)i], i});
  }
  auto add_entry = [&](const std::string &k, int32_t v) { for (auto &e : entries) if (e.first == k) return; entries.push_back({k, v}); };
  if (opts->add_rest) {
    for (int i = 0; i < nraw; i++) {  // p.known: ColumnName and ColPath of every raw field
      const tfgpu_colschema &sc = fields->cols[i];
      add_entry(sc.name ? sc.name : "", JS_KNOWN);
      if (!opts->ignore_column_paths) add_entry((sc.path && sc.path[0]) ? sc.path : (sc.name ? sc.name : ""), JS_KNOWN);
    }
    for (const char *a : {"_rest", "_timestamp", "_partition", "_offset", "_idx"}) add_entry(a, JS_KNOWN);
  }
  uint32_t nslots = 16;
  while (nslots < entries.size() * 2 + 2) nslots <<= 1;
  std::vector<JSlot> slots(nslots, JSlot{0, 0, 0, ~0u, 0});
  std::string names;
  for (auto &e : entries) {
    const uint32_t h = key_hash_host(e.first);
    uint32_t s = h & (nslots - 1);
    while (slots[s].slen != ~0u) s = (s + 1) & (nslots - 1);
    uint32_t is_aux = 0;
    if (opts->add_rest) for (const char *a : {"_rest", "_timestamp", "_partition", "_offset", "_idx"}) if (e.first == a) is_aux = 1;
    slots[s] = JSlot{h, e.second, (uint32_t)names.size(), (uint32_t)e.first.size(), is_aux};
    names += e.first;
    names.append((8 - names.size() % 8) % 8, '\0');  // whole 8-byte words: the device compares a word at a time
  }

  { size_t k = 0; for (auto &nb : nested_blobs) { cols[(size_t)nb.first].npath_off = (uint32_t)names.size(); names += nb.second; cols[(size_t)nb.first].scr_base = (uint32_t)(k++ * ((len + 64 + 63) & ~63ull)); } }
  names.append((8 - names.size() % 8) % 8 + 8, '\0');
  const bool any_nested = !nested_blobs.empty();

  // ---- input in HBM ----
  Buf staged;
  const uint8_t *data;
  if (mem == TFGPU_MEM_HOST) {
    staged = dalloc(len + 64);
    h2d(staged->p, bytes, len);
    TF_HIP(hipMemsetAsync((char *)staged->p + len, 0, 64, st));
    data = ptr<uint8_t>(staged);
  } else {
    data = (const uint8_t *)bytes;
    if (reinterpret_cast<uintptr_t>(data) & 15) return tf::fail(TFGPU_ERR_INVALID, "json: device buffer must be 16-byte aligned");
  }
  // ---- messages ----
  const uint32_t nmsg = msgs ? (uint32_t)msgs->nmsg : 1u;
  std::vector<uint32_t> ms((size_t)nmsg + 1);
  if (msgs) {
    if (msgs->nmsg < 1 || !msgs->start) return tf::fail(TFGPU_ERR_INVALID, "json: empty message batch");
    for (uint32_t m = 0; m <= nmsg; m++) {
      if (msgs->start[m] > len || (m && msgs->start[m] < msgs->start[m - 1])) return tf::fail(TFGPU_ERR_INVALID, "json: message offsets must be ascending and inside the buffer");
      ms[m] = (uint32_t)msgs->start[m];
    }
    if (ms[0] != 0 || ms[nmsg] != len) return tf::fail(TFGPU_ERR_INVALID, "json: messages must cover the buffer");
  } else { ms[0] = 0; ms[1] = (uint32_t)len; }
  Buf bms = dalloc(ms.size() * 4), bmoff, bmwt;
  h2d(bms->p, ms.data(), ms.size() * 4);
  if (msgs && msgs->offset) { bmoff = dalloc((size_t)nmsg * 8); h2d(bmoff->p, msgs->offset, (size_t)nmsg * 8); }
  if (msgs && msgs->write_time_ns) { bmwt = dalloc((size_t)nmsg * 8); h2d(bmwt->p, msgs->write_time_ns, (size_t)nmsg * 8); }

  // ---- 1-3: lines ----
  Buf nls;
  const uint32_t nnl = newline_starts(data, len, &nls);
  const int64_t nseg = (int64_t)nnl + nmsg;
  Buf bounds = dalloc((size_t)(nseg + 1) * 4), rank_ms = dalloc((size_t)nmsg * 4);
  Buf seg_start = dalloc((size_t)nseg * 4), seg_len = dalloc((size_t)nseg * 4), seg_ord = dalloc((size_t)(nseg + 1) * 4);
  {
    KernelTimer t("json_line_bounds");
    json_merge_bounds<<<jblocks(nseg, 256), 256, 0, st>>>(ptr<uint32_t>(nls) + 1, nnl, ptr<uint32_t>(bms), nmsg, (uint32_t)len, ptr<uint32_t>(bounds), ptr<uint32_t>(rank_ms));
    json_segments<<<jblocks(nseg, 256), 256, 0, st>>>(data, ptr<uint32_t>(bounds), nseg, ptr<uint32_t>(seg_start), ptr<uint32_t>(seg_len), ptr<uint32_t>(seg_ord));
  }
  exclusive_scan_u32(ptr<uint32_t>(seg_ord), ptr<uint32_t>(seg_ord), nseg, true);
  const uint32_t *hn = d2h_u32(ptr<uint32_t>(seg_ord) + nseg);
  tf::sync();  // also fences the message tables' pageable sources
  const int64_t nlines = *hn;
  const int64_t nalloc = std::max<int64_t>(nlines, 1);

  // ---- outputs ----
  auto db = std::make_unique<tfgpu_dbatch>();
  db->nrows = nlines;
  db->ns = "";
  {  // tableName(): GenericParser.name with '/' and '@' replaced (:569-574, :1236)
    std::string t = opts->topic ? opts->topic : "";
    for (char &ch : t) if (ch == '/' || ch == '@') ch = '_';
    db->table = t;
  }
  int nstr = 0;
  for (auto &c : cols) if (c.kind == JK_TEXT || c.kind == JK_ANY) nstr++;
  const int rest_seg = opts->add_rest ? nstr++ : -1;  // `_rest` is one more text column (its cells come from the whole line)
  int rest_col = -1;                                   // its index in db->cols
  const int64_t seg_stride = ((nlines + 1 + 3) / 4) * 4;
  Buf lens_all = dalloc_zero((size_t)std::max(nstr, 1) * (size_t)seg_stride * 4 + 16);
  Buf fstart_all = dalloc((size_t)std::max(nstr, 1) * (size_t)nalloc * 8);  // fstart | meta
  Buf cellst_all = dalloc_zero((size_t)std::max(nraw, 1) * (size_t)nalloc);
  Buf valid_all = dalloc((size_t)std::max(nraw, 1) * (size_t)((nalloc + 7) / 8 + 1));
  std::vector<int> str_col_index;
  std::vector<uint8_t *> valid_ptrs;
  int si = 0;
  // the fixed-width columns' values (and nanoseconds) as views of ONE zeroed block: a memset launch per column was ~90 launches per batch
  size_t fixed_bytes = 0;
  auto fixed_room = [&](size_t bytes) { const size_t at = fixed_bytes; fixed_bytes += (bytes + 255) & ~(size_t)255; return at; };
  std::vector<size_t> val_at((size_t)nraw, 0), nan_at((size_t)nraw, 0);
  for (int i = 0; i < nraw; i++) {
    const JCol &c = cols[(size_t)i];
    if (c.kind == JK_TEXT || c.kind == JK_ANY) continue;
    val_at[(size_t)i] = fixed_room((size_t)nalloc * (size_t)c.width);
    if (!(c.kind == JK_INT || c.kind == JK_UINT || c.kind == JK_F64 || c.kind == JK_BOOL)) nan_at[(size_t)i] = fixed_room((size_t)nalloc * 4);
  }
  Buf fixed_all = dalloc_zero(std::max<size_t>(fixed_bytes, 256));
  for (int i = 0; i < nraw; i++) {
    JCol &c = cols[(size_t)i];
    const tfgpu_colschema &sc = fields->cols[i];
    DColumn d;
    d.name = sc.name ? sc.name : ""; d.dtype = sc.dtype;
    c.cellst = ptr<uint8_t>(cellst_all) + (size_t)i * (size_t)nalloc;
    d.validity = subbuf(valid_all, (size_t)i * (size_t)((nalloc + 7) / 8 + 1), (size_t)((nalloc + 7) / 8));
    valid_ptrs.push_back(ptr<uint8_t>(d.validity));
    if (c.kind == JK_TEXT || c.kind == JK_ANY) {
      d.repr = c.kind == JK_ANY ? TFGPU_R_JSON : TFGPU_R_STRING;
      c.lens = ptr<uint32_t>(lens_all) + (int64_t)si * seg_stride;
      c.fstart = ptr<uint32_t>(fstart_all) + (int64_t)si * 2 * nalloc;
      c.meta = c.fstart + nalloc;
      str_col_index.push_back(i);
      si++;
    } else {
      switch (c.kind) {
        case JK_INT: d.repr = c.width == 1 ? TFGPU_R_INT8 : c.width == 2 ? TFGPU_R_INT16 : c.width == 4 ? TFGPU_R_INT32 : TFGPU_R_INT64; break;
        case JK_UINT: d.repr = c.width == 1 ? TFGPU_R_UINT8 : c.width == 2 ? TFGPU_R_UINT16 : c.width == 4 ? TFGPU_R_UINT32 : TFGPU_R_UINT64; break;
        case JK_F64: d.repr = TFGPU_R_FLOAT64; break;
        case JK_BOOL: d.repr = TFGPU_R_BOOL; break;
        default: d.repr = TFGPU_R_TIME;
      }
      d.values = subbuf(fixed_all, val_at[(size_t)i], (size_t)nalloc * (size_t)c.width);
      c.values = d.values->p;
      if (d.repr == TFGPU_R_TIME) { d.nanos = subbuf(fixed_all, nan_at[(size_t)i], (size_t)nalloc * 4); c.nanos = ptr<int32_t>(d.nanos); }
    }
    db->cols.push_back(std::move(d));
  }
  Buf bcols = upload_small(cols.data(), cols.size() * sizeof(JCol)), bslots = upload_small(slots.data(), slots.size() * sizeof(JSlot));
  Buf bnames = upload_small(names.data(), names.size());
  Buf linest = dalloc((size_t)nalloc + 16), line_pos = dalloc((size_t)nalloc * 4);

  JParams pp;
  std::memset(&pp, 0, sizeof pp);
  pp.data = data; pp.seg_start = ptr<uint32_t>(seg_start); pp.seg_len = ptr<uint32_t>(seg_len); pp.seg_ord = ptr<uint32_t>(seg_ord); pp.nseg = nseg;
  pp.cols = ptr<JCol>(bcols); pp.ncols = nraw; pp.slots = ptr<JSlot>(bslots); pp.slot_mask = nslots - 1; pp.names = ptr<uint8_t>(bnames);
  pp.pow10 = pow10_table(); pp.pow128 = reinterpret_cast<const uint64_t *>(pp.pow10 + 632); pp.linest = ptr<uint8_t>(linest); pp.line_pos = ptr<uint32_t>(line_pos);
  pp.add_rest = opts->add_rest; pp.use_numbers = opts->use_numbers_in_any;
  pp.format = opts->format; pp.tskv_unescape = tskv && opts->unescape_string_values;
  Buf scratch;
  if (any_nested) {  // one copy-sized region per nested column (cells are 32-bit offsets into it)
    const uint64_t per = (len + 64 + 63) & ~63ull;
    if (per * nested_blobs.size() >= 0xFFFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "json: nested ColSchema.Path columns x batch bytes exceed 4 GiB of scratch; split the batch");
    scratch = dalloc_zero(per * nested_blobs.size());
    pp.scratch = ptr<uint8_t>(scratch);
  }
  if (rest_seg >= 0) {
    pp.rest_lens = ptr<uint32_t>(lens_all) + (int64_t)rest_seg * seg_stride;
    pp.rest_fstart = ptr<uint32_t>(fstart_all) + (int64_t)rest_seg * 2 * nalloc;
    pp.rest_meta = pp.rest_fstart + nalloc;
  }
  // The per-line path is the default: 64 lines of one shape run in lockstep, so a wave pays each instruction once per
  // 64 lines (≈ 23 k wave-instructions per 64 lines of 2.3 KB).  The wave path (TFGPU_JSON_WAVEPATH=1) reads coalesced
  // and does stage 1 on the scalar unit, but its lanes are MEMBERS of one line — different column kinds side by side —
  // and the divergent typed parse costs ≈ 20 k wave-instructions per LINE: measured 2× slower at 2^18 hits rows
  // (13.6 vs 6.7 ms, profiles/r01y_json_paths.txt).  Kept as the parity cross-check of the grammar and as the
  // starting point for a kind-sorted member phase.
  static const bool wavepath = [] { const char *e = std::getenv("TFGPU_JSON_WAVEPATH"); return e && e[0] == '1'; }();
  // The tile path (default for Format json; TFGPU_JSON_TILES=0 for A/B runs): lines staged in LDS, cells by value kind.
  static const bool tilepath = [] { const char *e = std::getenv("TFGPU_JSON_TILES"); return !(e && e[0] == '0'); }();
  if (nlines && tilepath && !tskv && !wavepath && !any_nested) {  // (lookupComplex lives in the per-line parser)
    const uint64_t avg = std::max<uint64_t>(len / (uint64_t)std::max<int64_t>(nseg, 1), 1);
    const int32_t per_tile = (int32_t)std::min<uint64_t>(std::max<uint64_t>((uint64_t)(JT_BYTES - 16) * 8 / (avg * 9), 1), (uint64_t)JT_LINES);
    Buf slow = dalloc((size_t)(nseg + 1) * 4), heavy = dalloc((size_t)(nseg + 1) * 4);
    TF_HIP(hipMemsetAsync(slow->p, 0, 4, st));
    TF_HIP(hipMemsetAsync(heavy->p, 0, 4, st));
    // json_parse_quick (tf_jsonquick.inc) when no column needs what it leaves out (float tokens) and a line's members fit its map;
    // json_parse_tiles otherwise (and with TFGPU_JSON_QUICK=0, for A/B runs and as the cross-check of the two)
    static const bool quick_off = [] { const char *e = std::getenv("TFGPU_JSON_QUICK"); return e && e[0] == '0'; }();
    bool quick = !quick_off && nraw <= JQ_MEM;
    for (auto &c : cols) if (c.kind == JK_F64) quick = false;
    static const int jt_ablate = [] { const char *e = std::getenv("TFGPU_JT_ABLATE"); return e ? std::atoi(e) : 0; }();
    int32_t used_per_tile = per_tile;
    if (quick) {
      const int32_t qper = (int32_t)std::min<uint64_t>(std::max<uint64_t>((uint64_t)(JQ_BYTES - 16) * 8 / (avg * 9), 1), (uint64_t)JQ_LINES);
      used_per_tile = qper;
      Buf bmap = dalloc(sizeof(JqMap));
      { KernelTimer t("json_quick_map"); json_quick_map<<<1, JQ_THREADS, 0, st>>>(pp, ptr<JqMap>(bmap)); }
      { KernelTimer t("json_parse_quick"); json_parse_quick<<<jblocks(nseg, qper), JQ_THREADS, 0, st>>>(pp, ptr<JqMap>(bmap), qper, nalloc, ptr<uint32_t>(slow), ptr<uint32_t>(slow) + 1, jt_ablate); }
    } else {
    // persistent workgroups (two per CU fit its LDS): each walks every (2 * CUs)-th tile and keeps its member map
    const unsigned ntile = jblocks(nseg, per_tile), nblk = (unsigned)std::min<int64_t>((int64_t)ntile, (int64_t)cx.num_cus * 2);
    { KernelTimer t("json_parse_tiles"); json_parse_tiles<<<nblk, JT_THREADS, 0, st>>>(pp, per_tile, ptr<uint32_t>(slow), ptr<uint32_t>(slow) + 1); }
    }
    { KernelTimer t("json_parse_listed"); json_parse_listed_lean<<<jblocks(nseg, 256), 256, 0, st>>>(pp, ptr<uint32_t>(slow), ptr<uint32_t>(slow) + 1, ptr<uint32_t>(heavy), ptr<uint32_t>(heavy) + 1);
      json_parse_listed<<<jblocks(nseg, 256), 256, 0, st>>>(pp, ptr<uint32_t>(heavy), ptr<uint32_t>(heavy) + 1); }
    static const bool dbg = [] { const char *e = std::getenv("TFGPU_JSON_TILE_DEBUG"); return e && e[0] == '1'; }();
    if (dbg) {  // how many lines the tile path handed over (diagnostics only: costs a synchronisation)
      const uint32_t *a = d2h_u32(slow->p), *b = d2h_u32(heavy->p);
      tf::sync();
      std::fprintf(stderr, "[tfgpu] json tiles: %lld lines, %d per tile, %u to the per-line parser, %u of them to the map emitter\n", (long long)nseg, (int)used_per_tile, *a, *b);
    }
  }
  else if (nlines && (!wavepath || tskv || any_nested)) {
    Buf slow = dalloc((size_t)(nseg + 1) * 4);  // lines that need the map emitter: re-parsed by the kernel that carries it
    TF_HIP(hipMemsetAsync(slow->p, 0, 4, st));
    { KernelTimer t("json_parse_lines"); json_parse_lines<<<jblocks(nseg, 256), 256, 0, st>>>(pp, ptr<uint32_t>(slow), ptr<uint32_t>(slow) + 1); }
    if (!tskv || any_nested) { KernelTimer t("json_parse_listed"); json_parse_listed<<<jblocks(nseg, 256), 256, 0, st>>>(pp, ptr<uint32_t>(slow), ptr<uint32_t>(slow) + 1); }
  }
  else if (nlines) {
    Buf slow = dalloc((size_t)(nseg + 1) * 4);
    TF_HIP(hipMemsetAsync(slow->p, 0, 4, st));
    const unsigned nblk = (unsigned)std::min<int64_t>((nseg + JF_WAVES - 1) / JF_WAVES, (int64_t)cx.num_cus * 8);
    { KernelTimer t("json_parse_waves"); json_parse_waves<<<nblk, 256, 0, st>>>(pp, ptr<uint32_t>(slow), ptr<uint32_t>(slow) + 1); }
    { KernelTimer t("json_parse_listed"); json_parse_listed<<<jblocks(nseg, 256), 256, 0, st>>>(pp, ptr<uint32_t>(slow), ptr<uint32_t>(slow) + 1); }
  }

  // ---- aux columns + row rules ----
  Buf code = dalloc((size_t)nalloc + 16), ecol = dalloc((size_t)nalloc * 4), keep = dalloc((size_t)(nalloc + 1) * 4), nerr = dalloc_zero(4);
  Buf has_nil = dalloc_zero((size_t)std::max(nraw, 1) * 4);
  db->part_id = dalloc((size_t)nalloc * 4);
  JFinish fp;
  std::memset(&fp, 0, sizeof fp);
  fp.cols = ptr<JCol>(bcols); fp.ncols = nraw; fp.nlines = nlines; fp.linest = ptr<uint8_t>(linest); fp.null_keys_allowed = opts->null_keys_allowed;
  fp.code = ptr<uint8_t>(code); fp.ecol = ptr<int32_t>(ecol); fp.keep = ptr<uint32_t>(keep); fp.nerr = ptr<uint32_t>(nerr);
  fp.line_pos = ptr<uint32_t>(line_pos); fp.ms = ptr<uint32_t>(bms); fp.nmsg = nmsg; fp.rank_ms = ptr<uint32_t>(rank_ms); fp.seg_ord = ptr<uint32_t>(seg_ord);
  fp.msg_offset = ptr<uint64_t>(bmoff); fp.msg_wtime = ptr<int64_t>(bmwt); fp.part_id = ptr<uint32_t>(db->part_id);
  const std::string part = opts->partition ? opts->partition : "";
  for (auto &a : aux) {
    DColumn d;
    d.name = a.name; d.dtype = a.dtype;
    if (a.dtype == TFGPU_T_ANY) {  // `_rest`: a text cell per line, written by rest_emit
      d.repr = TFGPU_R_JSON;
      rest_col = (int)db->cols.size();
    } else if (a.dtype == TFGPU_T_BYTES) {  // `_partition` = PartID
      const std::string text = part;
      d.repr = TFGPU_R_STRING;
      if ((uint64_t)text.size() * (uint64_t)nalloc >= 0xFFFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "json: `_partition` column exceeds 4 GiB");
      d.offsets = dalloc((size_t)(nalloc + 1) * 4); d.data = dalloc(text.size() * (size_t)nalloc + 8); d.data_len = text.size() * (uint64_t)nlines;
      Buf bt = upload_small(text.data(), text.size());
      json_const_text<<<jblocks(nlines + 1, 256), 256, 0, st>>>(ptr<uint8_t>(bt), (uint32_t)text.size(), nlines, ptr<uint32_t>(d.offsets), ptr<uint8_t>(d.data));
    } else if (a.dtype == TFGPU_T_TIMESTAMP) {
      d.repr = TFGPU_R_TIME; d.values = dalloc((size_t)nalloc * 8); d.nanos = dalloc((size_t)nalloc * 4);
      fp.ts_sec = ptr<int64_t>(d.values); fp.ts_nanos = ptr<int32_t>(d.nanos);
    } else if (a.dtype == TFGPU_T_UINT64) { d.repr = TFGPU_R_UINT64; d.values = dalloc((size_t)nalloc * 8); fp.off_out = ptr<uint64_t>(d.values); }
    else { d.repr = TFGPU_R_UINT32; d.values = dalloc((size_t)nalloc * 4); fp.idx_out = ptr<uint32_t>(d.values); }
    db->cols.push_back(std::move(d));
  }
  if (nlines) {
    KernelTimer t("json_finish");
    json_finish<<<jblocks(nlines, 256), 256, 0, st>>>(fp);
    Buf bvp = upload_small(valid_ptrs.data(), valid_ptrs.size() * sizeof(uint8_t *));
    if (nraw) json_validity<<<dim3(jblocks((nlines + 7) / 8, 256), (unsigned)nraw), 256, 0, st>>>(ptr<JCol>(bcols), nlines, (uint8_t *const *)bvp->p, ptr<uint32_t>(keep), ptr<uint32_t>(has_nil));
  }
  if (nstr) exclusive_scan_u32_segments(ptr<uint32_t>(lens_all), nlines, nstr, seg_stride);
  Buf spec = dalloc_zero((size_t)std::max(nstr, 1) * 4);
  if (nlines && nstr) {
    std::vector<const uint32_t *> metas((size_t)nstr);
    for (int sg = 0; sg < nstr; sg++) metas[(size_t)sg] = sg == rest_seg ? pp.rest_meta : cols[(size_t)str_col_index[(size_t)sg]].meta;
    Buf bm = upload_small(metas.data(), metas.size() * sizeof(uint32_t *));
    json_mark_special<<<dim3(jblocks(nlines, 256 * 16), (unsigned)nstr), 256, 0, st>>>(ptr<uint32_t>(lens_all), seg_stride, (const uint32_t *const *)bm->p, nlines, ptr<uint32_t>(spec));
  }
  Buf summary = dalloc((size_t)(2 * nstr + nraw + 1) * 4);
  json_collect<<<1, 64, 0, st>>>(ptr<uint32_t>(nerr), ptr<uint32_t>(lens_all), seg_stride, nlines, nstr, ptr<uint32_t>(has_nil), nraw, ptr<uint32_t>(spec), ptr<uint32_t>(summary));
  const uint32_t *hsum = d2h_u32(summary->p, (size_t)2 * nstr + nraw + 1);
  tf::sync();
  const uint32_t hnerr = hsum[0];
  for (int i = 0; i < nraw; i++) if (!hsum[1 + nstr + i]) db->cols[(size_t)i].validity = nullptr;  // no nil among the kept lines

  // ---- text payloads ----
  std::vector<JCopyCol> cc;
  for (int s = 0; s < nstr; s++) {
    const bool is_rest = s == rest_seg;
    DColumn &d = db->cols[is_rest ? (size_t)rest_col : (size_t)str_col_index[(size_t)s]];
    d.data_len = hsum[1 + s];
    d.data = dalloc(d.data_len + 8);
    d.offsets = subbuf(lens_all, (size_t)s * (size_t)seg_stride * 4, (size_t)(nlines + 1) * 4);
    if (is_rest) cc.push_back(JCopyCol{ptr<uint32_t>(d.offsets), pp.rest_fstart, pp.rest_meta, ptr<uint8_t>(d.data)});
    else { const JCol &c = cols[(size_t)str_col_index[(size_t)s]]; cc.push_back(JCopyCol{ptr<uint32_t>(d.offsets), c.fstart, c.meta, ptr<uint8_t>(d.data)}); }
  }
  if (nlines && nstr) {
    Buf bcc = upload_small(cc.data(), cc.size() * sizeof(JCopyCol));
    KernelTimer t("json_copy_cells");
    static const bool words = [] { const char *e = std::getenv("TFGPU_JSON_COPY_WORDS"); return !(e && e[0] == '0'); }();  // 0: A/B runs
    if (words) json_copy_words<<<dim3(jblocks(nlines, 256), (unsigned)nstr), 256, 0, st>>>(data, ptr<JCopyCol>(bcc), nlines);
    // the walkers only where json_mark_special saw their cells (every column without TFGPU_JSON_COPY_WORDS: the plain cells are theirs then)
    std::vector<JCopyCol> light, heavy;
    for (int sg = 0; sg < nstr; sg++) {
      const uint32_t f = hsum[1 + nstr + nraw + sg];
      if ((f & 1u) || !words) light.push_back(cc[(size_t)sg]);
      if (f & 2u) heavy.push_back(cc[(size_t)sg]);
    }
    if (!light.empty()) { Buf bl = upload_small(light.data(), light.size() * sizeof(JCopyCol)); json_copy_cells<false><<<dim3(jblocks(nlines, 256), (unsigned)light.size()), 256, 0, st>>>(pp, ptr<JCopyCol>(bl), nlines, words ? 1 : 0); }
    if (!heavy.empty()) { Buf bh = upload_small(heavy.data(), heavy.size() * sizeof(JCopyCol)); json_copy_cells<true><<<dim3(jblocks(nlines, 256), (unsigned)heavy.size()), 256, 0, st>>>(pp, ptr<JCopyCol>(bh), nlines, words ? 1 : 0); }
  }

  // ---- dropped lines: `_unparsed` rows / host fallback are reported, skipped lines vanish ----
  std::vector<uint8_t> hcode((size_t)nlines);
  std::vector<int32_t> hecol((size_t)nlines);
  std::vector<uint32_t> hpart((size_t)nlines);
  int64_t ne = 0;
  if (hnerr) {
    d2h(hcode.data(), code->p, (size_t)nlines); d2h(hecol.data(), ecol->p, (size_t)nlines * 4); d2h(hpart.data(), db->part_id->p, (size_t)nlines * 4);
    tf::sync();
    for (int64_t r = 0; r < nlines; r++) {
      const int c = hcode[(size_t)r];
      if (!c || c == JCODE_SKIPPED) continue;
      if (errs && ne < errs_cap) errs[ne] = tfgpu_row_error{r, c, (int32_t)hpart[(size_t)r], hecol[(size_t)r]};
      ne++;
    }
  }
  std::unique_ptr<tfgpu_dbatch> result = compact_rows(*db, keep);  // identity (shared buffers) when every line is a row
  if (nerrs) *nerrs = ne;
  *out = result.release();
  return TFGPU_OK;
  TF_API_END
}
