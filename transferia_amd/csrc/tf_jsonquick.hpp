// tf_jsonquick.hpp — the front half of the "quick" JSON tile parsers (json_parse_quick in tf_jsonquick.inc, sr_parse_quick
// in tf_srquick.inc): one tile of consecutive lines / payloads per workgroup, staged in LDS, four byte classes (quote,
// backslash, comma, closing brace), unescaped quotes by the backslash-run carry trick, the in-string state by a prefix xor
// carried across lanes and waves, the separators ',' '}' outside strings indexed.  What a member's text must spell and what a
// (member, line) cell means is the caller's business (its member map rides in the same LDS block).
#pragma once
#include "tf_jsontile.hpp"

namespace tf {

static constexpr int JQ_THREADS = 448;
static constexpr int JQ_WAVES = JQ_THREADS / 64;
static constexpr int JQ_CPT = 4;                             // 16-byte chunks per thread (blocked): 64 bytes, one 64-bit mask per class
static constexpr int JQ_BYTES = JQ_THREADS * JQ_CPT * 16;    // 28 KiB of text per tile
static constexpr int JQ_LINES = 32;                          // lines per tile at most
static constexpr int JQ_SCAP = 2048;                         // separators (',' and '}' outside strings) indexed per tile
static constexpr int JQ_MEM = 128;                           // members per line
static constexpr int JQ_KREF = 3072;                         // bytes of member text the map keeps
static constexpr uint32_t JQ_NOLINE = 2;

template <class MAP, bool HIGH = false> struct JqLds {
  static constexpr bool kHigh = HIGH;
  __attribute__((aligned(16))) uint8_t sbuf[16 + JQ_BYTES + 48];
  __attribute__((aligned(16))) MAP map;
  uint16_t spos[JQ_SCAP + 2];
  uint32_t qmask[JQ_BYTES / 32 + 1], bmask[JQ_BYTES / 32 + 1];  // unescaped quotes; backslashes
  uint16_t qpre[JQ_BYTES / 32 + 2], bpre[JQ_BYTES / 32 + 2];    // … in front of each 32-byte word
  uint64_t hblk[JQ_WAVES];                                       // HIGH: bit t of word w = thread 64 w + t's 64 bytes hold a byte >= 0x80
  uint32_t wpar[JQ_WAVES], wcnt[JQ_WAVES], wqc[JQ_WAVES], wbc[JQ_WAVES];
  uint16_t lstart[JQ_LINES], lend[JQ_LINES], lbase[JQ_LINES];    // per line: bytes [lstart, lend), ordinal of its first separator
  int32_t lrow[JQ_LINES];
  uint8_t lslow[JQ_LINES];                                       // 0 taken, 1 handed over, 2 no line here
  uint32_t misc[4];                                              // 0: tile cannot be taken, 1: separators
};
template <class LDS> __device__ __forceinline__ uint32_t jq_quotes_in(const LDS &L, uint32_t a, uint32_t b) {
  const uint32_t ca = (uint32_t)L.qpre[a >> 5] + (uint32_t)__popc(L.qmask[a >> 5] & ((1u << (a & 31)) - 1u));
  const uint32_t cb = (uint32_t)L.qpre[b >> 5] + (uint32_t)__popc(L.qmask[b >> 5] & ((1u << (b & 31)) - 1u));
  return cb - ca;
}
template <class LDS> __device__ __forceinline__ uint32_t jq_backslashes_in(const LDS &L, uint32_t a, uint32_t b) {
  const uint32_t ca = (uint32_t)L.bpre[a >> 5] + (uint32_t)__popc(L.bmask[a >> 5] & ((1u << (a & 31)) - 1u));
  const uint32_t cb = (uint32_t)L.bpre[b >> 5] + (uint32_t)__popc(L.bmask[b >> 5] & ((1u << (b & 31)) - 1u));
  return cb - ca;
}
// does any 64-byte block that [a, b) touches hold a byte >= 0x80?  (a <= b; block granularity: a neighbour's bytes may answer yes —
// the caller only loses a shortcut then)
template <class LDS> __device__ __forceinline__ bool jq_high_near(const LDS &L, uint32_t a, uint32_t b) {
  if (b <= a) return false;
  const uint32_t t0 = a >> 6, t1 = (b - 1) >> 6;
  for (uint32_t w = t0 >> 6; w <= t1 >> 6; w++) {
    uint64_t m = L.hblk[w];
    if (w == t0 >> 6) m &= ~0ull << (t0 & 63);
    if (w == t1 >> 6) m &= ~0ull >> (63 - (t1 & 63));
    if (m) return true;
  }
  return false;
}
// first separator ordinal whose position is >= pos
__device__ __forceinline__ uint32_t jq_lower(const uint16_t *spos, uint32_t nsep, uint32_t pos) {
  uint32_t lo = 0, hi = nsep;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)spos[mid] < pos) lo = mid + 1; else hi = mid; }
  return lo;
}

// Stage the bytes [g0, last), classify, index the separators ',' '}' outside strings, frame the lines.  All threads call it
// (it holds the barriers); false (uniform): the tile cannot be taken.
// (STAGE = false: the caller has put the lines' bytes into L.sbuf + 16 itself — payloads gathered from apart — and left
// everything that belongs to no line zero.)
template <class LDS, bool STAGE = true> __device__ __forceinline__ bool jq_front(LDS &L, const uint8_t *data, const uint32_t first, const uint32_t last, const uint32_t g0, const int nl, const int ablate) {
  uint8_t *const sb = L.sbuf + 16;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if constexpr (STAGE) {
#pragma unroll
    for (int it = 0; it < JQ_CPT; it++) {
      const int chunk = it * JQ_THREADS + tid;
      const uint32_t gp = g0 + (uint32_t)chunk * 16;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gp < last) v = *reinterpret_cast<const uint4 *>(data + gp);  // the buffer is padded past its payload
      *reinterpret_cast<uint4 *>(sb + chunk * 16) = v;
    }
  }
  if (tid < 4) { reinterpret_cast<uint32_t *>(L.sbuf)[tid] = 0; L.misc[tid] = 0; }
  if (tid >= 64 && tid < 76) reinterpret_cast<uint32_t *>(L.sbuf + 16 + JQ_BYTES)[tid - 64] = 0;
  __syncthreads();
  if (ablate == 1) return false;  // (TFGPU_JT_ABLATE, profiling only: leave after phase n; the lines then go to the per-line parser)

  // ---- pass 1: byte classes of this thread's 64 bytes ----
  const uint32_t base_chunk = (uint32_t)tid * JQ_CPT, tpos = base_chunk * 16;
  uint64_t Q = 0, B = 0, S = 0, H = 0, C = 0;  // (LDS::kHigh only) H: nonzero = this thread's bytes hold one >= 0x80; C: the bytes < 0x20, one bit each
#pragma unroll
  for (int q = 0; q < JQ_CPT; q++) {
    const uint4 v = *reinterpret_cast<const uint4 *>(sb + (base_chunk + q) * 16);
    const Chunk16 ck = chunk16(v);
    if constexpr (LDS::kHigh) {
      H |= (uint64_t)((v.x | v.y | v.z | v.w) & 0x80808080u);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      uint32_t cm = 0;  // bytes below 0x20, one bit each (exact: what lies between two payloads — their 5-byte prefixes — is masked out below)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t t = w[k] & 0xE0E0E0E0u;                                      // zero where the byte is below 0x20
        const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;  // exact zero-byte flags
        cm |= __builtin_amdgcn_udot4(z >> 7, 0x08040201u, 0u, false) << (4 * k);
      }
      C |= (uint64_t)cm << (16 * q);
    }
    Q |= (uint64_t)class16c(ck, 0x22222222u) << (16 * q);
    B |= (uint64_t)class16c(ck, 0x5C5C5C5Cu) << (16 * q);
    S |= (uint64_t)(class16c(ck, 0x2C2C2C2Cu) | class16c(ck, 0x7D7D7D7Du)) << (16 * q);
  }
  {  // bytes that belong to no line of this tile (what lies between two messages, the head of the next line) take no part
    uint64_t cover = 0;
    for (int j = 0; j < nl; j++) {
      if (L.lslow[j] == JQ_NOLINE) continue;
      const uint32_t a = L.lstart[j], b = L.lend[j];
      if (b <= tpos || a >= tpos + 64u) continue;
      const uint32_t lo = a > tpos ? a - tpos : 0u, hi = min(b - tpos, 64u);
      cover |= (hi >= 64 ? ~0ull : (1ull << hi) - 1) & ~((1ull << lo) - 1);
    }
    Q &= cover; B &= cover; S &= cover;
    if constexpr (LDS::kHigh) {
      C &= cover;
      if (C) {  // a byte below 0x20 (a syntax error inside a string, a blank outside): the lines that hold one are not decided here
        for (int j = 0; j < nl; j++) {
          if (L.lslow[j] == JQ_NOLINE) continue;
          const uint32_t a = L.lstart[j], b = L.lend[j];
          if (b <= tpos || a >= tpos + 64u) continue;
          const uint32_t lo = a > tpos ? a - tpos : 0u, hi = min(b - tpos, 64u);
          if (C & (hi >= 64 ? ~0ull : (1ull << hi) - 1) & ~((1ull << lo) - 1)) L.lslow[j] = 1;
        }
      }
    }
  }
  // quotes escaped by an odd run of backslashes; a run that reaches back over this thread's first byte is counted in LDS
  uint32_t bad = 0;
  {
    uint32_t k = 0;  // backslashes directly in front of this thread's bytes (sb[-1] is a zero pad byte)
    while (k < 64 && sb[(int)tpos - 1 - (int)k] == '\\') k++;
    if (k >= 64) bad = 1;
    const uint64_t carry = k & 1u;
    const uint64_t bs = B & ~carry;
    const uint64_t follows = (bs << 1) | carry;
    const uint64_t even = 0x5555555555555555ull;
    const uint64_t odd_starts = bs & ~even & ~follows;
    const uint64_t sum = odd_starts + bs;
    const uint64_t escaped = (even ^ (sum << 1)) & follows;
    Q &= ~escaped;
  }
  const uint32_t par = (uint32_t)__popcll(Q) & 1u;
  const uint64_t pb = __ballot(par != 0);
  const uint32_t par_in = lanes_below(pb) & 1u;
  if (lane == 0) L.wpar[wv] = (uint32_t)__popcll(pb) & 1u;
  {
    uint16_t *q16 = reinterpret_cast<uint16_t *>(L.qmask), *b16 = reinterpret_cast<uint16_t *>(L.bmask);
#pragma unroll
    for (int q = 0; q < JQ_CPT; q++) { q16[base_chunk + q] = (uint16_t)(Q >> (16 * q)); b16[base_chunk + q] = (uint16_t)(B >> (16 * q)); }
  }
  if (__any(bad != 0) && lane == 0) L.misc[0] = 1u;
  const uint32_t qc = (uint32_t)__popcll(Q), bc = (uint32_t)__popcll(B);
  const uint32_t qinc = wave_scan_add(qc), binc = wave_scan_add(bc);
  if (lane == 63) { L.wqc[wv] = qinc; L.wbc[wv] = binc; }
  if constexpr (LDS::kHigh) {
    const uint64_t hb = __ballot(H != 0);
    if (lane == 0) L.hblk[wv] = hb;
  }
  __syncthreads();
  if (ablate == 2) { if ((Q ^ B ^ S) == 0x1234567ull) L.misc[3] = 1u; return false; }

  // ---- pass 2: inside-string state, separators outside strings ----
  uint32_t s_in = par_in;
  for (int i = 0; i < wv; i++) s_in ^= L.wpar[i];
  uint64_t px = Q;  // bit i = parity of the unescaped quotes in bytes [0, i]
  px ^= px << 1; px ^= px << 2; px ^= px << 4; px ^= px << 8; px ^= px << 16; px ^= px << 32;
  const uint64_t inside = px ^ (s_in ? ~0ull : 0ull);
  S &= ~inside;
  const uint32_t cnt = (uint32_t)__popcll(S);
  const uint32_t cinc = wave_scan_add(cnt);
  if (lane == 63) L.wcnt[wv] = cinc;
  uint32_t qb = qinc - qc, bb = binc - bc;
  for (int i = 0; i < wv; i++) { qb += L.wqc[i]; bb += L.wbc[i]; }
  // quotes / backslashes in front of the two 32-byte words of this thread's bytes
  L.qpre[tid * 2] = (uint16_t)qb; L.qpre[tid * 2 + 1] = (uint16_t)(qb + (uint32_t)__popc((uint32_t)Q));
  L.bpre[tid * 2] = (uint16_t)bb; L.bpre[tid * 2 + 1] = (uint16_t)(bb + (uint32_t)__popc((uint32_t)B));
  if (tid == JQ_THREADS - 1) { L.qpre[JQ_BYTES / 32] = (uint16_t)(qb + qc); L.bpre[JQ_BYTES / 32] = (uint16_t)(bb + bc); }
  __syncthreads();
  uint32_t cpre = 0, ctot = 0;
  for (int i = 0; i < JQ_WAVES; i++) { const uint32_t x = L.wcnt[i]; if (i < wv) cpre += x; ctot += x; }
  if (L.misc[0] || ctot > (uint32_t)JQ_SCAP) return false;  // uniform
  {
    uint32_t k = cpre + cinc - cnt;
    uint64_t m = S;
    while (m) {
      const uint32_t b = (uint32_t)__ffsll((long long)m) - 1; m &= m - 1;
      L.spos[k++] = (uint16_t)(tpos + b);
    }
    if (tid == 0) { L.misc[1] = ctot; L.spos[ctot] = 0xFFFFu; }
  }
  __syncthreads();
  if (ablate == 3) return false;
  (void)first;
  return true;
}

// the four bytes [ve - 4, ve) of the tile, byte 3 = sb[ve - 1]
__device__ __forceinline__ uint32_t jq_window4(const uint8_t *sb, uint32_t ve) {
  const uint32_t *w = reinterpret_cast<const uint32_t *>(sb + (int)((ve & ~3u) - 4u));
  return __builtin_amdgcn_alignbyte(w[1], w[0], ve);
}
// The escapes of a string body [a, b) as encoding/json's scanner takes them — \" \\ \/ \b \f \n \r \t \uXXXX — visiting only the
// backslashes (the bitmap says where they are) instead of every byte of the string.
template <class LDS> __device__ __forceinline__ bool jq_escapes_ok(const LDS &L, const uint8_t *sb, uint32_t a, uint32_t b) {
  uint32_t i = a;
  while (i < b) {
    uint32_t w = i >> 5;
    uint32_t m = L.bmask[w] & (~0u << (i & 31));
    while (!m && ((w + 1) << 5) < b) { w++; m = L.bmask[w]; }
    if (!m) return true;
    const uint32_t pos = (w << 5) + (uint32_t)__ffs((int)m) - 1;
    if (pos >= b) return true;
    if (pos + 1 >= b) return false;
    const uint32_t d = sb[pos + 1];
    if (d == 'u') {
      if (b - (pos + 2) < 4) return false;
      for (uint32_t k = 2; k <= 5; k++) { const uint32_t h = sb[pos + k]; if (!((h >= '0' && h <= '9') || ((h | 0x20u) >= 'a' && (h | 0x20u) <= 'f'))) return false; }
      i = pos + 6;
    } else {
      if (!(d == '"' || d == '\\' || d == '/' || d == 'b' || d == 'f' || d == 'n' || d == 'r' || d == 't')) return false;
      i = pos + 2;
    }
  }
  return true;
}
// unescapeStringBestEffort's output length for a string whose escapes are all of the one-character kind (\" \\ \/ \b \f \n \r \t
// give one byte, any other character but `u` keeps both); ~0u: a \u escape — the per-line parser decodes it
template <class LDS> __device__ __forceinline__ uint32_t jq_unescaped_len(const LDS &L, const uint8_t *sb, uint32_t ss, uint32_t sn) {
  uint32_t out = sn, i = ss;
  const uint32_t e = ss + sn;
  while (i < e) {
    // the next backslash at or after i
    uint32_t w = i >> 5;
    uint32_t m = L.bmask[w] & (~0u << (i & 31));
    while (!m && ((w + 1) << 5) < e) { w++; m = L.bmask[w]; }
    if (!m) break;
    const uint32_t pos = (w << 5) + (uint32_t)__ffs((int)m) - 1;
    if (pos + 1 >= e) break;  // (cannot happen inside a closed string)
    const uint32_t ch = sb[pos + 1];
    if (ch == 'u') return ~0u;
    if (ch == '"' || ch == '\\' || ch == '/' || ch == 'b' || ch == 'f' || ch == 'n' || ch == 'r' || ch == 't') out--;
    i = pos + 2;
  }
  return out;
}


}  // namespace tf
