// tf_jsontile.hpp — the front half of the JSON tile parsers (json_parse_tiles in tf_json.hip, sr_parse_tiles in
// tf_srjson.hip): consecutive lines staged in LDS, their bytes classified once with SWAR masks (unescaped quotes by the
// backslash-run carry trick, the in-string state by a prefix xor carried across lanes and waves), the separators
// { } [ ] , : outside strings indexed, every line framed.  What a (member, line) cell means is the caller's business.
#pragma once
#include "tf_swar.hpp"
#include "tf_wave.hpp"

namespace tf {

#ifndef TF_JT_THREADS
#define TF_JT_THREADS 512
#endif
static constexpr int JT_THREADS = TF_JT_THREADS;
static constexpr int JT_CPT = 4;                           // 16-byte chunks per thread
static constexpr int JT_BYTES = JT_THREADS * JT_CPT * 16;  // 32 KiB of text per tile
static constexpr uint64_t JT_MASK = JT_CPT == 4 ? ~0ull : (1ull << (16 * (JT_CPT & 3))) - 1;  // this thread's bytes as mask bits
static constexpr int JT_LINES = 32;                        // lines per tile at most
static constexpr int JT_SCAP = 9 * JT_THREADS;             // separators indexed per tile
static constexpr int JT_MEM = 192;                         // members per line
static constexpr int JT_OWN = 512;                         // columns (duplicate-key check)
static constexpr int JT_KREF = 4096;                       // bytes of key text the member map keeps

struct JtTile {
  const uint8_t *sb; const uint16_t *spos; const uint32_t *qmask; const uint16_t *qpre, *bpre;
  uint32_t g0;
};
// unescaped quotes / backslashes in tile positions [a, b), a <= b
__device__ __forceinline__ uint32_t jt_quotes_in(const JtTile &t, uint32_t a, uint32_t b) {
  const uint32_t ca = (uint32_t)t.qpre[a >> 5] + (uint32_t)__popc(t.qmask[a >> 5] & ((1u << (a & 31)) - 1u));
  const uint32_t cb = (uint32_t)t.qpre[b >> 5] + (uint32_t)__popc(t.qmask[b >> 5] & ((1u << (b & 31)) - 1u));
  return cb - ca;
}
// the 8 bytes at tile position a (any alignment), little-endian
__device__ __forceinline__ uint64_t jt_word(const uint8_t *sb, uint32_t a) {
  const uint32_t *w = reinterpret_cast<const uint32_t *>(sb + (a & ~3u));
  const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
  return (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, a) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, a) << 32);
}
__device__ __forceinline__ bool jt_same2(const uint8_t *sb, uint32_t a, const uint8_t *kr, uint32_t b, uint32_t n) {  // n bytes at sb + a and at kr + b (b a multiple of 8)
  for (uint32_t k = 0; k < n; k += 8) {
    const uint32_t nb = n - k < 8 ? n - k : 8u;
    const uint64_t m = nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1;
    if ((jt_word(sb, a + k) ^ *reinterpret_cast<const uint64_t *>(kr + b + k)) & m) return false;
  }
  return true;
}
// -?digits with at most 19 digits (their value fits 64 bits unsigned).  false: not that form.
__device__ __forceinline__ bool jt_int_token(const uint8_t *sb, uint32_t vs, uint32_t ve, bool *neg, uint64_t *mag, uint32_t *ndig) {
  const uint32_t n = ve - vs;
  const bool ng = sb[vs] == '-';
  const uint32_t nd = n - (ng ? 1u : 0u);
  if (nd - 1u >= 19u) return false;
  uint32_t g0v = 0, g1v = 0, g2v = 0, lo, hi;
  window8(sb, ve, &lo, &hi);
  bool ok = digits8_window(lo, hi, min(nd, 8u), &g0v);
  if (nd > 8) { window8(sb, ve - 8, &lo, &hi); ok = digits8_window(lo, hi, min(nd - 8, 8u), &g1v) && ok; }
  if (nd > 16) { window8(sb, ve - 16, &lo, &hi); ok = digits8_window(lo, hi, nd - 16, &g2v) && ok; }
  *neg = ng; *ndig = nd;
  *mag = ((uint64_t)g2v * 100000000ull + g1v) * 100000000ull + g0v;
  return ok;
}
// parseRawNumber's token: a run of [0-9.+-eE] that is not a lone sign (inf / nan spellings: per-line path)
__device__ __forceinline__ bool jt_number_chars(const uint8_t *sb, uint32_t vs, uint32_t ve) {
  if (ve - vs > 64u) return false;
  for (uint32_t i = vs; i < ve; i++) { const uint32_t d = (uint32_t)sb[i] - 0x2Bu; if (!(d < 64u && ((0x0400000004007FEDull >> d) & 1ull))) return false; }
  return !(ve - vs == 1 && (sb[vs] == '-' || sb[vs] == '+'));
}
__device__ __forceinline__ bool jt_ws(uint32_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }
__device__ __forceinline__ bool jt_lit(const uint8_t *sb, uint32_t vs, uint32_t n, uint32_t word, uint32_t len) {  // the 4 first bytes + length
  return n == len && (uint32_t)jt_word(sb, vs) == word && (len == 4 || sb[vs + 4] == 'e');
}

// what the front half leaves in LDS
struct JtLds {
  __attribute__((aligned(16))) uint8_t sbuf[16 + JT_BYTES + 48];
  uint16_t spos[JT_SCAP + 2];
  uint32_t qmask[JT_BYTES / 32 + 1], smask[JT_BYTES / 32 + 1];  // unescaped quotes; inside a string after the byte
  uint32_t bmask[JT_BYTES / 32 + 1];                            // backslashes
  uint16_t qpre[JT_BYTES / 32 + 2], bpre[JT_BYTES / 32 + 2];    // quotes / backslashes in front of each 32-byte word
  uint32_t wpar[JT_THREADS / 64], wcnt[JT_THREADS / 64], wqc[JT_THREADS / 64], wbc[JT_THREADS / 64];
  uint16_t lstart[JT_LINES], lend[JT_LINES], lbase[JT_LINES], lK[JT_LINES];  // per line: bytes [lstart, lend), first separator, members
  uint8_t lslow[JT_LINES];                                                   // 0 taken, 1 handed over, 2 no line here
  uint32_t misc[4];                                                          // 0: tile cannot be taken, 1: separators
};
__device__ __forceinline__ uint32_t jt_backslashes_in(const JtLds &L, uint32_t a, uint32_t b) {
  const uint32_t ca = (uint32_t)L.bpre[a >> 5] + (uint32_t)__popc(L.bmask[a >> 5] & ((1u << (a & 31)) - 1u));
  const uint32_t cb = (uint32_t)L.bpre[b >> 5] + (uint32_t)__popc(L.bmask[b >> 5] & ((1u << (b & 31)) - 1u));
  return cb - ca;
}

// Stage the bytes [g0, last) of `data` (g0 a multiple of 16, first = the first line's offset), classify, index, frame the
// lines.  The caller has written lstart / lend / lslow / lbase (= 0xFFFF) of its nl lines.  All threads of the workgroup
// call it (it holds the barriers); false (uniform): the tile cannot be taken.
__device__ __forceinline__ bool jt_front(JtLds &L, const uint8_t *data, const uint32_t first, const uint32_t last, const uint32_t g0, const int nl, const int ablate = 0) {
  uint8_t *const sbuf = L.sbuf, *const sb = L.sbuf + 16;
  uint16_t *const spos = L.spos, *const qpre = L.qpre, *const bpre = L.bpre, *const lstart = L.lstart, *const lend = L.lend, *const lbase = L.lbase, *const lK = L.lK;
  uint32_t *const qmask = L.qmask, *const smask = L.smask, *const bmask = L.bmask, *const wpar = L.wpar, *const wcnt = L.wcnt, *const wqc = L.wqc, *const wbc = L.wbc, *const misc = L.misc;
  uint8_t *const lslow = L.lslow;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // ---- stage: coalesced 16 B per lane ----
#pragma unroll
  for (int it = 0; it < JT_CPT; it++) {
    const int chunk = it * JT_THREADS + tid;
    const uint32_t gp = g0 + (uint32_t)chunk * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gp < last) v = *reinterpret_cast<const uint4 *>(data + gp);  // the buffer is padded past its payload
    *reinterpret_cast<uint4 *>(sb + chunk * 16) = v;
  }
  if (tid < 4) { reinterpret_cast<uint32_t *>(sbuf)[tid] = 0; misc[tid] = 0; }
  if (tid < 12) reinterpret_cast<uint32_t *>(sbuf + 16 + JT_BYTES)[tid] = 0;
  __syncthreads();
  if (ablate == 1) return false;  // (TFGPU_JT_ABLATE, profiling only: leave after phase n; the tile then goes to the per-line parser)


  // ---- pass 1: byte classes of this thread's 48 bytes ----
  const uint32_t base_chunk = (uint32_t)tid * JT_CPT, tpos = base_chunk * 16;
  uint64_t Q = 0, B = 0, S = 0;
  {
#pragma unroll
    for (int q = 0; q < JT_CPT; q++) {
      const uint4 v = *reinterpret_cast<const uint4 *>(sb + (base_chunk + q) * 16);
      const Chunk16 ck = chunk16(v);
      Q |= (uint64_t)class16(ck, 0x22222222u) << (16 * q);
      B |= (uint64_t)class16(ck, 0x5C5C5C5Cu) << (16 * q);
      const uint32_t st = class16(ck, 0x3A3A3A3Au) | class16(ck, 0x2C2C2C2Cu) | class16(ck, 0x7B7B7B7Bu) | class16(ck, 0x7D7D7D7Du) | class16(ck, 0x5B5B5B5Bu) | class16(ck, 0x5D5D5D5Du);
      S |= (uint64_t)st << (16 * q);
    }
  }
  {  // bytes that belong to no line of this tile (what lies between two frames, say) take no part
    uint64_t cover = 0;
    for (int j = 0; j < nl; j++) {
      if (lslow[j] == 2) continue;
      const uint32_t a = lstart[j], b = lend[j];
      if (b <= tpos || a >= tpos + (uint32_t)JT_CPT * 16u) continue;
      const uint32_t lo = a > tpos ? a - tpos : 0u, hi = min(b - tpos, (uint32_t)JT_CPT * 16u);
      cover |= (hi >= 64 ? ~0ull : (1ull << hi) - 1) & ~((1ull << lo) - 1);
    }
    Q &= cover; B &= cover; S &= cover;
  }
  {  // what precedes the tile's first line (up to 15 bytes, the tail of the line before) belongs to another tile
    const uint32_t frs = first - g0;
    if (tpos < frs) { const uint64_t keep = frs - tpos >= 64 ? 0ull : ~0ull << (frs - tpos); Q &= keep; B &= keep; S &= keep; }
    const uint32_t lim = last - g0;  // … and so does the head of the next line that the last 16-byte chunk brought along
    if (tpos + (uint32_t)JT_CPT * 16u > lim) { const uint64_t keep = lim <= tpos ? 0ull : (1ull << (lim - tpos)) - 1; Q &= keep; B &= keep; S &= keep; }
  }
  // quotes escaped by an odd run of backslashes; a run that reaches back over this thread's first byte is counted in LDS
  uint32_t bad = 0;
  {
    uint32_t k = 0;  // backslashes directly in front of this thread's bytes (sb[-1] is a zero pad byte)
    while (k < 64 && sb[(int)tpos - 1 - (int)k] == '\\') k++;
    if (k >= 64) bad = 1;
    const uint64_t carry = k & 1u;
    uint64_t bs = B & ~carry;
    const uint64_t follows = (bs << 1) | carry;
    const uint64_t even = 0x5555555555555555ull;
    const uint64_t odd_starts = bs & ~even & ~follows;
    const uint64_t sum = odd_starts + bs;
    const uint64_t escaped = (even ^ (sum << 1)) & follows;
    Q &= ~escaped;
  }
  uint32_t par = (uint32_t)__popcll(Q) & 1u;
  const uint64_t pb = __ballot(par != 0);
  const uint32_t par_in = lanes_below(pb) & 1u;
  if (lane == 0) wpar[wv] = (uint32_t)__popcll(pb) & 1u;
  // quote / backslash words of 32 bytes: this thread's 48 bytes are one and a half of them → write 16-bit halves
  {
    uint16_t *q16 = reinterpret_cast<uint16_t *>(qmask), *b16 = reinterpret_cast<uint16_t *>(bmask);
#pragma unroll
    for (int q = 0; q < JT_CPT; q++) { q16[base_chunk + q] = (uint16_t)(Q >> (16 * q)); b16[base_chunk + q] = (uint16_t)(B >> (16 * q)); }
  }
  if (__any(bad != 0) && lane == 0) misc[0] = 1u;
  __syncthreads();
  if (ablate == 2) { if ((Q ^ B ^ S) == 0x1234567ull) misc[3] = 1u; return false; }

  // ---- pass 2: inside-string state, separators outside strings ----
  uint32_t s_in = par_in;
  for (int i = 0; i < wv; i++) s_in ^= wpar[i];
  uint64_t px = Q;  // bit i = parity of the unescaped quotes in bytes [0, i]
  px ^= px << 1; px ^= px << 2; px ^= px << 4; px ^= px << 8; px ^= px << 16; px ^= px << 32;
  const uint64_t inside = (px ^ (s_in ? ~0ull : 0ull)) & JT_MASK;
  S &= ~inside & JT_MASK;
  {
    uint16_t *s16 = reinterpret_cast<uint16_t *>(smask);
#pragma unroll
    for (int q = 0; q < JT_CPT; q++) s16[base_chunk + q] = (uint16_t)(inside >> (16 * q));
  }
  const uint32_t cnt = (uint32_t)__popcll(S);
  const uint32_t cinc = wave_scan_add(cnt);
  // prefix counts per 32-byte word: words 3*tid/2 … — computed by the threads that own a word's first half
  const uint32_t qc = (uint32_t)__popcll(Q), bc = (uint32_t)__popcll(B);
  const uint32_t qinc = wave_scan_add(qc), binc = wave_scan_add(bc);
  if (lane == 63) { wcnt[wv] = cinc; wqc[wv] = qinc; wbc[wv] = binc; }
  __syncthreads();
  uint32_t cpre = 0, ctot = 0, qb = qinc - qc, bb = binc - bc;
  for (int i = 0; i < JT_THREADS / 64; i++) { const uint32_t x = wcnt[i]; if (i < wv) { cpre += x; qb += wqc[i]; bb += wbc[i]; } ctot += x; }
  if (misc[0] || ctot > (uint32_t)JT_SCAP) return false;  // uniform
  {
    // separator index
    uint32_t k = cpre + cinc - cnt;
    uint64_t m = S;
    while (m) {
      const uint32_t b = (uint32_t)__ffsll((long long)m) - 1; m &= m - 1;
      const uint32_t pos = tpos + b;
      if (sb[pos] == '{') for (int j = 0; j < nl; j++) if (lstart[j] == pos && lslow[j] == 0) lbase[j] = (uint16_t)k;  // a line's own brace: where its separators start
      spos[k++] = (uint16_t)pos;
    }
    if (tid == 0) { misc[1] = ctot; spos[ctot] = 0xFFFFu; }
    // quotes / backslashes in front of every 32-byte word that STARTS in this thread's bytes (tpos = 48 * tid: words start at
    // multiples of 32 → at tpos when tid is even, at tpos + 16 when tid is odd, and at tpos + 32 when tid is even)
#pragma unroll
    for (int q = 0; q < JT_CPT; q++) {
      const uint32_t cp = tpos + 16u * q;
      if ((cp & 31u) == 0) {
        const uint64_t below = q ? (1ull << (16 * q)) - 1 : 0ull;
        qpre[cp >> 5] = (uint16_t)(qb + (uint32_t)__popcll(Q & below));
        bpre[cp >> 5] = (uint16_t)(bb + (uint32_t)__popcll(B & below));
      }
    }
    if (tid == JT_THREADS - 1) { qpre[JT_BYTES / 32] = (uint16_t)(qb + qc); bpre[JT_BYTES / 32] = (uint16_t)(bb + bc); }
  }
  __syncthreads();

  if (ablate == 3) return false;
  const uint32_t nsep = misc[1];
  // ---- lines: where their separators start, how many, the frame { … } ----
  if (tid < nl && lslow[tid] == 0) {
    const uint32_t ls = lstart[tid], le = lend[tid];
    const uint32_t b0 = lbase[tid];  // written by the index sweep when the line's first byte is a brace outside strings
    uint32_t b1 = nsep;              // the next line's, or the end of the index
    for (int j = tid + 1; j < nl; j++) if (lslow[j] != 2) { b1 = lbase[j]; break; }
    const uint32_t c = (b0 != 0xFFFFu && b1 != 0xFFFFu && b1 > b0) ? b1 - b0 : 0u;
    const bool in_before = ls ? ((smask[(ls - 1) >> 5] >> ((ls - 1) & 31)) & 1u) : false;  // a string open across the line start
    const bool in_after = (smask[(le - 1) >> 5] >> ((le - 1) & 31)) & 1u;
    bool ok = !in_before && !in_after && c >= 3 && (c & 1u) && spos[b0 < nsep ? b0 : 0] == ls && spos[b0 + c - 1] == le - 1 && sb[ls] == '{' && sb[le - 1] == '}' && (c - 1) / 2 <= (uint32_t)JT_MEM;
    lbase[tid] = (uint16_t)b0; lK[tid] = (uint16_t)((c - 1) / 2);
    if (!ok) lslow[tid] = 1;
  }
  __syncthreads();
  return true;
}

}  // namespace tf
