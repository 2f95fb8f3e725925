// tf_segcopy.hpp — destination-centric packing of var-width cells (Arrow offsets + bytes), shared by the CSV
// ingest (cells cut out of the CSV text) and by row compaction (cells gathered through a selection vector).
//
// One workgroup owns a run of R = 256 * RPT consecutive rows of ONE column.  The run's cells form one contiguous
// destination range; it is cut into aligned 8-byte words, one word per lane, so a wave stores 512 contiguous bytes
// per instruction.  A word finds its row by binary search in the run's offsets (staged in LDS) and pulls its bytes
// from `src + src_off(row)` with two aligned 8-byte loads funnelled to the word's alignment (the bytes of one cell
// are contiguous, neighbouring lanes read neighbouring bytes); a word straddling rows takes one such read per row.
// A word belongs to the run holding its first byte; where it reaches past the run, offsets come from HBM.
//
// The kernels built on this are latency-bound by the dependent chain offsets → row → source bytes, so every lane
// carries SEG_U words at once: the searches, then all source loads, then the stores.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tf {

static constexpr uint32_t SEG_NONE = 0xFFFFFFFFu;  // src_off(row): the cell is zero-filled here and written elsewhere
static constexpr int SEG_U = 4;

// 8 bytes at `src + a` for any alignment (the buffers carry >= 16 bytes of slack past their payload)
__device__ __forceinline__ uint64_t seg_read8(const uint8_t *__restrict__ src, uint64_t a) {
  const uint32_t sh = (uint32_t)(a & 7) * 8;
  const uint64_t *q = reinterpret_cast<const uint64_t *>(src + (a & ~7ull));
  const uint64_t x = q[0];
  return sh ? (x >> sh) | (q[1] << (64 - sh)) : x;
}

// doff: LDS, R + 1 words; soff: LDS, R words.  All 256 threads of the workgroup call this.
template <int RPT, class SrcOff>
__device__ __forceinline__ void segcopy_run(const uint32_t *__restrict__ dst_off, const int64_t nrows, const int64_t k0, const uint8_t *__restrict__ src,
                                            uint8_t *__restrict__ dst, SrcOff src_off, uint32_t *doff, uint32_t *soff) {
  constexpr int R = 256 * RPT;
  const int t = threadIdx.x;
  const int nr = (int)(nrows - k0 < R ? nrows - k0 : R);
#pragma unroll
  for (int i = 0; i < RPT; i++) {
    const int r = i * 256 + t;
    if (r < nr) { doff[r] = dst_off[k0 + r]; soff[r] = src_off(k0 + r); }
  }
  if (t == 0) doff[nr] = dst_off[k0 + nr];
  const uint32_t Dend = dst_off[nrows];
  __syncthreads();
  const uint32_t D0 = doff[0], D1 = doff[nr];
  if (D0 == D1) return;
  const uint32_t wbeg = (uint32_t)(((uint64_t)D0 + 7) >> 3), wend = (uint32_t)(((uint64_t)D1 + 7) >> 3);  // words whose first byte lies in [D0, D1)
  for (uint32_t wb = wbeg + t; wb < wend; wb += 256 * SEG_U) {
    int row[SEG_U];
    uint64_t a0[SEG_U], a1[SEG_U];
    uint32_t shv[SEG_U];
    bool fast[SEG_U], live[SEG_U];
#pragma unroll
    for (int u = 0; u < SEG_U; u++) {
      const uint32_t w = wb + (uint32_t)u * 256;
      live[u] = w < wend;
      fast[u] = false; a0[u] = 0; a1[u] = 0; shv[u] = 0; row[u] = 0;
      if (!live[u]) continue;
      const uint32_t b0 = w << 3;
      int lo = 0, hi = nr;  // first row whose offset lies beyond b0; the row holding b0 is the one before
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (doff[mid] <= b0) lo = mid + 1; else hi = mid; }
      const int r = lo - 1;
      row[u] = r;
      if (doff[r + 1] >= b0 + 8) {
        fast[u] = true;
        const uint32_t so = soff[r];
        if (so != SEG_NONE) {
          const uint64_t a = (uint64_t)so + (b0 - doff[r]);
          const uint64_t *q = reinterpret_cast<const uint64_t *>(src + (a & ~7ull));
          shv[u] = (uint32_t)(a & 7) * 8;
          a0[u] = q[0]; a1[u] = q[1];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < SEG_U; u++) {
      if (!live[u]) continue;
      const uint32_t b0 = (wb + (uint32_t)u * 256) << 3;
      uint64_t word;
      if (fast[u]) word = shv[u] ? (a0[u] >> shv[u]) | (a1[u] << (64 - shv[u])) : a0[u];
      else {  // the word straddles rows: one funnelled read per row that owns bytes of it
        word = 0;
        int r = row[u];
        uint32_t rs = doff[r], re = doff[r + 1], so = soff[r];
        int64_t gr = k0 + r;
        const uint32_t bend = b0 + 8 < Dend ? b0 + 8 : Dend;
        uint32_t b = b0;
        while (b < bend) {
          while (re <= b) {
            gr++; rs = re;
            const int64_t lr = gr - k0;
            re = lr + 1 <= nr ? doff[lr + 1] : dst_off[gr + 1];
            so = lr < nr ? soff[lr] : src_off(gr);
          }
          const uint32_t take = (re < bend ? re : bend) - b;
          if (so != SEG_NONE) {
            uint64_t x = seg_read8(src, (uint64_t)so + (b - rs));
            if (take < 8) x &= (1ull << (8 * take)) - 1;
            word |= x << (8 * (b - b0));
          }
          b += take;
        }
      }
      *reinterpret_cast<uint64_t *>(dst + b0) = word;
    }
  }
}

}  // namespace tf
