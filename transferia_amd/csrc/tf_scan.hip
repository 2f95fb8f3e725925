// tf_scan.hip — exclusive prefix sums over uint32 arrays (row counts, string
// lengths → Arrow offsets, selection vectors).  Reduce-then-scan with 4096
// elements per 256-thread workgroup: every element is read twice and written
// once, all accesses 16 B/lane coalesced.  HBM-bound: 12 B/element.
#include "tf_common.hpp"

namespace tf {

static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_ITEMS = 16;
static constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 4096
static constexpr int64_t RAW_SUMS_MAX = 1024;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// Exclusive scan of one value per thread across a 256-thread block.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *total, uint32_t *lds /*>=4*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = wave_incl_scan(v);
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_THREADS / 64; i++) { uint32_t s = lds[i]; if (i < w) base += s; tot += s; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// seg = blockIdx.y; arrays of nseg segments, each seg_stride elements apart.
__global__ void __launch_bounds__(SCAN_THREADS) scan_reduce_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ sums,
                                                                  int64_t n, int64_t seg_stride, int64_t nblocks) {
  __shared__ uint32_t lds[4];
  const uint32_t *src = in + (int64_t)blockIdx.y * seg_stride;
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
  if (base + SCAN_ITEMS <= n && ((reinterpret_cast<uintptr_t>(src + base) & 15) == 0)) {
    const uint4 *p = reinterpret_cast<const uint4 *>(src + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; i++) { uint4 v = p[i]; s += v.x + v.y + v.z + v.w; }
  } else {
    for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) s += src[base + i];
  }
  uint32_t tot;
  block_excl_scan(s, &tot, lds);
  if (threadIdx.x == 0) sums[(int64_t)blockIdx.y * nblocks + blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_apply_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                                 const uint32_t *__restrict__ block_base, int64_t n,
                                                                 int64_t seg_stride, int64_t nblocks, int write_total, int raw_sums) {
  __shared__ uint32_t lds[4];
  // raw_sums: block_base holds the workgroups' own totals, not their scan — each workgroup adds up the ones in front of it (at most
  // RAW_SUMS_MAX words from L2) instead of a third launch scanning them
  uint32_t bb = 0;
  if (block_base) {
    const uint32_t *bs = block_base + (int64_t)blockIdx.y * nblocks;
    if (raw_sums) {
      uint32_t part = 0;
      for (int64_t i = threadIdx.x; i < (int64_t)blockIdx.x; i += SCAN_THREADS) part += bs[i];
      block_excl_scan(part, &bb, lds);
    } else bb = bs[blockIdx.x];
  }
  const uint32_t *src = in + (int64_t)blockIdx.y * seg_stride;
  uint32_t *dst = out + (int64_t)blockIdx.y * seg_stride;
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  bool fast = base + SCAN_ITEMS <= n && ((reinterpret_cast<uintptr_t>(src + base) & 15) == 0);
  if (fast) {
    const uint4 *p = reinterpret_cast<const uint4 *>(src + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; i++) { uint4 q = p[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) v[i] = (base + i < n) ? src[base + i] : 0;
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
  uint32_t tot;
  uint32_t ex = block_excl_scan(s, &tot, lds) + bb;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) { uint32_t t = v[i]; v[i] = ex; ex += t; }
  if (fast) {
    uint4 *p = reinterpret_cast<uint4 *>(dst + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; i++) p[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) dst[base + i] = v[i];
  }
  // the element one past the end receives the grand total (Arrow offsets[n])
  if (write_total && base <= n - 1 && n - 1 < base + SCAN_ITEMS) dst[n] = ex;
}

static void scan_impl(const uint32_t *in, uint32_t *out, int64_t n, int64_t seg_stride, int nseg, bool with_total) {
  if (n <= 0) {
    if (with_total) for (int s = 0; s < nseg; s++) TF_HIP(hipMemsetAsync(out + (int64_t)s * seg_stride, 0, 4, ctx().stream));
    return;
  }
  hipStream_t st = ctx().stream;
  int64_t nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
  dim3 grid((unsigned)nblocks, (unsigned)nseg);
  if (nblocks == 1) {
    scan_apply_kernel<<<grid, SCAN_THREADS, 0, st>>>(in, out, nullptr, n, seg_stride, 1, with_total ? 1 : 0, 0);
    return;
  }
  Buf sums = dalloc((size_t)nblocks * nseg * 4 + 4);
  scan_reduce_kernel<<<grid, SCAN_THREADS, 0, st>>>(in, ptr<uint32_t>(sums), n, seg_stride, nblocks);
  const bool raw = nblocks <= RAW_SUMS_MAX;  // two launches instead of three up to 4 Mi elements a segment
  if (!raw) scan_impl(ptr<uint32_t>(sums), ptr<uint32_t>(sums), nblocks, nblocks, nseg, false);
  scan_apply_kernel<<<grid, SCAN_THREADS, 0, st>>>(in, out, ptr<uint32_t>(sums), n, seg_stride, nblocks, with_total ? 1 : 0, raw ? 1 : 0);
}

__global__ void __launch_bounds__(SCAN_THREADS) sum_u64_kernel(const uint32_t *__restrict__ in, int64_t n, int64_t seg_stride, unsigned long long *totals) {
  const uint32_t *src = in + (int64_t)blockIdx.y * seg_stride;
  unsigned long long s = 0;
  for (int64_t i = (int64_t)blockIdx.x * SCAN_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SCAN_THREADS) s += src[i];
  for (int d = 32; d; d >>= 1) s += __shfl_down(s, d, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(totals + blockIdx.y, s);
}

void sum_u32_segments_u64(const uint32_t *in, int64_t seg_len, int nseg, int64_t seg_stride, unsigned long long *totals) {
  if (nseg <= 0 || seg_len <= 0) return;
  const int64_t nb = std::min<int64_t>((seg_len + SCAN_THREADS * 8 - 1) / (SCAN_THREADS * 8), 1024);
  sum_u64_kernel<<<dim3((unsigned)nb, (unsigned)nseg), SCAN_THREADS, 0, ctx().stream>>>(in, seg_len, seg_stride, totals);
}

__global__ void segment_totals_kernel(const uint32_t *lens_all, int64_t seg_stride, int64_t n, int nseg, uint32_t *out) {
  for (int t = threadIdx.x; t < nseg; t += blockDim.x) out[t] = lens_all[(int64_t)t * seg_stride + n];
}
// the totals exclusive_scan_u32_segments left at [seg_len] of every segment, gathered by one launch into one read-back (the pointer is
// valid after the next sync) — nseg four-byte copies of their own were 5 us each
const uint32_t *segment_totals_to_host(const uint32_t *scanned, int64_t seg_len, int nseg, int64_t seg_stride) {
  if (nseg <= 0) return nullptr;
  Buf out = dalloc((size_t)nseg * 4);
  segment_totals_kernel<<<1, 256, 0, ctx().stream>>>(scanned, seg_stride, seg_len, nseg, ptr<uint32_t>(out));
  return d2h_u32(out->p, (size_t)nseg);
}

__global__ void __launch_bounds__(256) any_nonzero_kernel(const uint32_t *v, int64_t n, uint32_t *flag) {
  bool hit = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) hit |= v[i] != 0;
  if (__ballot(hit) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
// is any of v[0 .. n) non-zero?  One word read back (valid after the next sync) instead of the array.
const uint32_t *any_nonzero_to_host(const uint32_t *v, int64_t n) {
  Buf flag = dalloc_zero(4);
  if (n > 0) any_nonzero_kernel<<<(unsigned)std::min<int64_t>((n + 255) / 256, 1024), 256, 0, ctx().stream>>>(v, n, ptr<uint32_t>(flag));
  return d2h_u32(flag->p, 1);
}

void exclusive_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, bool with_total) {
  KernelTimer t("scan_u32");
  scan_impl(in, out, n, 0, 1, with_total);
}

// Each segment is (seg_len + 1) entries long, segments seg_stride (>= seg_len+1,
// a multiple of 4 keeps the 16-byte path) apart; entry 0 of the input is ignored
// by construction: callers store length[r] at index r and receive offsets at
// index r, with the total at index seg_len.
void exclusive_scan_u32_segments(uint32_t *inout, int64_t seg_len, int nseg, int64_t seg_stride) {
  if (nseg <= 0) return;
  KernelTimer t("scan_u32_segments");
  scan_impl(inout, inout, seg_len, seg_stride, nseg, true);
}

}  // namespace tf
