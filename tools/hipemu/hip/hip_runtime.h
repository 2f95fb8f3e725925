// tools/hipemu — TEST INFRASTRUCTURE ONLY.  A stand-in for <hip/hip_runtime.h> that runs the kernels of libtfgpu as
// plain C++ in the GPU-less build container — every thread of a workgroup a fiber, wavefront operations and barriers
// exchanged in lockstep (see "wavefronts and workgroups" below) — so their byte-level logic can be checked against
// the oracle BEFORE a GPU box is spent on them.  Never shipped, never loaded by
// transferia_amd/: the product path is the hipcc build for gfx950 and fails loudly without a device.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __constant__ const
#define __launch_bounds__(...)
#define __shared__ static thread_local /* one instance per kernel and host thread: workgroups run one after another */
/* dynamic LDS named at file scope (`extern __shared__ T name[];` in HIP): here a fixed 160 KiB per array */
#define TF_DYNAMIC_LDS(type, name) static thread_local type name[(160 * 1024) / sizeof(type)]
#define TF_GLOBAL_PTR(T, p) ((T *)(p))
#define TF_CONST_PTR(T, p) ((const T *)(p))
#define TF_OPAQUE(x) ((void)(x))
#define TF_KEEP(x) ((void)(x))
#define __builtin_amdgcn_fence(...) ((void)0)

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
typedef struct emuStream *hipStream_t;
typedef struct emuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostMallocPortable = 1 };
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; char name[64]; };

inline hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipMallocAsync(void **p, size_t n, hipStream_t) { return hipMalloc(p, n); }
inline hipError_t hipFreeAsync(void *p, hipStream_t) { return hipFree(p); }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
/* HIPEMU_DEVICES=G fakes G devices (one address space, as under unified addressing): the several-devices-in-one-process logic
 * of tf_runtime.hip / tf_shard.hip runs here, nothing about peer copies is proven */
inline int &hipemu_current_device() { static thread_local int d = 0; return d; }
inline hipError_t hipGetDeviceCount(int *n) { const char *e = std::getenv("HIPEMU_DEVICES"); *n = e ? std::max(1, std::atoi(e)) : 1; return hipSuccess; }
inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipSetDevice(int d) { int n; hipGetDeviceCount(&n); if (d < 0 || d >= n) return hipErrorInvalidValue; hipemu_current_device() = d; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { std::memset(p, 0, sizeof *p); std::strcpy(p->gcnArchName, "gfx950:hipemu"); p->multiProcessorCount = 256; return hipSuccess; }

inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
inline long long __double_as_longlong(double d) { long long v; std::memcpy(&v, &d, 8); return v; }
inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
inline unsigned __float_as_uint(float f) { unsigned v; std::memcpy(&v, &f, 4); return v; }
inline float __uint_as_float(unsigned v) { float f; std::memcpy(&f, &v, 4); return f; }
inline float __uint2float_rz(unsigned x) { float f = (float)x; if ((double)f > (double)x) f = std::nextafterf(f, 0.0f); return f; }
inline int __float_as_int(float f) { int v; std::memcpy(&v, &f, 4); return v; }
inline float __int_as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
inline int __mul24(int a, int b) { return ((a << 8) >> 8) * ((b << 8) >> 8); }
// ---- wavefronts and workgroups: lockstep fibers ------------------------------------------------------------------------
// Every thread of a workgroup is a ucontext fiber; workgroups run one after another.  A fiber runs until it reaches a
// wave operation (ballot / any / all / shuffle / readfirstlane / bpermute) or __syncthreads(), where it parks.  When no
// lane of a wave can run any more, the lanes parked at a wave operation exchange their operands — they are the
// operation's active lanes, as on the hardware — and go on; when nothing in the workgroup can run, the barrier opens.
// The model is a 64-lane wavefront (gfx950), lanes = consecutive threadIdx.x.
#include <sys/mman.h>
#include <ucontext.h>
#include <vector>
namespace hipemu {
enum { RUN = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };
struct Fiber { ucontext_t ctx; int state; unsigned tid; };
struct Wave { unsigned long long slot[64]; unsigned long long snap[64]; unsigned long long mask; };
struct Sched {
  ucontext_t main;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  Fiber *cur = nullptr;
  void (*entry)(void *) = nullptr;
  void *entry_arg = nullptr;
};
inline thread_local Sched *g = nullptr;
inline constexpr size_t STACK = 256 * 1024;
inline void yield_(int st) { Fiber *f = g->cur; f->state = st; swapcontext(&f->ctx, &g->main); }
inline unsigned lane_() { return g->cur->tid & 63u; }
inline Wave &wave_() { return g->waves[g->cur->tid >> 6]; }
// park at a wave operation with operand v; returns after the exchange (snap / mask hold the active lanes' operands)
inline Wave &wave_sync(unsigned long long v) {
  Wave &w = wave_();
  w.slot[lane_()] = v;
  yield_(WAIT_WAVE);
  return w;
}
inline void trampoline() { g->entry(g->entry_arg); g->cur->state = DONE; swapcontext(&g->cur->ctx, &g->main); }
}  // namespace hipemu

inline unsigned __lane_id() { return hipemu::lane_(); }
inline unsigned long long __ballot(int p) {
  hipemu::Wave &w = hipemu::wave_sync(p ? 1ull : 0ull);
  unsigned long long r = 0;
  for (int i = 0; i < 64; i++) if (((w.mask >> i) & 1ull) && w.snap[i]) r |= 1ull << i;
  return r;
}
inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return __ballot(p ? 1 : 0); }
inline int __any(int p) { return __ballot(p) != 0; }
inline int __all(int p) { hipemu::Wave &w = hipemu::wave_sync(p ? 1ull : 0ull); for (int i = 0; i < 64; i++) if (((w.mask >> i) & 1ull) && !w.snap[i]) return 0; return 1; }
inline unsigned long long __activemask() { hipemu::Wave &w = hipemu::wave_sync(0); return w.mask; }
template <class T> inline T emu_lane_read(T own, int src) {
  static_assert(sizeof(T) <= 8, "shuffle operand");
  unsigned long long bits = 0; std::memcpy(&bits, &own, sizeof(T));
  hipemu::Wave &w = hipemu::wave_sync(bits);
  if (src < 0 || src > 63 || !((w.mask >> src) & 1ull)) return own;  // inactive source lane: the hardware returns garbage; own value here
  T out; std::memcpy(&out, &w.snap[src], sizeof(T)); return out;
}
template <class T> inline T __shfl(T v, int src, int width = 64) { const int l = (int)hipemu::lane_(); return emu_lane_read(v, (l & ~(width - 1)) | (src & (width - 1))); }
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) { const int l = (int)hipemu::lane_(); const int s = l - (int)d; return emu_lane_read(v, (s < (l & ~(width - 1))) ? l : s); }
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) { const int l = (int)hipemu::lane_(); const int s = l + (int)d; return emu_lane_read(v, (s > (l | (width - 1))) ? l : s); }
template <class T> inline T __shfl_xor(T v, int m, int width = 64) { const int l = (int)hipemu::lane_(); const int s = l ^ m; return emu_lane_read(v, (s > (l | (width - 1))) ? l : s); }
inline void __syncthreads() { hipemu::yield_(hipemu::WAIT_BLOCK); }
inline int __builtin_amdgcn_readfirstlane(int v) {
  hipemu::Wave &w = hipemu::wave_sync((unsigned long long)(unsigned)v);
  return (int)(unsigned)w.snap[__builtin_ctzll(w.mask)];
}
inline int __builtin_amdgcn_readlane(int v, int l) { return emu_lane_read(v, l); }
inline int __builtin_amdgcn_ds_bpermute(int byte_addr, int v) { return emu_lane_read(v, (byte_addr >> 2) & 63); }
// v_mov_b32 ... dpp: the controls libtfgpu uses (gfx9 rows of 16 lanes).  A lane without a source, or masked out by
// row_mask, keeps `old` (bound_ctrl:0 would give 0 — the library always passes old = 0).
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int l = (int)hipemu::lane_(), row = l >> 4;
  int from = -1;
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = l - (ctrl - 0x110); if (s >= 0 && (s >> 4) == row) from = s; }        // row_shr:n
  else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = l + (ctrl - 0x100); if (s < 64 && (s >> 4) == row) from = s; }    // row_shl:n
  else if (ctrl == 0x142) { if (row >= 1) from = row * 16 - 1; }                                                             // row_bcast:15
  else if (ctrl == 0x143) { if (l >= 32) from = 31; }                                                                        // row_bcast:31
  else if (ctrl == 0x138) { if (l >= 1) from = l - 1; }                                                                      // wave_shr:1
  else { std::fprintf(stderr, "hipemu: dpp_ctrl 0x%x not modelled\n", ctrl); std::abort(); }
  const int got = emu_lane_read(src, from < 0 ? l : from);
  if (!((row_mask >> row) & 1) || !((bank_mask >> ((l >> 2) & 3)) & 1)) return old;
  if (from < 0) return bound_ctrl ? 0 : old;
  return got;
}
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned m, unsigned add) { const unsigned l = hipemu::lane_(); return add + (unsigned)__builtin_popcount(m & (l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1u))); }
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned m, unsigned add) { const unsigned l = hipemu::lane_(); return add + (l > 32 ? (unsigned)__builtin_popcount(m & ((1u << (l - 32)) - 1u)) : 0u); }
inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool) {
  for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
  return c;
}
inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (sh & 3u))); }
inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31u)); }
inline unsigned __builtin_amdgcn_perm(unsigned a, unsigned b, unsigned sel) {
  const unsigned long long src = (((unsigned long long)a) << 32) | b; unsigned r = 0;
  for (int i = 0; i < 4; i++) { const unsigned s = (sel >> (8 * i)) & 0xFFu; unsigned byte = 0; if (s < 8) byte = (unsigned)(src >> (8 * s)) & 0xFFu; else if (s == 0x0C) byte = 0; else if (s >= 0x0D) byte = 0xFF; r |= byte << (8 * i); }
  return r;
}
inline unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned off, unsigned w) { return w >= 32 ? v >> off : (v >> off) & ((1u << w) - 1u); }
inline void __builtin_amdgcn_wave_barrier() { (void)hipemu::wave_sync(0); }  // lanes run one after another here: a real exchange point
inline void __builtin_amdgcn_s_sleep(int) {}
inline unsigned long long __builtin_amdgcn_s_memtime() { return 0; }
inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
inline unsigned __builtin_rotateright32(unsigned x, unsigned n) { n &= 31u; return n ? (x >> n) | (x << (32 - n)) : x; }
inline unsigned __builtin_rotateleft32(unsigned x, unsigned n) { n &= 31u; return n ? (x << n) | (x >> (32 - n)) : x; }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
template <class T> inline T __hip_atomic_load(const T *p, int, int) { return *p; }
template <class T> inline void __hip_atomic_store(T *p, T v, int, int) { *p = v; }
inline void __threadfence() {}
inline void __threadfence_block() {}
template <class T> inline T min(T a, T b) { return a < b ? a : b; }
template <class T> inline T max(T a, T b) { return a > b ? a : b; }
inline unsigned min(unsigned a, int b) { return a < (unsigned)b ? a : (unsigned)b; }
inline unsigned max(unsigned a, int b) { return a > (unsigned)b ? a : (unsigned)b; }
inline unsigned min(int a, unsigned b) { return (unsigned)a < b ? (unsigned)a : b; }
inline unsigned max(int a, unsigned b) { return (unsigned)a > b ? (unsigned)a : b; }
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
inline uint32_t atomicOr(uint32_t *p, uint32_t v) { uint32_t o = *p; *p |= v; return o; }
inline int atomicAdd(int *p, int v) { int o = *p; *p += v; return o; }
inline uint32_t atomicMax(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
inline uint32_t atomicMin(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v < o) *p = v; return o; }
inline int atomicMax(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }
inline uint32_t atomicCAS(uint32_t *p, uint32_t c, uint32_t v) { uint32_t o = *p; if (o == c) *p = v; return o; }
inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long c, unsigned long long v) { unsigned long long o = *p; if (o == c) *p = v; return o; }
inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p |= v; return o; }
inline uint32_t atomicExch(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = v; return o; }

// kernel<<<grid, block, shmem, stream>>>(args) is rewritten to this by tools/hipemu/build.py
template <class F> inline void emu_run_blocks(dim3 grid, dim3 block, F &&body) {
  using namespace hipemu;
  Sched sch;
  Sched *outer = g;
  g = &sch;
  const unsigned nt = block.x * block.y * block.z;
  sch.fibers.resize(nt);
  sch.waves.resize((nt + 63) / 64);
  // fiber stacks: one lazily committed mapping per host thread, reused by every launch
  static thread_local char *stacks = nullptr;
  if (!stacks) {
    stacks = (char *)mmap(nullptr, (size_t)1024 * STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == (char *)MAP_FAILED) { std::perror("hipemu: mmap"); std::abort(); }
  }
  if (nt > 1024) { std::fprintf(stderr, "hipemu: workgroup of %u threads\n", nt); std::abort(); }
  struct Thunk { F *f; } th{&body};
  sch.entry = [](void *p) { (*static_cast<Thunk *>(p)->f)(); };
  sch.entry_arg = &th;
  gridDim = grid; blockDim = block;
  for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
    for (unsigned t = 0; t < nt; t++) {
      Fiber &f = sch.fibers[t];
      f.state = RUN; f.tid = t;
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = stacks + (size_t)t * STACK;
      f.ctx.uc_stack.ss_size = STACK;
      f.ctx.uc_link = &sch.main;
      makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    unsigned done = 0;
    while (done < nt) {
      bool progressed = false;
      for (unsigned wv = 0; wv < sch.waves.size(); wv++) {
        const unsigned t0 = wv * 64, t1 = t0 + 64 < nt ? t0 + 64 : nt;
        for (;;) {
          bool ran = false;
          for (unsigned t = t0; t < t1; t++) {
            Fiber &f = sch.fibers[t];
            if (f.state != RUN) continue;
            sch.cur = &f;
            blockIdx = dim3(bx, by, bz);
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            swapcontext(&sch.main, &f.ctx);
            if (f.state == DONE) done++;
            ran = true; progressed = true;
          }
          if (ran) continue;
          // nobody in this wave can run: the lanes parked at a wave operation are its active lanes
          Wave &w = sch.waves[wv];
          unsigned long long m = 0;
          for (unsigned t = t0; t < t1; t++) if (sch.fibers[t].state == WAIT_WAVE) m |= 1ull << (t - t0);
          if (!m) break;
          w.mask = m;
          for (unsigned t = t0; t < t1; t++) if ((m >> (t - t0)) & 1ull) { w.snap[t - t0] = w.slot[t - t0]; sch.fibers[t].state = RUN; }
          progressed = true;
        }
      }
      if (done == nt) break;
      // every live fiber is parked at the workgroup barrier
      bool any_block = false;
      for (unsigned t = 0; t < nt; t++) if (sch.fibers[t].state == WAIT_BLOCK) { sch.fibers[t].state = RUN; any_block = true; }
      if (!any_block && !progressed) { std::fprintf(stderr, "hipemu: workgroup deadlock\n"); std::abort(); }
    }
  }
  g = outer;
}
template <class K, class... A> inline void emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
  emu_run_blocks(grid, block, [&]() { kernel(args...); });
}
