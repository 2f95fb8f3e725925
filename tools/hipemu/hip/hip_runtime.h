// tools/hipemu — TEST INFRASTRUCTURE ONLY.  A stand-in for <hip/hip_runtime.h> that lets the lane-per-item kernels of
// libtfgpu (no LDS, no barriers, no wave intrinsics) run as plain C++ loops in the GPU-less build container, so their
// byte-level logic can be checked against the oracle BEFORE a GPU box is spent on them.  Never shipped, never loaded by
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
#define __constant__ static const
#define __launch_bounds__(...)
#define __shared__ static  /* one instance per kernel: workgroups run one after another here */
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
enum { hipSuccess = 0 };
inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
typedef struct emuStream *hipStream_t;
typedef struct emuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
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
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { std::memset(p, 0, sizeof *p); std::strcpy(p->gcnArchName, "gfx950:hipemu"); p->multiProcessorCount = 256; return hipSuccess; }

inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
inline long long __double_as_longlong(double d) { long long v; std::memcpy(&v, &d, 8); return v; }
inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
inline int __float_as_int(float f) { int v; std::memcpy(&v, &f, 4); return v; }
inline float __int_as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
inline int __mul24(int a, int b) { return ((a << 8) >> 8) * ((b << 8) >> 8); }
// wave-level intrinsics have no meaning in a lane-at-a-time loop: a kernel that reaches one is not emulable
[[noreturn]] inline void emu_no_wave_ops(const char *what) { std::fprintf(stderr, "hipemu: %s needs a wavefront; this kernel cannot run under tools/hipemu\n", what); std::abort(); }
// __any / __all choose between two per-lane-correct code paths in the emulable kernels (digits_u64: "the whole wave parses
// short numbers"), so a one-lane wave is a faithful model of them; kernels that really exchange data across lanes reach a
// ballot / shuffle / barrier below and abort.
inline int __any(int x) { return x != 0; }
inline int __all(int x) { return x != 0; }
inline unsigned long long __ballot(int) { emu_no_wave_ops("__ballot"); }
template <class T> inline T __shfl(T, int, int = 64) { emu_no_wave_ops("__shfl"); }
template <class T> inline T __shfl_up(T, unsigned, int = 64) { emu_no_wave_ops("__shfl_up"); }
template <class T> inline T __shfl_xor(T, int, int = 64) { emu_no_wave_ops("__shfl_xor"); }
inline void __syncthreads() { emu_no_wave_ops("__syncthreads"); }

inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
inline uint32_t atomicOr(uint32_t *p, uint32_t v) { uint32_t o = *p; *p |= v; return o; }

// kernel<<<grid, block, shmem, stream>>>(args) is rewritten to this by tools/hipemu/build.py
template <class K, class... A> inline void emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
  gridDim = grid; blockDim = block;
  for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
    blockIdx = dim3(bx, by, bz);
    for (unsigned tz = 0; tz < block.z; tz++) for (unsigned ty = 0; ty < block.y; ty++) for (unsigned tx = 0; tx < block.x; tx++) {
      threadIdx = dim3(tx, ty, tz);
      kernel(args...);
    }
  }
}
