// host read-back latency on one stream: kernel -> 4-byte D2H into pinned memory -> wait, three ways of waiting
//   hipcc --offload-arch=gfx950 -O3 -o sync_latency sync_latency.hip && ./sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void work(uint32_t *p, uint32_t v) { *p = v; }
__global__ void flag(volatile uint32_t *f, uint32_t v) { *f = v; __threadfence_system(); }
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint32_t *d; CK(hipMalloc(&d, 64));
  uint32_t *h; CK(hipHostMalloc(&h, 64, hipHostMallocDefault));
  volatile uint32_t *fh; CK(hipHostMalloc((void **)&fh, 64, hipHostMallocMapped)); *fh = 0;
  uint32_t *fd; CK(hipHostGetDevicePointer((void **)&fd, (void *)fh, 0));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const int N = 2000;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (int mode = 0; mode < 4; mode++) {
    for (int w = 0; w < 2; w++) {  // warm, then timed
      auto t0 = now();
      for (int i = 1; i <= N; i++) {
        work<<<1, 64, 0, st>>>(d, (uint32_t)i);
        CK(hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, st));
        if (mode == 0) CK(hipStreamSynchronize(st));
        else if (mode == 1) { CK(hipEventRecord(ev, st)); while (hipEventQuery(ev) == hipErrorNotReady) _mm_pause(); }
        else if (mode == 2) { flag<<<1, 1, 0, st>>>(fd, (uint32_t)(i + w * N + mode * 100000)); while (*fh != (uint32_t)(i + w * N + mode * 100000)) _mm_pause(); }
        else { while (hipStreamQuery(st) == hipErrorNotReady) _mm_pause(); }
        if (h[0] != (uint32_t)i) { printf("mode %d: stale read-back at %d\n", mode, i); return 1; }
      }
      if (w) printf("mode %d (%s): %.2f us per kernel + read-back + wait\n", mode, mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "event query spin" : mode == 2 ? "flag kernel + host spin on mapped memory" : "hipStreamQuery spin", us(t0, now()) / N);
    }
  }
  return 0;
}
