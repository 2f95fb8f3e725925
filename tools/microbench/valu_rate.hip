// tools/microbench/valu_rate.hip — cycles per wave64 instruction per SIMD on gfx950 for the integer / byte VALU ops and the
// LDS reads the tile parsers are made of.  Measurement aid only (not part of libtfgpu.so).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Every kernel runs 8 waves per SIMD (256 CUs x 4 SIMDs x 8) with 8 independent chains per lane, N iterations unrolled.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

constexpr int ITERS = 2048;

// eight chains: each asm statement is one instruction on its own accumulator
#define OP8(STR) \
  asm volatile(STR : "+v"(a0) : "v"(b0), "v"(c0) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41"); asm volatile(STR : "+v"(a1) : "v"(b0), "v"(c0) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41"); \
  asm volatile(STR : "+v"(a2) : "v"(b0), "v"(c0) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41"); asm volatile(STR : "+v"(a3) : "v"(b0), "v"(c0) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41"); \
  asm volatile(STR : "+v"(a4) : "v"(b0), "v"(c0) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41"); asm volatile(STR : "+v"(a5) : "v"(b0), "v"(c0) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41"); \
  asm volatile(STR : "+v"(a6) : "v"(b0), "v"(c0) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41"); asm volatile(STR : "+v"(a7) : "v"(b0), "v"(c0) : "vcc", "s20", "s21", "s22", "s23", "s24", "v40", "v41");

#define KERNEL32(NAME, STR) \
  __global__ void __launch_bounds__(512) NAME(uint32_t *out, uint32_t seed) { \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15; \
    uint32_t b0 = seed | 1u, c0 = seed * 7u + 3u; \
    for (int i = 0; i < ITERS / 4; i++) { OP8(STR) OP8(STR) OP8(STR) OP8(STR) } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
  }

KERNEL32(k_add, "v_add_u32 %0, %0, %1")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_or, "v_or_b32 %0, %0, %1")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_not, "v_not_b32 %0, %0")
KERNEL32(k_lshr, "v_lshrrev_b32 %0, 3, %0")
KERNEL32(k_ashr, "v_ashrrev_i32 %0, 3, %0")
KERNEL32(k_lshlv, "v_lshlrev_b32 %0, %1, %0")
KERNEL32(k_max_u32, "v_max_u32 %0, %0, %1")
KERNEL32(k_min_i32, "v_min_i32 %0, %0, %1")
KERNEL32(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL32(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
KERNEL32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL32(k_cndmask_s, "v_cndmask_b32 %0, %0, %1, s[22:23]")
KERNEL32(k_cmp_eq, "v_cmp_eq_u32 s[20:21], %0, %1")
KERNEL32(k_addc, "v_addc_co_u32 %0, s[20:21], %0, %1, s[22:23]")
KERNEL32(k_subrev, "v_subrev_u32 %0, %0, %1")
KERNEL32(k_add_i, "v_add_u32 %0, 7, %0")
KERNEL32(k_add_lit, "v_add_u32 %0, 0x76767676, %0")
KERNEL32(k_and_lit, "v_and_b32 %0, 0x80808080, %0")
KERNEL32(k_xor_lit, "v_xor_b32 %0, 0x30303030, %0")
KERNEL32(k_add_s, "v_add_u32 %0, s24, %0")
KERNEL32(k_bfe_i, "v_bfe_i32 %0, %0, 3, 8")
KERNEL32(k_pk_add_f32, "v_pk_add_f32 v[40:41], v[40:41], v[40:41]")
KERNEL32(k_pk_fma_f32, "v_pk_fma_f32 v[40:41], v[40:41], v[40:41], v[40:41]")
KERNEL32(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_xad, "v_xad_u32 %0, %0, %1, %2")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 3, 8")
KERNEL32(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, %2")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
KERNEL32(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL32(k_sad_u8, "v_sad_u8 %0, %0, %1, %2")
KERNEL32(k_msad_u8, "v_msad_u8 %0, %0, %1, %2")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_cvt_ubyte, "v_cvt_f32_ubyte1 %0, %0")
KERNEL32(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
KERNEL32(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL32(k_med3, "v_med3_u32 %0, %0, %1, %2")
KERNEL32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL32(k_ffbl, "v_ffbl_b32 %0, %0")
KERNEL32(k_ffbh, "v_ffbh_u32 %0, %0")
KERNEL32(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %0, %1")
KERNEL32(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL32(k_pk_mul_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL32(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL32(k_pk_lshl_u16, "v_pk_lshlrev_b16 %0, %1, %0")
KERNEL32(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL32(k_mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_cmp_sgpr, "v_cmp_lt_u32 s[20:21], %0, %1")
KERNEL32(k_readlane, "v_readfirstlane_b32 s20, %0")
KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL32(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL32(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL32(k_mad_u64, "v_mad_u64_u32 v[40:41], vcc, %0, %1, v[40:41]")
KERNEL32(k_lshl64, "v_lshlrev_b64 v[40:41], %0, v[40:41]")
KERNEL32(k_addco, "v_add_co_u32 %0, vcc, %0, %1")

// LDS reads: MODE fixed at compile time, 8 independent reads in flight per lane
template <int MODE> __global__ void __launch_bounds__(512) k_lds(uint32_t *out, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[32768 + 64];
  for (int i = threadIdx.x; i < (32768 + 64) / 4; i += 512) reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u + seed;
  __syncthreads();
  uint32_t a = threadIdx.x * 97u + seed, acc = 0;
  for (int i = 0; i < ITERS / 8; i++) {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t idx = (a + (uint32_t)j * 1931u) & 32767u;
      if (MODE == 0) v[j] = *reinterpret_cast<const uint32_t *>(lds + (idx & ~3u));                 // ds_read_b32 random aligned
      else if (MODE == 1) v[j] = *reinterpret_cast<const uint16_t *>(lds + (idx & ~1u));            // ds_read_u16 random
      else if (MODE == 2) v[j] = lds[idx];                                                          // ds_read_u8 random
      else if (MODE == 3) { uint2 t = *reinterpret_cast<const uint2 *>(lds + (idx & ~7u)); v[j] = t.x ^ t.y; }  // ds_read_b64 random aligned
      else if (MODE == 4) { uint32_t t; __builtin_memcpy(&t, lds + idx, 4); v[j] = t; }             // ds_read_b32 at a byte address
      else if (MODE == 5) { uint64_t t; __builtin_memcpy(&t, lds + idx, 8); v[j] = (uint32_t)t ^ (uint32_t)(t >> 32); }  // ds_read_b64 at a byte address
      else if (MODE == 6) v[j] = *reinterpret_cast<const uint32_t *>(lds + (((threadIdx.x & 63) * 4 + j * 256 + i) & 32764u));   // ds_read_b32 conflict-free
      else { uint4 t = *reinterpret_cast<const uint4 *>(lds + (idx & ~15u)); v[j] = t.x ^ t.y ^ t.z ^ t.w; }  // ds_read_b128 random
    }
#pragma unroll
    for (int j = 0; j < 8; j++) acc += v[j];
    a += acc & 0xFFu;  // next addresses depend on this round: no hoisting, one dependency per 8 reads
    a += 7919u;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// the tile parser's access shapes: lane = line j, field f of a line-major u16 index with NF entries per line
template <int MODE> __global__ void __launch_bounds__(512) k_lds_pat(uint32_t *out, uint32_t seed, uint32_t NF) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u + seed;
  __syncthreads();
  const uint32_t j = threadIdx.x & 63;
  uint32_t acc = 0, f = seed & 7u;
  for (int i = 0; i < ITERS / 8; i++) {
    uint32_t v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t k = j * NF + ((f + q * 13u) % NF);
      if (MODE == 0) v[q] = *reinterpret_cast<const uint16_t *>(lds + 49216 + ((2u * k) & 16382u));              // u16 index entry
      else if (MODE == 1) v[q] = *reinterpret_cast<const uint32_t *>(lds + 32768 + ((4u * k) & 32764u));         // the same as u32 entries
      else if (MODE == 2) v[q] = *reinterpret_cast<const uint16_t *>(lds + ((800u * j + 2u * ((f + q * 13u) % 400u)) & 65534u));  // u16 at a line's stride of 800 B
      else if (MODE == 3) v[q] = *reinterpret_cast<const uint32_t *>(lds + ((800u * j + 4u * ((f + q * 13u) % 200u)) & 65532u));  // u32 at a line's stride of 800 B
      else if (MODE == 4) v[q] = lds[(800u * j + ((f + q * 13u) % 800u)) & 65535u];                                  // u8 at a line's stride
      else v[q] = *reinterpret_cast<const uint32_t *>(lds + ((808u * j + 4u * ((f + q * 13u) % 200u)) & 65532u));      // u32 at a stride of 808 B
    }
#pragma unroll
    for (int q = 0; q < 8; q++) acc += v[q];
    f = (f + (acc & 3u) + 1u) % NF;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_lds_unaligned_check(uint32_t *out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[64];
  if (threadIdx.x < 64) lds[threadIdx.x] = (uint8_t)threadIdx.x;
  __syncthreads();
  const uint32_t addr = threadIdx.x & 15u;
  uint32_t v; uint64_t w;
  __builtin_memcpy(&v, lds + addr, 4);
  __builtin_memcpy(&w, lds + addr, 8);
  out[threadIdx.x * 3] = v; out[threadIdx.x * 3 + 1] = (uint32_t)w; out[threadIdx.x * 3 + 2] = (uint32_t)(w >> 32);
}

template <class F> static double time_ms(F &&launch, int reps = 5) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < reps; r++) {
    CHECK(hipEventRecord(e0, 0)); launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate / 1e6;
  std::printf("device %s, %d CUs, clockRate %.2f GHz\n", prop.gcnArchName, cus, ghz);
  uint32_t *out; CHECK(hipMalloc(&out, (size_t)cus * 4 * 512 * 4 * 4));
  const int blocks = cus * 4;  // 4 x 512 threads per CU = 32 waves per CU = 8 per SIMD
  struct K { const char *name; void (*fn)(uint32_t *, uint32_t); };
  const K ks[] = {{"v_add_u32", k_add}, {"v_add_u32 inline", k_add_i}, {"v_add_u32 literal", k_add_lit}, {"v_and_b32 literal", k_and_lit}, {"v_xor_b32 literal", k_xor_lit}, {"v_add_u32 sgpr", k_add_s},
                  {"v_or_b32", k_or}, {"v_mov_b32", k_mov}, {"v_not_b32", k_not}, {"v_lshrrev_b32", k_lshr}, {"v_ashrrev_i32", k_ashr}, {"v_lshlrev_b32 vreg", k_lshlv}, {"v_max_u32", k_max_u32}, {"v_min_i32", k_min_i32},
                  {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_rcp_f32", k_rcp_f32}, {"v_cndmask_b32 sgpr", k_cndmask_s}, {"v_cmp_eq_u32", k_cmp_eq},
                  {"v_addc_co_u32", k_addc}, {"v_subrev_u32", k_subrev}, {"v_bfe_i32", k_bfe_i}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_sub_u32", k_sub}, {"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_lshlrev_b32", k_lshl}, {"v_lshl_add_u32", k_lshl_add},
                  {"v_lshl_or_b32", k_lshl_or}, {"v_add3_u32", k_add3}, {"v_or3_b32", k_or3}, {"v_and_or_b32", k_and_or}, {"v_xad_u32", k_xad}, {"v_bfe_u32", k_bfe}, {"v_bfi_b32", k_bfi},
                  {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_alignbyte}, {"v_alignbit_b32", k_alignbit}, {"v_mul_u32_u24", k_mul24},
                  {"v_mad_u32_u24", k_mad24}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_dot4_u32_u8", k_dot4}, {"v_sad_u8", k_sad_u8},
                  {"v_msad_u8", k_msad_u8}, {"v_fma_f32", k_fma_f32}, {"v_cvt_f32_ubyte1", k_cvt_ubyte}, {"v_cvt_u32_f32", k_cvt_u32_f32}, {"v_min_u32", k_min_u32},
                  {"v_med3_u32", k_med3}, {"v_bcnt_u32_b32", k_bcnt}, {"v_ffbl_b32", k_ffbl}, {"v_ffbh_u32", k_ffbh}, {"v_mbcnt_lo", k_mbcnt},
                  {"v_pk_add_u16", k_pk_add_u16}, {"v_pk_mul_lo_u16", k_pk_mul_u16}, {"v_pk_mad_u16", k_pk_mad_u16}, {"v_pk_lshlrev_b16", k_pk_lshl_u16},
                  {"v_add_u32_sdwa", k_add_sdwa}, {"v_mov_b32_dpp", k_mov_dpp}, {"v_add_u32_dpp", k_add_dpp}, {"v_cmp_lt_u32 vcc", k_cmp},
                  {"v_cmp_lt_u32 sgpr", k_cmp_sgpr}, {"v_readfirstlane", k_readlane}, {"v_mad_u64_u32", k_mad_u64}, {"v_lshlrev_b64", k_lshl64}, {"v_add_co_u32", k_addco}};
  std::printf("%-22s %10s %14s\n", "instruction", "ms", "cyc/inst/SIMD");
  for (const K &k : ks) {
    const double ms = time_ms([&] { hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(512), 0, 0, out, 12345u); });
    // per SIMD: 8 waves x ITERS x 8 instructions
    const double inst = 8.0 * ITERS * 8.0;
    std::printf("%-22s %10.4f %14.2f\n", k.name, ms, ms * 1e-3 * ghz * 1e9 / inst);
  }
  const char *ldsn[] = {"ds_read_b32 random", "ds_read_u16 random", "ds_read_u8 random", "ds_read_b64 random", "ds_read_b32 byte addr", "ds_read_b64 byte addr", "ds_read_b32 linear", "ds_read_b128 random"};
  void (*lk[])(uint32_t *, uint32_t) = {k_lds<0>, k_lds<1>, k_lds<2>, k_lds<3>, k_lds<4>, k_lds<5>, k_lds<6>, k_lds<7>};
  for (int mode = 0; mode < 8; mode++) {
    const double ms = time_ms([&] { hipLaunchKernelGGL(lk[mode], dim3(cus * 4), dim3(512), 0, 0, out, 777u); });
    const double inst = 32.0 * ITERS;  // per CU: 4 blocks x 8 waves
    std::printf("%-22s %10.4f %14.2f (cyc per wave-instruction per CU; each read also costs ~2 address VALU on its SIMD)\n", ldsn[mode], ms, ms * 1e-3 * ghz * 1e9 / inst);
  }
  {
    const char *pn[] = {"u16 index NF=105", "u32 index NF=105", "u16 line stride 800", "u32 line stride 800", "u8 line stride 800", "u32 line stride 808"};
    void (*pk[])(uint32_t *, uint32_t, uint32_t) = {k_lds_pat<0>, k_lds_pat<1>, k_lds_pat<2>, k_lds_pat<3>, k_lds_pat<4>, k_lds_pat<5>};
    for (int mode = 0; mode < 6; mode++) for (uint32_t NF : {105u, 104u, 128u}) {
      if (mode >= 2 && NF != 105u) continue;
      const double ms = time_ms([&] { hipLaunchKernelGGL(pk[mode], dim3(cus * 2), dim3(512), 0, 0, out, 777u, NF); });
      std::printf("%-22s NF=%3u %10.4f %14.2f (cyc per wave-instruction per CU, 2 blocks of 64 KiB)\n", pn[mode], NF, ms, ms * 1e-3 * ghz * 1e9 / (16.0 * ITERS));
    }
  }
  {
    std::vector<uint32_t> h(64 * 3);
    hipLaunchKernelGGL(k_lds_unaligned_check, dim3(1), dim3(64), 0, 0, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    bool ok = true;
    for (int t = 0; t < 16; t++) {
      const uint32_t a = t;
      const uint32_t want = a | ((a + 1) << 8) | ((a + 2) << 16) | ((a + 3) << 24);
      const uint32_t want_hi = (a + 4) | ((a + 5) << 8) | ((a + 6) << 16) | ((a + 7) << 24);
      if (h[t * 3] != want || h[t * 3 + 1] != want || h[t * 3 + 2] != want_hi) { ok = false; std::printf("byte addr %d: b32 %08x b64 %08x %08x (want %08x %08x)\n", t, h[t * 3], h[t * 3 + 1], h[t * 3 + 2], want, want_hi); }
    }
    std::printf("ds_read_b32 / ds_read_b64 at byte addresses (compiler-emitted for align-1 loads): %s\n", ok ? "CORRECT" : "WRONG");
  }
  return 0;
}
