// tf_wave.hpp — wavefront-wide primitives on DPP (gfx950: rows of 16 lanes, row_bcast across rows) and ballots:
// no LDS round trip, no s_waitcnt, six VALU instructions for a 64-lane prefix sum.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tf {

// inclusive prefix sum over the 64 lanes of the wave (all lanes must call it)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 → rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 → rows 2, 3
  return (uint32_t)x;
}
// does the predicate hold on every active lane?  (ballot of a bool: an s_andn2 / s_cmp, no VALU; __all goes through an int)
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0; }
// number of set bits of a 64-bit lane mask below this lane
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

}  // namespace tf
