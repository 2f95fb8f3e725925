// tf_textview.hpp — device side of late-materialised text cells (TextView, tf_common.hpp): how a cell's source
// position is encoded, and the two cell forms that are not a plain byte range of the source text.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tf {

// fstart word of a cell: offset into the source text; bit 31 = doubled quotes to collapse; low 31 bits all ones = the
// cell is the column's DefaultValue (no source bytes)
__device__ __forceinline__ bool cell_plain(uint32_t fsv) { return !(fsv & 0x80000000u) && fsv != 0x7FFFFFFFu; }

// Called by all 64 lanes of a wave, one candidate cell per lane (n = 0: nothing to do for this lane; fsv / o0 / n of a
// non-plain cell otherwise).  DefaultValue of a double: json.Number "0".  ""-collapse (swapToSingleQuotes,
// reader.go:307-320): every flagged cell is moved by the whole wave, 64 source bytes per step — a quote is dropped iff an
// odd number of quotes runs directly before it (non-overlapping pairs, left to right), and the survivors are packed
// through a ballot prefix count.
__device__ __forceinline__ void text_copy_special_wave(const uint8_t *__restrict__ src, uint32_t quote, uint8_t *__restrict__ out, bool is_jsonnum,
                                                       uint32_t fsv, uint32_t o0, uint32_t n, int lane) {
  bool special = false;
  if (n && !cell_plain(fsv)) {
    if ((fsv & 0x7FFFFFFFu) == 0x7FFFFFFFu) { if (is_jsonnum) out[o0] = '0'; }
    else special = true;
  }
  uint64_t todo = __ballot(special);
  const uint64_t below_me = (1ull << lane) - 1;
  while (todo) {
    const int l = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const uint64_t s0 = __shfl(fsv, l, 64) & 0x7FFFFFFFu;
    const uint32_t on = __shfl(n, l, 64);
    uint8_t *dst = out + __shfl(o0, l, 64);
    uint32_t produced = 0;
    bool carry = false;  // the previous step ended on a kept quote still waiting for its pair
    for (uint64_t sp = s0; produced < on; sp += 64) {
      const uint32_t ch = src[sp + lane];
      const bool isq = ch == quote;
      const uint64_t Q = __ballot(isq);
      const uint64_t nonq_below = ~Q & below_me;
      const uint32_t before = nonq_below ? (uint32_t)lane - (63u - (uint32_t)__clzll((long long)nonq_below)) - 1u : (uint32_t)lane + (carry ? 1u : 0u);
      const bool dropped = isq && (before & 1u);
      const uint64_t K = __ballot(!dropped);
      const uint32_t idx = produced + (uint32_t)__popcll(K & below_me);
      if (!dropped && idx < on) dst[idx] = (uint8_t)ch;
      carry = __shfl((int)(isq && !(before & 1u)), 63, 64) != 0;
      produced += (uint32_t)__popcll(K);
    }
  }
}

}  // namespace tf
