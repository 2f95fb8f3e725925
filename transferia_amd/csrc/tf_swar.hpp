// tf_swar.hpp — byte-class masks and decimal windows over text staged in LDS, shared by the tile parsers (tf_csv.hip,
// tf_json.hip): SWAR classification of 16-byte chunks, right-aligned digit windows summed with v_dot4.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace tf {

// four digit VALUES (most significant in byte 0) → their number: two v_dot4 and a mad
__device__ __forceinline__ uint32_t four_dot(uint32_t x) {
  return __umul24(__builtin_amdgcn_udot4(x, 0x0000010Au, 0u, false), 100u) + __builtin_amdgcn_udot4(x, 0x010A0000u, 0u, false);
}

// ASCII byte classes of 16 bytes by SWAR.  Per word: lo7 = the low seven bits of every byte, hi1 = all ones except
// bit 7 of the bytes below 0x80.  For an ASCII pattern byte P, (lo7 ^ P4) + 0x7F7F7F7F carries into bit 7 exactly
// where the seven bits differ (one v_xad_u32), and OR-ing hi1 leaves 0x7F in the bytes equal to P and 0xFF elsewhere.
// v_dot4_u32_u8 with weights 1,2,4,8 and the accumulator preset to -(0x7F * 15) then sums 128 * (the weights of the
// bytes that differ): the complement of the match nibble, already shifted by 7.
struct Chunk16 { uint32_t lo7[4], hi1[4]; };
__device__ __forceinline__ Chunk16 chunk16(uint4 v) {
  Chunk16 c;
  c.lo7[0] = v.x & 0x7F7F7F7Fu; c.lo7[1] = v.y & 0x7F7F7F7Fu; c.lo7[2] = v.z & 0x7F7F7F7Fu; c.lo7[3] = v.w & 0x7F7F7F7Fu;
  c.hi1[0] = v.x | 0x7F7F7F7Fu; c.hi1[1] = v.y | 0x7F7F7F7Fu; c.hi1[2] = v.z | 0x7F7F7F7Fu; c.hi1[3] = v.w | 0x7F7F7F7Fu;
  return c;
}
__device__ __forceinline__ uint32_t ne_bytes(const Chunk16 &c, int w, uint32_t pat4) { return ((c.lo7[w] ^ pat4) + 0x7F7F7F7Fu) | c.hi1[w]; }  // 0x7F = equal, 0xFF = not
// 16-bit mask of the bytes equal to the ASCII byte replicated in pat4
__device__ __forceinline__ uint32_t class16(const Chunk16 &c, uint32_t pat4) {
  const uint32_t W = 0x08040201u, B = 0u - 0x7Fu * 15u;
  const uint32_t d0 = __builtin_amdgcn_udot4(ne_bytes(c, 0, pat4), W, B, false), d1 = __builtin_amdgcn_udot4(ne_bytes(c, 1, pat4), W, B, false);
  const uint32_t d2 = __builtin_amdgcn_udot4(ne_bytes(c, 2, pat4), W, B, false), d3 = __builtin_amdgcn_udot4(ne_bytes(c, 3, pat4), W, B, false);
  return ~((d0 >> 7) | (d1 >> 3) | (d2 << 1) | (d3 << 5)) & 0xFFFFu;
}
// The same mask with the four v_dot4 chained in pairs through the accumulator: weights 1..8 for the first word of a
// pair and 16..128 for the second give 0x7F * 255 + 128 * (the 8-bit mask of the bytes that differ), so what is left to
// combine is one shift each way.
__device__ __forceinline__ uint32_t class16c(const Chunk16 &c, uint32_t pat4) {
  const uint32_t WL = 0x08040201u, WH = 0x80402010u, B = 0u - 0x7Fu * 255u;
  const uint32_t a = __builtin_amdgcn_udot4(ne_bytes(c, 1, pat4), WH, __builtin_amdgcn_udot4(ne_bytes(c, 0, pat4), WL, B, false), false);
  const uint32_t b = __builtin_amdgcn_udot4(ne_bytes(c, 3, pat4), WH, __builtin_amdgcn_udot4(ne_bytes(c, 2, pat4), WL, B, false), false);
  return ~((a >> 7) | (b << 1)) & 0xFFFFu;
}
// does any of the 16 bytes equal the ASCII byte replicated in pat4?
__device__ __forceinline__ bool any16(const Chunk16 &c, uint32_t pat4) {
  return ((ne_bytes(c, 0, pat4) & ne_bytes(c, 1, pat4) & ne_bytes(c, 2, pat4) & ne_bytes(c, 3, pat4)) & 0x80808080u) != 0x80808080u;
}


// The 8-byte window [end - 8, end) of the tile as two words (byte 0 of lo = sb[end - 8]): three aligned LDS words, two v_alignbyte.
__device__ __forceinline__ void window8(const uint8_t *sb, uint32_t end, uint32_t *lo, uint32_t *hi) {
  const int e8 = (int)end - 8;
  const uint32_t *w = reinterpret_cast<const uint32_t *>(sb + (e8 & ~3));
  const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
  *lo = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)e8); *hi = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)e8);
}
// Up to eight decimal digits that END the window (nd of them, 1..8): they sit right-aligned, so no shifting by the field
// length is needed — mask what precedes them, check, two v_dot4 per four.  Returns false on a non-digit.
template <bool NARROW = false>
__device__ __forceinline__ bool digits8_window(uint32_t wlo, uint32_t whi, uint32_t nd, uint32_t *out) {
  const uint32_t lo = wlo ^ 0x30303030u, hi = whi ^ 0x30303030u;
  const uint64_t keep = ~0ull << (8 * (8 - nd));  // nd >= 1
  const uint32_t tlo = lo & (uint32_t)keep, thi = hi & (uint32_t)(keep >> 32);
  if (NARROW) *out = __umul24(tlo >> 24, 10000u) + four_dot(thi);  // callers that reject nd > 5 anyway: the fifth digit is tlo's top byte
  else *out = __umul24(four_dot(tlo), 10000u) + four_dot(thi);     // < 10^8
  return (((tlo + 0x76767676u) | tlo | (thi + 0x76767676u) | thi) & 0x80808080u) == 0;
}
__device__ __forceinline__ bool digits8_end(const uint8_t *sb, uint32_t end, uint32_t nd, uint32_t *out) {
  uint32_t lo, hi;
  window8(sb, end, &lo, &hi);
  return digits8_window(lo, hi, nd, out);
}


}  // namespace tf
