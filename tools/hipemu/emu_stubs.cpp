// tools/hipemu/emu_stubs.cpp — TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h).  Host versions of the two primitives the
// emulated translation units call into (their real forms use wave scans / LDS), and loud stubs for the rest of the ABI.
#include "tf_common.hpp"

namespace tf {
void exclusive_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, bool with_total) {
  uint32_t s = 0;
  for (int64_t i = 0; i < n; i++) { const uint32_t v = in[i]; out[i] = s; s += v; }
  if (with_total) out[n] = s;
}
void exclusive_scan_u32_segments(uint32_t *inout, int64_t seg_len, int nseg, int64_t seg_stride) {
  for (int g = 0; g < nseg; g++) exclusive_scan_u32(inout + g * seg_stride, inout + g * seg_stride, seg_len, true);  // total at [seg_len], like tf_scan.hip
}
void materialize(const ::tfgpu_dbatch &, const std::vector<const DColumn *> *) {}
}  // namespace tf

