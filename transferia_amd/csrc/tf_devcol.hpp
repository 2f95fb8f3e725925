// tf_devcol.hpp — device-side view of one column of a batch (shared by the
// transformer and serializer kernels).
#pragma once
#include "tf_common.hpp"

namespace tf {

// ---- device view of a column ------------------------------------------------
struct DCol {
  const void *values;
  const uint32_t *offsets;
  const uint8_t *data;
  const int32_t *nanos;
  const uint8_t *validity;
  int32_t repr;
  int32_t dtype;
};
static inline DCol dcol_of(const DColumn &c) {
  DCol d;
  d.values = c.values ? c.values->p : nullptr;
  d.offsets = ptr<uint32_t>(c.offsets);
  if (c.lazy()) throw Error(TFGPU_ERR_INVALID, "internal: text column " + c.name + " read before materialize()");
  d.data = ptr<uint8_t>(c.payload());
  d.nanos = ptr<int32_t>(c.nanos);
  d.validity = ptr<uint8_t>(c.validity);
  d.repr = c.repr; d.dtype = c.dtype;
  return d;
}
__device__ __forceinline__ bool is_valid(const DCol &c, int64_t r) { return !c.validity || ((c.validity[r >> 3] >> (r & 7)) & 1); }


}  // namespace tf
