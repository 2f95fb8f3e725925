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

// tf_csv.hip's line index (wave scans there): row_start[0] = 0, row_start[k] = one past the k-th '\n'; returns their number
uint32_t newline_starts(const uint8_t *data, uint64_t len, Buf *out) {
  uint32_t n = 0;
  for (uint64_t i = 0; i < len; i++) n += data[i] == '\n';
  *out = dalloc_zero((size_t)(n + 2) * 4);
  uint32_t *rs = ptr<uint32_t>(*out), k = 0;
  for (uint64_t i = 0; i < len; i++) if (data[i] == '\n') rs[++k] = (uint32_t)(i + 1);
  return n;
}

// tf_transform.hip's row compaction (scan + gather kernels there): keep flags (uint32 0/1, n + 1 slots) → the kept rows
std::unique_ptr<tfgpu_dbatch> compact_rows(const tfgpu_dbatch &in, Buf keep) {
  const int64_t n = in.nrows;
  const uint32_t *kf = ptr<uint32_t>(keep);
  std::vector<int64_t> sel;
  for (int64_t r = 0; r < n; r++) if (kf[r]) sel.push_back(r);
  const int64_t m = (int64_t)sel.size();
  if (m == n) return std::make_unique<tfgpu_dbatch>(in);
  auto out = std::make_unique<tfgpu_dbatch>();
  out->nrows = m; out->ns = in.ns; out->table = in.table; out->schema = in.schema; out->key_names = in.key_names;
  auto bits = [&](const Buf &src) {
    Buf d = dalloc_zero((size_t)(m + 7) / 8 + 8);
    const uint8_t *s = ptr<uint8_t>(src); uint8_t *o = ptr<uint8_t>(d);
    for (int64_t k = 0; k < m; k++) if ((s[sel[(size_t)k] >> 3] >> (sel[(size_t)k] & 7)) & 1) o[k >> 3] |= (uint8_t)(1u << (k & 7));
    return d;
  };
  auto fixed = [&](const Buf &src, size_t w) {
    Buf d = dalloc((size_t)std::max<int64_t>(m, 1) * w);
    for (int64_t k = 0; k < m; k++) std::memcpy((char *)d->p + (size_t)k * w, (const char *)src->p + (size_t)sel[(size_t)k] * w, w);
    return d;
  };
  auto column = [&](const DColumn &c) {
    DColumn o;
    o.name = c.name; o.dtype = c.dtype; o.repr = c.repr;
    if (repr_is_var(c.repr)) {
      const uint32_t *off = ptr<uint32_t>(c.offsets); const uint8_t *dat = ptr<uint8_t>(c.payload());
      o.offsets = dalloc((size_t)(m + 1) * 4 + 16);
      uint32_t *oo = ptr<uint32_t>(o.offsets), tot = 0;
      for (int64_t k = 0; k < m; k++) { oo[k] = tot; tot += off[sel[(size_t)k] + 1] - off[sel[(size_t)k]]; }
      oo[m] = tot; o.data_len = tot;
      o.data = dalloc((size_t)tot + 8);
      for (int64_t k = 0; k < m; k++) std::memcpy(ptr<uint8_t>(o.data) + oo[k], dat + off[sel[(size_t)k]], oo[k + 1] - oo[k]);
    } else {
      o.values = fixed(c.values, repr_width(c.repr));
      if (c.nanos) o.nanos = fixed(c.nanos, 4);
    }
    if (c.validity) o.validity = bits(c.validity);
    return o;
  };
  for (auto &c : in.cols) out->cols.push_back(column(c));
  for (auto &c : in.old_keys) out->old_keys.push_back(column(c));
  if (in.old_present) out->old_present = bits(in.old_present);
  if (in.kind) out->kind = fixed(in.kind, 1);
  if (in.part_id) out->part_id = fixed(in.part_id, 4);
  out->src_row = dalloc((size_t)std::max<int64_t>(m, 1) * 4);
  for (int64_t k = 0; k < m; k++) ptr<int32_t>(out->src_row)[k] = in.src_row ? ptr<int32_t>(in.src_row)[sel[(size_t)k]] : (int32_t)sel[(size_t)k];
  return out;
}
}  // namespace tf
