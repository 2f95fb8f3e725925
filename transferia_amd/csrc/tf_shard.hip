// tf_shard.hip — one process, several GPUs: row-range shards of a device batch and their ordered concatenation.
//
// the path shards by rows with no data-path collective — a transformer sees one row at a time
// (transformation.go:131-135 runs one goroutine per table over the whole batch; nothing in filter / mask / cast looks at a
// neighbour row), so G devices each take a contiguous row range and the results are put back in range order.  One process per
// GPU does that with RANK-local batches (bench.py under torchrun); one process DRIVING several GPUs (one Go worker, G lanes,
// lane k on device k mod G — tf_runtime.hip) needs the split and the merge as operations on device batches:
//
//   tfgpu_dbatch_slice    rows [row0, row0 + n) of a batch as a batch of its own (same lane; row0 a multiple of 8 so that
//                         bitmaps are cut at byte boundaries)
//   tfgpu_dbatch_to_lane  a batch re-homed on another lane: a deep copy over xGMI / PCIe when the lane lives on another
//                         device (hipMemcpyAsync, hipMemcpyDefault — no peer mapping is assumed), shared buffers when not
//   tfgpu_shard_rows      G balanced slices cut at multiples of 64 rows, slice g on lane lanes[g]
//   tfgpu_dbatch_concat   the parts' rows in order on the calling lane; src_row of part g shifted by row_base[g], i.e. back
//                         to the row numbers of the batch that was sharded
//
// Everything here is copies plus three index kernels (offset rebasing, bitmap splicing at bit granularity, src_row shift): the
// HBM roofline of a copy, and off the timed path of the per-device work.
#include <algorithm>

#include "tf_common.hpp"

namespace tf {

__global__ void shard_rebase_offsets(const uint32_t *in, int64_t n1, uint32_t sub, uint32_t add, uint32_t *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n1) out[i] = in[i] - sub + add;
}
__global__ void shard_gather_u32(const uint32_t *const *arrays, const int64_t *idx, int64_t n, uint32_t *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = arrays[i][idx[i]];
}
// out bits [at, at + n) = in bits [0, n); other bits of the touched bytes are kept.  Launched part after part on one stream.
__global__ void shard_splice_bits(const uint8_t *in, int64_t n, int64_t at, uint8_t *out, int in_null_means_ones) {
  // one thread per OUTPUT byte of the range
  const int64_t b0 = at >> 3, b1 = (at + n + 7) >> 3;
  const int64_t ob = b0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ob >= b1) return;
  uint32_t v = out[ob], m = 0, bits = 0;
  for (int k = 0; k < 8; k++) {
    const int64_t j = ob * 8 + k - at;  // bit index in `in`
    if (j < 0 || j >= n) continue;
    m |= 1u << k;
    const uint32_t bit = in_null_means_ones ? 1u : (in[j >> 3] >> (j & 7)) & 1u;
    bits |= bit << k;
  }
  out[ob] = (uint8_t)((v & ~m) | bits);
}
__global__ void shard_shift_i32(const int32_t *in, int64_t n, int32_t add, int32_t *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + add;
}
__global__ void shard_iota_i32(int64_t n, int32_t add, int32_t *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)i + add;
}

static inline unsigned blocks_of(int64_t n) { return (unsigned)((n + 255) / 256); }
// any-to-any device copy on the calling lane's stream (the two ends may live on different GPUs)
static void xcopy(void *dst, const void *src, size_t n) { if (n) TF_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDefault, ctx().stream)); }

static int batch_device(const tfgpu_dbatch &b) {
  auto dev = [](const Buf &m) { return m ? m->device : -1; };
  for (auto &c : b.cols) for (const Buf *m : {&c.values, &c.offsets, &c.data, &c.validity}) if (*m) return dev(*m);
  for (const Buf *m : {&b.kind, &b.src_row, &b.part_id}) if (*m) return dev(*m);
  return -1;  // no buffers at all: lives anywhere
}

// ---- slice ------------------------------------------------------------------------------------------------------------
struct VarCut { uint32_t lo, hi; };
static DColumn slice_column(const DColumn &c, int64_t r0, int64_t n, const VarCut *cut) {
  DColumn d;
  d.name = c.name; d.dtype = c.dtype; d.repr = c.repr;
  hipStream_t st = ctx().stream;
  if (repr_is_var(c.repr)) {
    d.offsets = dalloc((size_t)(n + 1) * 4);
    shard_rebase_offsets<<<blocks_of(n + 1), 256, 0, st>>>(ptr<uint32_t>(c.offsets) + r0, n + 1, cut->lo, 0u, ptr<uint32_t>(d.offsets));
    d.data_len = cut->hi - cut->lo;
    d.data = dalloc(d.data_len);
    d2d(d.data->p, ptr<uint8_t>(c.payload()) + cut->lo, d.data_len);
  } else {
    const size_t w = (size_t)repr_width(c.repr);
    if (c.values) { d.values = dalloc((size_t)n * w); d2d(d.values->p, (const char *)c.values->p + (size_t)r0 * w, (size_t)n * w); }
    if (c.nanos) { d.nanos = dalloc((size_t)n * 4); d2d(d.nanos->p, ptr<int32_t>(c.nanos) + r0, (size_t)n * 4); }
  }
  if (c.validity) { d.validity = dalloc((size_t)(n + 7) / 8); d2d(d.validity->p, ptr<uint8_t>(c.validity) + (r0 >> 3), (size_t)(n + 7) / 8); }
  if (c.absent) { d.absent = dalloc((size_t)(n + 7) / 8); d2d(d.absent->p, ptr<uint8_t>(c.absent) + (r0 >> 3), (size_t)(n + 7) / 8); }  // rows keep their ColumnNames
  return d;
}

// rows [r0, r0 + n) of every cut (r0 multiples of 8), on the calling lane; ONE read-back for all var-width cut points
static std::vector<std::unique_ptr<tfgpu_dbatch>> slice_many(const tfgpu_dbatch &b, const std::vector<int64_t> &cuts) {
  if (b.col_order) throw Error(TFGPU_ERR_UNSUPPORTED, "the batch's rows carry their own ColumnNames order (a collapsed batch): it is not cut into shards");
  materialize(b);
  const size_t G = cuts.size() - 1;
  std::vector<const DColumn *> var;
  for (auto &c : b.cols) if (repr_is_var(c.repr)) var.push_back(&c);
  for (auto &c : b.old_keys) if (repr_is_var(c.repr)) var.push_back(&c);
  std::vector<uint32_t> at(var.size() * (G + 1), 0u);
  if (!var.empty()) {
    std::vector<const uint32_t *> arrays; std::vector<int64_t> idx;
    for (auto *c : var) for (size_t g = 0; g <= G; g++) { arrays.push_back(ptr<uint32_t>(c->offsets)); idx.push_back(cuts[g]); }
    Buf da = upload_small(arrays.data(), arrays.size() * sizeof(void *)), di = upload_small(idx.data(), idx.size() * 8), out = dalloc(arrays.size() * 4);
    shard_gather_u32<<<blocks_of((int64_t)arrays.size()), 256, 0, ctx().stream>>>(reinterpret_cast<const uint32_t *const *>(da->p), ptr<int64_t>(di), (int64_t)arrays.size(), ptr<uint32_t>(out));
    std::vector<uint32_t> h(arrays.size());
    d2h(h.data(), out->p, h.size() * 4);
    tf::sync();
    at = h;
  }
  std::vector<std::unique_ptr<tfgpu_dbatch>> res;
  for (size_t g = 0; g < G; g++) {
    const int64_t r0 = cuts[g], n = cuts[g + 1] - cuts[g];
    auto s = std::make_unique<tfgpu_dbatch>();
    s->nrows = n; s->schema = b.schema; s->key_names = b.key_names; s->ns = b.ns; s->table = b.table;
    size_t v = 0;
    auto one = [&](const DColumn &c) {
      VarCut cut{0, 0};
      if (repr_is_var(c.repr)) { cut.lo = at[v * (G + 1) + g]; cut.hi = at[v * (G + 1) + g + 1]; v++; }
      return slice_column(c, r0, n, &cut);
    };
    for (auto &c : b.cols) s->cols.push_back(one(c));
    for (auto &c : b.old_keys) s->old_keys.push_back(one(c));
    if (b.old_present) { s->old_present = dalloc((size_t)(n + 7) / 8); d2d(s->old_present->p, ptr<uint8_t>(b.old_present) + (r0 >> 3), (size_t)(n + 7) / 8); }
    if (b.kind) { s->kind = dalloc((size_t)n); d2d(s->kind->p, ptr<uint8_t>(b.kind) + r0, (size_t)n); }
    if (b.src_row) { s->src_row = dalloc((size_t)n * 4); d2d(s->src_row->p, ptr<int32_t>(b.src_row) + r0, (size_t)n * 4); }
    if (b.part_id) { s->part_id = dalloc((size_t)n * 4); d2d(s->part_id->p, ptr<uint32_t>(b.part_id) + r0, (size_t)n * 4); }
    res.push_back(std::move(s));
  }
  return res;
}

// ---- re-homing ----------------------------------------------------------------------------------------------------------
// `b` (made on the calling lane, or at least complete: the caller synchronised its lane) as a batch of lane `lane`
static std::unique_ptr<tfgpu_dbatch> to_lane(const tfgpu_dbatch &b, int lane) {
  if (b.col_order) throw Error(TFGPU_ERR_UNSUPPORTED, "the batch's rows carry their own ColumnNames order (a collapsed batch): it stays on its lane");
  materialize(b);
  auto r = std::make_unique<tfgpu_dbatch>();
  r->nrows = b.nrows; r->schema = b.schema; r->key_names = b.key_names; r->ns = b.ns; r->table = b.table;
  const int src_dev = batch_device(b);
  const bool share = src_dev < 0 || src_dev == lane_device(lane);
  LaneScope on(lane);  // allocations come from the target lane's cache, the copies run on its stream
  std::lock_guard<std::mutex> lk(ctx().mu);
  auto mv = [&](const Buf &m, size_t bytes) -> Buf {
    if (!m) return nullptr;
    if (share) return m;
    Buf d = dalloc(bytes);
    xcopy(d->p, m->p, bytes);
    return d;
  };
  auto col = [&](const DColumn &c) {
    DColumn d;
    d.name = c.name; d.dtype = c.dtype; d.repr = c.repr; d.data_len = c.data_len;
    const size_t n = (size_t)b.nrows;
    d.values = mv(c.values, n * (size_t)repr_width(c.repr));
    d.offsets = mv(c.offsets, (n + 1) * 4);
    d.data = mv(c.payload(), (size_t)c.data_len);
    d.nanos = mv(c.nanos, n * 4);
    d.validity = mv(c.validity, (n + 7) / 8);
    d.absent = mv(c.absent, (n + 7) / 8);
    return d;
  };
  for (auto &c : b.cols) r->cols.push_back(col(c));
  for (auto &c : b.old_keys) r->old_keys.push_back(col(c));
  const size_t n = (size_t)b.nrows;
  r->old_present = mv(b.old_present, (n + 7) / 8);
  r->kind = mv(b.kind, n); r->src_row = mv(b.src_row, n * 4); r->part_id = mv(b.part_id, n * 4);
  if (!share) tf::sync();  // the target lane's stream: the copy has landed when the handle is returned
  return r;
}

// ---- concat ---------------------------------------------------------------------------------------------------------------
static bool same_shape(const DColumn &a, const DColumn &b) { return a.name == b.name && a.dtype == b.dtype && a.repr == b.repr; }

static std::unique_ptr<tfgpu_dbatch> concat(const std::vector<const tfgpu_dbatch *> &parts, const int64_t *row_base) {
  const tfgpu_dbatch &p0 = *parts[0];
  for (auto *p : parts) {
    if (p->cols.size() != p0.cols.size() || p->old_keys.size() != p0.old_keys.size()) throw Error(TFGPU_ERR_INVALID, "tfgpu_dbatch_concat: the parts hold different columns");
    for (size_t i = 0; i < p0.cols.size(); i++) if (!same_shape(p->cols[i], p0.cols[i])) throw Error(TFGPU_ERR_INVALID, "tfgpu_dbatch_concat: column " + p0.cols[i].name + " differs between parts");
    for (size_t i = 0; i < p0.old_keys.size(); i++) if (!same_shape(p->old_keys[i], p0.old_keys[i])) throw Error(TFGPU_ERR_INVALID, "tfgpu_dbatch_concat: OldKeys column " + p0.old_keys[i].name + " differs between parts");
    if (p->schema != p0.schema || p->key_names != p0.key_names || p->ns != p0.ns || p->table != p0.table) throw Error(TFGPU_ERR_INVALID, "tfgpu_dbatch_concat: the parts belong to different tables / schemas");
    // late-materialised text (CSV cells still in their source text) is packed by kernels on the CALLING lane's stream: fine for a part
    // of this device, not for one whose text lives on another (no peer mapping is assumed) — such a part must be complete on its
    // own lane first (tfgpu_dbatch_to_lane packs before it copies)
    bool lazy_remote = false;
    for (auto &c : p->cols) if (c.view && !c.view->packed && !c.data && c.view->fstart && c.view->fstart->device != ctx().device) lazy_remote = true;
    if (lazy_remote) throw Error(TFGPU_ERR_INVALID, "tfgpu_dbatch_concat: a part on another device still holds unpacked text columns: move it with tfgpu_dbatch_to_lane (which packs) first");
    if (p->col_order) throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_dbatch_concat: a part's rows carry their own ColumnNames order (a collapsed batch)");
    materialize(*p);
  }
  hipStream_t st = ctx().stream;
  std::vector<int64_t> base(parts.size() + 1, 0);
  for (size_t g = 0; g < parts.size(); g++) base[g + 1] = base[g] + parts[g]->nrows;
  const int64_t n = base.back();
  if (n > 0x7FFFFFFFll) throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_dbatch_concat: more than 2^31-1 rows");
  auto r = std::make_unique<tfgpu_dbatch>();
  r->nrows = n; r->schema = p0.schema; r->key_names = p0.key_names; r->ns = p0.ns; r->table = p0.table;
  // a bitmap part that lives on another device is staged here first (the splice kernel reads it)
  auto local = [&](const Buf &m, size_t bytes) -> Buf {
    if (!m || m->device == ctx().device) return m;
    Buf d = dalloc(bytes);
    xcopy(d->p, m->p, bytes);
    return d;
  };
  auto bitmap = [&](auto pick, bool any, bool null_means_ones = true) -> Buf {  // pick(part) -> const Buf& ; a part without one = all ones (validity) or all zeros (ABSENT)
    if (!any) return nullptr;
    Buf out = dalloc_zero((size_t)(n + 7) / 8 + 8);
    for (size_t g = 0; g < parts.size(); g++) {
      const int64_t k = parts[g]->nrows;
      if (!k) continue;
      Buf in = local(pick(*parts[g]), (size_t)(k + 7) / 8);
      if (!in && !null_means_ones) continue;  // (the block starts zeroed)
      const int64_t nb = ((base[g] + k + 7) >> 3) - (base[g] >> 3);
      shard_splice_bits<<<blocks_of(nb), 256, 0, st>>>(ptr<uint8_t>(in), k, base[g], ptr<uint8_t>(out), in ? 0 : 1);
    }
    return out;
  };
  auto col = [&](size_t i, bool old) {
    auto of = [&](const tfgpu_dbatch &p) -> const DColumn & { return old ? p.old_keys[i] : p.cols[i]; };
    const DColumn &c0 = of(p0);
    DColumn d;
    d.name = c0.name; d.dtype = c0.dtype; d.repr = c0.repr;
    if (repr_is_var(c0.repr)) {
      uint64_t total = 0;
      for (auto *p : parts) total += of(*p).data_len;
      if (total >> 32) throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_dbatch_concat: column " + c0.name + " exceeds 4 GiB of text");
      d.data_len = total;
      d.data = dalloc((size_t)total);
      d.offsets = dalloc((size_t)(n + 1) * 4);
      uint64_t at = 0;
      for (size_t g = 0; g < parts.size(); g++) {
        const DColumn &c = of(*parts[g]);
        const int64_t k = parts[g]->nrows;
        xcopy((char *)d.data->p + at, c.payload() ? c.payload()->p : nullptr, (size_t)c.data_len);
        // the part's offsets, shifted, straight into place (its last entry is overwritten by the next part's first: same value)
        Buf off = local(c.offsets, (size_t)(k + 1) * 4);
        if (off) shard_rebase_offsets<<<blocks_of(k + 1), 256, 0, st>>>(ptr<uint32_t>(off), k + 1, 0u, (uint32_t)at, ptr<uint32_t>(d.offsets) + base[g]);
        at += c.data_len;
      }
      if (n == 0) TF_HIP(hipMemsetAsync(d.offsets->p, 0, 4, st));
    } else {
      const size_t w = (size_t)repr_width(c0.repr);
      d.values = dalloc((size_t)n * w);
      bool nanos = false;
      for (auto *p : parts) nanos = nanos || (bool)of(*p).nanos;
      if (nanos) d.nanos = dalloc_zero((size_t)n * 4);
      for (size_t g = 0; g < parts.size(); g++) {
        const DColumn &c = of(*parts[g]);
        const size_t k = (size_t)parts[g]->nrows;
        if (c.values) xcopy((char *)d.values->p + (size_t)base[g] * w, c.values->p, k * w);
        if (c.nanos) xcopy(ptr<int32_t>(d.nanos) + base[g], c.nanos->p, k * 4);
      }
    }
    bool anyv = false;
    for (auto *p : parts) anyv = anyv || (bool)of(*p).validity;
    d.validity = bitmap([&](const tfgpu_dbatch &p) -> const Buf & { return of(p).validity; }, anyv);
    bool anya = false;
    for (auto *p : parts) anya = anya || (bool)of(*p).absent;
    d.absent = bitmap([&](const tfgpu_dbatch &p) -> const Buf & { return of(p).absent; }, anya, false);  // the Bufferer's flush keeps every row's ColumnNames
    return d;
  };
  for (size_t i = 0; i < p0.cols.size(); i++) r->cols.push_back(col(i, false));
  for (size_t i = 0; i < p0.old_keys.size(); i++) r->old_keys.push_back(col(i, true));
  bool anyp = false, anyk = false, anys = false, anyid = false;
  for (auto *p : parts) { anyp = anyp || (bool)p->old_present; anyk = anyk || (bool)p->kind; anys = anys || (bool)p->src_row; anyid = anyid || (bool)p->part_id; }
  if (!p0.old_keys.empty()) r->old_present = bitmap([&](const tfgpu_dbatch &p) -> const Buf & { return p.old_present; }, anyp);
  if (anyk) {
    r->kind = dalloc((size_t)n);
    for (size_t g = 0; g < parts.size(); g++) {
      const size_t k = (size_t)parts[g]->nrows;
      if (parts[g]->kind) xcopy(ptr<uint8_t>(r->kind) + base[g], parts[g]->kind->p, k);
      else if (k) TF_HIP(hipMemsetAsync(ptr<uint8_t>(r->kind) + base[g], TFGPU_K_INSERT, k, st));
    }
  }
  if (anys || row_base) {  // src_row: the row of the batch that was sharded
    r->src_row = dalloc((size_t)n * 4);
    for (size_t g = 0; g < parts.size(); g++) {
      const int64_t k = parts[g]->nrows;
      const int32_t add = row_base ? (int32_t)row_base[g] : 0;
      if (!k) continue;
      if (parts[g]->src_row) {
        Buf in = local(parts[g]->src_row, (size_t)k * 4);
        shard_shift_i32<<<blocks_of(k), 256, 0, st>>>(ptr<int32_t>(in), k, add, ptr<int32_t>(r->src_row) + base[g]);
      } else shard_iota_i32<<<blocks_of(k), 256, 0, st>>>(k, add, ptr<int32_t>(r->src_row) + base[g]);
    }
  }
  if (anyid) {
    r->part_id = dalloc_zero((size_t)n * 4);
    for (size_t g = 0; g < parts.size(); g++) if (parts[g]->part_id) xcopy(ptr<uint32_t>(r->part_id) + base[g], parts[g]->part_id->p, (size_t)parts[g]->nrows * 4);
  }
  tf::sync();  // staged copies of remote bitmaps die with this scope
  return r;
}

// (the bufferer's flush: a merged batch whose parts' source rows cannot be lined up carries none)
void dbatch_drop_src_row(tfgpu_dbatch *b) { if (b) b->src_row = nullptr; }

}  // namespace tf

using namespace tf;

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

extern "C" {

int tfgpu_dbatch_slice(const tfgpu_dbatch *b, int64_t row0, int64_t nrows, tfgpu_dbatch **out) {
  TF_API_BEGIN
  tf::dense(b, true);  // its rows may still be a selection (tfgpu_dbatch::pending); ABSENT bitmaps are cut / copied with the rows
  if (!b || !out || row0 < 0 || nrows < 0 || row0 + nrows > b->nrows) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_slice: bad argument");
  if (row0 & 7) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_slice: row0 must be a multiple of 8 (bitmaps are cut at byte boundaries)");
  std::lock_guard<std::mutex> lk(ctx().mu);
  auto r = slice_many(*b, {row0, row0 + nrows});
  tf::sync();
  *out = r[0].release();
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_dbatch_to_lane(const tfgpu_dbatch *b, int lane, tfgpu_dbatch **out) {
  TF_API_BEGIN
  tf::dense(b, true);  // its rows may still be a selection (tfgpu_dbatch::pending); ABSENT bitmaps are cut / copied with the rows
  if (!b || !out || lane < 0 || lane >= tfgpu_lane_count()) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_to_lane: bad argument");
  { std::lock_guard<std::mutex> lk(ctx().mu); materialize(*b); tf::sync(); }  // everything enqueued for `b` on this lane has run
  *out = to_lane(*b, lane).release();
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_shard_rows(const tfgpu_dbatch *b, int nshards, const int *lanes, tfgpu_dbatch **out, int64_t *row0) {
  TF_API_BEGIN
  tf::dense(b, true);  // its rows may still be a selection (tfgpu_dbatch::pending); ABSENT bitmaps are cut / copied with the rows
  if (!b || !out || nshards < 1) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_shard_rows: bad argument");
  for (int g = 0; lanes && g < nshards; g++) if (lanes[g] < 0 || lanes[g] >= tfgpu_lane_count()) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_shard_rows: lane out of range");
  std::vector<int64_t> cuts((size_t)nshards + 1, 0);
  const int64_t groups = (b->nrows + 63) / 64;  // balanced, cut at multiples of 64 rows
  for (int g = 1; g < nshards; g++) cuts[(size_t)g] = std::min<int64_t>(b->nrows, (groups * g / nshards) * 64);
  cuts[(size_t)nshards] = b->nrows;
  std::vector<std::unique_ptr<tfgpu_dbatch>> parts;
  {
    std::lock_guard<std::mutex> lk(ctx().mu);
    parts = slice_many(*b, cuts);
    tf::sync();
  }
  const int here = current_lane();
  for (int g = 0; g < nshards; g++) {
    const int lane = lanes ? lanes[g] : g;
    if (lane != here) parts[(size_t)g] = to_lane(*parts[(size_t)g], lane);
    if (row0) row0[g] = cuts[(size_t)g];
  }
  for (int g = 0; g < nshards; g++) out[g] = parts[(size_t)g].release();
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_dbatch_concat(const tfgpu_dbatch *const *parts, int nparts, const int64_t *row_base, tfgpu_dbatch **out) {
  TF_API_BEGIN
  if (!parts || !out || nparts < 1) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_concat: bad argument");
  std::vector<const tfgpu_dbatch *> v;
  for (int g = 0; g < nparts; g++) { if (!parts[g]) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_concat: null part"); v.push_back(parts[g]); }
  for (auto *b : v) tf::dense(b, true);  // rows that are still a selection (tfgpu_dbatch::pending)
  std::lock_guard<std::mutex> lk(ctx().mu);
  *out = concat(v, row_base).release();
  return TFGPU_OK;
  TF_API_END
}

}  // extern "C"
