// tf_common.hpp — internal (non-ABI) types of libtfgpu: context, HBM buffers,
// device batches, error plumbing and the per-kernel event profiler.
// gfx950 / CDNA4 only; there is no CPU path in this library.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tfgpu.h"

// dynamic LDS named at file scope (the CPU emulator's hip_runtime.h defines its own form)
#ifndef TF_DYNAMIC_LDS
#define TF_DYNAMIC_LDS(type, name) extern __shared__ type name[]
#endif

// a pointer known to address device (global) memory, e.g. one read back from LDS as an integer: stores through it are
// global_store instead of flat_store (the CPU emulator's hip_runtime.h defines the plain cast)
#ifndef TF_GLOBAL_PTR
#define TF_GLOBAL_PTR(T, p) ((T *)(__attribute__((address_space(1))) T *)(p))
#endif

// a pointer to device memory that no kernel writes while this one runs (uploaded descriptor tables), read through the
// constant address space: a load whose address is wave-uniform becomes an s_load and its result lives in SGPRs
#ifndef TF_CONST_PTR
#define TF_CONST_PTR(T, p) ((const __attribute__((address_space(4))) T *)(p))
#endif

// a value the optimizer must treat as unknown (keeps two neighbouring narrow LDS reads from being merged into one
// misaligned wide read)
// a value the compiler must have in a vector register HERE although nothing reads it: the tail end of a load issued early only
// to bring its cache line in (the CPU emulator's hip_runtime.h defines the no-op)
#ifndef TF_KEEP
#define TF_KEEP(x) asm volatile("" ::"v"(x))
#endif
#ifndef TF_OPAQUE
#define TF_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

struct tfgpu_dbatch;

namespace tf {

// ---- error plumbing -------------------------------------------------------
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
void set_last_error(const std::string &m);
int fail(int code, const std::string &m);  // sets last error, returns code

#define TF_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess)                                                                        \
      throw ::tf::Error(TFGPU_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));    \
  } while (0)

// ---- context: one process ↔ one GPU ↔ one stream --------------------------
struct ProfEntry {
  std::string name;
  int64_t launches = 0;
  double total_ms = 0;
  int64_t units = 0;  // rows / values / bytes the timed launches were issued over (KernelTimer's second argument), summed
};

struct BlockCache {
  std::mutex mu;
  std::unordered_map<size_t, std::vector<void *>> free_;
  size_t cached = 0;
  void *get(size_t cls);
  void put(void *p, size_t cls);
  void trim();
};

struct Context {
  int device = -1;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // created on first use: a large upload in pieces beside the kernels that read the pieces already there (tf_parquet.hip)
  BlockCache blocks;
  // pinned ring for small host<->device tables and read-backs
  void *pin_base = nullptr;
  size_t pin_cap = 0, pin_off = 0;
  void *pin(size_t bytes);
  template <class T> T *pin_n(size_t n) { return reinterpret_cast<T *>(pin(n * sizeof(T))); }
  bool prof_on = false;
  std::vector<ProfEntry> prof;
  struct Pending { int idx; hipEvent_t a, b; int64_t units = 0; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
  hipEvent_t dense_event = nullptr;  // re-recorded behind every selection -> dense gather of this lane while other lanes are alive (tf::dense_locked)
  std::mutex mu;  // serialises API calls that enqueue on the stream
  int num_cus = 256;
  std::shared_ptr<struct DevMem> pow10tab;  // math.Pow10(n), n = -323..308 (tf_json.hip)
  // read-only descriptor tables already in HBM, by content (upload_const): a steady stream of same-shaped batches re-sends
  // the same column / plan tables call after call
  struct ConstEntry { uint64_t hash; std::vector<uint8_t> bytes; std::shared_ptr<struct DevMem> mem; uint64_t stamp; };
  std::vector<ConstEntry> consts;
  uint64_t const_clock = 0;
  // tfgpu_csv_parse's single-pass form: bytes per line of the last chunk this lane parsed, and that chunk's column count
  double csv_hint_bpl = 0.0;
  int csv_hint_ncols = -1;

  hipEvent_t get_event();
  void prof_begin(const char *name, Pending &p);
  void prof_end(Pending &p);
  void prof_flush();
};
Context &ctx();  // throws TFGPU_ERR_DEVICE if tfgpu_init has not succeeded
int bind_lane(int lane);   // binds the calling thread (and its HIP device) to `lane`; returns the lane it was bound to before
int current_lane();
int lane_device(int lane);
struct LaneScope {         // the calling thread works on another lane for a while (allocations for it, copies to it)
  int prev;
  explicit LaneScope(int lane) : prev(bind_lane(lane)) {}
  ~LaneScope() { try { bind_lane(prev); } catch (...) {} }
};

// RAII marker: times everything enqueued between ctor and dtor under `name`.
struct KernelTimer {
  Context::Pending p{-1, nullptr, nullptr};
  bool on;
  explicit KernelTimer(const char *name, int64_t units = 0) : on(ctx().prof_on) { if (on) { ctx().prof_begin(name, p); p.units = units; } }
  ~KernelTimer() { if (on) ctx().prof_end(p); }
};

// ---- HBM buffers (stream-ordered pool) -------------------------------------
struct DevMem {
  void *p = nullptr;
  size_t bytes = 0, cls = 0;
  std::shared_ptr<DevMem> parent;  // set for views into a larger block (see subbuf)
  struct BlockCache *owner = nullptr;  // the lane's cache the block returns to
  int device = -1;                     // the HIP device the block lives on
  explicit DevMem(size_t n);
  DevMem(std::shared_ptr<DevMem> owner, size_t off, size_t n);
  ~DevMem();
  DevMem(const DevMem &) = delete;
  DevMem &operator=(const DevMem &) = delete;
};
using Buf = std::shared_ptr<DevMem>;
Buf dalloc(size_t bytes);              // uninitialised
Buf subbuf(const Buf &owner, size_t off, size_t bytes);  // view sharing ownership of `owner`
Buf dalloc_zero(size_t bytes);
template <class T> inline T *ptr(const Buf &b) { return b ? reinterpret_cast<T *>(b->p) : nullptr; }

void h2d(void *dst, const void *src, size_t n);
void h2d_small(void *dst, const void *src, size_t n);  // via the pinned arena: truly async
Buf upload_small(const void *src, size_t n);
// the same for tables no kernel ever writes: an identical table uploaded earlier on this lane is handed out again (no copy)
void dbatch_drop_src_row(struct ::tfgpu_dbatch *b);  // tf_shard.hip
Buf upload_const(const void *src, size_t n);
// n uint32 words read back into the pinned arena; valid after the next sync()
const uint32_t *d2h_u32(const void *dev, size_t n = 1);
struct PinScope { PinScope() {} };  // (the pinned arena is a ring; nothing to do per call)
void d2h(void *dst, const void *src, size_t n);
void d2d(void *dst, const void *src, size_t n);
void sync();

// ---- device batches --------------------------------------------------------
inline bool repr_is_var(int r) { return r == TFGPU_R_STRING || r == TFGPU_R_BYTES || r == TFGPU_R_JSONNUM || r == TFGPU_R_JSON; }
inline size_t repr_width(int r) {
  switch (r) {
    case TFGPU_R_INT8: case TFGPU_R_UINT8: case TFGPU_R_BOOL: return 1;
    case TFGPU_R_INT16: case TFGPU_R_UINT16: return 2;
    case TFGPU_R_INT32: case TFGPU_R_UINT32: case TFGPU_R_FLOAT32: return 4;
    case TFGPU_R_INT64: case TFGPU_R_UINT64: case TFGPU_R_FLOAT64: case TFGPU_R_TIME: case TFGPU_R_DURATION: return 8;
    default: return 0;
  }
}

// Late-materialised text cells (CSV ingest): until a consumer needs the packed Arrow payload, a text column is its
// offsets (lengths already scanned, data_len known) plus, per row, where the cell sits in the SOURCE text — the columnar
// form of Go substrings that alias the chunk they were cut from.  Row compaction packs the kept cells straight from
// the source; everything else calls materialize() first.  `packed` is shared by every copy of the column.
struct TextView {
  Buf src;              // the CSV text, kept alive here
  Buf fstart;           // u32[nrows]: offset of the cell in src | bit 31: collapse doubled quotes; low 31 bits all ones: DefaultValue
  uint8_t quote = '"';
  bool jsonnum = false; // DefaultValue of a double is json.Number("0")
  bool has_special = true;  // some cell is not a plain byte range of src (doubled quotes, DefaultValue text)
  Buf packed;           // set once by materialize()
};
struct DColumn {
  std::string name;
  int dtype = TFGPU_T_INVALID;
  int repr = TFGPU_R_INVALID;
  Buf values, offsets, data, nanos, validity;
  Buf absent;  // bitmap, bit r = 1: row r's ColumnNames do not list this column (tfgpu_column.absent); null = every row lists it
  uint64_t data_len = 0;
  std::shared_ptr<TextView> view;  // non-null while `data` may still be unpacked
  const Buf &payload() const { return (!data && view) ? view->packed : data; }
  bool lazy() const { return !data && view && !view->packed; }
};
// Packs the lazy text columns of `b` (all of them, or only those listed) from their source text; no-op otherwise.
void materialize(const struct ::tfgpu_dbatch &b, const std::vector<const DColumn *> *only = nullptr);
// Gathers the rows of a batch whose rows are still a selection (tfgpu_dbatch::pending); no-op otherwise.  dense() takes the lane's
// mutex itself (call it BEFORE locking); dense_locked() is for callers that hold it.
// dense() also REFUSES (TFGPU_ERR_UNSUPPORTED, by name) a batch that holds ABSENT cells (DColumn::absent) unless the caller says it
// reads them (collapse, keys_changed, view / download, the row movers): an entry that computes on values would read them as nil.
void dense(const struct ::tfgpu_dbatch *b, bool absent_ok = false);
void dense_locked(const struct ::tfgpu_dbatch &b);
int lanes_created();   // lanes (stream + cache + pinned ring) this process has made so far (tf_runtime.hip)
bool has_absent(const struct ::tfgpu_dbatch &b);
// a fresh bitmap: (validity, or all ones when null) with the bits of `absent` cleared — an ABSENT cell reads nil (tf_runtime.hip)
Buf validity_minus_absent(const Buf &validity, const Buf &absent, int64_t nrows);
// The library-owned HBM block that starts at `p` (a tfgpu_dbuf), or null for foreign pointers.
Buf find_device_block(const void *p);
void register_device_block(const Buf &b);

}  // namespace tf

// Rows a row filter kept but nobody has read yet (late materialisation of ROWS, the counterpart of TextView): filter_rows and
// skip_events hand on "rows sel[0 .. nrows) of the batch I read" instead of gathering 100 columns for a sink that may drop them
// (devnull), a serializer that walks a selection anyway, or a next transformer that reads two of them.  The kept ChangeItems of the
// reference alias the input items the same way (filter_rows.go:99-125 appends the item, it copies no value).  tf::dense() gathers —
// every C-ABI entry that reads a batch's columns calls it first; transformers that can work through a selection (mask_field) do.
namespace tf {
struct PendingRows {
  std::shared_ptr<const struct ::tfgpu_dbatch> src;  // the dense batch the filter read
  Buf sel;                                           // int32[nrows]: the kept rows of src, ascending
};
}  // namespace tf

struct tfgpu_dbatch {
  int64_t nrows = 0;
  std::vector<tf::DColumn> cols;                    // EMPTY while `pending` is set
  std::shared_ptr<tf::PendingRows> pending;         // set: this batch is nrows selected rows of pending->src, not gathered yet
  void *dense_done = nullptr;                       // the gathering lane's hipEvent_t (Context::dense_event, lives as long as the lane), recorded behind the gather that ended
  int dense_lane = -1;                              //   `pending` while several lanes were alive: a reader on ANOTHER lane's stream waits for it (tf::dense)
  std::vector<tf::DColumn> replaced;                // with `pending`: columns already computed over the kept rows (dense, by name)
  std::vector<std::pair<std::string, int>> schema;  // TableSchema (name, DataType) in order; empty = same as cols
  std::vector<std::string> key_names;               // names of the PrimaryKey columns of that schema (MakeMapKeys)
  std::vector<tf::DColumn> old_keys;                // OldKeys.KeyValues by KeyNames; empty = no row has OldKeys
  tf::Buf old_present;                              // bitmap: the row has OldKeys; null with old_keys = every row
  std::string ns, table;
  tf::Buf kind, src_row, part_id;
  tf::Buf col_order;  // uint16 [nrows][ncols] or null: every row's own ColumnNames order (tfgpu_batch.col_order; tfgpu_collapse's merged rows)
};

struct tfgpu_dbuf {
  tf::Buf mem;
  uint64_t size = 0;
};

namespace tf {

// ---- shared device primitives (tf_scan.hip) --------------------------------
// out[i] = sum(in[0..i)), i in [0, n]; out has n+1 entries when `with_total`.
// in/out may alias.  All on ctx().stream.
void exclusive_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, bool with_total);
// Segmented variant for `nseg` equally-long arrays laid out back to back
// (used for the string columns of a batch): each segment scanned on its own.
void exclusive_scan_u32_segments(uint32_t *inout, int64_t seg_len, int nseg, int64_t seg_stride);
const uint32_t *any_nonzero_to_host(const uint32_t *v, int64_t n);  // one flag word read back; valid after the next sync
const uint32_t *segment_totals_to_host(const uint32_t *scanned, int64_t seg_len, int nseg, int64_t seg_stride);  // one launch + one read-back; valid after the next sync
// totals[s] += sum of segment s's seg_len entries, in 64 bits (the scans above wrap at 4 GiB: callers that size a 32-bit-offset
// column from a scan's total check this sum first).  totals must be zeroed; enqueued on the lane's stream.
void sum_u32_segments_u64(const uint32_t *in, int64_t seg_len, int nseg, int64_t seg_stride, unsigned long long *totals);

}  // namespace tf
