// tf_runtime.hip — context, HBM pool, profiler and the batch/buffer half of the
// C ABI (include/tfgpu.h).  One process drives one MI355X through one HIP
// stream; device memory comes from a stream-ordered pool whose release
// threshold is unlimited, so steady-state batches never hit hipMalloc.
#include <algorithm>
#include <cstdlib>

#include "tf_common.hpp"

namespace tf {

void executor_shutdown();  // tf_transformation.cpp
static thread_local std::string g_last_error;
// Lanes: independent (device, stream, HBM block cache, pinned ring, profiler) sets.  A host thread binds itself to a lane
// with tfgpu_lane_use(); calls on different lanes overlap — on one GPU the parse of batch N+1 beside transform / serialize /
// D2H of batch N (the parsequeue's parallel workers, parsequeue.go:57-154, map onto lanes), and with a device list
// (tfgpu_init_devices) lane k lives on device k mod G, so ONE process drives several GPUs, a row-range shard each
// (tfgpu_shard_rows / tfgpu_dbatch_concat, tf_shard.hip).  Buffers belong to the lane that made them.
static constexpr int MAX_LANES = 32;
static std::unique_ptr<Context> g_lanes[MAX_LANES];
static thread_local int g_lane = 0;
static std::vector<int> g_devices;      // the process's device list; lane k -> g_devices[k % size]
static std::vector<int> g_device_cus;   // multiProcessorCount of each
static std::mutex g_init_mu;
static int lane_device_index(int lane) { return g_devices.empty() ? 0 : lane % (int)g_devices.size(); }
#define g_ctx g_lanes[0]

void set_last_error(const std::string &m) { g_last_error = m; }
int fail(int code, const std::string &m) { g_last_error = m; return code; }

// the HIP device is a per-thread setting: a thread that reaches a lane (bound, or lane 0 by default) works on that lane's device
static thread_local int t_device = -1;
static void ensure_device(int device) {
  if (t_device == device) return;
  TF_HIP(hipSetDevice(device));
  t_device = device;
}
Context &ctx() {
  if (!g_ctx) throw Error(TFGPU_ERR_DEVICE, "tfgpu_init() has not been called (or no gfx950 device): no CPU fallback exists");
  Context *c = g_lanes[g_lane].get();
  if (!c) throw Error(TFGPU_ERR_INVALID, "this thread is bound to a lane that no longer exists");
  ensure_device(c->device);
  return *c;
}
static std::unique_ptr<Context> make_lane(int lane) {  // the caller's HIP device is the lane's when this returns
  auto c = std::make_unique<Context>();
  c->device = g_devices[(size_t)lane_device_index(lane)];
  c->num_cus = g_device_cus[(size_t)lane_device_index(lane)];
  ensure_device(c->device);
  TF_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->pin_cap = 8u << 20;
  // (portable pinned memory only when there is more than one device to be portable to: on a single device the plain form)
  TF_HIP(hipHostMalloc(&c->pin_base, c->pin_cap, g_devices.size() > 1 ? hipHostMallocPortable : hipHostMallocDefault));
  return c;
}
// binds the calling thread to `lane` (created on first use) and makes the lane's device the thread's HIP device
int bind_lane(int lane) {
  if (lane < 0 || lane >= MAX_LANES) throw Error(TFGPU_ERR_INVALID, "lane out of range");
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (!g_ctx) throw Error(TFGPU_ERR_DEVICE, "tfgpu_init() has not been called");
  if (!g_lanes[lane]) g_lanes[lane] = make_lane(lane);
  ensure_device(g_lanes[lane]->device);
  const int prev = g_lane;
  g_lane = lane;
  return prev;
}
int current_lane() { return g_lane; }
int lane_device(int lane) { return (lane < 0 || g_devices.empty()) ? -1 : g_devices[(size_t)lane_device_index(lane)]; }
static void destroy_lane(std::unique_ptr<Context> &c) {
  if (!c) return;
  t_device = -1;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  for (auto &p : c->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
  for (auto e : c->free_events) hipEventDestroy(e);
  c->pow10tab.reset();
  c->consts.clear();
  c->blocks.trim();
  if (c->pin_base) hipHostFree(c->pin_base);
  hipStreamDestroy(c->stream);
  c.reset();
}

hipEvent_t Context::get_event() {
  if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
  hipEvent_t e;
  TF_HIP(hipEventCreate(&e));
  return e;
}
void Context::prof_begin(const char *name, Pending &p) {
  int idx = -1;
  for (size_t i = 0; i < prof.size(); i++) if (prof[i].name == name) { idx = (int)i; break; }
  if (idx < 0) { prof.push_back(ProfEntry{name, 0, 0, 0}); idx = (int)prof.size() - 1; }
  p.idx = idx; p.a = get_event(); p.b = get_event();
  TF_HIP(hipEventRecord(p.a, stream));
}
void Context::prof_end(Pending &p) {
  hipEventRecord(p.b, stream);
  pending.push_back(p);
  if (pending.size() > 4096) prof_flush();
}
void Context::prof_flush() {
  if (pending.empty()) return;
  hipStreamSynchronize(stream);
  for (auto &p : pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { prof[p.idx].launches++; prof[p.idx].total_ms += ms; prof[p.idx].units += p.units; }
    free_events.push_back(p.a); free_events.push_back(p.b);
  }
  pending.clear();
}

// ---- HBM block cache ---------------------------------------------------------
// Every buffer of the library lives on ONE stream, so a block handed back by the
// host can be re-issued at once: whatever is enqueued next runs after the work
// that still reads it.  Blocks are binned in 8 size classes per octave (<= 12.5 %
// slack) and never returned to the driver while the context lives — the HIP pool
// (hipMallocAsync/hipFreeAsync) cost 40-80 us per call here (profiles/r01c), more
// than most kernels of the path.
static inline size_t size_class(size_t n) {
  if (n <= 256) return 256;
  int hi = 63 - __builtin_clzll((unsigned long long)n);  // floor(log2 n)
  size_t step = (size_t)1 << (hi > 3 ? hi - 3 : 0);
  return (n + step - 1) & ~(step - 1);
}
void *BlockCache::get(size_t cls) {
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = free_.find(cls);
    if (it != free_.end() && !it->second.empty()) { void *p = it->second.back(); it->second.pop_back(); cached -= cls; return p; }
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, cls);
  if (e != hipSuccess) {  // give everything cached back to the driver and retry once
    (void)hipGetLastError();
    trim();
    e = hipMalloc(&p, cls);
  }
  if (e != hipSuccess) throw Error(TFGPU_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(cls) + "): " + hipGetErrorString(e));
  return p;
}
void BlockCache::put(void *p, size_t cls) {
  std::lock_guard<std::mutex> lk(mu);
  free_[cls].push_back(p);
  cached += cls;
}
void BlockCache::trim() {
  std::lock_guard<std::mutex> lk(mu);
  hipDeviceSynchronize();
  for (auto &kv : free_) for (void *p : kv.second) hipFree(p);
  free_.clear();
  cached = 0;
}

DevMem::DevMem(size_t n) : bytes(n) {
  Context &c = ctx();
  device = c.device;
  cls = size_class(n + 16);  // 16 bytes of slack: kernels read whole aligned words around payloads
  p = c.blocks.get(cls);
  owner = &c.blocks;
}
DevMem::DevMem(std::shared_ptr<DevMem> owner, size_t off, size_t n) : p((char *)owner->p + off), bytes(n), cls(0), parent(std::move(owner)) { device = parent->device; }
DevMem::~DevMem() {
  if (parent) return;  // a view: the owner returns the block
  if (p && owner && g_ctx) owner->put(p, cls);  // back to the cache of the lane whose stream ordered its use
}
Buf subbuf(const Buf &owner, size_t off, size_t bytes) { return std::make_shared<DevMem>(owner, off, bytes); }

// Ring: slots are handed out in order and recycled only after a stream sync at wrap-around, so a
// pending async copy never sees its pinned source (or a read-back its destination) reused.
void *Context::pin(size_t bytes) {
  size_t a = (bytes + 63) & ~(size_t)63;
  if (a > pin_cap) return nullptr;
  if (pin_off + a > pin_cap) { hipStreamSynchronize(stream); pin_off = 0; }
  void *r = (char *)pin_base + pin_off;
  pin_off += a;
  return r;
}
Buf dalloc(size_t bytes) { return std::make_shared<DevMem>(bytes); }
Buf dalloc_zero(size_t bytes) {
  Buf b = dalloc(bytes);
  if (bytes) TF_HIP(hipMemsetAsync(b->p, 0, bytes, ctx().stream));
  return b;
}
void h2d(void *dst, const void *src, size_t n) { if (n) TF_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, ctx().stream)); }
// small host tables (plans, column descriptors): bounce through the pinned arena so the copy is
// asynchronous and the caller's vector may die right after the call
void h2d_small(void *dst, const void *src, size_t n) {
  if (!n) return;
  void *st = ctx().pin(n);
  if (!st) { TF_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, ctx().stream)); TF_HIP(hipStreamSynchronize(ctx().stream)); return; }
  std::memcpy(st, src, n);
  TF_HIP(hipMemcpyAsync(dst, st, n, hipMemcpyHostToDevice, ctx().stream));
}
const uint32_t *d2h_u32(const void *dev, size_t n) {
  uint32_t *h = ctx().pin_n<uint32_t>(n ? n : 1);
  if (!h) throw Error(TFGPU_ERR_NOMEM, "pinned read-back arena exhausted");
  d2h(h, dev, n * 4);
  return h;
}
Buf upload_small(const void *src, size_t n) {
  Buf b = dalloc(n + 16);
  h2d_small(b->p, src, n);
  return b;
}
Buf upload_const(const void *src, size_t n) {
  static const bool off = [] { const char *e = std::getenv("TFGPU_NO_CONST_CACHE"); return e && e[0] == '1'; }();  // A/B measurements
  if (n == 0 || n > (32u << 10) || off) return upload_small(src, n);
  Context &cx = ctx();
  uint64_t h = 1469598103934665603ull;  // FNV-1a over 8-byte steps, tail bytewise
  const uint8_t *b = static_cast<const uint8_t *>(src);
  size_t i = 0;
  for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, b + i, 8); h = (h ^ w) * 1099511628211ull; }
  for (; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  for (auto &e : cx.consts)
    if (e.hash == h && e.bytes.size() == n && !std::memcmp(e.bytes.data(), src, n)) { e.stamp = ++cx.const_clock; return e.mem; }
  Buf m = upload_small(src, n);
  if (cx.consts.size() >= 128) {  // drop the entry used longest ago
    size_t old = 0;
    for (size_t k = 1; k < cx.consts.size(); k++) if (cx.consts[k].stamp < cx.consts[old].stamp) old = k;
    cx.consts.erase(cx.consts.begin() + (long)old);
  }
  cx.consts.push_back(Context::ConstEntry{h, std::vector<uint8_t>(b, b + n), m, ++cx.const_clock});
  return m;
}
void d2h(void *dst, const void *src, size_t n) { if (n) TF_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, ctx().stream)); }
void d2d(void *dst, const void *src, size_t n) { if (n) TF_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, ctx().stream)); }
void sync() { TF_HIP(hipStreamSynchronize(ctx().stream)); }

// library-owned blocks handed out as tfgpu_dbuf, by base pointer: a lazy text column keeps its source alive through this
static std::mutex g_blocks_mu;
static std::unordered_map<const void *, std::weak_ptr<DevMem>> g_blocks;
void register_device_block(const Buf &b) {
  if (!b) return;
  std::lock_guard<std::mutex> lk(g_blocks_mu);
  if (g_blocks.size() > 4096) for (auto it = g_blocks.begin(); it != g_blocks.end();) it = it->second.expired() ? g_blocks.erase(it) : std::next(it);
  g_blocks[b->p] = b;
}
Buf find_device_block(const void *p) {
  std::lock_guard<std::mutex> lk(g_blocks_mu);
  auto it = g_blocks.find(p);
  if (it == g_blocks.end()) return nullptr;
  Buf b = it->second.lock();
  if (!b || b->p != p) { g_blocks.erase(it); return nullptr; }
  return b;
}

// An ABSENT cell reads nil: every kernel that reads key cells goes through the validity bitmap only, so the
// invariant "absent bit set => validity bit clear" is established HERE, not trusted (a caller's NULL validity means "no nils").
// One thread per bitmap byte; the bits past nrows of the last byte stay as the caller sent them (validity) or ones (no validity).
__global__ void upload_validity_minus_absent(const uint8_t *validity_in, const uint8_t *absent, int64_t nbytes, uint8_t *validity_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nbytes) return;
  validity_out[i] = (uint8_t)((validity_in ? validity_in[i] : 0xFFu) & ~absent[i]);
}

int lanes_created() { int n = 0; for (auto &l : g_lanes) if (l) n++; return n; }
Buf validity_minus_absent(const Buf &validity, const Buf &absent, int64_t nrows) {
  const int64_t nb = (nrows + 7) / 8;
  Buf v = dalloc((size_t)nb + 8);
  if (nb) upload_validity_minus_absent<<<(unsigned)((nb + 255) / 256), 256, 0, ctx().stream>>>(ptr<uint8_t>(validity), ptr<uint8_t>(absent), nb, ptr<uint8_t>(v));
  return v;
}

}  // namespace tf

using namespace tf;

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

extern "C" {

int tfgpu_abi_version(void) { return TFGPU_ABI_VERSION; }
const char *tfgpu_last_error(void) { return g_last_error.c_str(); }

int tfgpu_device_count(int *out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *out = 0; return tf::fail(TFGPU_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  *out = n;
  return TFGPU_OK;
}

int tfgpu_init_devices(const int *devices, int ndevices) {
  TF_API_BEGIN
  if (!devices || ndevices <= 0) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_init_devices: empty device list");
  std::lock_guard<std::mutex> lk(g_init_mu);
  const std::vector<int> want(devices, devices + ndevices);
  if (g_ctx) {
    if (g_devices == want) return TFGPU_OK;
    return tf::fail(TFGPU_ERR_INVALID, "tfgpu_init: process already bound to another device list (tfgpu_shutdown first)");
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return tf::fail(TFGPU_ERR_DEVICE, "no HIP device visible; this library has no CPU fallback");
  std::vector<int> cus;
  for (int d : want) {
    if (d < 0 || d >= n) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_init: device index out of range");
    hipDeviceProp_t prop;
    TF_HIP(hipGetDeviceProperties(&prop, d));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
      return tf::fail(TFGPU_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    cus.push_back(prop.multiProcessorCount);
  }
  g_devices = want; g_device_cus = cus;
  // Buffers move between the lanes' devices (tfgpu_dbatch_to_lane, tfgpu_dbatch_concat: hipMemcpyAsync, hipMemcpyDefault).  With the
  // peer mapping enabled that copy is one DMA over xGMI; without it the runtime stages through host memory.  Best effort: a pair
  // that cannot be mapped (or already is) keeps the staged copy, which is correct either way.
  for (size_t i = 0; i < want.size(); i++)
    for (size_t j = 0; j < want.size(); j++) {
      if (want[i] == want[j]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, want[i], want[j]) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
      if (hipSetDevice(want[i]) != hipSuccess || hipDeviceEnablePeerAccess(want[j], 0) != hipSuccess) (void)hipGetLastError();
    }
  t_device = -1;  // (the loop may have left another device current)
  g_ctx = make_lane(0);
  return TFGPU_OK;
  TF_API_END
}
int tfgpu_init(int device) { return tfgpu_init_devices(&device, 1); }

int tfgpu_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (!g_ctx) return TFGPU_OK;
  executor_shutdown();  // the push workers hold lanes
  for (int i = MAX_LANES - 1; i >= 0; i--) destroy_lane(g_lanes[i]);
  g_lane = 0;
  t_device = -1;
  g_devices.clear(); g_device_cus.clear();
  return TFGPU_OK;
}

int tfgpu_lane_count(void) { return MAX_LANES; }
int tfgpu_lane_use(int lane) {
  TF_API_BEGIN
  if (lane < 0 || lane >= MAX_LANES) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_lane_use: lane out of range");
  bind_lane(lane);
  return TFGPU_OK;
  TF_API_END
}
int tfgpu_lane_device(int lane, int *device) {
  TF_API_BEGIN
  if (!device || lane < 0 || lane >= MAX_LANES) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_lane_device: bad argument");
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (!g_ctx) return tf::fail(TFGPU_ERR_DEVICE, "tfgpu_init() has not been called");
  *device = lane_device(lane);
  return TFGPU_OK;
  TF_API_END
}
int tfgpu_lane_current(void) { return g_lane; }

int tfgpu_synchronize(void) {
  TF_API_BEGIN
  tf::sync();
  return TFGPU_OK;
  TF_API_END
}

void *tfgpu_stream(void) { return (g_ctx && g_lanes[g_lane]) ? (void *)g_lanes[g_lane]->stream : nullptr; }

int tfgpu_host_alloc(size_t bytes, void **out) {
  TF_API_BEGIN
  ctx();
  TF_HIP(hipHostMalloc(out, bytes ? bytes : 1, g_devices.size() > 1 ? hipHostMallocPortable : hipHostMallocDefault));
  return TFGPU_OK;
  TF_API_END
}
int tfgpu_host_free(void *p) {
  TF_API_BEGIN
  if (p) TF_HIP(hipHostFree(p));
  return TFGPU_OK;
  TF_API_END
}

// ---- batches ---------------------------------------------------------------
static thread_local bool g_upload_from_device = false;
static Buf upload(const void *src, size_t n) {
  if (!src) return nullptr;
  Buf b = dalloc(n);
  if (g_upload_from_device) d2d(b->p, src, n); else h2d(b->p, src, n);
  return b;
}

int tfgpu_batch_upload(const tfgpu_batch *h, tfgpu_dbatch **out) {
  TF_API_BEGIN
  if (!h || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_batch_upload: null argument");
  if (h->mem != TFGPU_MEM_HOST && h->mem != TFGPU_MEM_DEVICE) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_batch_upload: bad mem");
  if (h->col_order) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_batch_upload: col_order is an output of tfgpu_collapse (rows whose ColumnNames are not in batch order stay with the Go path)");
  std::lock_guard<std::mutex> lk(ctx().mu);
  const bool from_dev = h->mem == TFGPU_MEM_DEVICE;
  struct Flag { Flag(bool v) { g_upload_from_device = v; } ~Flag() { g_upload_from_device = false; } } flag(from_dev);
  auto b = std::make_unique<tfgpu_dbatch>();
  b->nrows = h->nrows;
  b->ns = h->table_ns ? h->table_ns : "";
  b->table = h->table_name ? h->table_name : "";
  size_t n = (size_t)h->nrows;
  for (int i = 0; i < h->ncols + h->n_old_keys; i++) {
    const bool is_old = i >= h->ncols;
    const tfgpu_column &c = is_old ? h->old_keys[i - h->ncols] : h->cols[i];
    DColumn d;
    d.name = c.name ? c.name : "";
    d.dtype = c.dtype; d.repr = c.repr;
    if (c.repr <= TFGPU_R_INVALID || c.repr >= TFGPU_R__COUNT) return tf::fail(TFGPU_ERR_INVALID, "column " + d.name + ": bad repr");
    if (repr_is_var(c.repr)) {
      if (!c.offsets) return tf::fail(TFGPU_ERR_INVALID, "column " + d.name + ": var-width column without offsets");
      d.offsets = upload(c.offsets, (n + 1) * 4);
      d.data_len = from_dev ? c.data_len : c.offsets[n];  // device offsets cannot be read here: the caller states the length
      d.data = dalloc(d.data_len);
      if (from_dev) d2d(d.data->p, c.data, d.data_len); else h2d(d.data->p, c.data, d.data_len);
    } else {
      if (!c.values && n) return tf::fail(TFGPU_ERR_INVALID, "column " + d.name + ": fixed-width column without values");
      d.values = upload(c.values, n * repr_width(c.repr));
      if (c.repr == TFGPU_R_TIME && c.nanos) d.nanos = upload(c.nanos, n * 4);
    }
    if (c.validity) d.validity = upload(c.validity, (n + 7) / 8);
    if (c.absent && !is_old && n) {  // (OldKeys say which rows have them through old_keys_present)
      d.absent = upload(c.absent, (n + 7) / 8);
      // validity = (validity or all-ones) & ~absent: an unlisted key must hash, compare and shard as nil (change_item.go CurrentKeysString)
      d.validity = validity_minus_absent(d.validity, d.absent, (int64_t)n);
    }
    (is_old ? b->old_keys : b->cols).push_back(std::move(d));
  }
  if (h->n_old_keys && h->old_keys_present) b->old_present = upload(h->old_keys_present, (n + 7) / 8);
  if (h->schema) for (int i = 0; i < h->schema->ncols; i++) {
    const tfgpu_colschema &sc = h->schema->cols[i];
    b->schema.push_back({sc.name ? sc.name : "", sc.dtype});
    if (sc.flags & TFGPU_COL_KEY) b->key_names.push_back(sc.name ? sc.name : "");
  }
  if (h->kind) b->kind = upload(h->kind, n);
  if (h->src_row) b->src_row = upload(h->src_row, n * 4);
  if (h->part_id) b->part_id = upload(h->part_id, n * 4);
  tf::sync();  // caller may release its buffers as soon as we return
  *out = b.release();
  return TFGPU_OK;
  TF_API_END
}

// Views hand out pointers into per-dbatch scratch arrays.
struct ViewStore { std::vector<tfgpu_column> cols; std::vector<tfgpu_colschema> scols; tfgpu_schema schema; };
static thread_local ViewStore g_view;

int64_t tfgpu_dbatch_nrows(const tfgpu_dbatch *b) { return b ? b->nrows : -1; }
int tfgpu_dbatch_dense(const tfgpu_dbatch *b) {
  TF_API_BEGIN
  if (!b) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_dense: null argument");
  tf::dense(b, true);
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_dbatch_view(const tfgpu_dbatch *b, tfgpu_batch *v) {
  TF_API_BEGIN
  tf::dense(b, true);  // its rows may still be a selection (tfgpu_dbatch::pending)
  if (!b || !v) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_view: null argument");
  { std::lock_guard<std::mutex> lk(ctx().mu); materialize(*b); }  // the view exposes packed payload pointers
  g_view.cols.assign(b->cols.size() + b->old_keys.size(), tfgpu_column{});
  for (size_t i = 0; i < b->cols.size() + b->old_keys.size(); i++) {
    const DColumn &d = i < b->cols.size() ? b->cols[i] : b->old_keys[i - b->cols.size()];
    tfgpu_column &c = g_view.cols[i];
    c.name = d.name.c_str(); c.dtype = d.dtype; c.repr = d.repr;
    c.values = d.values ? d.values->p : nullptr;
    c.offsets = ptr<uint32_t>(d.offsets); c.data = ptr<uint8_t>(d.payload()); c.data_len = d.data_len;
    c.nanos = ptr<int32_t>(d.nanos); c.validity = ptr<uint8_t>(d.validity); c.absent = ptr<uint8_t>(d.absent);
  }
  v->nrows = b->nrows; v->ncols = (int32_t)b->cols.size(); v->cols = g_view.cols.data();
  v->table_ns = b->ns.c_str(); v->table_name = b->table.c_str();
  v->kind = ptr<uint8_t>(b->kind); v->src_row = ptr<int32_t>(b->src_row); v->part_id = ptr<uint32_t>(b->part_id);
  v->mem = TFGPU_MEM_DEVICE;
  v->schema = nullptr;
  if (!b->schema.empty()) {  // the TableSchema, where it is not simply the columns: names, types, PrimaryKey flags
    g_view.scols.assign(b->schema.size(), tfgpu_colschema{});
    for (size_t i = 0; i < b->schema.size(); i++) {
      tfgpu_colschema &sc = g_view.scols[i];
      sc.name = b->schema[i].first.c_str(); sc.dtype = b->schema[i].second; sc.path = ""; sc.original_type = "";
      for (auto &k : b->key_names) if (k == b->schema[i].first) sc.flags |= TFGPU_COL_KEY;
    }
    g_view.schema.ncols = (int32_t)g_view.scols.size(); g_view.schema.cols = g_view.scols.data();
    v->schema = &g_view.schema;
  }
  v->n_old_keys = (int32_t)b->old_keys.size();
  v->old_keys = b->old_keys.empty() ? nullptr : g_view.cols.data() + b->cols.size();
  v->old_keys_present = ptr<uint8_t>(b->old_present);
  v->col_order = ptr<uint16_t>(b->col_order);
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_dbatch_download(const tfgpu_dbatch *b, tfgpu_batch *h) {
  TF_API_BEGIN
  tf::dense(b, true);  // its rows may still be a selection (tfgpu_dbatch::pending)
  if (!b || !h) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_download: null argument");
  if (h->ncols != (int32_t)b->cols.size() || h->nrows != b->nrows) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_download: shape mismatch");
  if (h->n_old_keys && h->n_old_keys != (int32_t)b->old_keys.size()) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_download: old-key shape mismatch");
  // rows that do not list every column, or list them in their own order, cannot come down as if they did: a caller that brought no room for the
  // bitmaps / the order (a binding older than tfgpu_column.absent) is told so instead of reading nils where the items had no column at all
  for (size_t i = 0; i < b->cols.size(); i++)
    if (b->cols[i].absent && !h->cols[i].absent) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_download: column " + b->cols[i].name + " holds ABSENT cells (see the view): host_out->cols[i].absent must point at (nrows + 7) / 8 bytes");
  if (b->col_order && !h->col_order) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_download: the rows carry their own ColumnNames order (see the view): host_out->col_order must point at nrows * ncols uint16");
  std::lock_guard<std::mutex> lk(ctx().mu);
  materialize(*b);
  size_t n = (size_t)b->nrows;
  for (size_t i = 0; i < b->cols.size() + (size_t)h->n_old_keys; i++) {
    const DColumn &d = i < b->cols.size() ? b->cols[i] : b->old_keys[i - b->cols.size()];
    tfgpu_column &c = i < b->cols.size() ? h->cols[i] : h->old_keys[i - b->cols.size()];
    if (repr_is_var(d.repr)) {
      if (c.offsets && d.offsets) d2h(c.offsets, d.offsets->p, (n + 1) * 4);
      if (c.data && d.payload()) d2h(c.data, d.payload()->p, d.data_len);
    } else {
      if (c.values && d.values) d2h(c.values, d.values->p, n * repr_width(d.repr));
      if (c.nanos && d.nanos) d2h(c.nanos, d.nanos->p, n * 4);
    }
    if (c.validity && d.validity) d2h(c.validity, d.validity->p, (n + 7) / 8);
    if (c.absent && d.absent) d2h(c.absent, d.absent->p, (n + 7) / 8);
  }
  if (h->kind && b->kind) d2h(h->kind, b->kind->p, n);
  if (h->src_row && b->src_row) d2h(h->src_row, b->src_row->p, n * 4);
  if (h->part_id && b->part_id) d2h(h->part_id, b->part_id->p, n * 4);
  if (h->old_keys_present && b->old_present) d2h(h->old_keys_present, b->old_present->p, (n + 7) / 8);
  if (h->col_order && b->col_order) d2h(h->col_order, b->col_order->p, n * b->cols.size() * 2);
  tf::sync();
  return TFGPU_OK;
  TF_API_END
}

void tfgpu_dbatch_free(tfgpu_dbatch *b) { delete b; }

// ---- raw byte buffers --------------------------------------------------------
int tfgpu_dbuf_upload(const void *host, uint64_t len, tfgpu_dbuf **out) {
  TF_API_BEGIN
  std::lock_guard<std::mutex> lk(ctx().mu);
  auto b = std::make_unique<tfgpu_dbuf>();
  b->size = len;
  b->mem = dalloc(len + 64);  // tail padding lets kernels issue whole 16-byte loads
  h2d(b->mem->p, host, len);
  TF_HIP(hipMemsetAsync((char *)b->mem->p + len, 0, 64, ctx().stream));
  tf::sync();
  register_device_block(b->mem);
  *out = b.release();
  return TFGPU_OK;
  TF_API_END
}
int tfgpu_dbuf_alloc(uint64_t len, tfgpu_dbuf **out) {
  TF_API_BEGIN
  std::lock_guard<std::mutex> lk(ctx().mu);
  auto b = std::make_unique<tfgpu_dbuf>();
  b->size = len;
  b->mem = dalloc(len + 64);
  TF_HIP(hipMemsetAsync((char *)b->mem->p + len, 0, 64, ctx().stream));
  register_device_block(b->mem);
  *out = b.release();
  return TFGPU_OK;
  TF_API_END
}
int tfgpu_dbuf_write(tfgpu_dbuf *b, uint64_t offset, const void *host, uint64_t len) {
  TF_API_BEGIN
  if (!b || offset + len > b->size) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbuf_write: out of range");
  std::lock_guard<std::mutex> lk(ctx().mu);
  if (b->mem.use_count() > 1) {  // a lazy text column still reads this text: write into a private copy
    Buf fresh = dalloc(b->size + 64);
    d2d(fresh->p, b->mem->p, b->size + 64);
    b->mem = fresh;
    register_device_block(b->mem);
  }
  h2d((char *)b->mem->p + offset, host, len);
  tf::sync();  // the staging buffer is reusable on return
  return TFGPU_OK;
  TF_API_END
}
int tfgpu_dbuf_size(const tfgpu_dbuf *b, uint64_t *out) { *out = b ? b->size : 0; return TFGPU_OK; }
void *tfgpu_dbuf_ptr(const tfgpu_dbuf *b) { return b && b->mem ? b->mem->p : nullptr; }
int tfgpu_dbuf_download(const tfgpu_dbuf *b, void *host, uint64_t cap) {
  TF_API_BEGIN
  if (!b) return tf::fail(TFGPU_ERR_INVALID, "null dbuf");
  std::lock_guard<std::mutex> lk(ctx().mu);
  d2h(host, b->mem->p, std::min<uint64_t>(cap, b->size));
  tf::sync();
  return TFGPU_OK;
  TF_API_END
}
void tfgpu_dbuf_free(tfgpu_dbuf *b) { delete b; }

// ---- profiler ---------------------------------------------------------------
int tfgpu_prof_enable(int on) {
  TF_API_BEGIN
  ctx().prof_flush();
  ctx().prof_on = on != 0;
  return TFGPU_OK;
  TF_API_END
}
int tfgpu_prof_reset(void) {
  TF_API_BEGIN
  ctx().prof_flush();
  ctx().prof.clear();
  return TFGPU_OK;
  TF_API_END
}
int tfgpu_prof_count(void) {
  if (!g_ctx || !g_lanes[g_lane]) return 0;
  g_lanes[g_lane]->prof_flush();
  return (int)g_lanes[g_lane]->prof.size();
}
int tfgpu_prof_get(int i, const char **name, int64_t *launches, double *total_ms) {
  Context *c = g_ctx ? g_lanes[g_lane].get() : nullptr;
  if (!c || i < 0 || i >= (int)c->prof.size()) return TFGPU_ERR_INVALID;
  *name = c->prof[i].name.c_str(); *launches = c->prof[i].launches; *total_ms = c->prof[i].total_ms;
  return TFGPU_OK;
}
int tfgpu_prof_get_units(int i, int64_t *units) {
  Context *c = g_ctx ? g_lanes[g_lane].get() : nullptr;
  if (!c || !units || i < 0 || i >= (int)c->prof.size()) return TFGPU_ERR_INVALID;
  *units = c->prof[i].units;
  return TFGPU_OK;
}

}  // extern "C"
