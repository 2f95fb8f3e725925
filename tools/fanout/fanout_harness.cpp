// fanout_harness.cpp — MEASUREMENT TOOL (bench.py --workload configs0 --from-rows / --workload fanin_hits): what the drop-in boundary costs
// when it is crossed the way the reference would cross it.
//
// transformation.do hands a transformer `[]abstract.ChangeItem` (pkg/transformer/transformation.go:252-257): every row a struct
// with a ColumnValues []interface{} — one (type, pointer) pair per cell, the value boxed on the heap.  The cgo binding of
// INTEGRATION.md §2 (plan.Apply) fans such a run OUT into the column buffers of a tfgpu_batch (one type switch per cell, values /
// offsets / validity written into pooled staging), crosses the C ABI once (tfgpu_batch_upload → tfgpu_apply → tfgpu_dbatch_download)
// and fans the result back IN (a new item per kept row: untouched cells re-use the input's boxed values by src_row, rewritten
// columns are boxed afresh).  There is no Go toolchain here, so this file restates that binding in C++ over a faithful model of
// the Go data: 16-byte interface words pointing at individually allocated boxes, a 24-byte slice header per row, items of ~200 bytes.
// It times every leg separately.  SURVEY §7 "hard part #1": ~10^8 type switches may dominate — this measures it.
//
// Not product code and not the oracle: it links nothing of either; libtfgpu.so is looked up in the process (the caller has loaded it).
#include <dlfcn.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tfgpu.h"

namespace {

// ---- the Go side, modelled -----------------------------------------------------------------------------------------------
enum Tag : uint32_t { T_NIL = 0, T_INT64, T_INT32, T_INT16, T_STRING, T_TIME, T_DATE /* a time.Time too, under a `date` column */ };   // the dynamic type word of an interface value
struct GoString { const char *p; int64_t n; };
struct GoTime { int64_t sec; int32_t nsec; int32_t pad; void *loc; };              // 24 bytes, like time.Time
struct Iface { uintptr_t type; void *data; };                                     // 16 bytes: (itab / type, pointer to the box)
struct Item {                                                                      // abstract.ChangeItem, the fields a transformer touches + padding to its size
  uint32_t id; uint64_t lsn, commit_time; int64_t counter; int kind;
  GoString schema, table, part_id;
  const std::vector<std::string> *column_names;   // shared by the run (a slice header pointing at one backing array)
  Iface *values; int64_t nvalues, cap;            // ColumnValues []interface{}
  const void *table_schema;
  uint8_t rest[72];                               // OldKeys, Size, TxID, Query, QueueMessageMeta
};

template <class T> void *box(T v) { T *p = (T *)std::malloc(sizeof(T)); *p = v; return p; }
void *box_string(const char *s, size_t n) {
  char *bytes = (char *)std::malloc(n ? n : 1);
  std::memcpy(bytes, s, n);
  return box(GoString{bytes, (int64_t)n});
}

struct Api {
  decltype(&tfgpu_plan_create) plan_create; decltype(&tfgpu_plan_destroy) plan_free; decltype(&tfgpu_batch_upload) upload; decltype(&tfgpu_apply) apply;
  decltype(&tfgpu_dbatch_view) view; decltype(&tfgpu_dbatch_download) download; decltype(&tfgpu_dbatch_free) dfree; decltype(&tfgpu_host_alloc) halloc;
  decltype(&tfgpu_host_free) hfree; decltype(&tfgpu_synchronize) sync; decltype(&tfgpu_last_error) last_error;
};
bool bind(const char *libpath, Api &a, std::string &why) {
  void *h = dlopen(libpath, RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen(libpath, RTLD_NOW);
  if (!h) { why = dlerror(); return false; }
#define B(field, name) a.field = (decltype(a.field))dlsym(h, name); if (!a.field) { why = std::string("missing ") + name; return false; }
  B(plan_create, "tfgpu_plan_create") B(plan_free, "tfgpu_plan_destroy") B(upload, "tfgpu_batch_upload") B(apply, "tfgpu_apply") B(view, "tfgpu_dbatch_view")
  B(download, "tfgpu_dbatch_download") B(dfree, "tfgpu_dbatch_free") B(halloc, "tfgpu_host_alloc") B(hfree, "tfgpu_host_free") B(sync, "tfgpu_synchronize")
  B(last_error, "tfgpu_last_error")
#undef B
  return true;
}

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct ColSpec { std::string name; int dtype, repr; Tag tag; };

// deterministic cell values
inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// one staged column of the fan-out (pinned staging from tfgpu_host_alloc, pooled across calls like the binding's arena)
struct Staged { void *values = nullptr; uint32_t *offsets = nullptr; uint8_t *data = nullptr; int32_t *nanos = nullptr; uint8_t *validity = nullptr; size_t data_cap = 0; };

}  // namespace

// rows of `ncols` columns (specs), fanned out for the columns listed in `touch` only ("only materialise columns the chain touches",
// SURVEY §7), through the chain (plan type / JSON config pairs), fanned back in.  Returns 0 and a JSON object of per-leg seconds.
extern "C" int fanout_run(const char *libpath, int64_t nrows, int32_t ncols, const char *const *col_names, const int32_t *col_tags, const int32_t *touch, int32_t ntouch,
                          const char *const *plan_types, const char *const *plan_configs, int32_t nplans, const char *table_ns, const char *table_name, int32_t repeats,
                          char *out_json, size_t out_cap) {
  Api api; std::string why;
  if (!bind(libpath, api, why)) { std::snprintf(out_json, out_cap, "{\"error\": \"%s\"}", why.c_str()); return 1; }
  static const int dtype_of[] = {TFGPU_T_ANY, TFGPU_T_INT64, TFGPU_T_INT32, TFGPU_T_INT16, TFGPU_T_UTF8, TFGPU_T_TIMESTAMP, TFGPU_T_DATE};
  static const int repr_of[] = {TFGPU_R_STRING, TFGPU_R_INT64, TFGPU_R_INT32, TFGPU_R_INT16, TFGPU_R_STRING, TFGPU_R_TIME, TFGPU_R_TIME};
  std::vector<ColSpec> cols;
  std::vector<std::string> names;
  for (int c = 0; c < ncols; c++) { cols.push_back(ColSpec{col_names[c], dtype_of[col_tags[c]], repr_of[col_tags[c]], (Tag)col_tags[c]}); names.push_back(col_names[c]); }

  // ---- the input run: nrows items, every cell its own heap box (built once; not a timed leg — a source produced it) ----
  double t0 = now();
  std::vector<Item> items((size_t)nrows);
  for (int64_t r = 0; r < nrows; r++) {
    Item &it = items[(size_t)r];
    std::memset(&it, 0, sizeof it);
    it.lsn = (uint64_t)r + 5; it.commit_time = 1700000000000000000ull; it.counter = r; it.column_names = &names;
    it.schema = GoString{table_ns, (int64_t)std::strlen(table_ns)}; it.table = GoString{table_name, (int64_t)std::strlen(table_name)};
    it.values = (Iface *)std::malloc(sizeof(Iface) * (size_t)ncols); it.nvalues = it.cap = ncols;
    for (int c = 0; c < ncols; c++) {
      const uint64_t h = mix((uint64_t)r * 1315423911ull + (uint64_t)c);
      Iface &v = it.values[c];
      v.type = cols[(size_t)c].tag;
      switch (cols[(size_t)c].tag) {
        case T_INT64: v.data = box<int64_t>((int64_t)(h >> 8)); break;
        case T_INT32: v.data = box<int32_t>((int32_t)h); break;
        case T_INT16: v.data = box<int16_t>((int16_t)h); break;
        case T_STRING: { char buf[40]; const int n = std::snprintf(buf, sizeof buf, "v%llu", (unsigned long long)(h % 100000000ull)); v.data = box_string(buf, (size_t)(h % 7 == 0 ? 0 : n)); break; }
        case T_TIME: v.data = box(GoTime{1372636800 + (int64_t)(h % 2678400), 0, 0, nullptr}); break;
        case T_DATE: v.data = box(GoTime{1372636800 + (int64_t)(h % 31) * 86400, 0, 0, nullptr}); break;   // July 2013, midnight
        default: v.type = T_NIL; v.data = nullptr;
      }
    }
  }
  const double t_build = now() - t0;

  std::vector<tfgpu_plan *> plans((size_t)nplans);
  for (int i = 0; i < nplans; i++) if (api.plan_create(plan_types[i], plan_configs[i], &plans[(size_t)i]) != 0) { std::snprintf(out_json, out_cap, "{\"error\": \"plan %d: %s\"}", i, api.last_error()); return 1; }

  std::vector<Staged> st((size_t)ntouch);
  double s_out = 0, s_up = 0, s_apply = 0, s_down = 0, s_in = 0;
  int64_t kept = 0, boxes = 0, switches = 0;
  for (int rep = 0; rep < repeats + 1; rep++) {  // the first round warms the pools and is not counted
    // ---- fanOut: one type switch per cell of the touched columns ----
    t0 = now();
    std::vector<tfgpu_column> hc((size_t)ntouch);
    bool ok = true;
    for (int k = 0; k < ntouch && ok; k++) {
      const ColSpec &cs = cols[(size_t)touch[k]];
      Staged &s = st[(size_t)k];
      tfgpu_column &c = hc[(size_t)k];
      std::memset(&c, 0, sizeof c);
      c.name = cs.name.c_str(); c.dtype = cs.dtype; c.repr = cs.repr;
      if (!s.validity) api.halloc((size_t)(nrows + 7) / 8 + 8, (void **)&s.validity);
      std::memset(s.validity, 0, (size_t)(nrows + 7) / 8);
      const Tag want = cs.tag;   // the first row fixes the column's representation; another dynamic type sends the run to the stock path
      switch (want) {
        case T_INT64: case T_INT32: case T_INT16: {
          const size_t w = want == T_INT64 ? 8 : want == T_INT32 ? 4 : 2;
          if (!s.values) api.halloc((size_t)nrows * w + 8, &s.values);
          for (int64_t r = 0; r < nrows; r++) {
            const Iface &v = items[(size_t)r].values[touch[k]];
            switches++;
            if (v.type == T_NIL) continue;
            if (v.type != want) { ok = false; break; }
            if (w == 8) ((int64_t *)s.values)[r] = *(const int64_t *)v.data; else if (w == 4) ((int32_t *)s.values)[r] = *(const int32_t *)v.data; else ((int16_t *)s.values)[r] = *(const int16_t *)v.data;
            s.validity[r >> 3] |= (uint8_t)(1u << (r & 7));
          }
          c.values = s.values;
          break;
        }
        case T_TIME: case T_DATE: {
          if (!s.values) { api.halloc((size_t)nrows * 8 + 8, &s.values); api.halloc((size_t)nrows * 4 + 8, (void **)&s.nanos); }
          for (int64_t r = 0; r < nrows; r++) {
            const Iface &v = items[(size_t)r].values[touch[k]];
            switches++;
            if (v.type == T_NIL) continue;
            if (v.type != want) { ok = false; break; }
            const GoTime *t = (const GoTime *)v.data;
            ((int64_t *)s.values)[r] = t->sec; s.nanos[r] = t->nsec;
            s.validity[r >> 3] |= (uint8_t)(1u << (r & 7));
          }
          c.values = s.values; c.nanos = s.nanos;
          break;
        }
        default: {  // strings: lengths first (offsets), then the bytes
          if (!s.offsets) api.halloc((size_t)(nrows + 1) * 4 + 8, (void **)&s.offsets);
          uint64_t tot = 0;
          for (int64_t r = 0; r < nrows; r++) {
            const Iface &v = items[(size_t)r].values[touch[k]];
            switches++;
            s.offsets[r] = (uint32_t)tot;
            if (v.type == T_NIL) continue;
            if (v.type != T_STRING) { ok = false; break; }
            tot += (uint64_t)((const GoString *)v.data)->n;
            s.validity[r >> 3] |= (uint8_t)(1u << (r & 7));
          }
          s.offsets[nrows] = (uint32_t)tot;
          if (tot + 8 > s.data_cap) { if (s.data) api.hfree(s.data); s.data_cap = (size_t)tot + (tot >> 2) + 64; api.halloc(s.data_cap, (void **)&s.data); }
          for (int64_t r = 0; r < nrows && ok; r++) {
            const Iface &v = items[(size_t)r].values[touch[k]];
            if (v.type != T_STRING) continue;
            const GoString *g = (const GoString *)v.data;
            std::memcpy(s.data + s.offsets[r], g->p, (size_t)g->n);
          }
          c.offsets = s.offsets; c.data = s.data; c.data_len = tot;
        }
      }
      c.validity = s.validity;
    }
    if (!ok) { std::snprintf(out_json, out_cap, "{\"error\": \"a cell's dynamic type differs from its column's: the stock path\"}"); return 1; }
    tfgpu_batch hb; std::memset(&hb, 0, sizeof hb);
    hb.nrows = nrows; hb.ncols = ntouch; hb.cols = hc.data(); hb.table_ns = table_ns; hb.table_name = table_name; hb.mem = TFGPU_MEM_HOST;
    const double d_out = now() - t0;

    // ---- one crossing: upload, the chain, download ----
    t0 = now();
    tfgpu_dbatch *in = nullptr, *out = nullptr;
    if (api.upload(&hb, &in) != 0) { std::snprintf(out_json, out_cap, "{\"error\": \"upload: %s\"}", api.last_error()); return 1; }
    api.sync();
    const double d_up = now() - t0;
    t0 = now();
    std::vector<tfgpu_row_error> errs(1024); int64_t nerr = 0;
    if (api.apply(plans.data(), nplans, in, &out, errs.data(), (int64_t)errs.size(), &nerr) != 0) { std::snprintf(out_json, out_cap, "{\"error\": \"apply: %s\"}", api.last_error()); return 1; }
    api.sync();
    const double d_apply = now() - t0;
    t0 = now();
    tfgpu_batch v; std::memset(&v, 0, sizeof v);
    if (api.view(out, &v) != 0) { std::snprintf(out_json, out_cap, "{\"error\": \"view: %s\"}", api.last_error()); return 1; }
    const int64_t m = v.nrows;
    std::vector<tfgpu_column> oc((size_t)v.ncols);
    std::vector<std::vector<uint8_t>> bufs;
    auto buf = [&](size_t n) { bufs.emplace_back(n + 8); return bufs.back().data(); };
    for (int c = 0; c < v.ncols; c++) {
      oc[(size_t)c] = v.cols[c];
      tfgpu_column &o = oc[(size_t)c];
      const bool var = o.repr == TFGPU_R_STRING || o.repr == TFGPU_R_BYTES || o.repr == TFGPU_R_JSON || o.repr == TFGPU_R_JSONNUM;
      if (var) { o.offsets = (uint32_t *)buf((size_t)(m + 1) * 4); o.data = buf((size_t)o.data_len); o.values = nullptr; }
      else { const size_t w = (o.repr == TFGPU_R_INT16 || o.repr == TFGPU_R_UINT16) ? 2 : (o.repr == TFGPU_R_INT32 || o.repr == TFGPU_R_UINT32 || o.repr == TFGPU_R_FLOAT32) ? 4 : (o.repr == TFGPU_R_INT8 || o.repr == TFGPU_R_UINT8 || o.repr == TFGPU_R_BOOL) ? 1 : 8;
             o.values = buf((size_t)m * w); if (o.nanos) o.nanos = (int32_t *)buf((size_t)m * 4); }
      if (o.validity) o.validity = buf((size_t)(m + 7) / 8);
    }
    tfgpu_batch ho = v;
    ho.cols = oc.data(); ho.mem = TFGPU_MEM_HOST; ho.n_old_keys = 0; ho.old_keys = nullptr; ho.old_keys_present = nullptr;
    std::vector<int32_t> src((size_t)std::max<int64_t>(m, 1));
    ho.src_row = v.src_row ? src.data() : nullptr; ho.kind = nullptr; ho.part_id = nullptr;
    if (api.download(out, &ho) != 0) { std::snprintf(out_json, out_cap, "{\"error\": \"download: %s\"}", api.last_error()); return 1; }
    const double d_down = now() - t0;

    // ---- fanIn: a new item per kept row; untouched cells re-use the input's boxes by src_row, rewritten columns are boxed ----
    t0 = now();
    std::vector<Item> res((size_t)m);
    std::vector<int> out_of((size_t)ncols, -1);   // input column → output column that replaces it (by name)
    for (int c = 0; c < ncols; c++) for (int oc_i = 0; oc_i < v.ncols; oc_i++) if (cols[(size_t)c].name == oc[(size_t)oc_i].name) out_of[(size_t)c] = oc_i;
    int64_t nb = 0;
    for (int64_t i = 0; i < m; i++) {
      const int64_t sr = ho.src_row ? src[(size_t)i] : i;
      const Item &from = items[(size_t)sr];
      Item &it = res[(size_t)i];
      it = from;                                                             // ID / LSN / CommitTime / OldKeys … ride on src_row
      it.table = GoString{v.table_name, (int64_t)std::strlen(v.table_name)}; it.schema = GoString{v.table_ns, (int64_t)std::strlen(v.table_ns)};
      it.values = (Iface *)std::malloc(sizeof(Iface) * (size_t)ncols);
      for (int c = 0; c < ncols; c++) {
        const int o = out_of[(size_t)c];
        bool rewritten = false;
        if (o >= 0) { for (int k = 0; k < ntouch; k++) if (touch[k] == c) rewritten = oc[(size_t)o].repr != cols[(size_t)c].repr || oc[(size_t)o].dtype != cols[(size_t)c].dtype; }
        if (!rewritten) { it.values[c] = from.values[c]; continue; }        // the boxed value is shared, as Go would share it
        const tfgpu_column &col = oc[(size_t)o];
        if (col.validity && !((col.validity[i >> 3] >> (i & 7)) & 1)) { it.values[c] = Iface{T_NIL, nullptr}; continue; }
        const uint32_t a = col.offsets[i], b = col.offsets[i + 1];
        it.values[c] = Iface{T_STRING, box_string((const char *)col.data + a, (size_t)(b - a))};
        nb++;
      }
    }
    const double d_in = now() - t0;
    // (the garbage collector's share is not modelled; the new boxes are freed outside the timed legs)
    for (int64_t i = 0; i < m; i++) {
      const int64_t sr = ho.src_row ? src[(size_t)i] : i;
      for (int c = 0; c < ncols; c++) if (res[(size_t)i].values[c].data && res[(size_t)i].values[c].data != items[(size_t)sr].values[c].data) { GoString *g = (GoString *)res[(size_t)i].values[c].data; std::free((void *)g->p); std::free(g); }
      std::free(res[(size_t)i].values);
    }
    api.dfree(in); api.dfree(out);
    if (rep > 0) { s_out += d_out; s_up += d_up; s_apply += d_apply; s_down += d_down; s_in += d_in; }
    kept = m; boxes = nb;
  }
  for (auto &s : st) { if (s.values) api.hfree(s.values); if (s.offsets) api.hfree(s.offsets); if (s.data) api.hfree(s.data); if (s.nanos) api.hfree(s.nanos); if (s.validity) api.hfree(s.validity); }
  for (auto *p : plans) api.plan_free(p);
  for (auto &it : items) {
    for (int c = 0; c < ncols; c++) { if (it.values[c].type == T_STRING) std::free((void *)((GoString *)it.values[c].data)->p); std::free(it.values[c].data); }
    std::free(it.values);
  }
  const double R = repeats;
  std::snprintf(out_json, out_cap,
                "{\"rows\": %lld, \"columns\": %d, \"columns_fanned_out\": %d, \"rows_out\": %lld, \"type_switches_per_call\": %lld, \"cells_boxed_per_call\": %lld, \"repeats\": %d, "
                "\"ms\": {\"fan_out\": %.3f, \"upload\": %.3f, \"apply\": %.3f, \"download\": %.3f, \"fan_in\": %.3f}, \"build_input_s\": %.2f}",
                (long long)nrows, ncols, ntouch, (long long)kept, (long long)(switches / (repeats + 1)), (long long)boxes, repeats,
                s_out / R * 1e3, s_up / R * 1e3, s_apply / R * 1e3, s_down / R * 1e3, s_in / R * 1e3, t_build);
  return 0;
}
