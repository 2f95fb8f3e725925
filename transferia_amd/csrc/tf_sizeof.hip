// tf_sizeof.hip — util.DeepSizeof(ColumnValues) for every row of a device batch.
//
// The reference measures items by reflection over their boxed values (pkg/util/sizeof.go:7-93): measurer.AsyncPush sets
// Size.Values from it for the Bufferer's byte trigger (pkg/middlewares/synchronizer/measurer.go:38-42) and the s3 CSV
// reader sets Size.Read from it on every row (reader_csv.go:336).  A columnar batch has no boxes to walk, but the sizes
// are a function of (Go type, length) alone: 24 for the []interface{} header, then per column 16 for the interface header
// plus 0 (nil) | the scalar's size | 16 + len (string, json.Number) | 24 + len ([]byte) | 24 (time.Time: three word-sized
// unexported fields, sizeof.go:49-51) | the decoded JSON value's size for `any` (map: 8 + per entry 16 + len(key) + 16 +
// value; []interface{}: 24 + per element 16 + value; numbers json.Number or float64).
//
// One lane per row; HBM traffic is the offsets and validity bitmaps only (text payloads are never read, so lazily
// materialised CSV text stays lazy) except for `any` cells, whose JSON text is walked once.  Per-row sizes are optional;
// the total is a wave reduction + one atomic per wave.
#include "tf_devcol.hpp"
#include "tf_wave.hpp"

using namespace tf;

namespace {

constexpr int SZ_MAXCOLS = 1024;
constexpr int SZ_DEPTH = 64;

struct SzCol {
  const uint32_t *offsets;
  const uint8_t *data;      // R_JSON only
  const uint8_t *validity;
  const uint8_t *absent;    // rows that do not list the column (DColumn::absent): no interface header either
  uint32_t fixed;           // size of a non-nil value without its payload bytes
  uint32_t var;             // 1: add the cell's byte length; 2: walk the JSON text
};

__device__ inline int hexval(uint8_t c) { return c <= '9' ? c - '0' : (c | 0x20) - 'a' + 10; }

// DeepSizeof of the value encoding/json decodes from valid JSON text.  `depth` containers at most SZ_DEPTH deep
// (deeper: *overflow set, the caller reports the row).  Strings: 16 + decoded length (escapes collapse, \uXXXX takes its
// UTF-8 length, a surrogate pair 4, a lone surrogate U+FFFD = 3).
__device__ uint64_t json_deepsize(const uint8_t *p, uint32_t n, bool float_numbers, bool *overflow) {
  uint64_t size = 0;
  uint64_t is_obj = 0;   // bit d: container at depth d is an object
  uint64_t want_key = 0; // bit d: the next string at depth d is a key
  int depth = 0;
  uint32_t i = 0;
  while (i < n) {
    uint8_t c = p[i];
    if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == ':') { i++; continue; }
    if (c == ',') { if (depth && ((is_obj >> (depth - 1)) & 1)) want_key |= 1ull << (depth - 1); i++; continue; }
    if (c == '{' || c == '[') {
      if (depth) size += 16;  // the interface header of an array element / a map value
      size += c == '{' ? 8 : 24;
      if (depth >= SZ_DEPTH) { *overflow = true; return 0; }
      if (c == '{') { is_obj |= 1ull << depth; want_key |= 1ull << depth; } else { is_obj &= ~(1ull << depth); want_key &= ~(1ull << depth); }
      depth++; i++;
      continue;
    }
    if (c == '}' || c == ']') { depth--; i++; continue; }
    const bool key = depth && ((want_key >> (depth - 1)) & 1);
    uint64_t v;
    if (c == '"') {
      uint32_t len = 0;
      i++;
      while (i < n && p[i] != '"') {
        if (p[i] == '\\' && i + 1 < n) {
          if (p[i + 1] == 'u' && i + 5 < n) {
            unsigned cp = (unsigned)(hexval(p[i + 2]) << 12 | hexval(p[i + 3]) << 8 | hexval(p[i + 4]) << 4 | hexval(p[i + 5]));
            i += 6;
            if (cp >= 0xD800 && cp < 0xDC00 && i + 5 < n && p[i] == '\\' && p[i + 1] == 'u') {
              unsigned lo = (unsigned)(hexval(p[i + 2]) << 12 | hexval(p[i + 3]) << 8 | hexval(p[i + 4]) << 4 | hexval(p[i + 5]));
              if (lo >= 0xDC00 && lo < 0xE000) { i += 6; len += 4; continue; }
            }
            len += cp < 0x80 ? 1 : cp < 0x800 ? 2 : 3;  // lone surrogates decode to U+FFFD (3 bytes)
          } else { i += 2; len += 1; }
        } else { i++; len++; }
      }
      i++;
      v = 16 + (uint64_t)len;
    } else if (c == 't') { i += 4; v = 1; }
    else if (c == 'f') { i += 5; v = 1; }
    else if (c == 'n') { i += 4; v = 0; }
    else {
      uint32_t s = i;
      while (i < n && p[i] != ',' && p[i] != '}' && p[i] != ']' && p[i] != ' ' && p[i] != '\t' && p[i] != '\n' && p[i] != '\r') i++;
      v = float_numbers ? 8 : 16 + (uint64_t)(i - s);
    }
    if (key) { size += v; want_key &= ~(1ull << (depth - 1)); }  // DeepSizeof(key): the string alone
    else size += (depth ? 16 : 0) + v;
  }
  return size;
}

__global__ void __launch_bounds__(256) deepsizeof_rows(const SzCol *cols, int ncols, int64_t n, uint32_t flags, uint64_t *per_row, unsigned long long *total, uint32_t *deep) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t s = 0;
  if (r < n) {
    s = 24 + 16ull * (uint64_t)ncols;
    for (int c = 0; c < ncols; c++) {
      const SzCol k = cols[c];
      if (k.absent && ((k.absent[r >> 3] >> (r & 7)) & 1)) { s -= 16; continue; }  // ColumnValues is that much shorter
      if (k.validity && !((k.validity[r >> 3] >> (r & 7)) & 1)) continue;
      s += k.fixed;
      if (k.var == 1) s += k.offsets[r + 1] - k.offsets[r];
      else if (k.var == 2) {
        bool over = false;
        s += json_deepsize(k.data + k.offsets[r], k.offsets[r + 1] - k.offsets[r], flags & 1u, &over);
        if (over) atomicAdd(deep, 1u);
      }
    }
    if (per_row) per_row[r] = s;
  }
  // wave sum, one atomic per wave
  uint64_t w = s;
  for (int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if ((threadIdx.x & 63) == 0 && w) atomicAdd(total, (unsigned long long)w);
}

}  // namespace

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

extern "C" int tfgpu_dbatch_deepsizeof(const tfgpu_dbatch *in, uint32_t flags, uint64_t *per_row, uint64_t *total) {
  TF_API_BEGIN
  tf::dense(in, true);  // its rows may still be a selection (tfgpu_dbatch::pending); a row's ColumnValues hold the columns it lists
  if (!in || !total) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbatch_deepsizeof: null argument");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  const int64_t n = in->nrows;
  const int ncols = (int)in->cols.size();
  if (ncols > SZ_MAXCOLS) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_dbatch_deepsizeof: more than 1024 columns");
  *total = 0;
  if (!n) return TFGPU_OK;
  std::vector<SzCol> h((size_t)std::max(ncols, 1));
  {
    std::vector<const DColumn *> need;  // only `any` cells are read; text stays unmaterialised
    for (auto &c : in->cols) if (c.repr == TFGPU_R_JSON) need.push_back(&c);
    if (!need.empty()) materialize(*in, &need);
  }
  for (int c = 0; c < ncols; c++) {
    const DColumn &d = in->cols[(size_t)c];
    SzCol &k = h[(size_t)c];
    k.offsets = ptr<uint32_t>(d.offsets);
    k.data = nullptr;
    k.validity = ptr<uint8_t>(d.validity);
    k.absent = ptr<uint8_t>(d.absent);
    k.var = 0;
    switch (d.repr) {
      case TFGPU_R_STRING: case TFGPU_R_JSONNUM: k.fixed = 16; k.var = 1; break;
      case TFGPU_R_BYTES: k.fixed = 24; k.var = 1; break;
      case TFGPU_R_JSON: k.fixed = 0; k.var = 2; k.data = ptr<uint8_t>(d.payload()); break;
      case TFGPU_R_TIME: k.fixed = 24; break;
      default:
        k.fixed = (uint32_t)repr_width(d.repr);
        if (!k.fixed) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_dbatch_deepsizeof: column " + d.name + " has no Go value form");
    }
  }
  KernelTimer t("deepsizeof_rows");
  Buf dcols = upload_small(h.data(), h.size() * sizeof(SzCol));
  Buf acc = dalloc_zero(16);
  Buf rows = per_row ? dalloc((size_t)n * 8) : Buf();
  deepsizeof_rows<<<(unsigned)((n + 255) / 256), 256, 0, cx.stream>>>(ptr<SzCol>(dcols), ncols, n, flags, ptr<uint64_t>(rows), ptr<unsigned long long>(acc),
                                                                   ptr<uint32_t>(acc) + 2);
  uint64_t back[2];
  d2h(back, ptr<uint8_t>(acc), 16);
  if (per_row) d2h(per_row, ptr<uint8_t>(rows), (size_t)n * 8);
  sync();
  if ((uint32_t)back[1]) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_dbatch_deepsizeof: an `any` value nests deeper than 64 containers");
  *total = back[0];
  return TFGPU_OK;
  TF_API_END
}
