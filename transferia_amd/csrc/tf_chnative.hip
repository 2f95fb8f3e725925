// tf_chnative.hip — a device batch as one ClickHouse `Native` format block.
//
// The reference's v2 ClickHouse sink marshals every ChangeItem into a []any row (pkg/providers/clickhouse/async/
// marshaller.go:62-190: nil stays nil, text / []byte / `any` become strings, date / datetime are clamped to
// [1970-01-01, 2106-01-01] by columntypes.Restore (types.go:15-29, 92-104), everything else passes through
// abstract.Restore) and appends it to a clickhouse-go batch, whose columns (github.com/ClickHouse/clickhouse-go/v2
// v2.46.0 over ch-go v0.71.0 — third-party, not vendored in the reference) encode the Native column layout and send it.
// The batch is already columnar in HBM, so the sink-side work collapses to writing that layout directly:
//
//   block  = varuint(ncolumns) varuint(nrows) column*
//   column = string(name) string(type) [null map: nrows bytes, 1 = NULL, when Nullable(T)] data
//   data   = T little-endian, nrows values (a NULL or nil cell holds T's zero value);  String: varuint(len) bytes per row
//   string = varuint(len) bytes
//
// (the layout of `INSERT … FORMAT Native` over HTTP and of the Data packet's block body).  Value mapping, restating the
// driver's Append for the Go types the marshaller hands it: intN / uintN / floatN / bool → the same-width column;
// time.Time → Date = days(Unix()/86400) u16, Date32 = i32 days, DateTime = u32 Unix(), DateTime64(p) = i64
// UnixNano()/10^(9-p) (Go division: toward zero); nil into a non-Nullable column → the zero value (the driver's
// `case nil`).  Values outside a column's range make the driver's Append fail (DateOverflowError) and the push with it:
// here the call fails the same way, naming the first offending row.  PARITY UNPINNED for those range edges and for the
// layout itself: the encoder is a dependency of the reference, not part of it; the oracle restates the same published
// format and the tests decode the block with an independent reader.
//
// Kernels are pure HBM streaming: one lane per value, 4/8-byte loads and stores; text columns are a length pass, one
// segmented scan over all of them, and a copy pass.
#include "tf_devcol.hpp"

#include <cstring>

using namespace tf;

namespace {

enum ChBase { CB_INT8, CB_INT16, CB_INT32, CB_INT64, CB_UINT8, CB_UINT16, CB_UINT32, CB_UINT64, CB_FLOAT32, CB_FLOAT64, CB_BOOL, CB_STRING, CB_DATE, CB_DATE32, CB_DATETIME, CB_DATETIME64 };

struct ChType { ChBase base; bool nullable = false; int precision = 0; };

std::string trim(const std::string &s) {
  size_t a = 0, b = s.size();
  while (a < b && s[a] == ' ') a++;
  while (b > a && s[b - 1] == ' ') b--;
  return s.substr(a, b - a);
}

// "Nullable(DateTime64(6, 'UTC'))" → {DATETIME64, nullable, 6}
ChType parse_type(const std::string &full, const std::string &col) {
  ChType t{};
  std::string s = trim(full);
  auto bad = [&]() -> Error { return Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_ch_native_block: column " + col + ": ClickHouse type " + full + " has no device encoder"); };
  if (s.rfind("Nullable(", 0) == 0 && s.back() == ')') { t.nullable = true; s = trim(s.substr(9, s.size() - 10)); }
  std::string name = s, args;
  size_t p = s.find('(');
  if (p != std::string::npos) { if (s.back() != ')') throw bad(); name = trim(s.substr(0, p)); args = s.substr(p + 1, s.size() - p - 2); }
  static const struct { const char *n; ChBase b; } T[] = {{"Int8", CB_INT8}, {"Int16", CB_INT16}, {"Int32", CB_INT32}, {"Int64", CB_INT64}, {"UInt8", CB_UINT8},
      {"UInt16", CB_UINT16}, {"UInt32", CB_UINT32}, {"UInt64", CB_UINT64}, {"Float32", CB_FLOAT32}, {"Float64", CB_FLOAT64}, {"Bool", CB_BOOL}, {"String", CB_STRING},
      {"Date", CB_DATE}, {"Date32", CB_DATE32}, {"DateTime", CB_DATETIME}, {"DateTime64", CB_DATETIME64}};
  bool found = false;
  for (auto &e : T) if (name == e.n) { t.base = e.b; found = true; }
  if (!found) throw bad();
  if (t.base == CB_DATETIME64) {
    std::string a = trim(args.substr(0, args.find(',')));
    if (a.empty() || a.size() > 1 || a[0] < '0' || a[0] > '9') throw bad();
    t.precision = a[0] - '0';
  } else if (!args.empty() && t.base != CB_DATETIME) throw bad();  // DateTime('tz') is the only other parametrised form taken
  return t;
}

int base_width(ChBase b) {
  switch (b) {
    case CB_INT8: case CB_UINT8: case CB_BOOL: return 1;
    case CB_INT16: case CB_UINT16: case CB_DATE: return 2;
    case CB_INT32: case CB_UINT32: case CB_FLOAT32: case CB_DATE32: case CB_DATETIME: return 4;
    case CB_STRING: return 0;
    default: return 8;
  }
}

// ---- kernels ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) chn_nullmap(const uint8_t *validity, int64_t n, uint8_t *out) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) out[r] = validity ? (uint8_t)(!((validity[r >> 3] >> (r & 7)) & 1)) : 0;
}

// same-width copy with nil → 0.  `out` is only byte-aligned (it follows variable-length headers): packed stores.
template <typename T>
__global__ void __launch_bounds__(256) chn_copy(const T *in, const uint8_t *validity, int64_t n, uint8_t *out) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  struct __attribute__((packed, aligned(1))) P { T v; };
  T v = (validity && !((validity[r >> 3] >> (r & 7)) & 1)) ? T(0) : in[r];
  reinterpret_cast<P *>(out + (size_t)r * sizeof(T))->v = v;
}

struct TimeParams {
  const int64_t *sec; const int32_t *nanos; const uint8_t *validity; int64_t n; uint8_t *out;
  int base;           // ChBase
  int clamp;          // columntypes.Restore's [1970-01-01, 2106-01-01] clamp (YT date / datetime)
  int64_t scale;      // DateTime64: 10^(9-p)
  unsigned long long *bad;  // lowest offending row + 1 (atomicMin on ~0)
};
__global__ void __launch_bounds__(256) chn_time(TimeParams p) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.n) return;
  struct __attribute__((packed, aligned(1))) P16 { uint16_t v; };
  struct __attribute__((packed, aligned(1))) P32 { uint32_t v; };
  struct __attribute__((packed, aligned(1))) P64 { int64_t v; };
  const bool nil = p.validity && !((p.validity[r >> 3] >> (r & 7)) & 1);
  int64_t s = nil ? 0 : p.sec[r];
  int32_t ns = (nil || !p.nanos) ? 0 : p.nanos[r];
  const int64_t MAXD = 4291747200ll;  // 2106-01-01T00:00:00Z
  if (!nil && p.clamp) {
    if (s > MAXD || (s == MAXD && ns > 0)) { s = MAXD; ns = 0; }
    if (s < 0) { s = 0; ns = 0; }
  }
  bool bad = false;
  switch (p.base) {
    case CB_DATE: {  // driver: 1970-01-01 … 2149-06-06; ToDate = Unix()/86400 (Go division)
      bad = !nil && (s < 0 || s > 5662224000ll || (s == 5662224000ll && ns > 0));
      reinterpret_cast<P16 *>(p.out + (size_t)r * 2)->v = (uint16_t)(bad ? 0 : s / 86400);
      break;
    }
    case CB_DATE32: {  // 1900-01-01 … 2299-12-31
      bad = !nil && (s < -2208988800ll || s > 10413705600ll || (s == 10413705600ll && ns > 0));
      int64_t d = s / 86400;
      reinterpret_cast<P32 *>(p.out + (size_t)r * 4)->v = (uint32_t)(int32_t)(bad ? 0 : d);
      break;
    }
    case CB_DATETIME: {  // 1970-01-01 00:00:00 … 2106-02-07 06:28:15
      bad = !nil && (s < 0 || s > 4294967295ll || (s == 4294967295ll && ns > 0));
      reinterpret_cast<P32 *>(p.out + (size_t)r * 4)->v = (uint32_t)(bad ? 0 : s);
      break;
    }
    default: {  // DateTime64(p): 1900-01-01 … 2262-04-11 23:47:16; UnixNano()/scale, toward zero
      bad = !nil && (s < -2208988800ll || s > 9223372036ll || (s == 9223372036ll && ns > 0));
      int64_t v = 0;
      if (!bad && !nil) { int64_t un = s * 1000000000ll + ns; v = un / p.scale; }
      reinterpret_cast<P64 *>(p.out + (size_t)r * 8)->v = v;
    }
  }
  if (bad) atomicMin(p.bad, (unsigned long long)r + 1);
}

__device__ __forceinline__ uint32_t varuint_len(uint32_t v) { return v < (1u << 7) ? 1 : v < (1u << 14) ? 2 : v < (1u << 21) ? 3 : v < (1u << 28) ? 4 : 5; }

// lengths of all text columns in one launch: slot (c, r) = varuint(len) + len of row r of text column c
struct StrCol { const uint32_t *offsets; const uint8_t *data; const uint8_t *validity; uint64_t section; int32_t is_json; };
__global__ void __launch_bounds__(256) chn_str_len(const StrCol *cols, int64_t n, int64_t stride, uint32_t *lens, unsigned long long *bad_any) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const StrCol c = cols[blockIdx.y];
  const bool nil = c.validity && !((c.validity[r >> 3] >> (r & 7)) & 1);
  const uint32_t len = nil ? 0 : c.offsets[r + 1] - c.offsets[r];
  // columntypes.Restore hands an `any` value that is a Go string to the driver as that string, not as its JSON text
  if (c.is_json && len && c.data[c.offsets[r]] == '"') atomicMin(bad_any, (unsigned long long)r + 1);
  lens[(size_t)blockIdx.y * stride + r] = varuint_len(len) + len;
}
__global__ void __launch_bounds__(256) chn_str_write(const StrCol *cols, int64_t n, int64_t stride, const uint32_t *offs, uint8_t *out) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const StrCol c = cols[blockIdx.y];
  const bool nil = c.validity && !((c.validity[r >> 3] >> (r & 7)) & 1);
  const uint32_t a = c.offsets[r];
  uint32_t len = nil ? 0 : c.offsets[r + 1] - a;
  uint8_t *dst = out + c.section + offs[(size_t)blockIdx.y * stride + r];
  uint32_t v = len;
  while (v >= 0x80) { *dst++ = (uint8_t)(v | 0x80); v >>= 7; }
  *dst++ = (uint8_t)v;
  const uint8_t *src = c.data + a;
  struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
  uint32_t i = 0;
  for (; i + 8 <= len; i += 8) reinterpret_cast<U64 *>(dst + i)->v = reinterpret_cast<const U64 *>(src + i)->v;
  for (; i < len; i++) dst[i] = src[i];
}

void put_varuint(std::vector<uint8_t> &b, uint64_t v) {
  while (v >= 0x80) { b.push_back((uint8_t)(v | 0x80)); v >>= 7; }
  b.push_back((uint8_t)v);
}
void put_string(std::vector<uint8_t> &b, const std::string &s) { put_varuint(b, s.size()); b.insert(b.end(), s.begin(), s.end()); }

bool repr_fits(int repr, ChBase b) {
  switch (b) {
    case CB_INT8: return repr == TFGPU_R_INT8;
    case CB_INT16: return repr == TFGPU_R_INT16;
    case CB_INT32: return repr == TFGPU_R_INT32;
    case CB_INT64: return repr == TFGPU_R_INT64 || repr == TFGPU_R_DURATION;
    case CB_UINT8: return repr == TFGPU_R_UINT8 || repr == TFGPU_R_BOOL;
    case CB_UINT16: return repr == TFGPU_R_UINT16;
    case CB_UINT32: return repr == TFGPU_R_UINT32;
    case CB_UINT64: return repr == TFGPU_R_UINT64;
    case CB_FLOAT32: return repr == TFGPU_R_FLOAT32;
    case CB_FLOAT64: return repr == TFGPU_R_FLOAT64;
    case CB_BOOL: return repr == TFGPU_R_BOOL;
    case CB_STRING: return repr == TFGPU_R_STRING || repr == TFGPU_R_BYTES || repr == TFGPU_R_JSON;
    default: return repr == TFGPU_R_TIME;
  }
}

}  // namespace

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

extern "C" int tfgpu_ch_native_block(const tfgpu_dbatch *in, const tfgpu_ch_native_column *cols, int32_t ncols, tfgpu_dbuf **out) {
  TF_API_BEGIN
  tf::dense(in);  // its rows may still be a selection (tfgpu_dbatch::pending)
  if (!in || !out || ncols < 0 || (ncols && !cols)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_ch_native_block: bad argument");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  hipStream_t st = cx.stream;
  const int64_t n = in->nrows;
  struct Plan { const DColumn *c; ChType t; std::vector<uint8_t> head; uint64_t null_at = 0, data_at = 0, data_bytes = 0; int str_slot = -1; };
  std::vector<Plan> plan((size_t)ncols);
  std::vector<const DColumn *> text;
  for (int i = 0; i < ncols; i++) {
    Plan &p = plan[(size_t)i];
    if (!cols[i].name || !cols[i].ch_type) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_ch_native_block: null column name / type");
    p.c = nullptr;
    for (auto &c : in->cols) if (c.name == cols[i].name) p.c = &c;
    if (!p.c) return tf::fail(TFGPU_ERR_INVALID, std::string("tfgpu_ch_native_block: the batch has no column ") + cols[i].name);
    p.t = parse_type(cols[i].ch_type, cols[i].name);
    if (!repr_fits(p.c->repr, p.t.base))
      return tf::fail(TFGPU_ERR_UNSUPPORTED, std::string("tfgpu_ch_native_block: column ") + cols[i].name + ": its Go values do not append to " + cols[i].ch_type + " without a conversion");
    put_string(p.head, cols[i].name);
    put_string(p.head, cols[i].ch_type);
    if (p.t.base == CB_STRING) {
      if (p.c->data_len + 5ull * (uint64_t)n >= 0xFFFFFFF0ull)  // the encoded lengths are scanned in 32 bits
        return tf::fail(TFGPU_ERR_UNSUPPORTED, std::string("tfgpu_ch_native_block: column ") + cols[i].name + " exceeds 4 GiB of text; split the batch by rows");
      p.str_slot = (int)text.size(); text.push_back(p.c);
    }
  }
  if (!text.empty()) materialize(*in, &text);
  KernelTimer timer("ch_native_block");
  // text columns: encoded length of every cell, one segmented scan, totals read back once
  const int nstr = (int)text.size();
  const int64_t stride = ((n + 1 + 3) / 4) * 4;
  Buf lens, dstr;
  Buf bad = dalloc(16);
  TF_HIP(hipMemsetAsync(bad->p, 0xff, 16, st));
  std::vector<StrCol> hstr((size_t)std::max(nstr, 1));
  std::vector<uint64_t> str_bytes((size_t)std::max(nstr, 1), 0);
  if (nstr) {
    lens = dalloc_zero((size_t)nstr * (size_t)stride * 4 + 16);
    for (int k = 0; k < nstr; k++) hstr[(size_t)k] = StrCol{ptr<uint32_t>(text[(size_t)k]->offsets), ptr<uint8_t>(text[(size_t)k]->payload()), ptr<uint8_t>(text[(size_t)k]->validity), 0,
                                                                 text[(size_t)k]->repr == TFGPU_R_JSON ? 1 : 0};
    dstr = upload_small(hstr.data(), hstr.size() * sizeof(StrCol));
    if (n) {
      chn_str_len<<<dim3((unsigned)((n + 255) / 256), (unsigned)nstr), 256, 0, st>>>(ptr<StrCol>(dstr), n, stride, ptr<uint32_t>(lens), ptr<unsigned long long>(bad) + 1);
      exclusive_scan_u32_segments(ptr<uint32_t>(lens), n, nstr, stride);
      const uint32_t *tot = segment_totals_to_host(ptr<uint32_t>(lens), n, nstr, stride);
      sync();
      for (int k = 0; k < nstr; k++) str_bytes[(size_t)k] = tot[k];
    }
  }
  // layout
  std::vector<uint8_t> head0;
  put_varuint(head0, (uint64_t)ncols);
  put_varuint(head0, (uint64_t)n);
  uint64_t at = head0.size();
  for (auto &p : plan) {
    at += p.head.size();
    if (p.t.nullable) { p.null_at = at; at += (uint64_t)n; }
    p.data_at = at;
    p.data_bytes = p.t.base == CB_STRING ? str_bytes[(size_t)p.str_slot] : (uint64_t)n * (uint64_t)base_width(p.t.base);
    at += p.data_bytes;
  }
  auto res = std::make_unique<tfgpu_dbuf>();
  res->size = at;
  res->mem = dalloc(at + 64);
  uint8_t *o = ptr<uint8_t>(res->mem);
  // headers: one small host-built run per column
  {
    uint64_t w = 0;
    h2d_small(o, head0.data(), head0.size());
    w = head0.size();
    for (auto &p : plan) { h2d_small(o + w, p.head.data(), p.head.size()); w = p.data_at + p.data_bytes; }
  }
  const unsigned g = (unsigned)((n + 255) / 256);
  if (n) {
    for (auto &p : plan) {
      const DColumn &c = *p.c;
      const uint8_t *val = ptr<uint8_t>(c.validity);
      if (p.t.nullable) chn_nullmap<<<g, 256, 0, st>>>(val, n, o + p.null_at);
      uint8_t *d = o + p.data_at;
      switch (p.t.base) {
        case CB_INT8: case CB_UINT8: case CB_BOOL: chn_copy<uint8_t><<<g, 256, 0, st>>>(ptr<uint8_t>(c.values), val, n, d); break;
        case CB_INT16: case CB_UINT16: chn_copy<uint16_t><<<g, 256, 0, st>>>(ptr<uint16_t>(c.values), val, n, d); break;
        case CB_INT32: case CB_UINT32: case CB_FLOAT32: chn_copy<uint32_t><<<g, 256, 0, st>>>(ptr<uint32_t>(c.values), val, n, d); break;
        case CB_INT64: case CB_UINT64: case CB_FLOAT64: chn_copy<uint64_t><<<g, 256, 0, st>>>(ptr<uint64_t>(c.values), val, n, d); break;
        case CB_STRING: hstr[(size_t)p.str_slot].section = p.data_at; break;
        default: {
          int64_t scale = 1;
          for (int k = p.t.precision; k < 9; k++) scale *= 10;
          TimeParams tp{ptr<int64_t>(c.values), ptr<int32_t>(c.nanos), val, n, d, (int)p.t.base, (c.dtype == TFGPU_T_DATE || c.dtype == TFGPU_T_DATETIME) ? 1 : 0, scale,
                        ptr<unsigned long long>(bad)};
          chn_time<<<g, 256, 0, st>>>(tp);
        }
      }
    }
    if (nstr) {
      dstr = upload_small(hstr.data(), hstr.size() * sizeof(StrCol));
      chn_str_write<<<dim3(g, (unsigned)nstr), 256, 0, st>>>(ptr<StrCol>(dstr), n, stride, ptr<uint32_t>(lens), o);
    }
  }
  uint64_t first_bad_[2] = {0, 0};
  d2h(first_bad_, bad->p, 16);
  sync();
  const uint64_t first_bad = first_bad_[0];
  if (first_bad_[1] != ~0ull)
    return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_ch_native_block: row " + std::to_string(first_bad_[1] - 1) + " holds an `any` value that is a Go string (sent unquoted by the reference): host step");
  if (first_bad != ~0ull)
    return tf::fail(TFGPU_ERR_INVALID, "tfgpu_ch_native_block: row " + std::to_string(first_bad - 1) + " holds a time outside its ClickHouse column's range (the driver's DateOverflowError)");
  *out = res.release();
  return TFGPU_OK;
  TF_API_END
}
