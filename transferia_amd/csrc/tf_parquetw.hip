// tf_parquetw.hip — device columns → a Parquet object (the emit half; the S3 sink's `parquet` output format).
//
// Reference: pkg/serializer/parquet.go:53-200 (parquetBatchSerializer / parquetStreamSerializer over parquet-go's GenericWriter,
// CodecFromString: SNAPPY | GZIP | ZSTD | uncompressed, row groups cut by RowGroupMaxRows / RowGroupMaxBytes) and
// pkg/serializer/parquet_format.go:13-135 (BuildParquetSchema: a group "table" whose fields are the TableSchema's columns — a Go
// map, so parquet-go orders them BY NAME — typed by primitiveTypesMap, OPTIONAL unless ColSchema.Required; toParquetValue: nil is
// a null, float64 travels as its decimal text, `any` as json.Marshal text).  The reference boxes one parquet.Value per cell and
// lets the library lay pages out; here a column is already an array:
//
//   device  per column: the rows that hold a value get the ordinal of their value (validity → exclusive scan), the values are
//           packed in Parquet's PLAIN form in one pass — fixed widths widened to the physical type (int8/16 → INT32, time.Time →
//           days | nanoseconds), byte arrays behind their 4-byte lengths — and come back in one copy per column;
//   host    definition levels (the validity bitmap IS the bit-packed run of 1-bit levels), page headers and the footer
//           (Thrift compact protocol), the codec (SNAPPY: the raw format with a greedy 4-byte-hash matcher; GZIP / ZSTD through
//           the system's libz / libzstd, bound at first use), row groups cut at multiples of eight rows.
//
// One data page (v1, PLAIN, RLE levels) per column chunk.  PARITY UNPINNED against the reference: the byte layout of a Parquet
// object is the writing library's choice (parquet-go is not under /root/reference, the one canon file holds a schema of one
// string column); what is pinned (tests/test_parquet_write.py) is that pyarrow reads back every value, null, logical type and
// the field order, and that tfgpu_parquet_read round-trips it.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "tf_common.hpp"

namespace tf {
namespace pqw {

// ---- Thrift compact protocol, writing ---------------------------------------------------------------------------------------
struct TWriter {
  std::string out;
  std::vector<int> last{0};
  void byte(uint8_t b) { out.push_back((char)b); }
  void varint(uint64_t v) { while (v >= 0x80) { byte((uint8_t)(v | 0x80)); v >>= 7; } byte((uint8_t)v); }
  void zz(int64_t v) { varint(((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
  void field(int id, int type) {
    const int d = id - last.back();
    if (d > 0 && d <= 15) byte((uint8_t)(d << 4 | type)); else { byte((uint8_t)type); zz(id); }
    last.back() = id;
  }
  void i32(int id, int64_t v) { field(id, 5); zz(v); }
  void i64(int id, int64_t v) { field(id, 6); zz(v); }
  void i8(int id, int v) { field(id, 3); byte((uint8_t)v); }
  void boolean(int id, bool v) { field(id, v ? 1 : 2); }
  void binary(int id, const std::string &s) { field(id, 8); varint(s.size()); out += s; }
  void begin(int id) { field(id, 12); last.push_back(0); }
  void begin_elem() { last.push_back(0); }  // a struct that is a list element
  void end() { byte(0); last.pop_back(); }
  void list(int id, int etype, size_t n) { field(id, 9); if (n < 15) byte((uint8_t)(n << 4 | etype)); else { byte((uint8_t)(0xF0 | etype)); varint(n); } }
};

enum { T_BOOLEAN = 0, T_INT32 = 1, T_INT64 = 2, T_FLOAT = 4, T_BYTE_ARRAY = 6 };
enum { C_UNCOMPRESSED = 0, C_SNAPPY = 1, C_GZIP = 2, C_ZSTD = 6 };
// what a column is in the file
struct Leaf {
  std::string name;
  int dtype = 0, phys = 0, width = 0;   // physical type, bytes of a PLAIN value (0: byte array)
  bool required = false;
  int col = -1;                         // index in the batch, -1: every row is nil
};

// ---- codecs (host) ------------------------------------------------------------------------------------------------------------
// Snappy's raw format: varint length, then literals and copies found with a 4-byte hash over 64 KiB blocks (every copy offset
// fits two bytes, as the format's reference compressor arranges it).
static void snappy_literal(std::string &o, const uint8_t *p, size_t n) {
  while (n) {
    const size_t k = std::min<size_t>(n, 1u << 16);
    if (k <= 60) o.push_back((char)((k - 1) << 2));
    else if (k <= 256) { o.push_back((char)(60 << 2)); o.push_back((char)(k - 1)); }
    else { o.push_back((char)(61 << 2)); o.push_back((char)((k - 1) & 0xFF)); o.push_back((char)((k - 1) >> 8)); }
    o.append((const char *)p, k);
    p += k; n -= k;
  }
}
static std::string snappy_deflate(const uint8_t *p, size_t n) {
  std::string o;
  { uint64_t v = n; while (v >= 0x80) { o.push_back((char)(v | 0x80)); v >>= 7; } o.push_back((char)v); }
  std::vector<uint16_t> table(1 << 14);
  for (size_t b0 = 0; b0 < n; b0 += 1u << 16) {
    const size_t bn = std::min<size_t>(n - b0, 1u << 16);
    const uint8_t *s = p + b0;
    std::fill(table.begin(), table.end(), 0);
    size_t lit = 0, i = 0;
    auto load32 = [&](size_t k) { uint32_t v; std::memcpy(&v, s + k, 4); return v; };
    while (i + 4 <= bn) {
      const uint32_t h = (load32(i) * 0x1E35A7BDu) >> 18;
      const size_t cand = table[h];
      table[h] = (uint16_t)i;
      if (cand < i && load32(cand) == load32(i) && i - cand < (1u << 16) && !(cand == 0 && i == 0)) {
        size_t len = 4;
        while (i + len < bn && s[cand + len] == s[i + len]) len++;
        snappy_literal(o, s + lit, i - lit);
        const size_t off = i - cand;
        size_t left = len;
        while (left) {  // copies of at most 64 bytes, two-byte offsets
          size_t k = std::min<size_t>(left, 64);
          if (left - k > 0 && left - k < 4) k = left - 4;  // never leave a tail shorter than a copy may be
          o.push_back((char)(((k - 1) << 2) | 2)); o.push_back((char)(off & 0xFF)); o.push_back((char)(off >> 8));
          left -= k;
        }
        i += len; lit = i;
      } else i++;
    }
    snappy_literal(o, s + lit, bn - lit);
  }
  return o;
}
static bool gzip_deflate(const uint8_t *p, size_t n, std::string &o, std::string &why) {
  struct ZS { const uint8_t *next_in; unsigned avail_in; unsigned long total_in; uint8_t *next_out; unsigned avail_out; unsigned long total_out; const char *msg; void *state; void *zalloc, *zfree, *opaque; int data_type; unsigned long adler, reserved; };
  using Init2 = int (*)(ZS *, int, int, int, int, int, const char *, int); using Def = int (*)(ZS *, int); using End = int (*)(ZS *); using Bound = unsigned long (*)(ZS *, unsigned long);
  static void *h = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
  static Init2 init2 = h ? (Init2)dlsym(h, "deflateInit2_") : nullptr; static Def def = h ? (Def)dlsym(h, "deflate") : nullptr; static End end = h ? (End)dlsym(h, "deflateEnd") : nullptr;
  static Bound bound = h ? (Bound)dlsym(h, "deflateBound") : nullptr; static const char *(*ver)() = h ? (const char *(*)())dlsym(h, "zlibVersion") : nullptr;
  if (!init2 || !def || !end || !bound || !ver) { why = "GZIP needs libz.so.1"; return false; }
  if (n >> 31) { why = "a GZIP page of 2 GiB"; return false; }
  ZS z; std::memset(&z, 0, sizeof z);
  if (init2(&z, -1, 8 /* Z_DEFLATED */, 15 + 16, 8, 0, ver(), (int)sizeof z) != 0) { why = "deflateInit2 failed"; return false; }
  o.resize((size_t)bound(&z, (unsigned long)n) + 32);
  z.next_in = p; z.avail_in = (unsigned)n; z.next_out = (uint8_t *)&o[0]; z.avail_out = (unsigned)o.size();
  const int rc = def(&z, 4 /* Z_FINISH */);
  const bool ok = rc == 1;
  if (ok) o.resize(z.total_out);
  end(&z);
  if (!ok) why = "deflate failed";
  return ok;
}
static bool zstd_deflate(const uint8_t *p, size_t n, std::string &o, std::string &why) {
  using Comp = size_t (*)(void *, size_t, const void *, size_t, int); using Bound = size_t (*)(size_t); using IsErr = unsigned (*)(size_t);
  static void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
  static Comp comp = h ? (Comp)dlsym(h, "ZSTD_compress") : nullptr; static Bound bound = h ? (Bound)dlsym(h, "ZSTD_compressBound") : nullptr; static IsErr iserr = h ? (IsErr)dlsym(h, "ZSTD_isError") : nullptr;
  if (!comp || !bound || !iserr) { why = "ZSTD needs libzstd.so.1"; return false; }
  o.resize(bound(n));
  const size_t got = comp(&o[0], o.size(), p, n, 3);
  if (iserr(got)) { why = "ZSTD_compress failed"; return false; }
  o.resize(got);
  return true;
}

// ---- device: PLAIN values of the rows that hold one ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pqw_present(const uint8_t *validity, int64_t n, uint32_t *present) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) present[r] = validity ? (validity[r >> 3] >> (r & 7)) & 1u : 1u;
}
enum : int { PK_COPY = 0, PK_SEXT = 1, PK_ZEXT = 2, PK_DATE = 3, PK_NANOS = 4 };
// rank = exclusive scan of present; a present row's value goes to slot rank[r] of `out` (ow bytes each)
__global__ void __launch_bounds__(256) pqw_pack_fixed(const uint8_t *values, const int32_t *nanos, int iw, int ow, int kind, const uint32_t *rank, int64_t n, uint8_t *out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || rank[r + 1] == rank[r]) return;
  int64_t v = 0;
  switch (iw) {
    case 1: v = kind == PK_SEXT ? (int64_t)((const int8_t *)values)[r] : (int64_t)values[r]; break;
    case 2: v = kind == PK_SEXT ? (int64_t)((const int16_t *)values)[r] : (int64_t)((const uint16_t *)values)[r]; break;
    case 4: v = kind == PK_SEXT ? (int64_t)((const int32_t *)values)[r] : (int64_t)((const uint32_t *)values)[r]; break;
    default: v = ((const int64_t *)values)[r];
  }
  if (kind == PK_DATE) { const int64_t d = v / 86400; v = d - ((v % 86400) < 0 ? 1 : 0); }  // days since the epoch, floor
  else if (kind == PK_NANOS) v = v * 1000000000ll + (nanos ? (int64_t)nanos[r] : 0ll);
  uint8_t *o = out + (size_t)rank[r] * (size_t)ow;
  if (ow == 1) o[0] = (uint8_t)v; else if (ow == 4) *reinterpret_cast<uint32_t *>(o) = (uint32_t)v; else *reinterpret_cast<uint64_t *>(o) = (uint64_t)v;
}
__global__ void __launch_bounds__(256) pqw_text_len(const uint32_t *offsets, const uint32_t *rank, int64_t n, uint32_t *len4) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) len4[r] = rank[r + 1] != rank[r] ? 4u + (offsets[r + 1] - offsets[r]) : 0u;
}
// dst = exclusive scan of len4: a present row's <u32 length><bytes> starts at dst[r]; one wave per row group of 64 rows, lanes copy bytes
__global__ void __launch_bounds__(256) pqw_text_copy(const uint32_t *offsets, const uint8_t *data, const uint32_t *dst, int64_t n, uint8_t *out) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n || dst[r + 1] == dst[r]) return;
  const uint32_t a = offsets[r], len = offsets[r + 1] - a;
  uint8_t *o = out + dst[r];
  if (lane < 4) o[lane] = (uint8_t)(len >> (8 * lane));
  for (uint32_t i = (uint32_t)lane; i < len; i += 64) o[4 + i] = data[a + i];
}

}  // namespace pqw
}  // namespace tf

using namespace tf;
using namespace tf::pqw;

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

extern "C" int tfgpu_parquet_write(const tfgpu_dbatch *b, const tfgpu_schema *schema, const char *codec_name, int64_t row_group_max_rows, uint64_t row_group_max_bytes, void **bytes, uint64_t *len) {
  TF_API_BEGIN
  tf::dense(b);  // its rows may still be a selection (tfgpu_dbatch::pending)
  if (!b || !schema || !bytes || !len || (schema->ncols && !schema->cols)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_write: null argument");
  const std::string cn = codec_name ? codec_name : "";
  const int codec = cn == "SNAPPY" ? C_SNAPPY : cn == "GZIP" ? C_GZIP : cn == "ZSTD" ? C_ZSTD : C_UNCOMPRESSED;  // CodecFromString: anything else is uncompressed
  const int64_t n = b->nrows;
  // ---- the schema: the TableSchema's columns by name (parquet.Group is a Go map) ----
  std::vector<Leaf> leaves;
  for (int i = 0; i < schema->ncols; i++) {
    const tfgpu_colschema &sc = schema->cols[i];
    Leaf l; l.name = sc.name ? sc.name : ""; l.dtype = sc.dtype; l.required = (sc.flags & TFGPU_COL_REQUIRED) != 0;
    switch (sc.dtype) {
      case TFGPU_T_INT8: case TFGPU_T_INT16: case TFGPU_T_INT32: case TFGPU_T_UINT8: case TFGPU_T_UINT16: case TFGPU_T_UINT32: case TFGPU_T_DATE: l.phys = T_INT32; l.width = 4; break;
      case TFGPU_T_INT64: case TFGPU_T_UINT64: case TFGPU_T_DATETIME: case TFGPU_T_TIMESTAMP: case TFGPU_T_INTERVAL: l.phys = T_INT64; l.width = 8; break;
      case TFGPU_T_BOOLEAN: l.phys = T_BOOLEAN; l.width = 1; break;
      case TFGPU_T_FLOAT32: l.phys = T_FLOAT; l.width = 4; break;
      case TFGPU_T_FLOAT64: case TFGPU_T_BYTES: case TFGPU_T_UTF8: case TFGPU_T_ANY: l.phys = T_BYTE_ARRAY; l.width = 0; break;
      default: return tf::fail(TFGPU_ERR_CONFIG, "serializer:parquet: field " + l.name + " type not recognised");
    }
    for (size_t k = 0; k < b->cols.size(); k++) if (b->cols[k].name == l.name) l.col = (int)k;
    leaves.push_back(l);
  }
  std::sort(leaves.begin(), leaves.end(), [](const Leaf &x, const Leaf &y) { return x.name < y.name; });
  for (size_t i = 1; i < leaves.size(); i++) if (leaves[i].name == leaves[i - 1].name) return tf::fail(TFGPU_ERR_CONFIG, "serializer:parquet: column " + leaves[i].name + " twice in the TableSchema");

  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  hipStream_t st = cx.stream;
  materialize(*b);
  // ---- per column: ordinals, PLAIN values, one copy back ----
  struct ColOut { std::vector<uint8_t> valid; std::vector<uint8_t> payload; std::vector<uint32_t> rank; std::vector<uint32_t> tdst; bool all_null = false; };
  std::vector<ColOut> outs(leaves.size());
  const int64_t nb = (n + 255) / 256;
  for (size_t li = 0; li < leaves.size(); li++) {
    const Leaf &l = leaves[li];
    ColOut &o = outs[li];
    if (l.col < 0 || n == 0) { o.all_null = l.col < 0; o.rank.assign((size_t)n + 1, 0); if (l.phys == T_BYTE_ARRAY) o.tdst.assign((size_t)n + 1, 0); continue; }
    const DColumn &c = b->cols[(size_t)l.col];
    // the Go dynamic types the reference's writer accepts for the DataType (parquet.ValueOf of the boxed value)
    int kind = PK_COPY, iw = (int)repr_width(c.repr);
    bool ok;
    switch (l.dtype) {
      case TFGPU_T_INT8: ok = c.repr == TFGPU_R_INT8; kind = PK_SEXT; break;
      case TFGPU_T_INT16: ok = c.repr == TFGPU_R_INT16; kind = PK_SEXT; break;
      case TFGPU_T_INT32: ok = c.repr == TFGPU_R_INT32; kind = PK_SEXT; break;
      case TFGPU_T_INT64: ok = c.repr == TFGPU_R_INT64; break;
      case TFGPU_T_UINT8: ok = c.repr == TFGPU_R_UINT8; kind = PK_ZEXT; break;
      case TFGPU_T_UINT16: ok = c.repr == TFGPU_R_UINT16; kind = PK_ZEXT; break;
      case TFGPU_T_UINT32: ok = c.repr == TFGPU_R_UINT32; kind = PK_ZEXT; break;
      case TFGPU_T_UINT64: ok = c.repr == TFGPU_R_UINT64; break;
      case TFGPU_T_BOOLEAN: ok = c.repr == TFGPU_R_BOOL; break;
      case TFGPU_T_FLOAT32: ok = c.repr == TFGPU_R_FLOAT32; break;
      case TFGPU_T_DATE: ok = c.repr == TFGPU_R_TIME; kind = PK_DATE; break;
      case TFGPU_T_DATETIME: case TFGPU_T_TIMESTAMP: ok = c.repr == TFGPU_R_TIME; kind = PK_NANOS; break;
      case TFGPU_T_INTERVAL: ok = c.repr == TFGPU_R_DURATION; break;
      case TFGPU_T_FLOAT64: ok = c.repr == TFGPU_R_JSONNUM || c.repr == TFGPU_R_STRING; break;  // the decimal text (json.Number after Strictify); a float64 VALUE would need fmt's %v: stock writer
      case TFGPU_T_ANY: ok = c.repr == TFGPU_R_JSON; break;
      default: ok = c.repr == TFGPU_R_STRING || c.repr == TFGPU_R_BYTES;
    }
    if (!ok) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_write: column " + l.name + ": values of representation " + std::to_string(c.repr) + " under this DataType are written by the stock serializer (run Strictify first)");
    Buf rank = dalloc((size_t)(n + 2) * 4);
    pqw_present<<<(unsigned)nb, 256, 0, st>>>(ptr<uint8_t>(c.validity), n, ptr<uint32_t>(rank));
    exclusive_scan_u32(ptr<uint32_t>(rank), ptr<uint32_t>(rank), n, true);
    o.rank.resize((size_t)n + 1);
    d2h(o.rank.data(), rank->p, (size_t)(n + 1) * 4);
    if (c.validity) { o.valid.resize((size_t)(n + 7) / 8); d2h(o.valid.data(), c.validity->p, o.valid.size()); }
    if (l.phys != T_BYTE_ARRAY) {
      const int ow = l.phys == T_BOOLEAN ? 1 : l.width;
      Buf packed = dalloc((size_t)n * (size_t)ow + 16);
      KernelTimer t("pqw_pack");
      pqw_pack_fixed<<<(unsigned)nb, 256, 0, st>>>(ptr<uint8_t>(c.values), ptr<int32_t>(c.nanos), iw, ow, kind, ptr<uint32_t>(rank), n, ptr<uint8_t>(packed));
      tf::sync();  // rank[n] is on the host now
      const size_t nv = o.rank[(size_t)n];
      o.payload.resize(nv * (size_t)ow);
      if (nv) { d2h(o.payload.data(), packed->p, o.payload.size()); tf::sync(); }
    } else {
      Buf len4 = dalloc((size_t)(n + 2) * 4);
      KernelTimer t("pqw_text");
      pqw_text_len<<<(unsigned)nb, 256, 0, st>>>(ptr<uint32_t>(c.offsets), ptr<uint32_t>(rank), n, ptr<uint32_t>(len4));
      exclusive_scan_u32(ptr<uint32_t>(len4), ptr<uint32_t>(len4), n, true);
      o.tdst.resize((size_t)n + 1);
      d2h(o.tdst.data(), len4->p, (size_t)(n + 1) * 4);
      tf::sync();
      const uint64_t total = o.tdst[(size_t)n];
      if ((uint64_t)c.data_len + 4ull * (uint64_t)n >= 0xFFFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_write: column " + l.name + " exceeds 4 GiB: write the batch in pieces");
      Buf packed = dalloc((size_t)total + 16);
      pqw_text_copy<<<(unsigned)((n + 3) / 4), 256, 0, st>>>(ptr<uint32_t>(c.offsets), ptr<uint8_t>(c.payload()), ptr<uint32_t>(len4), n, ptr<uint8_t>(packed));
      o.payload.resize((size_t)total);
      if (total) d2h(o.payload.data(), packed->p, (size_t)total);
      tf::sync();
    }
  }
  // ---- row groups: RowGroupMaxRows, else RowGroupMaxBytes over the PLAIN sizes (the default: 128 MiB), cut at multiples of eight rows ----
  uint64_t plain_total = 0;
  for (auto &o : outs) plain_total += o.payload.size();
  if (row_group_max_rows == 0 && row_group_max_bytes == 0) row_group_max_bytes = 128ull << 20;
  int64_t per_group = n;
  if (row_group_max_rows > 0) per_group = row_group_max_rows;
  else if (row_group_max_bytes > 0 && plain_total > row_group_max_bytes) per_group = std::max<int64_t>(1, (int64_t)((double)n * (double)row_group_max_bytes / (double)plain_total));
  per_group = std::max<int64_t>(8, (per_group + 7) / 8 * 8);

  std::string file = "PAR1";
  struct ChunkMeta { int64_t offset, usize, csize, nvalues; };
  std::vector<std::vector<ChunkMeta>> groups;
  std::vector<int64_t> group_rows;
  for (int64_t r0 = 0; r0 < n || (n == 0 && groups.empty()); r0 += per_group) {
    if (n == 0) break;
    const int64_t r1 = std::min(n, r0 + per_group), gr = r1 - r0;
    std::vector<ChunkMeta> metas;
    for (size_t li = 0; li < leaves.size(); li++) {
      const Leaf &l = leaves[li];
      const ColOut &o = outs[li];
      std::string page;
      const size_t v0 = o.all_null ? 0 : o.rank[(size_t)r0], v1 = o.all_null ? 0 : o.rank[(size_t)r1];
      const size_t nv = v1 - v0;
      // toParquetValue (parquet_format.go:43-52): a nil is parquet.ValueOf(nil) at definition level 0 whatever the field's repetition, and
      // parquet-go writes a Required field's null as the type's ZERO value — pinned by the reference's own object
      // (canondata/reference.reference.TestBatchSerializer_parquet_default/result: `__primary_key` is 0 in the rows of tables that have no
      // such column; tests/test_parquet_write.py).  Rare: the page's values are expanded here, nils as zeros / empty byte arrays.
      const bool fill_required = l.required && nv != (size_t)gr;
      if (fill_required) {
        auto is_valid = [&](int64_t r) { return !o.all_null && (o.valid.empty() || ((o.valid[(size_t)r >> 3] >> (r & 7)) & 1)); };
        if (l.phys == T_BOOLEAN) {
          size_t v = v0;
          for (int64_t k = 0; k < gr; k += 8) { uint8_t byte = 0; for (int64_t j = 0; j < 8 && k + j < gr; j++) if (is_valid(r0 + k + j) && o.payload[v++]) byte |= (uint8_t)(1u << j); page.push_back((char)byte); }
        } else if (l.phys == T_BYTE_ARRAY) {
          for (int64_t r = r0; r < r1; r++) {
            if (is_valid(r)) page.append((const char *)o.payload.data() + o.tdst[(size_t)r], (size_t)(o.tdst[(size_t)r + 1] - o.tdst[(size_t)r]));
            else page.append(4, '\0');
          }
        } else {
          size_t v = v0;
          for (int64_t r = r0; r < r1; r++) {
            if (is_valid(r)) { page.append((const char *)o.payload.data() + v * (size_t)l.width, (size_t)l.width); v++; }
            else page.append((size_t)l.width, '\0');
          }
        }
      }
      if (!l.required) {  // definition levels: RLE / bit-packed hybrid of 1-bit levels behind a 4-byte length
        std::string lv;
        auto hv = [&](uint64_t v) { while (v >= 0x80) { lv.push_back((char)(v | 0x80)); v >>= 7; } lv.push_back((char)v); };
        if (o.all_null || nv == 0) { hv((uint64_t)gr << 1); lv.push_back((char)0); }
        else if (o.valid.empty() || nv == (size_t)gr) { hv((uint64_t)gr << 1); lv.push_back((char)1); }
        else { const uint64_t g8 = (uint64_t)(gr + 7) / 8; hv(g8 << 1 | 1); lv.append((const char *)o.valid.data() + r0 / 8, (size_t)g8); if (gr & 7) lv.back() = (char)((uint8_t)lv.back() & ((1u << (gr & 7)) - 1)); }
        const uint32_t L = (uint32_t)lv.size();
        page.append((const char *)&L, 4); page += lv;
      }
      if (fill_required) {}
      else if (l.phys == T_BOOLEAN) { for (size_t k = 0; k < nv; k += 8) { uint8_t byte = 0; for (size_t j = 0; j < 8 && k + j < nv; j++) if (o.payload[v0 + k + j]) byte |= (uint8_t)(1u << j); page.push_back((char)byte); } }
      else if (l.phys == T_BYTE_ARRAY) { if (!o.all_null && !o.payload.empty()) page.append((const char *)o.payload.data() + o.tdst[(size_t)r0], (size_t)(o.tdst[(size_t)r1] - o.tdst[(size_t)r0])); }
      else if (nv) page.append((const char *)o.payload.data() + v0 * (size_t)l.width, nv * (size_t)l.width);
      std::string comp; const std::string *body = &page; std::string why;
      if (codec == C_SNAPPY) { comp = snappy_deflate((const uint8_t *)page.data(), page.size()); body = &comp; }
      else if (codec == C_GZIP) { if (!gzip_deflate((const uint8_t *)page.data(), page.size(), comp, why)) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_write: " + why); body = &comp; }
      else if (codec == C_ZSTD) { if (!zstd_deflate((const uint8_t *)page.data(), page.size(), comp, why)) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_write: " + why); body = &comp; }
      if (page.size() >> 31 || body->size() >> 31) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_write: a page of 2 GiB: lower RowGroupMaxRows / RowGroupMaxBytes");
      TWriter h;
      h.i32(1, 0);                       // DATA_PAGE
      h.i32(2, (int64_t)page.size()); h.i32(3, (int64_t)body->size());
      h.begin(5); h.i32(1, gr); h.i32(2, 0 /* PLAIN */); h.i32(3, 3 /* RLE */); h.i32(4, 3); h.end();
      h.end();
      ChunkMeta m{(int64_t)file.size(), (int64_t)(h.out.size() + page.size()), (int64_t)(h.out.size() + body->size()), gr};
      file += h.out; file += *body;
      metas.push_back(m);
    }
    groups.push_back(metas); group_rows.push_back(gr);
  }
  // ---- footer ----
  TWriter f;
  f.i32(1, 1);
  f.list(2, 12, leaves.size() + 1);
  f.begin_elem(); f.binary(4, "table"); f.i32(5, (int64_t)leaves.size()); f.end();
  for (auto &l : leaves) {
    f.begin_elem();
    f.i32(1, l.phys); f.i32(3, l.required ? 0 : 1); f.binary(4, l.name);
    int conv = -1;
    switch (l.dtype) {
      case TFGPU_T_INT8: conv = 15; break; case TFGPU_T_INT16: conv = 16; break; case TFGPU_T_INT32: conv = 17; break; case TFGPU_T_INT64: conv = 18; break;
      case TFGPU_T_UINT8: conv = 11; break; case TFGPU_T_UINT16: conv = 12; break; case TFGPU_T_UINT32: conv = 13; break; case TFGPU_T_UINT64: conv = 14; break;
      case TFGPU_T_UTF8: case TFGPU_T_FLOAT64: conv = 0; break; case TFGPU_T_DATE: conv = 6; break; case TFGPU_T_ANY: conv = 19; break;
      default: break;
    }
    if (conv >= 0) f.i32(6, conv);
    // LogicalType (field 10), a union
    auto integer = [&](int bits, bool sign) { f.begin(10); f.begin(10); f.i8(1, bits); f.boolean(2, sign); f.end(); f.end(); };
    auto empty = [&](int id) { f.begin(10); f.begin(id); f.end(); f.end(); };
    switch (l.dtype) {
      case TFGPU_T_INT8: integer(8, true); break; case TFGPU_T_INT16: integer(16, true); break; case TFGPU_T_INT32: integer(32, true); break; case TFGPU_T_INT64: integer(64, true); break;
      case TFGPU_T_UINT8: integer(8, false); break; case TFGPU_T_UINT16: integer(16, false); break; case TFGPU_T_UINT32: integer(32, false); break; case TFGPU_T_UINT64: integer(64, false); break;
      case TFGPU_T_UTF8: case TFGPU_T_FLOAT64: empty(1); break;
      case TFGPU_T_DATE: empty(6); break;
      case TFGPU_T_ANY: empty(12); break;
      case TFGPU_T_DATETIME: case TFGPU_T_TIMESTAMP: case TFGPU_T_INTERVAL:
        f.begin(10); f.begin(8); f.boolean(1, true); f.begin(2); f.begin(3); f.end(); f.end(); f.end(); f.end(); break;  // TIMESTAMP(isAdjustedToUTC, NANOS)
      default: break;
    }
    f.end();
  }
  f.i64(3, n);
  f.list(4, 12, groups.size());
  for (size_t g = 0; g < groups.size(); g++) {
    f.begin_elem();
    f.list(1, 12, leaves.size());
    int64_t total = 0, total_c = 0;
    for (size_t li = 0; li < leaves.size(); li++) {
      const ChunkMeta &m = groups[g][li];
      f.begin_elem();
      f.i64(2, m.offset);
      f.begin(3);
      f.i32(1, leaves[li].phys);
      f.list(2, 5, 2); f.zz(0); f.zz(3);  // encodings: PLAIN, RLE
      f.list(3, 8, 1); f.varint(leaves[li].name.size()); f.out += leaves[li].name;
      f.i32(4, codec); f.i64(5, m.nvalues); f.i64(6, m.usize); f.i64(7, m.csize); f.i64(9, m.offset);
      f.end();
      f.end();
      total += m.usize; total_c += m.csize;
    }
    f.i64(2, total); f.i64(3, group_rows[g]);
    f.i64(5, groups[g].empty() ? 4 : groups[g][0].offset); f.i64(6, total_c);
    f.end();
  }
  f.binary(6, "transferia_amd tfgpu_parquet_write");
  f.end();
  file += f.out;
  const uint32_t flen = (uint32_t)f.out.size();
  file.append((const char *)&flen, 4);
  file += "PAR1";
  void *hp = nullptr;
  const int rc = tfgpu_host_alloc(file.size(), &hp);
  if (rc) return rc;
  std::memcpy(hp, file.data(), file.size());
  *bytes = hp; *len = file.size();
  return TFGPU_OK;
  TF_API_END
}
