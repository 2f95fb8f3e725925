// tf_transform.hip — device kernels behind abstract.Transformer.Apply for the
// row transformers of SURVEY.md §8a (a4–a13), plus the apply dispatcher.
//
// All kernels are byte/row kernels: one lane per row, column-major (Arrow)
// buffers so that neighbouring lanes touch neighbouring addresses.  None of
// them is a contraction, so there is no MFMA here; mask is INT32-ALU bound
// (2–3 SHA-256 compressions per value), everything else is HBM bound.
#include <algorithm>

#include "tf_devfmt.hpp"
#include "tf_devfloat.hpp"
#include "tf_plan.hpp"
#include "tf_devcol.hpp"
#include "tf_segcopy.hpp"
#include "tf_textview.hpp"
#include "tf_emit.hpp"

namespace tf {

static inline unsigned grid_for(int64_t n, int threads) {
  int64_t b = (n + threads - 1) / threads;
  return (unsigned)std::max<int64_t>(1, b);
}

// ============================================================================
// SerializeToString on device (to_string.go:145-178).  Formats value r of
// column c into `buf` (>= 64 bytes) unless the value is var-width text, in
// which case *ext points at the bytes in HBM.  Returns the length.
// ============================================================================
__device__ __forceinline__ int serialize_small(const DCol &c, int64_t r, uint8_t *buf, const uint8_t **ext) {
  *ext = nullptr;
  if (!is_valid(c, r)) { buf[0] = '<'; buf[1] = 'n'; buf[2] = 'i'; buf[3] = 'l'; buf[4] = '>'; return 5; }
  switch (c.repr) {
    case TFGPU_R_INT8: return dev::fmt_i64(buf, ((const int8_t *)c.values)[r]);
    case TFGPU_R_INT16: return dev::fmt_i64(buf, ((const int16_t *)c.values)[r]);
    case TFGPU_R_INT32: return dev::fmt_i64(buf, ((const int32_t *)c.values)[r]);
    case TFGPU_R_INT64: return dev::fmt_i64(buf, ((const int64_t *)c.values)[r]);
    case TFGPU_R_UINT8: return dev::fmt_u64(buf, ((const uint8_t *)c.values)[r]);
    case TFGPU_R_UINT16: return dev::fmt_u64(buf, ((const uint16_t *)c.values)[r]);
    case TFGPU_R_UINT32: return dev::fmt_u64(buf, ((const uint32_t *)c.values)[r]);
    case TFGPU_R_UINT64: return dev::fmt_u64(buf, ((const uint64_t *)c.values)[r]);
    case TFGPU_R_BOOL:
      if (((const uint8_t *)c.values)[r]) { buf[0] = 't'; buf[1] = 'r'; buf[2] = 'u'; buf[3] = 'e'; return 4; }
      buf[0] = 'f'; buf[1] = 'a'; buf[2] = 'l'; buf[3] = 's'; buf[4] = 'e'; return 5;
    case TFGPU_R_TIME: {
      int64_t s = ((const int64_t *)c.values)[r];
      int32_t ns = c.nanos ? c.nanos[r] : 0;
      if (c.dtype == TFGPU_T_DATE) return dev::fmt_date(buf, s);
      if (c.dtype == TFGPU_T_DATETIME || c.dtype == TFGPU_T_TIMESTAMP) return dev::fmt_rfc3339nano(buf, s, ns);
      return dev::fmt_time_string(buf, s, ns);
    }
    case TFGPU_R_DURATION: return dev::fmt_duration(buf, ((const int64_t *)c.values)[r]);
    // fmt.Sprintf("%v", float): %g with the shortest digits (to_string.go:170), at most 24 bytes
    case TFGPU_R_FLOAT32: { dev::StoreOut so{buf}; dev::fmt_float(so, (double)((const float *)c.values)[r], 'g', 32); return (int)so.n; }
    case TFGPU_R_FLOAT64: { dev::StoreOut so{buf}; dev::fmt_float(so, ((const double *)c.values)[r], 'g', 64); return (int)so.n; }
    case TFGPU_R_STRING: case TFGPU_R_JSONNUM: case TFGPU_R_JSON: case TFGPU_R_BYTES: {
      uint32_t a = c.offsets[r], b = c.offsets[r + 1];
      *ext = c.data + a;
      return (int)(b - a);
    }
  }
  return 0;
}
// Host-side check: can serialize_small reproduce SerializeToString for this column?
static void require_serializable(const DColumn &c, const char *what) {
  if (c.repr == TFGPU_R_BYTES && c.dtype != TFGPU_T_BYTES)
    throw Error(TFGPU_ERR_UNSUPPORTED, std::string(what) + ": column " + c.name + " holds []byte under a non-\"string\" DataType (%v prints a byte list)");
  if (c.repr == TFGPU_R_STRING && c.dtype == TFGPU_T_ANY)
    throw Error(TFGPU_ERR_UNSUPPORTED, std::string(what) + ": column " + c.name + " is `any` holding Go strings (json.Marshal quoting)");
}

// ============================================================================
// a4  mask_field: hex(HMAC_SHA256(salt, SerializeToString(v))) — hmac_hasher.go:29-33
// ============================================================================
__constant__ uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_rotateright32(x, n); }

__device__ __forceinline__ void sha256_compress(uint32_t st[8], uint32_t w[16]) {
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    if (i >= 16) {
      uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
      uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
    }
    uint32_t t1 = h + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + w[i & 15];
    uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

struct MaskParams {
  DCol col;
  uint32_t ipad[8], opad[8];
  int64_t nrows;
  uint8_t *out;  // nrows * 64 hex bytes
  const int32_t *sel;  // non-null: output row r hashes the column's row sel[r] (the batch's rows are still a selection)
};

// The text of an integer (or "<nil>") is at most 21 bytes: built in three registers as a little-endian byte string — the emitters of
// tf_emit.hpp hand over eight digits per word — it becomes the one message block of the inner hash with six byte swaps, instead of a
// scratch byte buffer read back byte by byte into a dynamically indexed w[] (that detour was a fifth of the kernel's instructions).
struct Text24 {
  uint64_t t0 = 0, t1 = 0, t2 = 0; uint32_t n = 0;
  __device__ __forceinline__ void put_word(uint64_t w, uint32_t k) {  // the low k (1..8) bytes of w, the rest zero
    const uint32_t at = n & 7u, sh = at * 8, seg = n >> 3;
    const uint64_t lo = w << sh, hi = (w >> 1) >> (63 - sh);            // (w >> 1) >> 63 == 0 when at == 0
    if (seg == 0) { t0 |= lo; t1 |= hi; } else if (seg == 1) { t1 |= lo; t2 |= hi; } else t2 |= lo;
    n += k;
  }
  __device__ __forceinline__ void put(uint32_t c) { put_word(c & 0xFFu, 1); }
};
__device__ __forceinline__ bool mask_small_int(const DCol &c, int64_t r, Text24 &s) {  // false: not an integer column
  if (!is_valid(c, r)) { s.put_word(0x3E6C696E3Cull /* "<nil>" */, 5); return true; }
  switch (c.repr) {
    case TFGPU_R_INT8: emit_i64(s, ((const int8_t *)c.values)[r]); return true;
    case TFGPU_R_INT16: emit_i64(s, ((const int16_t *)c.values)[r]); return true;
    case TFGPU_R_INT32: emit_i64(s, ((const int32_t *)c.values)[r]); return true;
    case TFGPU_R_INT64: emit_i64(s, ((const int64_t *)c.values)[r]); return true;
    case TFGPU_R_UINT8: emit_u64(s, ((const uint8_t *)c.values)[r]); return true;
    case TFGPU_R_UINT16: emit_u64(s, ((const uint16_t *)c.values)[r]); return true;
    case TFGPU_R_UINT32: emit_u64(s, ((const uint32_t *)c.values)[r]); return true;
    case TFGPU_R_UINT64: emit_u64(s, ((const uint64_t *)c.values)[r]); return true;
    default: return false;
  }
}

__global__ void __launch_bounds__(256) mask_hmac_kernel(MaskParams p) {
  const int64_t ro = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // the row written
  if (ro >= p.nrows) return;
  const int64_t r = p.sel ? (int64_t)p.sel[ro] : ro;                   // the row read
  uint32_t st[8], w[16];
#pragma unroll
  for (int i = 0; i < 8; i++) st[i] = p.ipad[i];
  const bool small_int = p.col.repr >= TFGPU_R_INT8 && p.col.repr <= TFGPU_R_UINT64;  // (a property of the column: a scalar branch)
  if (small_int) {
    Text24 s;
    mask_small_int(p.col, r, s);
    const uint32_t len = s.n;
    s.put_word(0x80, 1);  // the padding byte right behind the text; the block's tail is zeros and the bit length (len <= 21 < 56)
    w[0] = __builtin_bswap32((uint32_t)s.t0); w[1] = __builtin_bswap32((uint32_t)(s.t0 >> 32));
    w[2] = __builtin_bswap32((uint32_t)s.t1); w[3] = __builtin_bswap32((uint32_t)(s.t1 >> 32));
    w[4] = __builtin_bswap32((uint32_t)s.t2); w[5] = __builtin_bswap32((uint32_t)(s.t2 >> 32));
#pragma unroll
    for (int i = 6; i < 15; i++) w[i] = 0;
    w[15] = (64 + len) * 8;
    sha256_compress(st, w);
  } else {
  uint8_t buf[64];
  const uint8_t *ext;
  int len = serialize_small(p.col, r, buf, &ext);
  // inner hash: the ipad block is already absorbed; stream the message
  int off = 0;
  uint64_t bits = (uint64_t)(64 + len) * 8;
  bool pad_done = false, len_done = false;
  while (!len_done) {
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = 0;
    int take = len - off; if (take > 64) take = 64; if (take < 0) take = 0;
    for (int i = 0; i < take; i++) {
      uint32_t b = ext ? ext[off + i] : buf[off + i];
      w[i >> 2] |= b << (24 - 8 * (i & 3));
    }
    off += take;
    if (take < 64 && !pad_done) { w[take >> 2] |= 0x80u << (24 - 8 * (take & 3)); pad_done = true; if (take < 56) { w[14] = (uint32_t)(bits >> 32); w[15] = (uint32_t)bits; len_done = true; } }
    else if (pad_done) { w[14] = (uint32_t)(bits >> 32); w[15] = (uint32_t)bits; len_done = true; }
    sha256_compress(st, w);
  }
  }
  // outer hash: opad block absorbed; message = 32-byte inner digest
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = st[i];
  w[8] = 0x80000000u;
#pragma unroll
  for (int i = 9; i < 15; i++) w[i] = 0;
  w[15] = (64 + 32) * 8;
  uint32_t so[8];
#pragma unroll
  for (int i = 0; i < 8; i++) so[i] = p.opad[i];
  sha256_compress(so, w);
  // hex.EncodeToString: 64 lower-case hex chars, stored as 4 x 16 bytes
  uint4 *dst = reinterpret_cast<uint4 *>(p.out + ro * 64);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // 2 bytes of digest → 4 hex chars (little-endian packing of the output bytes)
      uint32_t word = so[q * 2 + (k >> 1)];
      uint32_t half = (k & 1) ? (word & 0xFFFF) : (word >> 16);
      uint32_t n0 = (half >> 12) & 15, n1 = (half >> 8) & 15, n2 = (half >> 4) & 15, n3 = half & 15;
      auto hx = [](uint32_t n) { return n + (n < 10 ? '0' : 'a' - 10); };
      o[k] = hx(n0) | hx(n1) << 8 | hx(n2) << 16 | hx(n3) << 24;
    }
    dst[q] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

__global__ void fill_offsets_stride_kernel(uint32_t *off, int64_t n, uint32_t stride) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) off[i] = (uint32_t)(i * stride);
}

static std::unique_ptr<tfgpu_dbatch> shallow_copy(const tfgpu_dbatch &in) { return std::make_unique<tfgpu_dbatch>(in); }

// what apply_mask refuses for a whole batch, before anything is computed
void mask_precheck(const tfgpu_plan &p, const tfgpu_dbatch &in) {
  for (auto &c : (in.pending ? in.pending->src->cols : in.cols)) {
    if (!p.mask_has(c.name)) continue;
    bool done = false;
    for (auto &r : in.replaced) if (r.name == c.name) done = true;  // (masked already: a string, always serializable)
    if (done) continue;
    require_serializable(c, "mask_field");
    if ((uint64_t)in.nrows * 64 > 0xFFFFFFFFull) throw Error(TFGPU_ERR_UNSUPPORTED, "mask_field: batch too large for 32-bit offsets; split the batch by rows");
  }
}
// transformation.do runs the transformers in their configured order.  A filter_rows directly behind mask_field transformers whose
// columns it does not read gives the same Transformed rows, the same row errors and the same failed inputs (once those are masked:
// push_run does) when it runs FIRST — mask_field raises no row errors, changes no other column and drops no row — and the HMACs of
// the rows it drops are never computed (45 % of them on configs[1]).  Returns the execution sequence; hopped[k] = the masks the
// k-th executed plan (a filter) went in front of.
std::vector<int> chain_sequence(const tfgpu_plan *const *plans, int n, std::vector<std::vector<int>> *hopped) {
  static const bool off = [] { const char *e = std::getenv("TFGPU_CHAIN_REORDER"); return e && e[0] == '0'; }();
  std::vector<int> seq((size_t)n);
  for (int i = 0; i < n; i++) seq[(size_t)i] = i;
  if (hopped) hopped->assign((size_t)n, {});
  if (off) return seq;
  for (int j = 1; j < n; j++) {
    if (plans[seq[(size_t)j]]->kind != PK_FILTER_ROWS) continue;
    const tfgpu_plan &f = *plans[seq[(size_t)j]];
    int k = j;
    std::vector<int> over;
    while (k > 0 && plans[seq[(size_t)k - 1]]->kind == PK_MASK) {
      const tfgpu_plan &m = *plans[seq[(size_t)k - 1]];
      bool reads = false;
      for (auto &e : f.exprs) for (auto &t : e.terms) if (m.mask_has(t.attr)) reads = true;
      if (reads) break;
      over.insert(over.begin(), seq[(size_t)k - 1]);
      std::swap(seq[(size_t)k - 1], seq[(size_t)k]);
      k--;
    }
    if (hopped) (*hopped)[(size_t)k] = over;
  }
  return seq;
}

static std::unique_ptr<tfgpu_dbatch> apply_mask(const tfgpu_plan &p, const tfgpu_dbatch &in) {
  auto out = shallow_copy(in);
  for (auto &sc : out->schema) if (p.mask_has(sc.first)) sc.second = TFGPU_T_UTF8;  // hmac_hasher.go:35-46
  hipStream_t st = ctx().stream;
  // The batch's rows may still be a selection over the batch a filter_rows read (tfgpu_dbatch::pending): the hash reads the masked
  // column's kept rows THROUGH the selection and its 64-byte digests are the first column of the result that exists densely —
  // the other hundred columns stay ungathered until somebody reads them.
  const tfgpu_dbatch &from = in.pending ? *in.pending->src : in;
  const int32_t *sel = in.pending ? ptr<int32_t>(in.pending->sel) : nullptr;
  auto replaced_at = [&](const std::string &name) -> int { for (size_t i = 0; i < in.replaced.size(); i++) if (in.replaced[i].name == name) return (int)i; return -1; };
  {
    std::vector<const DColumn *> need;
    for (auto &c : from.cols) if (p.mask_has(c.name) && replaced_at(c.name) < 0) need.push_back(&c);
    materialize(from, &need);
  }
  const int64_t n = in.nrows;
  auto mask_one = [&](const DColumn &c, const int32_t *through) {
    require_serializable(c, "mask_field");
    if ((uint64_t)n * 64 > 0xFFFFFFFFull) throw Error(TFGPU_ERR_UNSUPPORTED, "mask_field: batch too large for 32-bit offsets; split the batch by rows");
    DColumn o;
    o.name = c.name; o.dtype = TFGPU_T_UTF8; o.repr = TFGPU_R_STRING;
    o.data_len = (uint64_t)n * 64;
    o.data = dalloc(o.data_len);
    o.offsets = dalloc((size_t)(n + 1) * 4);
    MaskParams mp;
    mp.col = dcol_of(c);
    std::memcpy(mp.ipad, p.ipad_state, sizeof mp.ipad);
    std::memcpy(mp.opad, p.opad_state, sizeof mp.opad);
    mp.nrows = n; mp.out = ptr<uint8_t>(o.data); mp.sel = through;
    {
      KernelTimer t("mask_hmac_sha256", n);
      if (n) mask_hmac_kernel<<<grid_for(n, 256), 256, 0, st>>>(mp);
    }
    fill_offsets_stride_kernel<<<grid_for(n + 1, 256), 256, 0, st>>>(ptr<uint32_t>(o.offsets), n, 64);
    return o;
  };
  if (!in.pending) {
    for (auto &c : out->cols) if (p.mask_has(c.name)) {
      // HmacHasher.Apply walks the item's OWN ColumnNames (hmac_hasher.go:56-63): a row that does not list the column is left as it is — the digest
      // the kernel wrote for it is nobody's, the cell stays ABSENT (and reads nil)
      const Buf ab = c.absent;
      c = mask_one(c, nullptr);
      if (ab) { c.absent = ab; c.validity = validity_minus_absent(nullptr, ab, n); }
    }
    return out;
  }
  for (auto &c : from.cols) {
    if (!p.mask_has(c.name)) continue;
    const int ri = replaced_at(c.name);
    if (ri >= 0) out->replaced[(size_t)ri] = mask_one(in.replaced[(size_t)ri], nullptr);  // (already dense over the kept rows)
    else out->replaced.push_back(mask_one(c, sel));
  }
  return out;
}

// ============================================================================
// a11  filter_rows — filter_rows.go:99-365
// ============================================================================
struct DTerm {
  int32_t col;      // batch column index, -1 = not present in ColumnNames
  int32_t op, vtype, is_list;
  int32_t nvals;
  int32_t ioff;     // into ints / floats
  int32_t soff;     // into str_off (nvals+1 entries) for FV_STRING
};
struct FilterParams {
  const DCol *cols;
  int32_t ncols;
  const DTerm *terms;
  const int32_t *expr_start;  // nexpr+1
  int32_t nexpr;
  const int64_t *ints;
  const double *floats;
  const uint32_t *str_off;
  const uint8_t *str_data;
  const uint8_t *kind;
  int64_t nrows;
  const uint8_t *const *absent;  // [ncols] ABSENT bitmaps (DColumn::absent) or null entries; null: every row lists every column
  uint32_t *keep;   // 0/1 per row
  uint8_t *err;     // tfgpu_rowerr per row
  int32_t *err_term;
  uint32_t *nerr;   // global counter
};

__device__ __forceinline__ int cmp_op_i(int64_t a, int64_t b, int op) {
  switch (op) { case F_EQ: return a == b; case F_NE: return a != b; case F_LT: return a < b; case F_LE: return a <= b; case F_GT: return a > b; case F_GE: return a >= b; }
  return -1;
}
__device__ __forceinline__ int cmp_op_f(double a, double b, int op) {
  switch (op) { case F_EQ: return a == b; case F_NE: return a != b; case F_LT: return a < b; case F_LE: return a <= b; case F_GT: return a > b; case F_GE: return a >= b; }
  return -1;
}
__device__ __forceinline__ int bytes_compare(const uint8_t *a, uint32_t an, const uint8_t *b, uint32_t bn) {
  uint32_t m = an < bn ? an : bn;
  for (uint32_t i = 0; i < m; i++) { if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1; }
  return an < bn ? -1 : an > bn ? 1 : 0;
}
__device__ __forceinline__ bool bytes_contains(const uint8_t *h, uint32_t hn, const uint8_t *n, uint32_t nn) {
  if (nn == 0) return true;
  if (nn > hn) return false;
  for (uint32_t i = 0; i + nn <= hn; i++) {
    uint32_t k = 0;
    while (k < nn && h[i + k] == n[k]) k++;
    if (k == nn) return true;
  }
  return false;
}

// time.Parse over the layouts of stringToTime (filter_rows/util.go:15-39) that
// are fixed-shape numeric: "2006-01-02", "2006-01-02 15:04:05", "2006-01-02T15:04:05",
// RFC3339 / RFC3339Nano.  Returns 1 ok, 0 = does not look like any of them
// (the row is then handed back to the host path).
__device__ __forceinline__ bool is_dg(uint8_t c) { return c >= '0' && c <= '9'; }
__device__ int parse_time_subset(const uint8_t *s, uint32_t n, int64_t *sec, int32_t *nsec) {
  if (n < 10) return 0;
  if (!(is_dg(s[0]) && is_dg(s[1]) && is_dg(s[2]) && is_dg(s[3]) && s[4] == '-' && is_dg(s[5]) && is_dg(s[6]) && s[7] == '-' && is_dg(s[8]) && is_dg(s[9]))) return 0;
  int64_t y = (s[0] - '0') * 1000 + (s[1] - '0') * 100 + (s[2] - '0') * 10 + (s[3] - '0');
  int mo = (s[5] - '0') * 10 + (s[6] - '0'), d = (s[8] - '0') * 10 + (s[9] - '0');
  int h = 0, mi = 0, se = 0; int64_t ns = 0; int off = 0;
  uint32_t k = 10;
  if (n > 10) {
    if (!(s[10] == 'T' || s[10] == ' ')) return 0;
    bool tform = s[10] == 'T';
    if (n < 19) return 0;
    if (!(is_dg(s[11]) && is_dg(s[12]) && s[13] == ':' && is_dg(s[14]) && is_dg(s[15]) && s[16] == ':' && is_dg(s[17]) && is_dg(s[18]))) return 0;
    h = (s[11] - '0') * 10 + (s[12] - '0'); mi = (s[14] - '0') * 10 + (s[15] - '0'); se = (s[17] - '0') * 10 + (s[18] - '0');
    k = 19;
    if (k + 1 < n && (s[k] == '.' || s[k] == ',') && is_dg(s[k + 1])) {
      k++; int nd = 0;
      while (k < n && is_dg(s[k])) { if (nd < 9) { ns = ns * 10 + (s[k] - '0'); nd++; } k++; }
      while (nd < 9) { ns *= 10; nd++; }
    }
    if (k < n) {
      if (!tform) return 0;  // "2006-01-02 15:04:05 -0700 MST" etc: host path
      if (s[k] == 'Z') k++;
      else if ((s[k] == '+' || s[k] == '-') && k + 6 <= n && is_dg(s[k + 1]) && is_dg(s[k + 2]) && s[k + 3] == ':' && is_dg(s[k + 4]) && is_dg(s[k + 5])) {
        int hh = (s[k + 1] - '0') * 10 + (s[k + 2] - '0'), mm = (s[k + 4] - '0') * 10 + (s[k + 5] - '0');
        if (hh > 24 || mm > 60) return 0;
        off = (s[k] == '-' ? -1 : 1) * (hh * 3600 + mm * 60);
        k += 6;
      } else return 0;
      if (k != n) return 0;
    }
  }
  if (mo < 1 || mo > 12 || d < 1 || d > dev::days_in_month(mo, y) || h > 23 || mi > 59 || se > 59) return 0;
  *sec = dev::days_from_civil(y, mo, d) * 86400 + h * 3600 + mi * 60 + se - off;
  *nsec = (int32_t)ns;
  return 1;
}

// matchValue (filter_rows.go:180-365).  Returns 1/0, or -(tfgpu_rowerr).
__device__ int match_value(const FilterParams &p, const DTerm &t, int64_t r) {
  const DCol &c = p.cols[t.col];
  const bool valid = is_valid(c, r);
  const bool is_set = t.op == F_IN || t.op == F_NOTIN;
  bool isInt1 = false, isFloat1 = false, maybeFloat = false;
  int64_t int1 = 0; double float1 = 0;
  if (valid) {
    switch (c.repr) {  // toInt64E util.go:42-82, then cast.ToFloat64E
      case TFGPU_R_INT8: int1 = ((const int8_t *)c.values)[r]; isInt1 = true; break;
      case TFGPU_R_INT16: int1 = ((const int16_t *)c.values)[r]; isInt1 = true; break;
      case TFGPU_R_INT32: int1 = ((const int32_t *)c.values)[r]; isInt1 = true; break;
      case TFGPU_R_INT64: int1 = ((const int64_t *)c.values)[r]; isInt1 = true; break;
      case TFGPU_R_UINT8: int1 = ((const uint8_t *)c.values)[r]; isInt1 = true; break;
      case TFGPU_R_UINT16: int1 = ((const uint16_t *)c.values)[r]; isInt1 = true; break;
      case TFGPU_R_UINT32: int1 = ((const uint32_t *)c.values)[r]; isInt1 = true; break;
      case TFGPU_R_UINT64: { uint64_t u = ((const uint64_t *)c.values)[r]; if (u > 0x7FFFFFFFFFFFFFFFull) return -TFGPU_ROW_INT_OVERFLOW; int1 = (int64_t)u; isInt1 = true; break; }
      case TFGPU_R_FLOAT32: float1 = ((const float *)c.values)[r]; isFloat1 = true; break;
      case TFGPU_R_FLOAT64: float1 = ((const double *)c.values)[r]; isFloat1 = true; break;
      case TFGPU_R_BOOL: float1 = ((const uint8_t *)c.values)[r] ? 1.0 : 0.0; isFloat1 = true; break;
      case TFGPU_R_STRING: case TFGPU_R_JSONNUM: maybeFloat = true; break;  // strconv.ParseFloat of text
    }
  } else { float1 = 0; isFloat1 = true; }  // cast.ToFloat64E(nil) == 0
  int res;
  switch (t.vtype) {
    case FV_INT:
      if (isInt1) {
        if (is_set) { bool f = false; for (int i = 0; i < t.nvals; i++) f = f || p.ints[t.ioff + i] == int1; return t.op == F_IN ? f : !f; }
        res = cmp_op_i(int1, p.ints[t.ioff], t.op); return res < 0 ? -TFGPU_ROW_TYPE_PAIR : res;
      }
      if (maybeFloat) return -TFGPU_ROW_HOST_FALLBACK;
      if (isFloat1) {
        if (is_set) {
          if (trunc(float1) == float1) { int64_t iv = (int64_t)float1; bool f = false; for (int i = 0; i < t.nvals; i++) f = f || p.ints[t.ioff + i] == iv; return t.op == F_IN ? f : !f; }
          return 0;
        }
        res = cmp_op_f(float1, (double)p.ints[t.ioff], t.op); return res < 0 ? -TFGPU_ROW_TYPE_PAIR : res;
      }
      break;
    case FV_FLOAT:
      if (maybeFloat && !isInt1) return -TFGPU_ROW_HOST_FALLBACK;
      if (isInt1 || isFloat1) {
        double a = isInt1 ? (double)int1 : float1;
        if (is_set) { bool f = false; for (int i = 0; i < t.nvals; i++) f = f || p.floats[t.ioff + i] == a; return t.op == F_IN ? f : !f; }
        res = cmp_op_f(a, p.floats[t.ioff], t.op); return res < 0 ? -TFGPU_ROW_TYPE_PAIR : res;
      }
      break;
    case FV_BOOL:
      if (valid && c.repr == TFGPU_R_BOOL) {
        res = cmp_op_i(((const uint8_t *)c.values)[r] ? 1 : 0, p.ints[t.ioff], t.op); return res < 0 ? -TFGPU_ROW_TYPE_PAIR : res;
      }
      break;
    case FV_STRING:
      if (valid && (c.repr == TFGPU_R_BYTES || c.repr == TFGPU_R_STRING)) {
        const uint8_t *a = c.data + c.offsets[r]; uint32_t an = c.offsets[r + 1] - c.offsets[r];
        const uint32_t *so = p.str_off + t.soff;
        if (t.op == F_MATCH || t.op == F_NOTMATCH) { bool m = bytes_contains(a, an, p.str_data + so[0], so[1] - so[0]); return t.op == F_MATCH ? m : !m; }
        if (is_set) {
          bool f = false;
          for (int i = 0; i < t.nvals; i++) { uint32_t bn = so[i + 1] - so[i]; if (bn == an && bytes_compare(a, an, p.str_data + so[i], bn) == 0) f = true; }
          return t.op == F_IN ? f : !f;
        }
        int cr = bytes_compare(a, an, p.str_data + so[0], so[1] - so[0]);
        res = cmp_op_i(cr, 0, t.op); return res < 0 ? -TFGPU_ROW_TYPE_PAIR : res;
      }
      break;
    case FV_TIME: {
      int64_t us1 = 0; bool have = false;
      if (valid && c.repr == TFGPU_R_TIME) { us1 = ((const int64_t *)c.values)[r] * 1000000 + (c.nanos ? c.nanos[r] : 0) / 1000; have = true; }
      else if (valid && c.repr == TFGPU_R_STRING) {
        int64_t s; int32_t ns;
        if (!parse_time_subset(c.data + c.offsets[r], c.offsets[r + 1] - c.offsets[r], &s, &ns)) return -TFGPU_ROW_HOST_FALLBACK;
        us1 = s * 1000000 + ns / 1000; have = true;
      }
      if (have) {
        if (is_set) { bool f = false; for (int i = 0; i < t.nvals; i++) f = f || p.ints[t.ioff + i] == us1; return t.op == F_IN ? f : !f; }
        res = cmp_op_i(us1, p.ints[t.ioff], t.op); return res < 0 ? -TFGPU_ROW_TYPE_PAIR : res;
      }
      break;
    }
    case FV_NULL:
      if (t.op == F_EQ) return !valid;
      if (t.op == F_NE) return valid;
      break;
  }
  return -TFGPU_ROW_TYPE_PAIR;
}

__global__ void __launch_bounds__(256) filter_eval_kernel(FilterParams p) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  int kind = p.kind ? p.kind[r] : TFGPU_K_INSERT;
  uint32_t keep = 0; int err = 0, eterm = -1;
  if (kind == TFGPU_K_UPDATE || kind == TFGPU_K_DELETE) err = TFGPU_ROW_UNSUPPORTED_KIND;  // :103-107
  else if (kind != TFGPU_K_INSERT || p.nexpr < 0) keep = 1;                              // :110-113 (nexpr<0: table not matched)
  else {
    for (int e = 0; e < p.nexpr && !keep && !err; e++) {  // matchItem :132-143
      bool ok = true;
      for (int k = p.expr_start[e]; k < p.expr_start[e + 1] && ok && !err; k++) {  // matchExpression :145-178
        const DTerm &t = p.terms[k];
        if (p.ncols == 0) continue;  // a row without columns never enters the name loop
        if (p.absent) {  // rows that list their own columns: the name loop runs over THIS row's ColumnNames (filter_rows.go:147-154)
          bool any = false;
          for (int c = 0; c < p.ncols && !any; c++) any = !(p.absent[c] && ((p.absent[c][r >> 3] >> (r & 7)) & 1));
          if (!any) continue;
          if (t.col >= 0 && p.absent[t.col] && ((p.absent[t.col][r >> 3] >> (r & 7)) & 1)) { err = TFGPU_ROW_COLUMN_NOT_FOUND; eterm = k; break; }
        }
        if (t.col < 0) { err = TFGPU_ROW_COLUMN_NOT_FOUND; eterm = k; break; }
        int m = match_value(p, t, r);
        if (m < 0) { err = -m; eterm = k; } else if (!m) ok = false;
      }
      if (!err && ok) keep = 1;
    }
  }
  p.keep[r] = err ? 0 : keep;
  p.err[r] = (uint8_t)err;
  if (err) { p.err_term[r] = eterm; atomicAdd(p.nerr, 1u); }
}

// ============================================================================
// compaction: keep flags → selection vector → gather of every column
// ============================================================================
__global__ void __launch_bounds__(256) build_selection_kernel(const uint32_t *__restrict__ keep_scan, int64_t n, int32_t *__restrict__ sel) {
  // keep_scan[i] = exclusive prefix of keep; row i kept iff keep_scan[i+1] != keep_scan[i]
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t a = keep_scan[i], b = keep_scan[i + 1];
  if (a != b) sel[a] = (int32_t)i;
}

// All fixed-width arrays of a batch (values, nanos, kind, part_id, src_row) in ONE launch: lane = output
// row, wave-uniform loop over the arrays, so every store is coalesced and the selection vector is read once.
struct GFix { const void *in; void *out; int32_t width; int32_t pad; };
// the arrays arrive sorted by element width (n1 of one byte, then n2, n4, n8), so every loop below has one element type and
// no branch between its loads: a lane requests eight arrays' values before it stores the first — one wait for memory per
// eight arrays instead of one per array (the kernel used to be a chain of 78 round trips per lane)
template <class T> __device__ __forceinline__ void gather_run(const GFix *__restrict__ a, int lo, int hi, int32_t s, int64_t k) {
  constexpr int U = 8;
  int i = lo;
  for (; i + U <= hi; i += U) {
    T v[U];
#pragma unroll
    for (int q = 0; q < U; q++) v[q] = TF_GLOBAL_PTR(const T, a[i + q].in)[s];
#pragma unroll
    for (int q = 0; q < U; q++) TF_GLOBAL_PTR(T, a[i + q].out)[k] = v[q];
  }
  for (; i < hi; i++) TF_GLOBAL_PTR(T, a[i].out)[k] = TF_GLOBAL_PTR(const T, a[i].in)[s];
}
__global__ void __launch_bounds__(256) gather_fixed_all(const GFix *__restrict__ a, int n1, int n2, int n4, int n8, uint32_t *__restrict__ ident, const int32_t *__restrict__ sel, int64_t m) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const int32_t s = sel[k];
  if (ident) ident[k] = (uint32_t)s;  // src_row of a batch that had none: the identity
  gather_run<uint8_t>(a, 0, n1, s, k);
  gather_run<uint16_t>(a, n1, n1 + n2, s, k);
  gather_run<uint32_t>(a, n1 + n2, n1 + n2 + n4, s, k);
  gather_run<uint64_t>(a, n1 + n2 + n4, n1 + n2 + n4 + n8, s, k);
}
// validity bitmaps of all columns in one launch: one thread per output byte (8 rows), blockIdx.y = bitmap
struct GBit { const uint8_t *in; uint8_t *out; };
__global__ void __launch_bounds__(256) gather_bitmap_all(const GBit *__restrict__ a, const int32_t *__restrict__ sel, int64_t m) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b * 8 >= m) return;
  const GBit g = a[blockIdx.y];
  uint32_t v = 0;
  for (int j = 0; j < 8; j++) { int64_t k = b * 8 + j; if (k < m) { int32_t s = sel[k]; v |= ((g.in[s >> 3] >> (s & 7)) & 1u) << j; } }
  g.out[b] = (uint8_t)v;
}
// Var-width columns: lengths of all columns in one launch (segmented layout, then one segmented scan),
// payload bytes of all columns in one launch.
// fstart != null: a late-materialised text column (TextView) — in_data is the SOURCE text and a kept cell is fetched from
// in_data + fstart[row]; cells that are not a plain byte range are left zero-filled for gather_text_special.
struct GVar { const uint32_t *in_off; const uint8_t *in_data; uint32_t *out_off; uint8_t *out_data; const uint32_t *fstart; uint32_t quote, jsonnum, has_special; };
__global__ void __launch_bounds__(256) gather_len_all(const GVar *__restrict__ v, int nv, const int32_t *__restrict__ sel, int64_t m) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const int32_t s = sel[k];
  constexpr int U = 8;  // eight columns' offsets requested per wait (as in gather_fixed_all)
  int i = 0;
  for (; i + U <= nv; i += U) {
    uint32_t a[U], b[U];
#pragma unroll
    for (int q = 0; q < U; q++) { const uint32_t *off = TF_GLOBAL_PTR(const uint32_t, v[i + q].in_off); a[q] = off[s]; b[q] = off[s + 1]; }
#pragma unroll
    for (int q = 0; q < U; q++) TF_GLOBAL_PTR(uint32_t, v[i + q].out_off)[k] = b[q] - a[q];
  }
  for (; i < nv; i++) { const uint32_t *off = v[i].in_off; v[i].out_off[k] = off[s + 1] - off[s]; }
}
// Payload bytes of the kept rows, destination-centric (tf_segcopy.hpp): blockIdx.y = column, blockIdx.x = a run of
// 256 * RPT kept rows, whose cells are contiguous in the output and are fetched through the selection vector.
template <int RPT> __global__ void __launch_bounds__(256) gather_bytes_all(const GVar *__restrict__ v, const int32_t *__restrict__ sel, int64_t m) {
  __shared__ uint32_t doff[256 * RPT + 1];
  __shared__ uint32_t soff[256 * RPT];
  const GVar g = v[blockIdx.y];
  auto so = [&](int64_t k) {
    const int32_t s = sel[k];
    if (!g.fstart) return g.in_off[s];
    const uint32_t f = g.fstart[s];
    return cell_plain(f) ? f : SEG_NONE;
  };
  segcopy_run<RPT>(g.out_off, m, (int64_t)blockIdx.x * 256 * RPT, g.in_data, g.out_data, so, doff, soff);
}
// short, mostly empty cells: cell-centric, lane = kept row, (unaligned) 8-byte words
__global__ void __launch_bounds__(256) gather_bytes_cells(const GVar *__restrict__ v, const int32_t *__restrict__ sel, int64_t m) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const GVar g = v[blockIdx.y];
  const uint32_t o0 = g.out_off[k], n = g.out_off[k + 1] - o0;
  if (!n) return;
  struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
  uint32_t so = 0;
  if (g.fstart) { so = g.fstart[sel[k]]; if (!cell_plain(so)) return; }  // gather_text_special
  else so = g.in_off[sel[k]];
  const uint8_t *src = g.in_data + so;
  uint8_t *dst = g.out_data + o0;
  uint32_t i = 0;
  for (; i + 8 <= n; i += 8) reinterpret_cast<U64 *>(dst + i)->v = reinterpret_cast<const U64 *>(src + i)->v;
  if (i < n) { uint64_t x = reinterpret_cast<const U64 *>(src + i)->v; for (; i < n; i++) { dst[i] = (uint8_t)x; x >>= 8; } }
}
// the kept cells of late-materialised columns that need more than a byte copy (doubled quotes, DefaultValue)
__global__ void __launch_bounds__(256) gather_text_special(const GVar *__restrict__ v, const int32_t *__restrict__ sel, int64_t m) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const GVar g = v[blockIdx.y];
  uint32_t fsv = 0, o0 = 0, n = 0;
  if (k < m) {
    o0 = g.out_off[k]; n = g.out_off[k + 1] - o0;  // coalesced; most cells of a sparse column are empty
    if (n) { fsv = g.fstart[sel[k]]; if (cell_plain(fsv)) n = 0; }
  }
  text_copy_special_wave(g.in_data, g.quote, g.out_data, g.jsonnum != 0, fsv, o0, n, threadIdx.x & 63);
}
__global__ void collect_totals(const uint32_t *lens_all, int64_t seg_stride, int64_t m, int nv, uint32_t *out) {
  for (int s = threadIdx.x; s < nv; s += blockDim.x) out[s] = lens_all[(int64_t)s * seg_stride + m];
}

// Gather all columns of `in` through `sel` (m output rows).
static std::unique_ptr<tfgpu_dbatch> gather_batch(const tfgpu_dbatch &in, const Buf &sel, int64_t m, const std::vector<DColumn> *skip = nullptr) {
  // (skip: columns — by name — whose kept rows exist already: they are not gathered, their slot takes the given column)
  if (in.col_order) throw Error(TFGPU_ERR_UNSUPPORTED, "the batch's rows carry their own ColumnNames order (tfgpu_batch.col_order: a collapsed TOAST batch); row-moving steps do not carry it — "
                                                       "view / download, the native queue format and the Debezium emitter read it");
  auto out = std::make_unique<tfgpu_dbatch>();
  out->nrows = m; out->ns = in.ns; out->table = in.table; out->schema = in.schema;
  hipStream_t st = ctx().stream;
  const int32_t *sp = ptr<int32_t>(sel);
  unsigned g = grid_for(m, 256);
  KernelTimer t("compact_gather");
  static const bool eager_text = [] { const char *e = std::getenv("TFGPU_CSV_EAGER"); return e && e[0] == '1'; }();
  std::vector<GFix> fx; std::vector<GVar> vr; std::vector<size_t> var_cols; std::vector<GBit> bits;
  // ColumnValues and OldKeys columns move the same way; `all` lists them in that order
  const size_t ncur = in.cols.size(), nall = ncur + in.old_keys.size();
  auto col_at = [&](size_t i) -> const DColumn & { return i < ncur ? in.cols[i] : in.old_keys[i - ncur]; };
  out->cols.reserve(ncur); out->old_keys.reserve(in.old_keys.size());
  out->key_names = in.key_names;
  int nvar = 0;
  for (size_t i = 0; i < nall; i++) if (repr_is_var(col_at(i).repr) && !(skip && i < ncur && [&] { for (auto &r : *skip) if (r.name == in.cols[i].name) return true; return false; }())) nvar++;
  const int64_t seg_stride = ((m + 1 + 3) / 4) * 4;
  Buf lens_all = nvar ? dalloc((size_t)nvar * (size_t)seg_stride * 4 + 16) : nullptr;
  auto skipped = [&](size_t ai) -> const DColumn * {
    if (!skip || ai >= ncur) return nullptr;
    for (auto &r : *skip) if (r.name == in.cols[ai].name) return &r;
    return nullptr;
  };
  for (size_t ai = 0; ai < nall; ai++) {
    const DColumn &c = col_at(ai);
    if (const DColumn *have = skipped(ai)) { out->cols.push_back(*have); continue; }
    DColumn o;
    o.name = c.name; o.dtype = c.dtype; o.repr = c.repr;
    if (repr_is_var(c.repr)) {
      size_t si = vr.size();
      o.offsets = subbuf(lens_all, si * (size_t)seg_stride * 4, (size_t)(m + 1) * 4);
      if (!c.lazy() || eager_text) o.data = dalloc(c.data_len + 8);  // the source size bounds the kept payload; exact length read back below
      o.data_len = c.data_len;
      if (c.lazy() && !eager_text) {
        // A kept row keeps pointing INTO the source text, as the kept ChangeItem's Go strings keep aliasing the chunk they
        // were cut from (filter_rows.go:99-125 appends the item, it copies no string bytes): the compacted column is again
        // (offsets, position of each cell in the text), packed when — and if — a consumer reads its bytes.
        if (!c.view->src) throw Error(TFGPU_ERR_INVALID, "internal: text column " + c.name + " lost its source text");
        auto v = std::make_shared<TextView>();
        v->src = c.view->src; v->fstart = dalloc((size_t)std::max<int64_t>(m, 1) * 4);
        v->has_special = c.view->has_special; v->quote = c.view->quote; v->jsonnum = c.view->jsonnum;
        fx.push_back(GFix{c.view->fstart->p, v->fstart->p, 4, 0});
        o.view = std::move(v);
        o.data = nullptr;
        vr.push_back(GVar{ptr<uint32_t>(c.offsets), nullptr, ptr<uint32_t>(o.offsets), nullptr, nullptr, 0, 0, 0});
      } else if (c.lazy()) {  // TFGPU_CSV_EAGER=1: kept cells are packed now, straight from the source text
        if (!c.view->src) throw Error(TFGPU_ERR_INVALID, "internal: text column " + c.name + " lost its source text");
        vr.push_back(GVar{ptr<uint32_t>(c.offsets), ptr<uint8_t>(c.view->src), ptr<uint32_t>(o.offsets), ptr<uint8_t>(o.data), ptr<uint32_t>(c.view->fstart), c.view->quote, c.view->jsonnum ? 1u : 0u, c.view->has_special ? 1u : 0u});
      } else vr.push_back(GVar{ptr<uint32_t>(c.offsets), ptr<uint8_t>(c.payload()), ptr<uint32_t>(o.offsets), ptr<uint8_t>(o.data), nullptr, 0, 0, 0});
      var_cols.push_back(ai);
    } else {
      size_t w = repr_width(c.repr);
      o.values = dalloc((size_t)m * w);
      fx.push_back(GFix{c.values->p, o.values->p, (int32_t)w, 0});
      if (c.nanos) { o.nanos = dalloc((size_t)m * 4); fx.push_back(GFix{c.nanos->p, o.nanos->p, 4, 0}); }
    }
    if (c.validity) { o.validity = dalloc((size_t)(m + 7) / 8 + 1); bits.push_back(GBit{ptr<uint8_t>(c.validity), ptr<uint8_t>(o.validity)}); }
    if (c.absent) { o.absent = dalloc((size_t)(m + 7) / 8 + 1); bits.push_back(GBit{ptr<uint8_t>(c.absent), ptr<uint8_t>(o.absent)}); }  // a moved row keeps its ColumnNames
    (ai < ncur ? out->cols : out->old_keys).push_back(std::move(o));
  }
  auto out_at = [&](size_t i) -> DColumn & { return i < ncur ? out->cols[i] : out->old_keys[i - ncur]; };
  if (in.old_present) { out->old_present = dalloc((size_t)(m + 7) / 8 + 1); bits.push_back(GBit{ptr<uint8_t>(in.old_present), ptr<uint8_t>(out->old_present)}); }
  if (in.kind) { out->kind = dalloc((size_t)m); fx.push_back(GFix{in.kind->p, out->kind->p, 1, 0}); }
  if (in.part_id) { out->part_id = dalloc((size_t)m * 4); fx.push_back(GFix{in.part_id->p, out->part_id->p, 4, 0}); }
  out->src_row = dalloc((size_t)m * 4);
  if (in.src_row) fx.push_back(GFix{in.src_row->p, out->src_row->p, 4, 0});
  uint32_t *const ident = in.src_row ? nullptr : ptr<uint32_t>(out->src_row);

  if (m && !bits.empty()) {
    Buf bb = upload_const(bits.data(), bits.size() * sizeof(GBit));
    gather_bitmap_all<<<dim3(grid_for((m + 7) / 8, 256), (unsigned)bits.size()), 256, 0, st>>>(ptr<GBit>(bb), sp, m);
  }
  std::stable_sort(fx.begin(), fx.end(), [](const GFix &x, const GFix &y) { return x.width < y.width; });
  int nw[4] = {0, 0, 0, 0};
  for (auto &x : fx) nw[x.width == 1 ? 0 : x.width == 2 ? 1 : x.width == 4 ? 2 : 3]++;
  const int nfx[4] = {nw[0], nw[1], nw[2], nw[3]};
  if (fx.empty()) fx.push_back(GFix{nullptr, nullptr, 0, 0});  // (never read: every count is zero)
  Buf bfx = upload_const(fx.data(), fx.size() * sizeof(GFix));
  if (m) gather_fixed_all<<<g, 256, 0, st>>>(ptr<GFix>(bfx), nfx[0], nfx[1], nfx[2], nfx[3], ident, sp, m);
  if (nvar) {
    Buf bvr = upload_const(vr.data(), vr.size() * sizeof(GVar));
    if (m) gather_len_all<<<g, 256, 0, st>>>(ptr<GVar>(bvr), nvar, sp, m);
    exclusive_scan_u32_segments(ptr<uint32_t>(lens_all), m, nvar, seg_stride);
    if (m) {  // long cells: short runs of rows; short cells: long runs (the fixed latency of a run needs bytes to amortise over)
      std::vector<GVar> lng, sht;
      for (size_t i = 0; i < vr.size(); i++) if (vr[i].out_data) ((col_at(var_cols[i]).data_len >= (uint64_t)in.nrows * 8) ? lng : sht).push_back(vr[i]);
      Buf blng = upload_const(lng.data(), lng.size() * sizeof(GVar)), bsht = upload_const(sht.data(), sht.size() * sizeof(GVar));
      if (!lng.empty()) gather_bytes_all<1><<<dim3(grid_for(m, 256), (unsigned)lng.size()), 256, 0, st>>>(ptr<GVar>(blng), sp, m);
      if (!sht.empty()) gather_bytes_cells<<<dim3(grid_for(m, 256), (unsigned)sht.size()), 256, 0, st>>>(ptr<GVar>(bsht), sp, m);
      std::vector<GVar> lz;
      for (auto &x : vr) if (x.fstart && x.has_special) lz.push_back(x);
      if (!lz.empty()) {
        Buf blz = upload_const(lz.data(), lz.size() * sizeof(GVar));
        gather_text_special<<<dim3(grid_for(m, 256), (unsigned)lz.size()), 256, 0, st>>>(ptr<GVar>(blz), sp, m);
      }
    }
    Buf tot = dalloc((size_t)nvar * 4);
    collect_totals<<<1, 64, 0, st>>>(ptr<uint32_t>(lens_all), seg_stride, m, nvar, ptr<uint32_t>(tot));
    const uint32_t *h = d2h_u32(tot->p, (size_t)nvar);
    sync();
    for (int i = 0; i < nvar; i++) out_at(var_cols[(size_t)i]).data_len = h[i];
  }
  return out;
}

std::unique_ptr<tfgpu_dbatch> gather_rows(const tfgpu_dbatch &in, const Buf &sel, int64_t m) { return gather_batch(in, sel, m); }  // for tf_collapse.hip

// TFGPU_LAZY_ROWS=0: the row filters gather their kept rows at once (the form of rounds 1-4; A/B measurements)
static bool lazy_rows_on() {
  static const bool off = [] { const char *e = std::getenv("TFGPU_LAZY_ROWS"); return e && e[0] == '0'; }();
  return !off;
}
// keep flags (uint32 0/1, n+1 slots) → compacted batch; identity if all kept.  lazy: the kept rows are handed on as a selection
// over `in` (tfgpu_dbatch::pending) and gathered when — and if — somebody reads them.
static std::unique_ptr<tfgpu_dbatch> compact(const tfgpu_dbatch &in, Buf keep /* n+1 u32 */, bool lazy = false) {
  int64_t n = in.nrows;
  exclusive_scan_u32(ptr<uint32_t>(keep), ptr<uint32_t>(keep), n, true);
  const uint32_t *hm = d2h_u32(ptr<uint32_t>(keep) + n);
  Buf sel = dalloc((size_t)n * 4 + 4);  // sized for every row: the selection is built while the count travels to the host
  if (n) build_selection_kernel<<<grid_for(n, 256), 256, 0, ctx().stream>>>(ptr<uint32_t>(keep), n, ptr<int32_t>(sel));
  sync();
  const uint32_t m = *hm;
  if ((int64_t)m == n) return shallow_copy(in);
  if (lazy && lazy_rows_on() && !in.pending && m) {
    auto out = std::make_unique<tfgpu_dbatch>();
    out->nrows = m; out->ns = in.ns; out->table = in.table; out->schema = in.schema; out->key_names = in.key_names;
    auto pr = std::make_shared<PendingRows>();
    pr->src = std::make_shared<const tfgpu_dbatch>(in);  // (a shallow copy: the buffers are shared)
    pr->sel = sel;
    out->pending = std::move(pr);
    return out;
  }
  return gather_batch(in, sel, m);
}

// The selection → dense transition mutates a handle its callers hold as const, and one handle can be reached from two threads on two lanes (the
// Bufferer's collector concatenating a batch the pusher is serializing): the lane's mutex is the CALLING thread's, so the transition has a lock of
// its own, and with more than one lane alive the gather is complete — not merely queued on the gathering lane's stream — when the lock is released
// (the next reader may sit on another stream).  ADVICE r5.
static std::mutex g_dense_mu;
void dense_locked(const tfgpu_dbatch &b) {
  std::lock_guard<std::mutex> dl(g_dense_mu);
  if (!b.pending) return;
  tfgpu_dbatch &mb = const_cast<tfgpu_dbatch &>(b);  // (the handle's observable value does not change: the same rows, now gathered)
  std::shared_ptr<PendingRows> pr = b.pending;
  std::unique_ptr<tfgpu_dbatch> g = gather_batch(*pr->src, pr->sel, b.nrows, b.replaced.empty() ? nullptr : &b.replaced);
  mb.cols = std::move(g->cols); mb.old_keys = std::move(g->old_keys); mb.old_present = g->old_present;
  mb.kind = g->kind; mb.src_row = g->src_row; mb.part_id = g->part_id;
  mb.replaced.clear();
  if (lanes_created() > 1) {  // the gather is QUEUED on this lane's stream: leave an event behind it for readers on other streams (a host sync here cost configs[4] 0.2 ms a pass)
    // ONE event a lane, recorded again at every transition: a reader that waits on a later record than its batch's waits for more than it must, never for less
    Context &cx = ctx();
    if (!cx.dense_event) TF_HIP(hipEventCreateWithFlags(&cx.dense_event, hipEventDisableTiming));
    TF_HIP(hipEventRecord(cx.dense_event, cx.stream));
    mb.dense_done = (void *)cx.dense_event;
    mb.dense_lane = current_lane();
  }
  mb.pending.reset();
}
// a batch another lane made dense: this lane's stream waits for that gather (no-op on the gathering lane, or once nobody else is alive)
static void wait_dense_nolock(const tfgpu_dbatch &b) {
  if (b.dense_done && b.dense_lane != current_lane()) TF_HIP(hipStreamWaitEvent(ctx().stream, (hipEvent_t)b.dense_done, 0));
}
bool has_absent(const tfgpu_dbatch &b) {
  std::lock_guard<std::mutex> dl(g_dense_mu);
  if (b.col_order) return true;  // rows with their own ColumnNames order are ragged rows too
  for (auto &c : (b.pending ? b.pending->src->cols : b.cols)) if (c.absent) return true;
  return false;
}
void refuse_absent(const tfgpu_dbatch &b) {
  if (has_absent(b))
    throw Error(TFGPU_ERR_UNSUPPORTED, "the batch holds ABSENT cells (rows whose ColumnNames leave a column out: TOASTed updates, tfgpu_column.absent); this entry computes on values and "
                                       "does not read them — tfgpu_collapse, tfgpu_keys_changed, tfgpu_partition, view / download do");
}
void dense(const tfgpu_dbatch *b, bool absent_ok) {
  if (!b) return;
  if (!absent_ok) refuse_absent(*b);
  { std::lock_guard<std::mutex> dl(g_dense_mu); if (!b->pending) { wait_dense_nolock(*b); return; } }
  std::lock_guard<std::mutex> lk(ctx().mu);
  dense_locked(*b);
}

std::unique_ptr<tfgpu_dbatch> compact_rows(const tfgpu_dbatch &in, Buf keep) { return compact(in, keep); }

static void collect_row_errors(const Buf &err, const Buf &err_term, int64_t n, ApplyCtx &ax) {
  std::vector<uint8_t> he((size_t)n);
  std::vector<int32_t> ht((size_t)n);
  d2h(he.data(), err->p, (size_t)n);
  if (err_term) d2h(ht.data(), err_term->p, (size_t)n * 4);
  sync();
  for (int64_t r = 0; r < n; r++) if (he[(size_t)r]) ax.errs.push_back(tfgpu_row_error{r, he[(size_t)r], ax.step, err_term ? ht[(size_t)r] : -1});
}

// The predicate program (an OR of ANDs of terms) over a batch, then the compaction of the kept rows.  `check_kinds`:
// filter_rows' rule that only Inserts may be filtered (Update / Delete rows are fatal row errors); the sql transformer
// evaluates its WHERE on every row event.
static std::unique_ptr<tfgpu_dbatch> run_filter(const std::vector<FExpr> &exprs, bool table_applies, bool check_kinds, const tfgpu_dbatch &in, ApplyCtx &ax);
static std::unique_ptr<tfgpu_dbatch> apply_filter_rows(const tfgpu_plan &p, const tfgpu_dbatch &in, ApplyCtx &ax) {
  // pass-through conditions that hold for the whole batch (one table per batch)
  if (!p.tables.match_table(in.ns, in.table) || is_system_table(in.table)) {
    // impossible kinds still raise errors before the table check (filter_rows.go:103-113)
    if (!in.kind) return shallow_copy(in);
  }
  const bool table_applies = p.tables.match_table(in.ns, in.table) && !is_system_table(in.table);
  return run_filter(p.exprs, table_applies, true, in, ax);
}
static std::unique_ptr<tfgpu_dbatch> run_filter(const std::vector<FExpr> &p_exprs, bool table_applies, bool check_kinds, const tfgpu_dbatch &in, ApplyCtx &ax) {
  int64_t n = in.nrows;
  hipStream_t st = ctx().stream;
  // device program
  std::vector<DTerm> terms; std::vector<int32_t> expr_start{0};
  std::vector<int64_t> ints; std::vector<double> floats; std::vector<uint32_t> soff; std::string sdata;
  if (table_applies) {
    for (auto &e : p_exprs) {
      for (auto &t : e.terms) {
        DTerm d{};
        d.col = -1;
        for (size_t i = 0; i < in.cols.size(); i++) if (in.cols[i].name == t.attr) { d.col = (int32_t)i; break; }
        d.op = t.op; d.vtype = t.vtype; d.is_list = t.is_list;
        switch (t.vtype) {
          case FV_FLOAT: d.nvals = (int32_t)t.floats.size(); d.ioff = (int32_t)floats.size(); floats.insert(floats.end(), t.floats.begin(), t.floats.end()); break;
          case FV_STRING:
            d.nvals = (int32_t)t.strs.size(); d.soff = (int32_t)soff.size();
            for (auto &s : t.strs) { soff.push_back((uint32_t)sdata.size()); sdata += s; }
            soff.push_back((uint32_t)sdata.size());
            break;
          default: d.nvals = (int32_t)t.ints.size(); d.ioff = (int32_t)ints.size(); ints.insert(ints.end(), t.ints.begin(), t.ints.end());
        }
        terms.push_back(d);
      }
      expr_start.push_back((int32_t)terms.size());
    }
  }
  {
    std::vector<const DColumn *> need;
    for (auto &t : terms) if (t.col >= 0) need.push_back(&in.cols[(size_t)t.col]);
    materialize(in, &need);
  }
  std::vector<DCol> cols;
  for (auto &c : in.cols) {
    if (c.lazy()) { DColumn shell = c; shell.view = nullptr; cols.push_back(dcol_of(shell)); }  // no term reads it: its payload stays unpacked
    else cols.push_back(dcol_of(c));
  }
  auto up = [&](const void *src, size_t bytes) { return upload_const(src, bytes); };  // tables the kernels only read
  Buf bcols = up(cols.data(), cols.size() * sizeof(DCol)), bterms = up(terms.data(), terms.size() * sizeof(DTerm));
  Buf bexpr = up(expr_start.data(), expr_start.size() * 4), bints = up(ints.data(), ints.size() * 8), bfl = up(floats.data(), floats.size() * 8);
  Buf bsoff = up(soff.data(), soff.size() * 4), bsd = up(sdata.data(), sdata.size());
  Buf keep = dalloc((size_t)(n + 1) * 4), err = dalloc((size_t)n + 1), eterm = dalloc((size_t)n * 4 + 4), nerr = dalloc_zero(4);
  FilterParams fp;
  fp.cols = ptr<DCol>(bcols); fp.ncols = (int32_t)cols.size(); fp.terms = ptr<DTerm>(bterms); fp.expr_start = ptr<int32_t>(bexpr);
  fp.nexpr = table_applies ? (int32_t)p_exprs.size() : 0;
  fp.ints = ptr<int64_t>(bints); fp.floats = ptr<double>(bfl); fp.str_off = ptr<uint32_t>(bsoff); fp.str_data = ptr<uint8_t>(bsd);
  fp.absent = nullptr;
  Buf babs;
  {
    std::vector<const uint8_t *> abs;
    bool any = false;
    for (auto &c : in.cols) { abs.push_back(ptr<uint8_t>(c.absent)); any = any || c.absent; }
    if (any) { babs = up(abs.data(), abs.size() * sizeof(const uint8_t *)); fp.absent = reinterpret_cast<const uint8_t *const *>(babs->p); }
  }
  fp.kind = check_kinds ? ptr<uint8_t>(in.kind) : nullptr; fp.nrows = n; fp.keep = ptr<uint32_t>(keep); fp.err = ptr<uint8_t>(err); fp.err_term = ptr<int32_t>(eterm); fp.nerr = ptr<uint32_t>(nerr);
  if (!table_applies) {
    // only the kind check applies: emulate with zero expressions and keep-all for inserts
    fp.nexpr = -1;
  }
  {
    KernelTimer t("filter_rows_eval");
    if (n) filter_eval_kernel<<<grid_for(n, 256), 256, 0, st>>>(fp);
  }
  const uint32_t *hn = d2h_u32(nerr->p);
  auto out = compact(in, keep, true);  // syncs; the kept rows stay a selection until somebody reads them
  if (*hn) collect_row_errors(err, eterm, n, ax);
  return out;
}

// ============================================================================
// a10 skip_events — skip_events.go:52-62
// ============================================================================
__global__ void kind_keep_kernel(const uint8_t *kind, int64_t n, uint32_t skip_mask, uint32_t *keep) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) keep[r] = ((skip_mask >> kind[r]) & 1u) ? 0u : 1u;
}
static std::unique_ptr<tfgpu_dbatch> apply_skip_events(const tfgpu_plan &p, const tfgpu_dbatch &in) {
  uint32_t mask = (p.skip[0] ? 1u : 0) | (p.skip[1] ? 2u : 0) | (p.skip[2] ? 4u : 0);
  int64_t n = in.nrows;
  if (!in.kind) {  // all rows are inserts
    if (!(mask & 1u)) return shallow_copy(in);
    Buf sel = dalloc(4);
    return gather_batch(in, sel, 0);
  }
  Buf keep = dalloc((size_t)(n + 1) * 4);
  if (n) kind_keep_kernel<<<grid_for(n, 256), 256, 0, ctx().stream>>>(ptr<uint8_t>(in.kind), n, mask, ptr<uint32_t>(keep));
  return compact(in, keep, true);
}

// ============================================================================
// a6 convert_to_string — to_string.go:58-97
// ============================================================================
// nil_empty: a nil value has no text (strictify leaves nil alone; convert_to_string prints "<nil>")
__global__ void __launch_bounds__(256) tostring_len_kernel(DCol c, int64_t n, uint32_t *len, int nil_empty) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  uint8_t buf[64]; const uint8_t *ext;
  if (nil_empty && c.validity && !((c.validity[r >> 3] >> (r & 7)) & 1)) { len[r] = 0; return; }
  len[r] = (uint32_t)serialize_small(c, r, buf, &ext);
}
__global__ void __launch_bounds__(256) tostring_write_kernel(DCol c, int64_t n, const uint32_t *off, uint8_t *data, int nil_empty) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  uint8_t buf[64]; const uint8_t *ext;
  if (nil_empty && c.validity && !((c.validity[r >> 3] >> (r & 7)) & 1)) return;
  int len = serialize_small(c, r, buf, &ext);
  uint8_t *dst = data + off[r];
  const uint8_t *src = ext ? ext : buf;
  struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
  int i = 0;
  for (; i + 8 <= len; i += 8) reinterpret_cast<U64 *>(dst + i)->v = reinterpret_cast<const U64 *>(src + i)->v;  // unaligned 8-byte moves
  for (; i < len; i++) dst[i] = src[i];
}

static DColumn column_to_string(const DColumn &c, int64_t n, bool to_bytes, int max_len_hint, int nil_empty = 0) {
  hipStream_t st = ctx().stream;
  DColumn o;
  o.name = c.name; o.dtype = to_bytes ? TFGPU_T_BYTES : TFGPU_T_UTF8; o.repr = to_bytes ? TFGPU_R_BYTES : TFGPU_R_STRING;
  o.offsets = dalloc((size_t)(n + 1) * 4);
  DCol dc = dcol_of(c);  // callers materialise text columns first
  KernelTimer t("to_string");
  if (n) tostring_len_kernel<<<grid_for(n, 256), 256, 0, st>>>(dc, n, ptr<uint32_t>(o.offsets), nil_empty);
  exclusive_scan_u32(ptr<uint32_t>(o.offsets), ptr<uint32_t>(o.offsets), n, true);
  uint64_t cap = repr_is_var(c.repr) ? c.data_len + (uint64_t)n * 5 : (uint64_t)n * (uint64_t)max_len_hint;
  o.data = dalloc(cap);
  if (n) tostring_write_kernel<<<grid_for(n, 256), 256, 0, st>>>(dc, n, ptr<uint32_t>(o.offsets), ptr<uint8_t>(o.data), nil_empty);
  const uint32_t *tot = d2h_u32(ptr<uint32_t>(o.offsets) + n);
  sync();
  o.data_len = *tot;
  return o;
}

DColumn column_to_text(const DColumn &c, int64_t n, bool to_bytes) { DColumn o = column_to_string(c, n, to_bytes, 64, 1); o.validity = c.validity; return o; }  // for tfgpu_strictify (tf_csv.hip)

static std::unique_ptr<tfgpu_dbatch> apply_to_string(const tfgpu_plan &p, const tfgpu_dbatch &in) {
  if (p.skip_utc) {
    for (auto &c : in.cols) if (p.columns.match(c.name) && c.repr == TFGPU_R_TIME)
      throw Error(TFGPU_ERR_UNSUPPORTED, "convert_to_string skip_utc_conversion=true needs per-value time zones, which the columnar batch does not carry");
  }
  {
    std::vector<const DColumn *> need;
    for (auto &c : in.cols) if (p.columns.match(c.name) && c.validity) need.push_back(&c);
    materialize(in, &need);
  }
  auto out = shallow_copy(in);
  for (auto &sc : out->schema) if (p.columns.match(sc.first)) sc.second = p.to_bytes ? TFGPU_T_BYTES : TFGPU_T_UTF8;  // to_string.go:114-127
  for (auto &c : out->cols) {
    if (!p.columns.match(c.name)) continue;
    require_serializable(c, "convert_to_string");
    // already text with identical bytes: only the type tag changes
    if (repr_is_var(c.repr) && !c.validity) { c.dtype = p.to_bytes ? TFGPU_T_BYTES : TFGPU_T_UTF8; c.repr = p.to_bytes ? TFGPU_R_BYTES : TFGPU_R_STRING; continue; }
    const Buf ab = c.absent;   // the transformer converts the values of the names a row LISTS (to_string.go: it walks item.ColumnNames): an ABSENT cell stays one
    c = column_to_string(c, in.nrows, p.to_bytes, 64);
    if (ab) { c.absent = ab; c.validity = validity_minus_absent(nullptr, ab, in.nrows); }
  }
  return out;
}

// ============================================================================
// a7 convert_to_datetime — to_datetime.go:89-149
// ============================================================================
template <typename T>
__global__ void todatetime_kernel(const T *in, int64_t n, int64_t *out) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) out[r] = (int64_t)in[r];
}
static std::unique_ptr<tfgpu_dbatch> apply_to_datetime(const tfgpu_plan &p, const tfgpu_dbatch &in) {
  auto out = shallow_copy(in);
  int64_t n = in.nrows;
  for (auto &sc : out->schema) if (p.columns.match(sc.first) && (sc.second == TFGPU_T_INT32 || sc.second == TFGPU_T_UINT32)) sc.second = TFGPU_T_DATETIME;  // to_datetime.go:125-133
  for (auto &c : out->cols) {
    if (!(p.columns.match(c.name) && (c.dtype == TFGPU_T_INT32 || c.dtype == TFGPU_T_UINT32))) continue;
    DColumn o;
    o.name = c.name; o.dtype = TFGPU_T_DATETIME; o.repr = TFGPU_R_TIME;
    o.values = dalloc_zero((size_t)n * 8);  // SerializeToDateTime falls back to time.Unix(0,0) on a type mismatch
    KernelTimer t("to_datetime");
    if (n && c.dtype == TFGPU_T_INT32 && c.repr == TFGPU_R_INT32) todatetime_kernel<int32_t><<<grid_for(n, 256), 256, 0, ctx().stream>>>(ptr<int32_t>(c.values), n, ptr<int64_t>(o.values));
    else if (n && c.dtype == TFGPU_T_UINT32 && c.repr == TFGPU_R_UINT32) todatetime_kernel<uint32_t><<<grid_for(n, 256), 256, 0, ctx().stream>>>(ptr<uint32_t>(c.values), n, ptr<int64_t>(o.values));
    // nil values also become time.Unix(0,0): the value.(int32) assertion fails — an ABSENT cell is no value at all (the loop walks item.ColumnNames): it stays one
    if (c.absent) { o.absent = c.absent; o.validity = validity_minus_absent(nullptr, c.absent, n); }
    c = std::move(o);
  }
  return out;
}

// ============================================================================
// a13 sharder_transformer — sharder.go:130-145 (CRC32-IEEE of '.'-joined strings)
// ============================================================================
__device__ __forceinline__ uint32_t crc32_update(uint32_t crc, uint8_t b, const uint32_t *tab) { return tab[(crc ^ b) & 0xFF] ^ (crc >> 8); }

struct SharderParams {
  const DCol *cols;   // in schema order, already resolved (repr==0 → missing → "<nil>")
  int32_t ncols;
  int64_t nrows;
  uint32_t shards;
  uint32_t *part_id;
};
__global__ void __launch_bounds__(256) sharder_kernel(SharderParams p) {
  __shared__ uint32_t tab[256];
  {  // build the IEEE table (reflected 0xEDB88320) once per block
    uint32_t c = threadIdx.x;
    for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    tab[threadIdx.x] = c;
  }
  __syncthreads();
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  uint32_t crc = 0xFFFFFFFFu;
  for (int i = 0; i < p.ncols; i++) {
    if (i) crc = crc32_update(crc, '.', tab);
    {  // an integer key (the usual one): its digits go from registers into the CRC, most significant first — no text buffer
      const DCol &c = p.cols[i];
      bool isint = true, neg = false;
      uint64_t mag = 0;
      if (c.repr == 0 || !is_valid(c, r)) isint = false;
      else switch (c.repr) {
        case TFGPU_R_INT8: { const int64_t v = ((const int8_t *)c.values)[r]; neg = v < 0; mag = neg ? (uint64_t)(-v) : (uint64_t)v; break; }
        case TFGPU_R_INT16: { const int64_t v = ((const int16_t *)c.values)[r]; neg = v < 0; mag = neg ? (uint64_t)(-v) : (uint64_t)v; break; }
        case TFGPU_R_INT32: { const int64_t v = ((const int32_t *)c.values)[r]; neg = v < 0; mag = neg ? (uint64_t)(-v) : (uint64_t)v; break; }
        case TFGPU_R_INT64: { const int64_t v = ((const int64_t *)c.values)[r]; neg = v < 0; mag = neg ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v; break; }
        case TFGPU_R_UINT8: mag = ((const uint8_t *)c.values)[r]; break;
        case TFGPU_R_UINT16: mag = ((const uint16_t *)c.values)[r]; break;
        case TFGPU_R_UINT32: mag = ((const uint32_t *)c.values)[r]; break;
        case TFGPU_R_UINT64: mag = ((const uint64_t *)c.values)[r]; break;
        default: isint = false;
      }
      if (isint) {
        if (neg) crc = crc32_update(crc, '-', tab);
        bool started = false;
        auto piece = [&](uint32_t x, bool last) {  // nine digits of x < 10^9; leading zeros of the number are not part of its text
          uint32_t pw = 100000000u;
#pragma unroll
          for (int k = 0; k < 9; k++) {
            const uint32_t d = x / pw; x -= d * pw; pw /= 10u;
            started = started || d != 0 || (last && k == 8);
            if (started) crc = crc32_update(crc, (uint8_t)('0' + d), tab);
          }
        };
        if (mag >> 32) {
          const uint64_t q = mag / 1000000000ull;
          const uint32_t lo = (uint32_t)(mag - q * 1000000000ull);
          if (q >> 32) {
            const uint32_t q2 = (uint32_t)(q / 1000000000ull);
            const uint32_t d1 = q2 / 10u;  // q2 <= 18
            if (d1) crc = crc32_update(crc, (uint8_t)('0' + d1), tab);
            crc = crc32_update(crc, (uint8_t)('0' + (q2 - d1 * 10u)), tab);
            started = true;
            piece((uint32_t)(q - (uint64_t)q2 * 1000000000ull), false);
          } else {
            const uint32_t qq = (uint32_t)q;  // 1 .. 4 294 967 295: ten digits at most
            const uint32_t d9 = qq / 1000000000u;
            if (d9) { crc = crc32_update(crc, (uint8_t)('0' + d9), tab); started = true; }
            piece(qq - d9 * 1000000000u, false);
          }
          piece(lo, true);
        } else {
          const uint32_t x = (uint32_t)mag;
          const uint32_t d9 = x / 1000000000u;
          if (d9) { crc = crc32_update(crc, (uint8_t)('0' + d9), tab); started = true; }
          piece(x - d9 * 1000000000u, true);
        }
        continue;
      }
    }
    uint8_t buf[64]; const uint8_t *ext = nullptr; int len;
    if (p.cols[i].repr == 0) { buf[0] = '<'; buf[1] = 'n'; buf[2] = 'i'; buf[3] = 'l'; buf[4] = '>'; len = 5; }
    else len = serialize_small(p.cols[i], r, buf, &ext);
    if (ext) for (int k = 0; k < len; k++) crc = crc32_update(crc, ext[k], tab);
    else for (int k = 0; k < len; k++) crc = crc32_update(crc, buf[k], tab);
  }
  p.part_id[r] = (crc ^ 0xFFFFFFFFu) % p.shards;
}

static std::unique_ptr<tfgpu_dbatch> apply_sharder(const tfgpu_plan &p, const tfgpu_dbatch &in, const tfgpu_schema *schema_order) {
  auto out = shallow_copy(in);
  int64_t n = in.nrows;
  // Columns are visited in TableSchema order; without a separate schema the
  // batch column order stands in for it (they coincide for every source that
  // builds ColumnNames from the schema).
  {
    std::vector<const DColumn *> need;
    for (auto &c : in.cols) if (p.columns.match(c.name)) need.push_back(&c);
    materialize(in, &need);
  }
  std::vector<DCol> cols;
  if (in.schema.empty()) {
    for (auto &c : in.cols) {
      if (!p.columns.match(c.name)) continue;
      require_serializable(c, "sharder_transformer");
      cols.push_back(dcol_of(c));
    }
  } else {  // item.TableSchema.Columns() in order, values through AsMap()[name]: a schema column without a value is nil
    for (auto &sc : in.schema) {
      if (!p.columns.match(sc.first)) continue;
      const DColumn *found = nullptr;
      for (auto &c : in.cols) if (c.name == sc.first) found = &c;  // AsMap: the last duplicate name wins
      if (!found) { DCol nil{}; nil.repr = 0; nil.dtype = sc.second; cols.push_back(nil); continue; }
      require_serializable(*found, "sharder_transformer");
      DCol d = dcol_of(*found);
      d.dtype = sc.second;
      cols.push_back(d);
    }
  }
  Buf bc = upload_small(cols.data(), cols.size() * sizeof(DCol));
  out->part_id = dalloc((size_t)n * 4 + 4);
  SharderParams sp{ptr<DCol>(bc), (int32_t)cols.size(), n, (uint32_t)p.shards, ptr<uint32_t>(out->part_id)};
  KernelTimer t("sharder_crc32");
  if (n) sharder_kernel<<<grid_for(n, 256), 256, 0, ctx().stream>>>(sp);
  return out;
}

// ============================================================================
// metadata-only transformers
// ============================================================================
static std::unique_ptr<tfgpu_dbatch> apply_rename(const tfgpu_plan &p, const tfgpu_dbatch &in) {  // rename.go:46-61
  auto out = shallow_copy(in);
  for (auto &r : p.renames) if (r[0] == in.ns && r[1] == in.table) { out->ns = r[2]; out->table = r[3]; break; }
  return out;
}
static std::unique_ptr<tfgpu_dbatch> apply_filter_columns(const tfgpu_plan &p, const tfgpu_dbatch &in) {  // filter_columns_transformer.go:51-79
  auto out = shallow_copy(in);
  out->cols.clear();
  for (auto &c : in.cols) if (p.columns.match(c.name)) out->cols.push_back(c);
  out->schema.clear();
  for (auto &sc : in.schema) if (p.columns.match(sc.first)) out->schema.push_back(sc);
  out->old_keys.clear();  // trimChangeItem also trims OldKeys (filter_columns_transformer.go:187-213)
  for (auto &c : in.old_keys) if (p.columns.match(c.name)) out->old_keys.push_back(c);
  return out;
}

__global__ void kind_is_kernel(const uint8_t *kind, int64_t n, uint8_t k, uint8_t *bits) {  // one thread per output byte
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b * 8 >= n) return;
  uint32_t v = 0;
  for (int j = 0; j < 8; j++) { const int64_t r = b * 8 + j; if (r < n && kind[r] == k) v |= 1u << j; }
  bits[b] = (uint8_t)v;
}
__global__ void kind_any_kernel(const uint8_t *kind, int64_t n, uint8_t k, uint32_t *flag) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && kind[r] == k) *flag = 1u;
}
__global__ void bitmap_or_kernel(uint8_t *a, const uint8_t *b, int64_t nbytes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nbytes) a[i] |= b[i];
}
// OldKeys of a run whose rows already carry some: an Update takes the key column's current value, any other row keeps its old one
// (one thread per validity byte = 8 rows)
__global__ void replace_pk_select_kernel(const uint8_t *kind, int64_t n, int w, const uint8_t *cur, const uint8_t *cur_valid, const uint8_t *old, const uint8_t *old_valid,
                                         uint8_t *out, uint8_t *out_valid) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b * 8 >= n) return;
  uint32_t v = 0;
  for (int j = 0; j < 8; j++) {
    const int64_t r = b * 8 + j;
    if (r >= n) break;
    const bool upd = kind[r] == TFGPU_K_UPDATE;
    const uint8_t *src = upd ? cur : old, *sv = upd ? cur_valid : old_valid;
    for (int k = 0; k < w; k++) out[r * w + k] = src[r * w + k];
    if (!sv || ((sv[r >> 3] >> (r & 7)) & 1)) v |= 1u << j;
  }
  out_valid[b] = (uint8_t)v;
}
// replace_primary_key.go:82-101: the TableSchema is replaced (keys first / flags rewritten); ColumnNames and ColumnValues
// stay as they are; an Update gets OldKeys = the NEW keys' current values (createOldKeys :51-80).
static std::unique_ptr<tfgpu_dbatch> apply_replace_pk(const tfgpu_plan &p, const tfgpu_dbatch &in) {
  auto out = shallow_copy(in);
  std::vector<SchemaCol> cols;
  if (!in.schema.empty()) for (auto &c : in.schema) cols.push_back(SchemaCol{c.first, c.second, 0u});
  else for (auto &c : in.cols) cols.push_back(SchemaCol{c.name, c.dtype, 0u});
  plan_result_columns(p, cols);
  out->schema.clear(); out->key_names.clear();
  for (auto &c : cols) { out->schema.emplace_back(c.name, c.dtype); if (c.flags & TFGPU_COL_KEY) out->key_names.push_back(c.name); }
  bool has_update = false;
  if (in.kind && in.nrows) {  // one flag word read back, not the kinds
    Buf flag = dalloc_zero(4);
    kind_any_kernel<<<grid_for(in.nrows, 256), 256, 0, ctx().stream>>>(ptr<uint8_t>(in.kind), in.nrows, (uint8_t)TFGPU_K_UPDATE, ptr<uint32_t>(flag));
    const uint32_t *h = d2h_u32(flag->p);
    sync();
    has_update = *h != 0;
  }
  if (has_update) {
    std::vector<DColumn> cur;
    for (auto &k : p.new_keys)  // key order; a key missing from ColumnNames leaves a nil value in the reference — not modelled
      for (auto &c : in.cols) if (c.name == k) { cur.push_back(c); break; }
    if (cur.size() != p.new_keys.size()) throw Error(TFGPU_ERR_UNSUPPORTED, "replace_primary_key: a new key is not among the batch's columns");
    if (in.old_keys.empty()) {
      out->old_keys = cur;
      out->old_present = dalloc((size_t)(in.nrows + 7) / 8 + 8);
      kind_is_kernel<<<grid_for((in.nrows + 7) / 8, 256), 256, 0, ctx().stream>>>(ptr<uint8_t>(in.kind), in.nrows, (uint8_t)TFGPU_K_UPDATE, ptr<uint8_t>(out->old_present));
    } else {
      // the batch already carries OldKeys (every pg / Debezium CDC batch does): Updates get the NEW keys' current values, the other
      // rows keep theirs.  Columnar only when both sets have the same names, types and fixed-width representations (per-item
      // KeyNames otherwise: that run stays with the stock transformer)
      bool same = in.old_keys.size() == cur.size();
      for (size_t i = 0; same && i < cur.size(); i++)
        same = in.old_keys[i].name == cur[i].name && in.old_keys[i].repr == cur[i].repr && in.old_keys[i].dtype == cur[i].dtype && !repr_is_var(cur[i].repr) && cur[i].repr != TFGPU_R_TIME;
      if (!same) throw Error(TFGPU_ERR_UNSUPPORTED, "replace_primary_key: a run that mixes Updates with rows carrying OldKeys of other key names (or text / time keys) needs per-item KeyNames; not columnar");
      const int64_t n = in.nrows;
      out->old_keys.clear();
      for (size_t i = 0; i < cur.size(); i++) {
        DColumn d = in.old_keys[i];
        const int w = repr_width(cur[i].repr);
        d.values = dalloc((size_t)n * (size_t)w);
        d.validity = dalloc((size_t)(n + 7) / 8 + 8);
        replace_pk_select_kernel<<<grid_for((n + 7) / 8, 256), 256, 0, ctx().stream>>>(ptr<uint8_t>(in.kind), n, w, (const uint8_t *)cur[i].values->p, ptr<uint8_t>(cur[i].validity),
                                                                                      (const uint8_t *)in.old_keys[i].values->p, ptr<uint8_t>(in.old_keys[i].validity), (uint8_t *)d.values->p, ptr<uint8_t>(d.validity));
        out->old_keys.push_back(std::move(d));
      }
      if (in.old_present) {  // present = Update, or it was
        out->old_present = dalloc((size_t)(n + 7) / 8 + 8);
        kind_is_kernel<<<grid_for((n + 7) / 8, 256), 256, 0, ctx().stream>>>(ptr<uint8_t>(in.kind), n, (uint8_t)TFGPU_K_UPDATE, ptr<uint8_t>(out->old_present));
        bitmap_or_kernel<<<grid_for((n + 7) / 8, 256), 256, 0, ctx().stream>>>(ptr<uint8_t>(out->old_present), ptr<uint8_t>(in.old_present), (n + 7) / 8);
      }
    }
  }
  return out;
}

// ============================================================================
// a14 sql — clickhouse/clickhouse_local.go:97-294, the predicate + cast subset of tf_sql.cpp
//   Apply: SplitUpdatedPKeys → Collapse → rows as JSONEachRow (nil → the ClickHouse type's default) → query → rows back,
//   each re-attached to the input row with the same primary key: meta, kind, and for Update / Delete the result's values as
//   OldKeys (Delete: no column values).  Here the query is evaluated row by row on the device, so "the input row with the
//   same key" is the row the result row came from; that is the same thing as long as the key columns pass through the
//   select list under their own names (required below) and no key repeats in the sub-batch (Collapse sees to that).
// ============================================================================
std::unique_ptr<tfgpu_dbatch> collapse_rows(const tfgpu_dbatch &in);        // tf_collapse.hip
int64_t keys_changed_rows(const tfgpu_dbatch &in, uint8_t *host_flags);   // tf_collapse.hip

struct SqlIntParams { const void *src; int32_t src_repr; const uint8_t *validity; int64_t n; int64_t addend[8]; int32_t ty[8]; int32_t nops; void *out; int32_t out_ty; };
__device__ __forceinline__ int64_t sql_wrap(int64_t v, int ty) {
  switch (ty) {
    case SQL_I8: return (int8_t)v; case SQL_I16: return (int16_t)v; case SQL_I32: return (int32_t)v;
    case SQL_U8: return (uint8_t)v; case SQL_U16: return (uint16_t)v; case SQL_U32: return (uint32_t)v;
    default: return v;  // 64 bits either way
  }
}
__global__ void __launch_bounds__(256) sql_int_kernel(SqlIntParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.n) return;
  int64_t v = 0;
  const bool nil = p.validity && !((p.validity[r >> 3] >> (r & 7)) & 1);  // a nil value reaches ClickHouse as the type's default
  if (!nil) switch (p.src_repr) {
    case TFGPU_R_INT8: v = ((const int8_t *)p.src)[r]; break; case TFGPU_R_INT16: v = ((const int16_t *)p.src)[r]; break;
    case TFGPU_R_INT32: v = ((const int32_t *)p.src)[r]; break; case TFGPU_R_INT64: v = ((const int64_t *)p.src)[r]; break;
    case TFGPU_R_UINT8: case TFGPU_R_BOOL: v = ((const uint8_t *)p.src)[r]; break; case TFGPU_R_UINT16: v = ((const uint16_t *)p.src)[r]; break;
    case TFGPU_R_UINT32: v = ((const uint32_t *)p.src)[r]; break; default: v = (int64_t)((const uint64_t *)p.src)[r];
  }
  for (int i = 0; i < p.nops; i++) v = sql_wrap((int64_t)((uint64_t)v + (uint64_t)p.addend[i]), p.ty[i]);
  switch (p.out_ty) {
    case SQL_I8: case SQL_U8: ((uint8_t *)p.out)[r] = (uint8_t)v; break;
    case SQL_I16: case SQL_U16: ((uint16_t *)p.out)[r] = (uint16_t)v; break;
    case SQL_I32: case SQL_U32: ((uint32_t *)p.out)[r] = (uint32_t)v; break;
    default: ((uint64_t *)p.out)[r] = (uint64_t)v;
  }
}
__global__ void __launch_bounds__(256) sql_const_text_kernel(uint32_t *off, uint8_t *data, int64_t n, const uint8_t *text, uint32_t len) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  off[r] = (uint32_t)r * len;
  if (r < n) for (uint32_t k = 0; k < len; k++) data[(uint64_t)r * len + k] = text[k];
}
struct SqlNilCol { const uint8_t *src; uint8_t *dst; const uint8_t *validity; int32_t width; };
// dst[r] = the value, or the type's zero where the row is nil; item = column * n + row
__global__ void __launch_bounds__(256) sql_default_nils_kernel(const SqlNilCol *cols, int32_t ncols, int64_t n) {
  const int32_t j = (int32_t)blockIdx.y; const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (the column is the grid's y: a scalar)
  if (r >= n || j >= ncols) return;
  const SqlNilCol c = cols[j];
  const bool ok = (c.validity[r >> 3] >> (r & 7)) & 1;
  switch (c.width) {
    case 1: c.dst[r] = ok ? c.src[r] : 0; break;
    case 2: ((uint16_t *)c.dst)[r] = ok ? ((const uint16_t *)c.src)[r] : 0; break;
    case 4: ((uint32_t *)c.dst)[r] = ok ? ((const uint32_t *)c.src)[r] : 0u; break;
    default: ((uint64_t *)c.dst)[r] = ok ? ((const uint64_t *)c.src)[r] : 0ull;
  }
}
__global__ void __launch_bounds__(256) sql_kind_bitmaps_kernel(const uint8_t *kind, int64_t n, uint8_t *has_old, uint8_t *has_cols) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b * 8 >= n) return;
  uint8_t o = 0, c = 0;
  for (int k = 0; k < 8 && b * 8 + k < n; k++) {
    const uint8_t kd = kind[b * 8 + k];
    if (kd == TFGPU_K_UPDATE || kd == TFGPU_K_DELETE) o |= (uint8_t)(1u << k);
    if (kd != TFGPU_K_DELETE) c |= (uint8_t)(1u << k);
  }
  has_old[b] = o; has_cols[b] = c;
}
static int sql_repr_of(int ch) {
  switch (ch) {
    case SQL_I8: return TFGPU_R_INT8; case SQL_I16: return TFGPU_R_INT16; case SQL_I32: return TFGPU_R_INT32; case SQL_I64: return TFGPU_R_INT64;
    case SQL_U8: return TFGPU_R_UINT8; case SQL_U16: return TFGPU_R_UINT16; case SQL_U32: return TFGPU_R_UINT32; case SQL_U64: return TFGPU_R_UINT64;
    case SQL_F64: return TFGPU_R_FLOAT64; case SQL_STRING: return TFGPU_R_STRING; default: return TFGPU_R_TIME;
  }
}

// ---- the expression program: SQL_EXPR items and a WHERE that is a tree (tf_sql.cpp).  One lane evaluates one row: a postfix
//      program over int64 slots, every arithmetic result wrapped to its ClickHouse type; text enters through leaves
//      (length, cityHash64, comparison against literals).  Text-valued nodes (lower / upper / toString) are whole columns. ----
struct SqlLeaf { const void *values; const uint32_t *offsets; const uint8_t *data; int32_t repr; int32_t pad; };
struct SqlIns { int32_t op, ty, a, b; int64_t imm; };
enum { BI_COL, BI_IMM, BI_ADD, BI_SUB, BI_MUL, BI_NEG, BI_WRAP, BI_CMP, BI_AND, BI_OR, BI_NOT, BI_LEN, BI_CITY, BI_SCMP, BI_IN, BI_SIN };
struct SqlProg { const SqlIns *ins; int32_t nins; const SqlLeaf *leaves; const int64_t *ints; const uint32_t *soff; const uint8_t *sdata; int64_t n; void *out; int32_t out_ty; uint32_t *keep; };
constexpr int SQL_STACK = 16;

// CityHash64 of CityHash v1.0.2 (the version ClickHouse carries as CityHash_v1_0_2; cityHash64(String) is CityHash64(data, size))
namespace city {
constexpr uint64_t k0 = 0xc3a5c85c97cb3127ull, k1 = 0xb492b66fbe98f273ull, k2 = 0x9ae16a3b2f90404full, k3 = 0xc949d7c7509e6557ull;
__device__ __forceinline__ uint64_t f64(const uint8_t *p) { uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | p[i]; return v; }
__device__ __forceinline__ uint64_t f32(const uint8_t *p) { return (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24); }
__device__ __forceinline__ uint64_t rot(uint64_t v, int s) { return s == 0 ? v : (v >> s) | (v << (64 - s)); }
__device__ __forceinline__ uint64_t smix(uint64_t v) { return v ^ (v >> 47); }
__device__ __forceinline__ uint64_t h16(uint64_t u, uint64_t v) {
  const uint64_t kMul = 0x9ddfea08eb382d69ull;
  uint64_t a = (u ^ v) * kMul; a ^= a >> 47;
  uint64_t b = (v ^ a) * kMul; b ^= b >> 47;
  return b * kMul;
}
struct P { uint64_t first, second; };
__device__ __forceinline__ P weak(uint64_t w, uint64_t x, uint64_t y, uint64_t z, uint64_t a, uint64_t b) {
  a += w; b = rot(b + a + z, 21);
  const uint64_t c = a;
  a += x; a += y; b += rot(a, 44);
  return P{a + z, b + c};
}
__device__ __forceinline__ P weak(const uint8_t *s, uint64_t a, uint64_t b) { return weak(f64(s), f64(s + 8), f64(s + 16), f64(s + 24), a, b); }
__device__ uint64_t hash64(const uint8_t *s, uint64_t len) {
  if (len <= 16) {
    if (len > 8) { const uint64_t a = f64(s), b = f64(s + len - 8); return h16(a, rot(b + len, (int)len)) ^ b; }
    if (len >= 4) { const uint64_t a = f32(s); return h16(len + (a << 3), f32(s + len - 4)); }
    if (len > 0) { const uint8_t a = s[0], b = s[len >> 1], c = s[len - 1]; const uint32_t y = (uint32_t)a + ((uint32_t)b << 8), z = (uint32_t)len + ((uint32_t)c << 2); return smix(y * k2 ^ z * k3) * k2; }
    return k2;
  }
  if (len <= 32) {
    const uint64_t a = f64(s) * k1, b = f64(s + 8), c = f64(s + len - 8) * k2, d = f64(s + len - 16) * k0;
    return h16(rot(a - b, 43) + rot(c, 30) + d, a + rot(b ^ k3, 20) - c + len);
  }
  if (len <= 64) {
    uint64_t z = f64(s + 24), a = f64(s) + (len + f64(s + len - 16)) * k0, b = rot(a + z, 52), c = rot(a, 37);
    a += f64(s + 8); c += rot(a, 7); a += f64(s + 16);
    const uint64_t vf = a + z, vs = b + rot(a, 31) + c;
    a = f64(s + 16) + f64(s + len - 32); z = f64(s + len - 8); b = rot(a + z, 52); c = rot(a, 37);
    a += f64(s + len - 24); c += rot(a, 7); a += f64(s + len - 16);
    const uint64_t wf = a + z, ws = b + rot(a, 31) + c;
    const uint64_t r = smix((vf + ws) * k2 + (wf + vs) * k0);
    return smix(r * k0 + vs) * k2;
  }
  uint64_t x = f64(s), y = f64(s + len - 16) ^ k1, z = f64(s + len - 56) ^ k0;
  P v = weak(s + len - 64, len, y), w = weak(s + len - 32, len * k1, k0);
  z += smix(v.second) * k1;
  x = rot(z + x, 39) * k1;
  y = rot(y, 33) * k1;
  len = (len - 1) & ~(uint64_t)63;
  do {
    x = rot(x + y + v.first + f64(s + 16), 37) * k1;
    y = rot(y + v.second + f64(s + 48), 42) * k1;
    x ^= w.second; y ^= v.first;
    z = rot(z ^ w.first, 33);
    v = weak(s, v.second * k1, x + w.first);
    w = weak(s + 32, z + w.second, y);
    const uint64_t t = z; z = x; x = t;
    s += 64; len -= 64;
  } while (len != 0);
  return h16(h16(v.first, w.first) + smix(y) * k1 + z, h16(v.second, w.second) + x);
}
}  // namespace city

// integers of different signedness compare by value (ClickHouse's accurate comparison): only UInt64 does not fit the int64 slot
__device__ __forceinline__ int sql_order(int64_t a, bool au, int64_t b, bool bu) {
  if (au == bu) return au ? ((uint64_t)a < (uint64_t)b ? -1 : (uint64_t)a > (uint64_t)b ? 1 : 0) : (a < b ? -1 : a > b ? 1 : 0);
  if (au) return a < 0 ? 1 : (a < b ? -1 : a > b ? 1 : 0);   // a >= 2^63 is above every signed value
  return b < 0 ? -1 : (a < b ? -1 : a > b ? 1 : 0);
}
__device__ __forceinline__ bool sql_cmp_holds(int c, int op) {
  switch (op) { case 0: return c == 0; case 1: return c != 0; case 2: return c < 0; case 3: return c <= 0; case 4: return c > 0; default: return c >= 0; }
}
__global__ void __launch_bounds__(256) sql_expr_kernel(SqlProg p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.n) return;
  int64_t st[SQL_STACK];
  int sp = 0;
  for (int i = 0; i < p.nins; i++) {
    const SqlIns in = p.ins[i];
    switch (in.op) {
      case BI_COL: {
        const SqlLeaf l = p.leaves[in.a];
        int64_t v;
        switch (l.repr) {
          case TFGPU_R_INT8: v = ((const int8_t *)l.values)[r]; break; case TFGPU_R_INT16: v = ((const int16_t *)l.values)[r]; break;
          case TFGPU_R_INT32: v = ((const int32_t *)l.values)[r]; break; case TFGPU_R_INT64: v = ((const int64_t *)l.values)[r]; break;
          case TFGPU_R_UINT8: case TFGPU_R_BOOL: v = ((const uint8_t *)l.values)[r]; break; case TFGPU_R_UINT16: v = ((const uint16_t *)l.values)[r]; break;
          case TFGPU_R_UINT32: v = ((const uint32_t *)l.values)[r]; break; default: v = (int64_t)((const uint64_t *)l.values)[r];
        }
        st[sp++] = v;
        break;
      }
      case BI_IMM: st[sp++] = in.imm; break;
      case BI_ADD: sp--; st[sp - 1] = sql_wrap((int64_t)((uint64_t)st[sp - 1] + (uint64_t)st[sp]), in.ty); break;
      case BI_SUB: sp--; st[sp - 1] = sql_wrap((int64_t)((uint64_t)st[sp - 1] - (uint64_t)st[sp]), in.ty); break;
      case BI_MUL: sp--; st[sp - 1] = sql_wrap((int64_t)((uint64_t)st[sp - 1] * (uint64_t)st[sp]), in.ty); break;
      case BI_NEG: st[sp - 1] = sql_wrap((int64_t)(0 - (uint64_t)st[sp - 1]), in.ty); break;
      case BI_WRAP: st[sp - 1] = sql_wrap(st[sp - 1], in.ty); break;
      case BI_CMP: sp--; st[sp - 1] = sql_cmp_holds(sql_order(st[sp - 1], (in.b & 1) != 0, st[sp], (in.b & 2) != 0), in.a) ? 1 : 0; break;
      case BI_AND: sp--; st[sp - 1] = (st[sp - 1] != 0 && st[sp] != 0) ? 1 : 0; break;
      case BI_OR: sp--; st[sp - 1] = (st[sp - 1] != 0 || st[sp] != 0) ? 1 : 0; break;
      case BI_NOT: st[sp - 1] = st[sp - 1] == 0 ? 1 : 0; break;
      case BI_LEN: { const SqlLeaf l = p.leaves[in.a]; st[sp++] = (int64_t)(l.offsets[r + 1] - l.offsets[r]); break; }
      case BI_CITY: { const SqlLeaf l = p.leaves[in.a]; const uint32_t a = l.offsets[r]; st[sp++] = (int64_t)city::hash64(l.data + a, l.offsets[r + 1] - a); break; }
      case BI_SCMP: {
        const SqlLeaf l = p.leaves[in.a];
        const uint32_t a = l.offsets[r], la = l.offsets[r + 1] - a, b = p.soff[in.b], lb = p.soff[in.b + 1] - b;
        st[sp++] = sql_cmp_holds(bytes_compare(l.data + a, la, p.sdata + b, lb), (int)in.imm) ? 1 : 0;
        break;
      }
      case BI_IN: {
        const int64_t x = st[sp - 1];
        const bool xu = (in.imm & 1) != 0;
        bool hit = false;
        for (int k = 0; k < in.b; k++) hit |= sql_order(x, xu, p.ints[in.a + k], false) == 0;
        st[sp - 1] = (hit != ((in.imm & 2) != 0)) ? 1 : 0;
        break;
      }
      case BI_SIN: {
        const SqlLeaf l = p.leaves[in.a];
        const uint32_t a = l.offsets[r], la = l.offsets[r + 1] - a;
        const int cnt = (int)(in.imm & 0xFFFFFFFFll);
        bool hit = false;
        for (int k = 0; k < cnt; k++) { const uint32_t b = p.soff[in.b + k], lb = p.soff[in.b + k + 1] - b; hit |= la == lb && bytes_compare(l.data + a, la, p.sdata + b, lb) == 0; }
        st[sp++] = (hit != ((in.imm >> 32) != 0)) ? 1 : 0;
        break;
      }
    }
  }
  const int64_t v = sp > 0 ? st[sp - 1] : 0;
  if (p.keep) { p.keep[r] = v != 0 ? 1u : 0u; return; }
  switch (p.out_ty) {
    case SQL_I8: case SQL_U8: ((uint8_t *)p.out)[r] = (uint8_t)v; break;
    case SQL_I16: case SQL_U16: ((uint16_t *)p.out)[r] = (uint16_t)v; break;
    case SQL_I32: case SQL_U32: ((uint32_t *)p.out)[r] = (uint32_t)v; break;
    default: ((uint64_t *)p.out)[r] = (uint64_t)v;
  }
}
__global__ void __launch_bounds__(256) sql_case_kernel(const uint8_t *in, uint8_t *out, uint64_t n, int upper) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t c = in[i];
  out[i] = upper ? ((c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c) : ((c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c);  // ASCII only: lowerUTF8 / upperUTF8 are other functions
}
static int sql_repr_of(int ch);
// the trees of one query over one batch: text nodes become columns, integer trees become programs
struct SqlEval {
  const tfgpu_plan &p; const std::vector<int> &ty; const tfgpu_dbatch &b; hipStream_t st;
  std::vector<DColumn> keep_alive;
  std::vector<SqlLeaf> leaves; std::vector<SqlIns> ins; std::vector<int64_t> ints; std::vector<uint32_t> soff{0}; std::string sdata;
  int depth = 0, max_depth = 0;
  const DColumn &column(const std::string &name) const {
    for (auto &c : b.cols) if (c.name == name) return c;
    throw Error(TFGPU_ERR_UNSUPPORTED, "sql: column " + name + " is in the TableSchema and not among the batch's ColumnNames");
  }
  DColumn text(int i) {
    const SqlNode &n = p.sql_nodes[(size_t)i];
    const int64_t rows = b.nrows;
    switch (n.op) {
      case SN_COL: {
        const DColumn &c = column(n.s);
        if (!(c.repr == TFGPU_R_STRING || c.repr == TFGPU_R_BYTES)) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: a text function over column " + n.s + ", which is not held as plain text (JSON values reach ClickHouse re-marshalled): host step");
        return c;
      }
      case SN_STR: {
        DColumn c; c.repr = TFGPU_R_STRING;
        const uint64_t total = (uint64_t)n.s.size() * (uint64_t)rows;
        if (total >> 32) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: a constant text column of more than 4 GiB");
        Buf t = upload_small(n.s.data(), n.s.size());
        c.offsets = dalloc((size_t)(rows + 1) * 4 + 16); c.data = dalloc(total + 8); c.data_len = total;
        sql_const_text_kernel<<<grid_for(rows + 1, 256), 256, 0, st>>>(ptr<uint32_t>(c.offsets), ptr<uint8_t>(c.data), rows, ptr<uint8_t>(t), (uint32_t)n.s.size());
        return c;
      }
      case SN_LOWER: case SN_UPPER: {
        DColumn src = text(n.kids[0]);
        DColumn c; c.repr = TFGPU_R_STRING; c.offsets = src.offsets; c.data_len = src.data_len;
        c.data = dalloc((size_t)src.data_len + 8);
        if (src.data_len) sql_case_kernel<<<grid_for((int64_t)src.data_len, 256), 256, 0, st>>>(ptr<uint8_t>(src.payload()), ptr<uint8_t>(c.data), src.data_len, n.op == SN_UPPER ? 1 : 0);
        keep_alive.push_back(src);
        return c;
      }
      case SN_TOSTR: {
        if (ty[(size_t)n.kids[0]] == SQL_STRING) return text(n.kids[0]);
        SqlEval sub{p, ty, b, st};  // (its own program: this one may be half emitted — toString inside length() inside an expression)
        DColumn v = sub.integer(n.kids[0]);
        for (auto &k : sub.keep_alive) keep_alive.push_back(k);
        return column_to_string(v, rows, false, 24);
      }
      default: throw Error(TFGPU_ERR_INVALID, "sql: internal text node");
    }
  }
  int text_leaf(int i) {
    DColumn c = text(i);
    SqlLeaf l{}; l.offsets = ptr<uint32_t>(c.offsets); l.data = ptr<uint8_t>(c.payload()); l.repr = c.repr;
    keep_alive.push_back(std::move(c));
    leaves.push_back(l);
    return (int)leaves.size() - 1;
  }
  int literal(const std::string &s) { sdata += s; soff.push_back((uint32_t)sdata.size()); return (int)soff.size() - 2; }
  void push() { if (++depth > max_depth) max_depth = depth; }
  void emit(int i) {
    const SqlNode &n = p.sql_nodes[(size_t)i];
    const int t = ty[(size_t)i];
    auto u64 = [&](int k) { return ty[(size_t)n.kids[(size_t)k]] == SQL_U64; };
    switch (n.op) {
      case SN_COL: {
        const DColumn &c = column(n.s);
        if (repr_is_var(c.repr) || c.repr == TFGPU_R_TIME || c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64 || !c.values) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: column " + n.s + " is not held as an integer");
        SqlLeaf l{}; l.values = c.values->p; l.repr = c.repr;
        leaves.push_back(l);
        ins.push_back(SqlIns{BI_COL, t, (int32_t)leaves.size() - 1, 0, 0}); push();
        break;
      }
      case SN_INT: ins.push_back(SqlIns{BI_IMM, t, 0, 0, n.ival}); push(); break;
      case SN_ADD: case SN_SUB: case SN_MUL: emit(n.kids[0]); emit(n.kids[1]); ins.push_back(SqlIns{n.op == SN_ADD ? BI_ADD : n.op == SN_SUB ? BI_SUB : BI_MUL, t, 0, 0, 0}); depth--; break;
      case SN_NEG: emit(n.kids[0]); ins.push_back(SqlIns{BI_NEG, t, 0, 0, 0}); break;
      case SN_CAST: emit(n.kids[0]); ins.push_back(SqlIns{BI_WRAP, t, 0, 0, 0}); break;
      case SN_LEN: ins.push_back(SqlIns{BI_LEN, t, text_leaf(n.kids[0]), 0, 0}); push(); break;
      case SN_CITY64: ins.push_back(SqlIns{BI_CITY, t, text_leaf(n.kids[0]), 0, 0}); push(); break;
      case SN_EQ: case SN_NE: case SN_LT: case SN_LE: case SN_GT: case SN_GE: {
        int op = n.op - SN_EQ;
        if (ty[(size_t)n.kids[0]] == SQL_STRING) {
          const bool lit_left = p.sql_nodes[(size_t)n.kids[0]].op == SN_STR;
          if (lit_left) { static const int mirror[6] = {0, 1, 4, 5, 2, 3}; op = mirror[op]; }  // 'a' < x  is  x > 'a'
          const int leaf = text_leaf(n.kids[lit_left ? 1 : 0]);
          ins.push_back(SqlIns{BI_SCMP, t, leaf, literal(p.sql_nodes[(size_t)n.kids[lit_left ? 0 : 1]].s), op}); push();
        } else {
          emit(n.kids[0]); emit(n.kids[1]);
          ins.push_back(SqlIns{BI_CMP, t, op, (u64(0) ? 1 : 0) | (u64(1) ? 2 : 0), 0}); depth--;
        }
        break;
      }
      case SN_AND: case SN_OR: emit(n.kids[0]); emit(n.kids[1]); ins.push_back(SqlIns{n.op == SN_AND ? BI_AND : BI_OR, t, 0, 0, 0}); depth--; break;
      case SN_NOT: emit(n.kids[0]); ins.push_back(SqlIns{BI_NOT, t, 0, 0, 0}); break;
      case SN_IN: case SN_NOTIN: {
        const bool neg = n.op == SN_NOTIN;
        if (ty[(size_t)n.kids[0]] == SQL_STRING) {
          const int leaf = text_leaf(n.kids[0]);
          const int first = (int)soff.size() - 1;
          for (size_t k = 1; k < n.kids.size(); k++) literal(p.sql_nodes[(size_t)n.kids[k]].s);
          ins.push_back(SqlIns{BI_SIN, t, leaf, first, (int64_t)(n.kids.size() - 1) | ((int64_t)(neg ? 1 : 0) << 32)}); push();
        } else {
          emit(n.kids[0]);
          const int first = (int)ints.size();
          for (size_t k = 1; k < n.kids.size(); k++) ints.push_back(p.sql_nodes[(size_t)n.kids[k]].ival);
          ins.push_back(SqlIns{BI_IN, t, first, (int32_t)n.kids.size() - 1, (int64_t)((u64(0) ? 1 : 0) | (neg ? 2 : 0))});
        }
        break;
      }
      default: throw Error(TFGPU_ERR_INVALID, "sql: internal integer node");
    }
  }
  void run(int root, void *out, int out_ty, uint32_t *keep) {
    leaves.clear(); ins.clear(); ints.clear(); soff.assign(1, 0u); sdata.clear(); depth = max_depth = 0;
    emit(root);
    if (max_depth > SQL_STACK) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: an expression nests deeper than the device program's sixteen slots");
    Buf bi = upload_small(ins.data(), ins.size() * sizeof(SqlIns)), bl = upload_small(leaves.data(), leaves.size() * sizeof(SqlLeaf));
    Buf bn = upload_small(ints.data(), ints.size() * 8), bo = upload_small(soff.data(), soff.size() * 4), bd = upload_small(sdata.data(), sdata.size());
    SqlProg g{};
    g.ins = reinterpret_cast<const SqlIns *>(bi->p); g.nins = (int32_t)ins.size(); g.leaves = reinterpret_cast<const SqlLeaf *>(bl->p);
    g.ints = ptr<int64_t>(bn); g.soff = ptr<uint32_t>(bo); g.sdata = ptr<uint8_t>(bd); g.n = b.nrows; g.out = out; g.out_ty = out_ty; g.keep = keep;
    KernelTimer t("sql_expr");
    if (b.nrows) sql_expr_kernel<<<grid_for(b.nrows, 256), 256, 0, st>>>(g);
  }
  DColumn integer(int root) {
    DColumn c;
    c.repr = sql_repr_of(ty[(size_t)root]);
    c.values = dalloc((size_t)std::max<int64_t>(b.nrows, 1) * repr_width(c.repr));
    run(root, c.values->p, ty[(size_t)root], nullptr);
    return c;
  }
};

static std::unique_ptr<tfgpu_dbatch> apply_sql(const tfgpu_plan &p, const tfgpu_dbatch &in0, ApplyCtx &ax) {
  hipStream_t st = ctx().stream;
  // the input schema as ResultSchema sees it (clickhouse_local.go:351-421)
  std::vector<SchemaCol> sc;
  auto is_key = [&](const std::string &n) { for (auto &k : in0.key_names) if (k == n) return true; return false; };
  if (!in0.schema.empty()) for (auto &c : in0.schema) sc.push_back(SchemaCol{c.first, c.second, is_key(c.first) ? (uint32_t)TFGPU_COL_KEY : 0u});
  else for (auto &c : in0.cols) sc.push_back(SchemaCol{c.name, c.dtype, is_key(c.name) ? (uint32_t)TFGPU_COL_KEY : 0u});
  const std::vector<SqlOut> outs = sql_resolve(p, sc);
  bool has_key = false;
  for (auto &o : outs) has_key |= o.key;
  if (!has_key) throw Error(TFGPU_ERR_CONFIG, "sql: result table has no primary key");  // ResultSchema :417-419
  for (auto &k : in0.key_names) {  // see the header of this section
    bool through = false;
    for (auto &o : outs) through |= o.kind == SQL_COLUMN && o.name == k && o.src >= 0 && sc[(size_t)o.src].name == k;
    if (!through) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: the primary key column " + k + " must pass through the select list unchanged (the reference re-attaches row meta by key)");
  }
  for (auto &c : sc) if (c.dtype == TFGPU_T_FLOAT32 || c.dtype == TFGPU_T_INTERVAL) {
    for (auto &o : outs) if (o.src >= 0 && sc[(size_t)o.src].name == c.name) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: column " + c.name + " (" + type_name(c.dtype) + ") reaches ClickHouse as text of another type: host step");
  }
  if (in0.nrows == 0) { auto e = shallow_copy(in0); e->cols.clear(); return e; }
  // SplitUpdatedPKeys (utils.go:75-128): an Update that moves its primary key cuts the batch into sub-batches with a
  // Delete + Insert pair of their own; such batches stay on the host
  std::vector<uint8_t> kflags((size_t)in0.nrows);
  if (in0.kind && !in0.old_keys.empty() && keys_changed_rows(in0, kflags.data()) > 0)
    throw Error(TFGPU_ERR_UNSUPPORTED, "sql: the batch holds Updates that change their primary key (SplitUpdatedPKeys sub-batches): host step");
  std::unique_ptr<tfgpu_dbatch> col = collapse_rows(in0);  // abstract.Collapse (last write per key wins)
  const tfgpu_dbatch &in = *col;
  const int64_t n0 = in.nrows;
  {
    std::vector<const DColumn *> need;
    for (auto &c : in.cols) need.push_back(&c);
    materialize(in);
  }
  // nil → the ClickHouse default (MarshalCItoJSON omits nil columns, JSONEachRow fills the type's default): 0 / ''
  auto w = shallow_copy(in);
  {
    std::vector<SqlNilCol> nil;  // every nullable fixed-width array in ONE launch (an SR / JSON batch has ~100 of them: a copy + a kernel each before)
    for (auto &c : w->cols) {
      if (!c.validity) continue;
      if (!repr_is_var(c.repr)) {
        const size_t width = repr_width(c.repr);
        Buf v = dalloc((size_t)std::max<int64_t>(n0, 1) * width);
        nil.push_back(SqlNilCol{(const uint8_t *)c.values->p, ptr<uint8_t>(v), ptr<uint8_t>(c.validity), (int32_t)width});
        c.values = v;
        if (c.nanos) { Buf nn = dalloc((size_t)std::max<int64_t>(n0, 1) * 4); nil.push_back(SqlNilCol{(const uint8_t *)c.nanos->p, ptr<uint8_t>(nn), ptr<uint8_t>(c.validity), 4}); c.nanos = nn; }
      }
      c.validity = nullptr;  // (a nil text cell has no bytes: it is '' already)
    }
    if (!nil.empty() && n0) {
      Buf bn = upload_small(nil.data(), nil.size() * sizeof(SqlNilCol));
      KernelTimer t("sql_default_nils");
      sql_default_nils_kernel<<<dim3(grid_for(n0, 256), (unsigned)nil.size()), 256, 0, st>>>(reinterpret_cast<const SqlNilCol *>(bn->p), (int32_t)nil.size(), n0);
    }
  }
  // WHERE on every row event
  std::vector<int> node_ty;
  if (!p.sql_nodes.empty()) node_ty = sql_node_types(p, sc);
  std::unique_ptr<tfgpu_dbatch> kept;
  if (sql_where_as_tree(p, sc)) {  // a WHERE that is a tree (or names a UInt64 column): one program, rows whose value is not zero stay
    Buf keep = dalloc((size_t)(n0 + 1) * 4);
    SqlEval ev{p, node_ty, *w, st};
    ev.run(p.sql_where_tree, nullptr, SQL_U8, ptr<uint32_t>(keep));
    kept = compact(*w, keep);  // syncs
  } else kept = p.sql_has_where ? run_filter(p.exprs, true, false, *w, ax) : std::move(w);
  dense_locked(*kept);  // (the select list reads the kept rows' columns right away)
  const int64_t n = kept->nrows;
  // the select list
  auto out = std::make_unique<tfgpu_dbatch>();
  out->nrows = n; out->ns = kept->ns; out->table = kept->table; out->kind = kept->kind; out->src_row = kept->src_row; out->part_id = kept->part_id;
  if (!out->src_row && p.sql_has_where) out->src_row = kept->src_row;
  for (const SqlOut &o : outs) {
    DColumn c;
    const DColumn *src = nullptr;
    if (o.src >= 0) {
      for (auto &kc : kept->cols) if (kc.name == sc[(size_t)o.src].name) { src = &kc; break; }
      if (!src) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: column " + sc[(size_t)o.src].name + " is in the TableSchema and not among the batch's ColumnNames");
    }
    switch (o.kind) {
      case SQL_COLUMN:
        c = *src;
        if (o.ch == SQL_STRING) { c.repr = TFGPU_R_STRING; }                           // Restore keeps a Go string under DataType "string"
        else if (src->repr == TFGPU_R_BOOL) { c.repr = TFGPU_R_UINT8; }                // boolean → UInt8
        break;
      case SQL_CONST_INT: case SQL_INT_EXPR: {
        c.repr = sql_repr_of(o.ch);
        const size_t width = repr_width(c.repr);
        c.values = dalloc((size_t)std::max<int64_t>(n, 1) * width);
        SqlIntParams ip{};
        ip.src = src ? src->values->p : nullptr; ip.src_repr = src ? src->repr : TFGPU_R_INT64; ip.validity = nullptr; ip.n = n; ip.out = c.values->p; ip.out_ty = o.ch;
        if (o.kind == SQL_CONST_INT) { ip.nops = 1; ip.addend[0] = o.ival; ip.ty[0] = o.ch; static const int64_t zero = 0; (void)zero; }
        else {
          if (o.ops.size() > 8) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: more than eight cast / arithmetic steps in one expression");
          ip.nops = (int32_t)o.ops.size();
          for (size_t k = 0; k < o.ops.size(); k++) { ip.addend[k] = o.ops[k].addend; ip.ty[k] = o.ops[k].ty; }
        }
        Buf zeros;
        if (!src) { zeros = dalloc_zero((size_t)std::max<int64_t>(n, 1) * 8); ip.src = zeros->p; }
        KernelTimer t("sql_int_expr");
        if (n) sql_int_kernel<<<grid_for(n, 256), 256, 0, st>>>(ip);
        break;
      }
      case SQL_EXPR: {
        SqlEval ev{p, node_ty, *kept, st};
        c = o.ch == SQL_STRING ? ev.text(o.root) : ev.integer(o.root);
        if (o.ch == SQL_STRING) c.repr = TFGPU_R_STRING;
        break;
      }
      case SQL_CONST_STR: {
        c.repr = TFGPU_R_STRING;
        const uint64_t total = (uint64_t)o.sval.size() * (uint64_t)n;
        if (total >> 32) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: a constant text column of more than 4 GiB");
        Buf text = upload_small(o.sval.data(), o.sval.size());
        c.offsets = dalloc((size_t)(n + 1) * 4 + 16); c.data = dalloc(total + 8); c.data_len = total;
        sql_const_text_kernel<<<grid_for(n + 1, 256), 256, 0, st>>>(ptr<uint32_t>(c.offsets), ptr<uint8_t>(c.data), n, ptr<uint8_t>(text), (uint32_t)o.sval.size());
        break;
      }
      case SQL_TO_STRING:
        if (repr_is_var(src->repr)) { c = *src; c.repr = TFGPU_R_STRING; }
        else c = column_to_string(*src, n, false, 24);
        break;
      case SQL_TO_DATETIME:
        if (src->repr == TFGPU_R_TIME) { c = *src; break; }
        c.repr = TFGPU_R_TIME;
        c.values = dalloc_zero((size_t)std::max<int64_t>(n, 1) * 8);
        if (n && src->repr == TFGPU_R_INT32) todatetime_kernel<int32_t><<<grid_for(n, 256), 256, 0, st>>>(ptr<int32_t>(src->values), n, ptr<int64_t>(c.values));
        else if (n && src->repr == TFGPU_R_UINT32) todatetime_kernel<uint32_t><<<grid_for(n, 256), 256, 0, st>>>(ptr<uint32_t>(src->values), n, ptr<int64_t>(c.values));
        else if (n && (src->repr == TFGPU_R_INT64 || src->repr == TFGPU_R_UINT64)) todatetime_kernel<int64_t><<<grid_for(n, 256), 256, 0, st>>>(ptr<int64_t>(src->values), n, ptr<int64_t>(c.values));
        else if (n) throw Error(TFGPU_ERR_UNSUPPORTED, "sql: toDateTime() of a value that is not held as a 32 / 64-bit integer");
        break;
      default: throw Error(TFGPU_ERR_INVALID, "sql: internal output kind");
    }
    c.name = o.name; c.dtype = o.yt; c.validity = nullptr;
    out->cols.push_back(std::move(c));
    out->schema.emplace_back(o.name, o.yt);
    if (o.key) out->key_names.push_back(o.name);
  }
  // Update / Delete: OldKeys = the result row (names and values); Delete: no column values (clickhouse_local.go:277-285)
  if (out->kind && n) {
    std::vector<uint8_t> hk((size_t)n);
    d2h(hk.data(), out->kind->p, hk.size());
    sync();
    bool any_old = false, any_del = false;
    for (uint8_t k : hk) { any_old |= k == TFGPU_K_UPDATE || k == TFGPU_K_DELETE; any_del |= k == TFGPU_K_DELETE; }
    if (any_old) {
      Buf has_old = dalloc((size_t)(n + 7) / 8 + 8), has_cols = dalloc((size_t)(n + 7) / 8 + 8);
      sql_kind_bitmaps_kernel<<<grid_for((n + 7) / 8, 256), 256, 0, st>>>(ptr<uint8_t>(out->kind), n, ptr<uint8_t>(has_old), ptr<uint8_t>(has_cols));
      out->old_keys = out->cols;  // buffers shared
      out->old_present = has_old;
      if (any_del) for (auto &c : out->cols) c.validity = has_cols;
    }
  }
  return out;
}

// ============================================================================
// hash-partition, local half (config 5): rows grouped by PartID, original order kept inside a part
// ============================================================================
__global__ void part_keep_kernel(const uint32_t *part_id, int64_t n, uint32_t d, uint32_t *keep) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) keep[r] = part_id[r] == d ? 1u : 0u;
}
__global__ void part_scatter_kernel(const uint32_t *keep_scan, int64_t n, uint32_t base, int32_t *sel) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t a = keep_scan[i], b = keep_scan[i + 1];
  if (a != b) sel[base + a] = (int32_t)i;
}
std::unique_ptr<tfgpu_dbatch> partition_rows(const tfgpu_dbatch &in, int nparts, int64_t *counts) {
  if (!in.part_id) throw Error(TFGPU_ERR_INVALID, "tfgpu_partition: the batch has no part_id (apply sharder_transformer first)");
  const int64_t n = in.nrows;
  hipStream_t st = ctx().stream;
  Buf sel = dalloc((size_t)n * 4 + 4), keep = dalloc((size_t)(n + 1) * 4);
  uint32_t base = 0;
  KernelTimer t("partition_rows");
  for (int d = 0; d < nparts; d++) {
    if (n) part_keep_kernel<<<grid_for(n, 256), 256, 0, st>>>(ptr<uint32_t>(in.part_id), n, (uint32_t)d, ptr<uint32_t>(keep));
    exclusive_scan_u32(ptr<uint32_t>(keep), ptr<uint32_t>(keep), n, true);
    const uint32_t *h = d2h_u32(ptr<uint32_t>(keep) + n);
    sync();
    const uint32_t m = *h;
    if (n && m) part_scatter_kernel<<<grid_for(n, 256), 256, 0, st>>>(ptr<uint32_t>(keep), n, base, ptr<int32_t>(sel));
    counts[d] = m;
    base += m;
  }
  if ((int64_t)base != n) throw Error(TFGPU_ERR_INVALID, "tfgpu_partition: part_id outside [0, nparts)");
  return gather_batch(in, sel, n);
}

std::unique_ptr<tfgpu_dbatch> apply_plan(const tfgpu_plan &p, const tfgpu_dbatch &in, ApplyCtx &ax) {
  // transformers compute on values: an ABSENT cell is not a nil (the stock path takes the batch) — but for the sharder, which reads its key
  // columns through AsMap()[name] (sharder.go:134-143: a name the item does not list IS nil there) and passes every column on untouched
  // Since round 6 the transformers that walk an item's own ColumnNames do so here too: mask_field / convert_to_string / convert_to_datetime leave a
  // cell the row does not list as it is (hmac_hasher.go:56-63), filter_rows fails such a row ("Unable to find column", filter_rows.go:147-154),
  // the column droppers and the row filters carry the bitmaps.  `sql` (it serializes whole rows for clickhouse-local) and batches whose rows carry
  // their own name ORDER (col_order indexes the column list these transformers change) stay with the stock path.
  if (p.kind == PK_SQL || (in.col_order && p.kind != PK_SHARDER)) refuse_absent(in);
  if (in.pending && (p.kind != PK_MASK || has_absent(in))) dense_locked(in);
  else if (!in.pending) { std::lock_guard<std::mutex> dl(g_dense_mu); wait_dense_nolock(in); }  // (the callers hold the lane's mutex) only mask_field reads through a selection
  switch (p.kind) {
    case PK_MASK: return apply_mask(p, in);
    case PK_RENAME: return apply_rename(p, in);
    case PK_FILTER_COLUMNS: return apply_filter_columns(p, in);
    case PK_SKIP_EVENTS: return apply_skip_events(p, in);
    case PK_FILTER_ROWS: return apply_filter_rows(p, in, ax);
    case PK_TO_STRING: return apply_to_string(p, in);
    case PK_TO_DATETIME: return apply_to_datetime(p, in);
    case PK_SHARDER: return apply_sharder(p, in, nullptr);
    case PK_REPLACE_PK: return apply_replace_pk(p, in);
    case PK_SQL: return apply_sql(p, in, ax);
  }
  throw Error(TFGPU_ERR_INVALID, "unknown plan kind");
}

}  // namespace tf
