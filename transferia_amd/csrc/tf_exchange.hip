// tf_exchange.hip — the one real exchange of the path behind the C ABI: the all-to-all half of the hash partition
// (BASELINE.json configs[4]: sharder_transformer -> tfgpu_partition -> THIS -> Collapse -> queue serializer), one
// process per GPU, RCCL point-to-point over xGMI.
//
// The reference has no such step inside one process: a sharded transfer gives every worker its own source partitions
// (pkg/abstract/coordinator sharding) and sharder_transformer only stamps PartID = crc32 % shards on each item
// (sharder.go:130-145) for the sink to route by.  Regrouping rows so that one key lands on one GPU is what the
// data-parallel version of "one PartID -> one sink shard" needs, and it is the only data-path collective in the library.
//
// Shape of one exchange (all ranks call it together, rows already grouped by destination by tfgpu_partition):
//   1. a header all-gather (column counts) and a descriptor all-gather (reprs, optional-buffer flags, rows and var-width
//      bytes per destination) — two small collectives, one host sync: every rank sees the same matrix, so a mismatch
//      (different schemas) fails on ALL ranks before any payload moves and nobody is left waiting;
//   2. ONE grouped RCCL call with a send / recv per (column buffer, peer): fixed-width values move in place, text moves
//      as lengths + bytes, bitmaps as one byte per row (a destination's slice of a bitmap is not byte aligned).  No
//      staging copy: xGMI is the bound (7 links x ~153 GB/s per GPU), HBM only sees each byte once per side;
//   3. offsets rebuilt from the received lengths (device scan), bitmaps repacked, part_id = this rank.
// Everything is enqueued on the calling lane's stream, so it orders with the partition kernels before it and the
// collapse after it without a device sync.
//
// RCCL is bound at run time (dlopen of librccl.so.1, or $TFGPU_RCCL_LIB): single-GPU users never load it, and the
// lock-step CPU emulator substitutes a socket-backed stand-in for the world_size-2 tests.
#include "tf_common.hpp"

#include <dlfcn.h>

#include <cstring>

using namespace tf;

namespace {

// ---- the RCCL entry points this file binds (rccl.h: ncclGetUniqueId … ncclGroupEnd).  The types are restated here
// instead of including <rccl/rccl.h> so the emulator build needs no ROCm headers; they are RCCL's public ABI.
struct NcclUniqueId { char internal[128]; };
typedef struct ncclComm *NcclComm;
enum { NCCL_UINT8 = 1 };  // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(NcclUniqueId *) = nullptr;
  int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};

Rccl &rccl() {
  static Rccl r = [] {
    Rccl x;
    const char *env = std::getenv("TFGPU_RCCL_LIB");
    const char *names[] = {env, "librccl.so.1", "librccl.so"};
    std::string tried;
    for (const char *n : names) {
      if (!n || !*n) continue;
      x.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (x.h) break;
      tried += std::string(tried.empty() ? "" : "; ") + dlerror();
    }
    if (!x.h) throw Error(TFGPU_ERR_DEVICE, "RCCL is not loadable (" + tried + ")");
    auto sym = [&](const char *s) {
      void *p = dlsym(x.h, s);
      if (!p) throw Error(TFGPU_ERR_DEVICE, std::string("RCCL symbol missing: ") + s);
      return p;
    };
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
    x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
    x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    return x;
  }();
  return r;
}

void nccl_check(int rc, const char *what) {
  if (rc != 0) throw Error(TFGPU_ERR_DEVICE, std::string(what) + ": " + rccl().GetErrorString(rc));
}
#define TF_NCCL(expr) nccl_check((expr), #expr)

inline unsigned grid_for(int64_t n, int threads) { return (unsigned)((n + threads - 1) / threads); }

// ---- small device helpers ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) xc_lengths(const uint32_t *off, int64_t n, uint32_t *len) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) len[i] = off[i + 1] - off[i];
}
// bytes[d] = off[bound[d+1]] - off[bound[d]]: the payload one destination receives from a text column
__global__ void xc_dest_bytes(const uint32_t *off, const int64_t *bound, int world, int64_t *bytes) {
  int d = threadIdx.x;
  if (d < world) bytes[d] = (int64_t)off[bound[d + 1]] - (int64_t)off[bound[d]];
}
__global__ void __launch_bounds__(256) xc_unpack_bits(const uint8_t *bits, int64_t n, uint8_t *bytes) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) bytes[i] = (bits[i >> 3] >> (i & 7)) & 1u;
}
// one thread per output byte of the bitmap; the tail byte's unused bits stay 0
__global__ void __launch_bounds__(256) xc_pack_bits(const uint8_t *bytes, int64_t n, uint8_t *bits) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b * 8 >= n) return;
  unsigned v = 0;
  for (int k = 0; k < 8 && b * 8 + k < n; k++) v |= (unsigned)(bytes[b * 8 + k] & 1u) << k;
  bits[b] = (uint8_t)v;
}
__global__ void __launch_bounds__(256) xc_fill_u8(uint8_t *p, int64_t n, uint8_t v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void __launch_bounds__(256) xc_fill_u32(uint32_t *p, int64_t n, uint32_t v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void __launch_bounds__(256) xc_iota_i32(int32_t *p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int32_t)i;
}

// descriptor flags
enum : int64_t { CF_VALIDITY = 1, CF_NANOS = 2, CF_ABSENT = 4 /* some row does not list the column (DColumn::absent) */ };
enum : int64_t { BF_KIND = 1, BF_SRC_ROW = 2, BF_OLD_PRESENT = 4 };

}  // namespace

struct tfgpu_comm {
  NcclComm comm = nullptr;
  int rank = 0, world = 1;
  std::mutex mu;
};

namespace {

// The part of every buffer that stays on this rank (at world 1: all of it) does not go through RCCL — a send / recv pair to oneself is
// one channel's copy loop, 0.20 ms for configs[4]'s ~20 buffers — but through ONE launch that moves all of them: a workgroup per
// 64 KiB piece, 16-byte words where source and destination allow, 4-byte words or bytes where they do not.
struct LocalPiece { const uint8_t *src; uint8_t *dst; uint32_t bytes; uint32_t pad; };
constexpr uint32_t XC_PIECE = 64u << 10;
__global__ void __launch_bounds__(256) xc_copy_local(const LocalPiece *pieces) {
  const LocalPiece pc = pieces[blockIdx.x];
  const uint32_t t = threadIdx.x;
  if ((((uintptr_t)pc.src | (uintptr_t)pc.dst) & 15u) == 0) {
    const uint32_t nq = pc.bytes >> 4;
    const uint4 *s = reinterpret_cast<const uint4 *>(pc.src); uint4 *d = reinterpret_cast<uint4 *>(pc.dst);
    for (uint32_t i = t; i < nq; i += 256) d[i] = s[i];
    for (uint32_t i = (nq << 4) + t; i < pc.bytes; i += 256) pc.dst[i] = pc.src[i];
  } else if ((((uintptr_t)pc.src | (uintptr_t)pc.dst) & 3u) == 0) {
    const uint32_t nw = pc.bytes >> 2;
    const uint32_t *s = reinterpret_cast<const uint32_t *>(pc.src); uint32_t *d = reinterpret_cast<uint32_t *>(pc.dst);
    for (uint32_t i = t; i < nw; i += 256) d[i] = s[i];
    for (uint32_t i = (nw << 2) + t; i < pc.bytes; i += 256) pc.dst[i] = pc.src[i];
  } else for (uint32_t i = t; i < pc.bytes; i += 256) pc.dst[i] = pc.src[i];
}

// One buffer of the exchange: `elem` bytes per row (or per byte for text payloads), split[d] units to destination d,
// got[s] units from source s.
struct Move {
  const uint8_t *send;
  uint8_t *recv;
  size_t elem;
  std::vector<int64_t> split, got;
};

std::unique_ptr<tfgpu_dbatch> exchange(tfgpu_comm &cm, const tfgpu_dbatch &in, const int64_t *counts, int64_t *recv_counts) {
  Rccl &R = rccl();
  Context &cx = ctx();
  hipStream_t st = cx.stream;
  const int W = cm.world, me = cm.rank;
  const int64_t n_in = in.nrows;
  {
    int64_t s = 0;
    for (int d = 0; d < W; d++) { if (counts[d] < 0) throw Error(TFGPU_ERR_INVALID, "tfgpu_exchange: negative count"); s += counts[d]; }
    if (s != n_in) throw Error(TFGPU_ERR_INVALID, "tfgpu_exchange: counts do not add up to the batch's rows (group it with tfgpu_partition first)");
  }
  materialize(in);  // text still aliasing its CSV chunk is packed now: a peer cannot read this rank's source text
  KernelTimer timer("exchange");

  // ---- 1a. header: [ncols, n_old_keys, n_key_names] of every rank
  std::vector<int64_t> hdr_all((size_t)W * 3);
  {
    int64_t h[3] = {(int64_t)in.cols.size(), (int64_t)in.old_keys.size(), (int64_t)in.key_names.size()};
    Buf dh = upload_small(h, sizeof h), da = dalloc((size_t)W * sizeof h);
    TF_NCCL(R.AllGather(ptr<uint8_t>(dh), ptr<uint8_t>(da), sizeof h, NCCL_UINT8, cm.comm, st));
    d2h(hdr_all.data(), ptr<uint8_t>(da), hdr_all.size() * 8);
    sync();
  }
  const int64_t ncols = (int64_t)in.cols.size();
  int64_t nold = 0;
  bool cols_agree = true;
  for (int r = 0; r < W; r++) { cols_agree &= hdr_all[(size_t)r * 3] == ncols; nold = std::max(nold, hdr_all[(size_t)r * 3 + 1]); }
  bool old_agree = true;
  for (int r = 0; r < W; r++) old_agree &= hdr_all[(size_t)r * 3 + 1] == nold || hdr_all[(size_t)r * 3 + 1] == 0;
  // the failures below are derived from data every rank holds: all ranks throw together
  if (!cols_agree) throw Error(TFGPU_ERR_INVALID, "tfgpu_exchange: ranks hold batches with different column counts");
  if (!old_agree) throw Error(TFGPU_ERR_INVALID, "tfgpu_exchange: ranks hold OldKeys of different widths");
  const bool local_old = !in.old_keys.empty();
  const int64_t C = ncols + nold;

  // ---- 1b. descriptor row of this rank: [batch flags | per column: repr, dtype, flags | rows per dest | per column: bytes per dest]
  const size_t o_col = 1, o_rows = o_col + (size_t)C * 3, o_bytes = o_rows + (size_t)W, L = o_bytes + (size_t)C * (size_t)W;
  std::vector<int64_t> row(L, 0), bound((size_t)W + 1, 0);
  for (int d = 0; d < W; d++) bound[(size_t)d + 1] = bound[(size_t)d] + counts[d];
  auto col_at = [&](int64_t j) -> const DColumn * { return j < ncols ? &in.cols[(size_t)j] : (local_old ? &in.old_keys[(size_t)(j - ncols)] : nullptr); };
  row[0] = (in.kind ? BF_KIND : 0) | (in.src_row ? BF_SRC_ROW : 0) | ((local_old && in.old_present) ? BF_OLD_PRESENT : 0);
  for (int64_t j = 0; j < C; j++) {
    const DColumn *c = col_at(j);
    if (!c) { row[o_col + (size_t)j * 3] = -1; continue; }  // this rank has no OldKeys: it takes the others' description
    row[o_col + (size_t)j * 3] = c->repr;
    row[o_col + (size_t)j * 3 + 1] = c->dtype;
    row[o_col + (size_t)j * 3 + 2] = (c->validity ? CF_VALIDITY : 0) | (c->nanos ? CF_NANOS : 0) | (c->absent ? CF_ABSENT : 0);
  }
  for (int d = 0; d < W; d++) row[o_rows + (size_t)d] = counts[d];
  Buf drow = upload_small(row.data(), L * 8), dbound = upload_small(bound.data(), bound.size() * 8);
  for (int64_t j = 0; j < C; j++) {
    const DColumn *c = col_at(j);
    if (c && repr_is_var(c->repr)) xc_dest_bytes<<<1, 64 * ((W + 63) / 64), 0, st>>>(ptr<uint32_t>(c->offsets), ptr<int64_t>(dbound), W, ptr<int64_t>(drow) + o_bytes + (size_t)j * W);
  }
  std::vector<int64_t> mat((size_t)W * L);
  {
    Buf dall = dalloc(mat.size() * 8);
    TF_NCCL(R.AllGather(ptr<uint8_t>(drow), ptr<uint8_t>(dall), L * 8, NCCL_UINT8, cm.comm, st));
    d2h(mat.data(), ptr<uint8_t>(dall), mat.size() * 8);
    sync();
  }
  auto M = [&](int r, size_t k) -> int64_t { return mat[(size_t)r * L + k]; };

  // ---- the agreed description of every column, and what this rank receives
  struct Desc { int repr = TFGPU_R_INVALID, dtype = TFGPU_T_INVALID; bool validity = false, nanos = false, absent = false; };
  std::vector<Desc> desc((size_t)C);
  for (int64_t j = 0; j < C; j++) {
    Desc &d = desc[(size_t)j];
    for (int r = 0; r < W; r++) {
      int64_t rp = M(r, o_col + (size_t)j * 3);
      if (rp < 0) continue;
      if (d.repr == TFGPU_R_INVALID) { d.repr = (int)rp; d.dtype = (int)M(r, o_col + (size_t)j * 3 + 1); }
      else if (d.repr != (int)rp || d.dtype != (int)M(r, o_col + (size_t)j * 3 + 1))
        throw Error(TFGPU_ERR_INVALID, "tfgpu_exchange: ranks disagree on the representation / type of column " + std::to_string(j));
      d.validity |= (M(r, o_col + (size_t)j * 3 + 2) & CF_VALIDITY) != 0;
      d.nanos |= (M(r, o_col + (size_t)j * 3 + 2) & CF_NANOS) != 0;
      d.absent |= (M(r, o_col + (size_t)j * 3 + 2) & CF_ABSENT) != 0;
    }
  }
  // Every failure is evaluated for EVERY rank from what the two all-gathers brought (the same matrix on all ranks), so all
  // ranks throw together before any payload moves: a rank that failed alone would leave its peers inside ncclGroupEnd.
  for (int r = 0; r < W; r++)
    if (nold && hdr_all[(size_t)r * 3 + 1] == 0 && hdr_all[(size_t)r * 3 + 2] != nold)
      throw Error(TFGPU_ERR_INVALID, "tfgpu_exchange: some ranks carry OldKeys and rank " + std::to_string(r) + " cannot name them (no TableSchema keys on its batch)");
  for (int d = 0; d < W; d++) {
    int64_t rows_d = 0;
    for (int r = 0; r < W; r++) rows_d += M(r, o_rows + (size_t)d);
    if (rows_d > 0x7fffffff) throw Error(TFGPU_ERR_INVALID, "tfgpu_exchange: more than 2^31-1 rows land on rank " + std::to_string(d));
    for (int64_t j = 0; j < C; j++) {
      if (!repr_is_var(desc[(size_t)j].repr)) continue;
      uint64_t bytes_d = 0;
      for (int r = 0; r < W; r++) bytes_d += (uint64_t)M(r, o_bytes + (size_t)j * W + (size_t)d);
      if (bytes_d > 0xffffffffull) throw Error(TFGPU_ERR_INVALID, "tfgpu_exchange: text column " + std::to_string(j) + " of more than 4 GiB lands on rank " + std::to_string(d));
    }
  }
  int64_t bflags = 0;
  bool some_lacks_old = false;
  for (int r = 0; r < W; r++) { bflags |= M(r, 0); some_lacks_old |= nold && hdr_all[(size_t)r * 3 + 1] == 0; }
  if (some_lacks_old) {  // the OldKeys a rank without any makes up are nil and absent: every rank sends both bitmaps
    bflags |= BF_OLD_PRESENT;
    for (int64_t j = ncols; j < C; j++) desc[(size_t)j].validity = true;
  }
  std::vector<int64_t> got_rows((size_t)W);
  int64_t n_out = 0;
  for (int r = 0; r < W; r++) { got_rows[(size_t)r] = M(r, o_rows + (size_t)me); n_out += got_rows[(size_t)r]; }
  std::vector<int64_t> send_rows(counts, counts + W);

  // ---- 2. plan every buffer's move, then one grouped RCCL call
  auto out = std::make_unique<tfgpu_dbatch>();
  out->nrows = n_out;
  out->ns = in.ns; out->table = in.table; out->schema = in.schema; out->key_names = in.key_names;
  std::vector<Move> moves;
  std::vector<Buf> keep;  // send-side temporaries: alive until the stream has run the collective
  struct Post { int kind; Buf tmp; DColumn *col; Buf *dst; };  // 0: lengths -> offsets, 1: bytes -> bitmap
  std::vector<Post> post;
  auto rows_move = [&](const void *send, void *recv, size_t elem) {
    moves.push_back(Move{static_cast<const uint8_t *>(send), static_cast<uint8_t *>(recv), elem, send_rows, got_rows});
  };
  auto ones_or_bits = [&](const Buf &bits, bool all_ones_if_null, uint8_t fill) -> Buf {  // bitmap -> one byte per row
    Buf b = dalloc((size_t)std::max<int64_t>(n_in, 1));
    if (n_in) {
      if (bits) xc_unpack_bits<<<grid_for(n_in, 256), 256, 0, st>>>(ptr<uint8_t>(bits), n_in, ptr<uint8_t>(b));
      else xc_fill_u8<<<grid_for(n_in, 256), 256, 0, st>>>(ptr<uint8_t>(b), n_in, all_ones_if_null ? 1 : fill);
    }
    keep.push_back(b);
    return b;
  };
  auto bitmap_move = [&](const Buf &bits, uint8_t fill_if_null, Buf *dst) {
    Buf s = ones_or_bits(bits, false, fill_if_null), r = dalloc((size_t)std::max<int64_t>(n_out, 1));
    rows_move(ptr<uint8_t>(s), ptr<uint8_t>(r), 1);
    post.push_back(Post{1, r, nullptr, dst});
  };
  auto plan_column = [&](const DColumn *c, const Desc &d, const std::string &name, int64_t j, DColumn &o) {
    o.name = name; o.repr = d.repr; o.dtype = d.dtype;
    if (repr_is_var(d.repr)) {
      Buf len = dalloc((size_t)std::max<int64_t>(n_in, 1) * 4), rlen = dalloc((size_t)(n_out + 1) * 4);
      if (n_in) {
        if (c) xc_lengths<<<grid_for(n_in, 256), 256, 0, st>>>(ptr<uint32_t>(c->offsets), n_in, ptr<uint32_t>(len));
        else xc_fill_u32<<<grid_for(n_in, 256), 256, 0, st>>>(ptr<uint32_t>(len), n_in, 0u);
      }
      keep.push_back(len);
      rows_move(ptr<uint8_t>(len), ptr<uint8_t>(rlen), 4);
      Move mv{c ? ptr<uint8_t>(c->payload()) : nullptr, nullptr, 1, std::vector<int64_t>((size_t)W), std::vector<int64_t>((size_t)W)};
      uint64_t total = 0;
      for (int r = 0; r < W; r++) {
        mv.split[(size_t)r] = M(me, o_bytes + (size_t)j * W + (size_t)r);
        mv.got[(size_t)r] = M(r, o_bytes + (size_t)j * W + (size_t)me);
        total += (uint64_t)mv.got[(size_t)r];
      }
      o.data = dalloc(std::max<uint64_t>(total, 1));
      o.data_len = total;
      mv.recv = ptr<uint8_t>(o.data);
      // the payload a destination gets starts where its first row starts
      moves.push_back(std::move(mv));
      o.offsets = rlen;
      post.push_back(Post{0, rlen, &o, nullptr});
    } else {
      const size_t w = repr_width(d.repr);
      if (!w) throw Error(TFGPU_ERR_INVALID, "tfgpu_exchange: column " + name + " has no exchangeable representation");
      o.values = dalloc((size_t)std::max<int64_t>(n_out, 1) * w);
      Buf sv = c ? c->values : dalloc_zero((size_t)std::max<int64_t>(n_in, 1) * w);
      keep.push_back(sv);
      rows_move(ptr<uint8_t>(sv), ptr<uint8_t>(o.values), w);
      if (d.nanos) {
        o.nanos = dalloc((size_t)std::max<int64_t>(n_out, 1) * 4);
        Buf sn = (c && c->nanos) ? c->nanos : dalloc_zero((size_t)std::max<int64_t>(n_in, 1) * 4);
        keep.push_back(sn);
        rows_move(ptr<uint8_t>(sn), ptr<uint8_t>(o.nanos), 4);
      }
    }
    if (d.validity) bitmap_move(c ? c->validity : Buf(), c ? 1 : 0, &o.validity);
    if (d.absent) bitmap_move(c ? c->absent : Buf(), 0, &o.absent);  // a rank whose rows all list the column sends zeros
  };
  out->cols.resize((size_t)ncols);
  out->old_keys.resize((size_t)nold);
  for (int64_t j = 0; j < ncols; j++) plan_column(&in.cols[(size_t)j], desc[(size_t)j], in.cols[(size_t)j].name, j, out->cols[(size_t)j]);
  for (int64_t k = 0; k < nold; k++)
    plan_column(local_old ? &in.old_keys[(size_t)k] : nullptr, desc[(size_t)(ncols + k)], local_old ? in.old_keys[(size_t)k].name : in.key_names[(size_t)k], ncols + k,
                out->old_keys[(size_t)k]);
  if (nold && (bflags & BF_OLD_PRESENT)) bitmap_move(local_old ? in.old_present : Buf(), local_old ? 1 : 0, &out->old_present);
  if (bflags & BF_KIND) {  // a batch without kinds is all inserts (tfgpu.h)
    out->kind = dalloc((size_t)std::max<int64_t>(n_out, 1));
    Buf sk = in.kind;
    if (!sk) { sk = dalloc((size_t)std::max<int64_t>(n_in, 1)); if (n_in) xc_fill_u8<<<grid_for(n_in, 256), 256, 0, st>>>(ptr<uint8_t>(sk), n_in, (uint8_t)TFGPU_K_INSERT); }
    keep.push_back(sk);
    rows_move(ptr<uint8_t>(sk), ptr<uint8_t>(out->kind), 1);
  }
  if (bflags & BF_SRC_ROW) {
    out->src_row = dalloc((size_t)std::max<int64_t>(n_out, 1) * 4);
    Buf ss = in.src_row;
    if (!ss) { ss = dalloc((size_t)std::max<int64_t>(n_in, 1) * 4); if (n_in) xc_iota_i32<<<grid_for(n_in, 256), 256, 0, st>>>(ptr<int32_t>(ss), n_in); }
    keep.push_back(ss);
    rows_move(ptr<uint8_t>(ss), ptr<uint8_t>(out->src_row), 4);
  }

  {  // what stays here: one launch over all buffers (split[me] == got[me]: both are this rank's rows for itself)
    std::vector<LocalPiece> pieces;
    for (const Move &m : moves) {
      size_t so = 0, ro = 0;
      for (int p = 0; p < me; p++) { so += (size_t)m.split[(size_t)p] * m.elem; ro += (size_t)m.got[(size_t)p] * m.elem; }
      const size_t sb = (size_t)m.split[(size_t)me] * m.elem;
      if ((size_t)m.got[(size_t)me] * m.elem != sb) throw Error(TFGPU_ERR_DEVICE, "tfgpu_exchange: internal: a rank's rows for itself differ between the send and the receive plan");
      for (size_t o = 0; o < sb; o += XC_PIECE) pieces.push_back(LocalPiece{m.send + so + o, m.recv + ro + o, (uint32_t)std::min<size_t>(XC_PIECE, sb - o), 0u});
    }
    if (!pieces.empty()) {
      Buf bp = upload_small(pieces.data(), pieces.size() * sizeof(LocalPiece));
      keep.push_back(bp);
      xc_copy_local<<<(unsigned)pieces.size(), 256, 0, st>>>(ptr<LocalPiece>(bp));
    }
  }
  if (W > 1) {
    TF_NCCL(R.GroupStart());
    for (const Move &m : moves) {
      size_t so = 0, ro = 0;
      for (int p = 0; p < W; p++) {
        const size_t sb = (size_t)m.split[(size_t)p] * m.elem, rb = (size_t)m.got[(size_t)p] * m.elem;
        if (p != me) {
          if (sb) TF_NCCL(R.Send(m.send + so, sb, NCCL_UINT8, p, cm.comm, st));
          if (rb) TF_NCCL(R.Recv(m.recv + ro, rb, NCCL_UINT8, p, cm.comm, st));
        }
        so += sb; ro += rb;
      }
    }
    TF_NCCL(R.GroupEnd());
  }

  // ---- 3. rebuild what travelled in another form
  for (Post &p : post) {
    if (p.kind == 0) {
      exclusive_scan_u32(ptr<uint32_t>(p.tmp), ptr<uint32_t>(p.tmp), n_out, true);
    } else {
      Buf bits = dalloc((size_t)std::max<int64_t>((n_out + 7) / 8, 1));
      if (n_out) xc_pack_bits<<<grid_for((n_out + 7) / 8, 256), 256, 0, st>>>(ptr<uint8_t>(p.tmp), n_out, ptr<uint8_t>(bits));
      *p.dst = bits;
    }
  }
  out->part_id = dalloc((size_t)std::max<int64_t>(n_out, 1) * 4);
  if (n_out) xc_fill_u32<<<grid_for(n_out, 256), 256, 0, st>>>(ptr<uint32_t>(out->part_id), n_out, (uint32_t)me);
  // send-side temporaries go back to the stream-ordered pool only after the stream has passed the collective
  sync();
  if (recv_counts) for (int r = 0; r < W; r++) recv_counts[r] = got_rows[(size_t)r];
  return out;
}

}  // namespace

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

extern "C" {

int tfgpu_comm_unique_id(uint8_t id[TFGPU_COMM_ID_BYTES]) {
  TF_API_BEGIN
  if (!id) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_comm_unique_id: null argument");
  (void)ctx();
  static_assert(sizeof(NcclUniqueId) == TFGPU_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  NcclUniqueId u;
  TF_NCCL(rccl().GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_comm_init(const uint8_t id[TFGPU_COMM_ID_BYTES], int rank, int world, tfgpu_comm **out) {
  TF_API_BEGIN
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_comm_init: bad argument");
  Context &cx = ctx();
  TF_HIP(hipSetDevice(cx.device));
  NcclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  auto c = std::make_unique<tfgpu_comm>();
  c->rank = rank; c->world = world;
  TF_NCCL(rccl().CommInitRank(&c->comm, world, u, rank));
  *out = c.release();
  return TFGPU_OK;
  TF_API_END
}

void tfgpu_comm_destroy(tfgpu_comm *c) {
  if (!c) return;
  try { if (c->comm) rccl().CommDestroy(c->comm); } catch (...) {}
  delete c;
}

int tfgpu_comm_rank(const tfgpu_comm *c) { return c ? c->rank : -1; }
int tfgpu_comm_world(const tfgpu_comm *c) { return c ? c->world : 0; }

int tfgpu_exchange(tfgpu_comm *c, const tfgpu_dbatch *in, const int64_t *counts, tfgpu_dbatch **out, int64_t *recv_counts) {
  TF_API_BEGIN
  tf::dense(in, true);  // its rows may still be a selection (tfgpu_dbatch::pending); ABSENT cells travel with their rows
  if (!c || !in || !counts || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_exchange: null argument");
  if (in->col_order) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_exchange: the batch's rows carry their own ColumnNames order (a collapsed batch): the exchange runs in front of Collapse");
  std::lock_guard<std::mutex> lk(c->mu);       // a communicator carries one collective at a time
  std::lock_guard<std::mutex> lk2(ctx().mu);
  *out = exchange(*c, *in, counts, recv_counts).release();
  return TFGPU_OK;
  TF_API_END
}

}  // extern "C"
