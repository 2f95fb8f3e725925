// tf_collapse.hip — abstract.Collapse on device (pkg/abstract/changeitem/change_item_collapse.go:48-134).
//
// The reference walks the batch once, in order, over three Go maps keyed by a row's key STRING — json.Marshal of its
// PrimaryKey values in key-name order, taken from OldKeys for an Update / Delete that carries them
// (OldOrCurrentKeysString / CurrentKeysString, change_item.go:314-358).  Every key string that can ever appear is one of
// two per row: K_cur(i) (from ColumnValues) and K_old(i) (from OldKeys), because a collapsed row's key is either its
// first item's OldKeys key or the last writer's current key.  That makes the walk a state machine over at most 2n
// integer key ids, and rows whose keys never meet are independent:
//
//   1. collapse_hash    lane = (row, cur|old): the key string is rendered by the json.Marshal emitter (tf_emit.hpp) into
//                       a hashing sink — 2 x 64 bits; equal strings <=> equal hashes up to a 2^-128 collision chance.
//   2. collapse_intern  open-addressing table in HBM (4n..8n slots): a key's id is the slot its hash claims.
//   3. collapse_link    union-find over key ids: a row ties K_cur(i) to K_old(i).  A component = the keys one chain of
//                       primary-key changes can reach; everything the reference's maps do for one key stays inside it.
//   4. radix sort of the rows by component root (rocPRIM, stable: input order survives inside a component).
//   5. collapse_walk    ONE lane per component replays the reference's switch over its rows in input order; the three
//                       maps are arrays indexed by key id (rows[k] = (first item m, last writer v), toDelete[k],
//                       hashKToIdx[k]).  Parallelism = number of components (a batch of n distinct keys: n lanes).
//   6. flags → one scan → three selection vectors → gathers: result = non-row items, rows by hashKToIdx, deletes.
//
// A collapsed row takes Kind / OldKeys / PartID / source position from its first item and ColumnValues from its last
// writer; a Delete that follows an item with OldKeys inherits those OldKeys (change_item_collapse.go:111-118).
//
// The compareColumns merge (:7-35, :86-100; TOAST updates that leave columns out): the columnar batch has ONE column list, and a
// row whose ColumnNames leave a column out marks the cell ABSENT (DColumn::absent).  The walk links every merged item to the item
// it merged into (prev[]); a surviving row then takes each column from the LAST item of its chain that lists it
// (collapse_merge_cols: one lane per (row, column that can be absent) walks the chain back), and lists the union.  The reference
// APPENDS the names a later item brings (`total`): where some chain's merged name order is not batch order (collapse_merge_order
// says so) the result carries every row's own order (collapse_merge_names → tfgpu_dbatch::col_order).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>

#include "tf_emit.hpp"

namespace tf {

std::unique_ptr<tfgpu_dbatch> gather_rows(const tfgpu_dbatch &in, const Buf &sel, int64_t m);  // tf_transform.hip

static inline unsigned cgrid(int64_t n) { return (unsigned)((n + 255) / 256); }
static constexpr uint32_t NOKEY = 0xFFFFFFFFu;
__device__ __forceinline__ int repr_width_dev(int r) {
  switch (r) {
    case TFGPU_R_INT8: case TFGPU_R_UINT8: case TFGPU_R_BOOL: return 1;
    case TFGPU_R_INT16: case TFGPU_R_UINT16: return 2;
    case TFGPU_R_INT32: case TFGPU_R_UINT32: case TFGPU_R_FLOAT32: return 4;
    default: return 8;
  }
}

// ---- 1. key strings → 128-bit hashes ----------------------------------------------------------------------------
struct HashSink {
  uint64_t h1 = 0x243F6A8885A308D3ull, h2 = 0x13198A2E03707344ull, acc = 0, len = 0;
  uint32_t n = 0;
  __device__ __forceinline__ void mix(uint64_t w) {
    h1 = (h1 ^ w) * 0x9E3779B97F4A7C15ull; h1 ^= h1 >> 32;
    h2 = (((h2 << 31) | (h2 >> 33)) ^ w) * 0xC2B2AE3D27D4EB4Full; h2 ^= h2 >> 29;
  }
  __device__ __forceinline__ void put(uint32_t c) {
    acc |= (uint64_t)(c & 0xFFu) << (8 * n); len++;
    if (++n == 8) { mix(acc); acc = 0; n = 0; }
  }
  __device__ __forceinline__ void finish() {
    mix(acc); mix(len);
    h1 ^= h1 >> 33; h1 *= 0xFF51AFD7ED558CCDull; h1 ^= h1 >> 33;
    h2 ^= h2 >> 33; h2 *= 0xC4CEB9FE1A85EC53ull; h2 ^= h2 >> 33;
  }
};

struct KCol { DCol cur, old; int32_t has_cur, has_old; };  // one PrimaryKey name, in sorted-name order
struct HashParams {
  const KCol *keys; int32_t nkeys;
  int64_t n;
  const uint8_t *old_present; int32_t has_old_keys;  // bitmap (null = every row) / the batch carries OldKeys at all
  uint64_t *h;        // [2n][2]: (h1 | 1, h2); h1 == 0: the row has no OldKeys
  uint32_t *badflag;  // a float key that json.Marshal refuses (NaN / Inf)
  uint32_t *dbg;      // TFGPU_COLLAPSE_DEBUG=1 (measurement only): [0] the longest probe sequence, [1] all probes
  uint32_t *defer_n, *defer_list;  // entries whose place only a compare of the key texts decides (collapse_intern_text)
  int32_t weak;
};
__device__ __forceinline__ bool float_bad(const DCol &c, int64_t r) {
  if (c.repr != TFGPU_R_FLOAT32 && c.repr != TFGPU_R_FLOAT64) return false;
  if (!is_valid(c, r)) return false;
  const double v = c.repr == TFGPU_R_FLOAT32 ? (double)((const float *)c.values)[r] : ((const double *)c.values)[r];
  return v != v || v == INFINITY || v == -INFINITY;
}
// the key string of entry j = (row j >> 1, current keys | OldKeys): CurrentKeysString / OldOrCurrentKeysString (change_item.go:314-358)
template <class S> __device__ __forceinline__ bool emit_key(const HashParams &p, int64_t j, S &s) {
  const int64_t r = j >> 1;
  const bool old = j & 1;
  bool bad = false;
  s.put('[');
  for (int k = 0; k < p.nkeys; k++) {
    if (k) s.put(',');
    const KCol &kc = p.keys[k];
    const bool has = old ? kc.has_old : kc.has_cur;
    if (!has) { put_lit(s, "null"); continue; }  // keys[k] = nil: the key column is not among the row's names
    const DCol &c = old ? kc.old : kc.cur;
    if (float_bad(c, r)) { bad = true; continue; }
    emit_json_cell(s, c, r, 0, true);
  }
  s.put(']');
  return bad;
}
__global__ void __launch_bounds__(256) collapse_hash(HashParams p) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * p.n) return;
  const int64_t r = j >> 1;
  const bool old = j & 1;
  if (old && (!p.has_old_keys || (p.old_present && !((p.old_present[r >> 3] >> (r & 7)) & 1)))) { p.h[2 * j] = 0; p.h[2 * j + 1] = 0; return; }
  HashSink s;
  if (emit_key(p, j, s)) *p.badflag = 1;
  s.finish();
  if (p.weak) { s.h1 &= 3ull; s.h2 = 0; }  // TFGPU_COLLAPSE_WEAK_HASH=1 (tests): almost every key collides, the string compare decides
  p.h[2 * j] = s.h1 | 1ull;
  p.h[2 * j + 1] = s.h2;
}

// ---- 2. key ids: the slot a hash claims — and, hash for hash, the key strings compared, so that two entries share an id
//      exactly when the reference's map would file them under one string ----------------------------------------
struct WindowSink {  // the bytes [lo, lo + 64) of an emitted text
  uint8_t *buf; uint32_t lo, pos = 0;
  __device__ __forceinline__ void put(uint32_t c) { if (pos - lo < 64u) buf[pos - lo] = (uint8_t)c; pos++; }
};
__device__ bool key_text_equal(const HashParams &p, int64_t a, int64_t b) {
  CountSink ca, cb;
  emit_key(p, a, ca); emit_key(p, b, cb);
  if (ca.n != cb.n) return false;
  uint8_t wa[64], wb[64];
  for (uint32_t lo = 0; lo < ca.n; lo += 64) {
    WindowSink sa{wa, lo}, sb{wb, lo};
    emit_key(p, a, sa); emit_key(p, b, sb);
    const uint32_t m = ca.n - lo < 64u ? ca.n - lo : 64u;
    for (uint32_t i = 0; i < m; i++) if (wa[i] != wb[i]) return false;
  }
  return true;
}
// Same key string?  Equal raw values of every key ⇒ equal strings; a fixed-width value that differs ⇒ different strings
// (integers, bools, times and finite floats print injectively); var-width values that differ may still print alike
// (invalid UTF-8 → U+FFFD): only then are the texts themselves compared.
// TEXT = false (collapse_intern, the kernel every entry runs): 2 = "only the texts can tell" — the entry is put off to
// collapse_intern_text.  The text compare renders both keys three times through the json.Marshal emitter: inlined into the kernel of
// every entry it cost 248 VGPRs and 224 bytes of scratch a lane (two waves a SIMD for a kernel of dependent random reads: 175 us
// whatever the batch); on its own it is paid by the entries that need it — hash twins whose raw values differ, once in a blue moon.
template <bool TEXT> __device__ int keys_equal(const HashParams &p, int64_t a, int64_t b) {
  const int64_t ra = a >> 1, rb = b >> 1;
  bool need_text = false;
  for (int k = 0; k < p.nkeys; k++) {
    const KCol &kc = p.keys[k];
    const bool ha = (a & 1) ? kc.has_old : kc.has_cur, hb = (b & 1) ? kc.has_old : kc.has_cur;
    const DCol &x = (a & 1) ? kc.old : kc.cur, &y = (b & 1) ? kc.old : kc.cur;
    const bool na = !ha || !is_valid(x, ra), nb = !hb || !is_valid(y, rb);
    if (na || nb) { if (na != nb) need_text = true; continue; }  // (a nil prints "null"; a JSON value may too: compare the text)
    if (x.repr != y.repr) { need_text = true; continue; }
    if (x.offsets) {
      const uint32_t oa = x.offsets[ra], la = x.offsets[ra + 1] - oa, ob = y.offsets[rb], lb = y.offsets[rb + 1] - ob;
      bool same = la == lb;
      for (uint32_t i = 0; same && i < la; i++) same = x.data[oa + i] == y.data[ob + i];
      if (!same) need_text = true;
    } else {
      const int w = repr_width_dev(x.repr);
      const uint8_t *va = (const uint8_t *)x.values + ra * w, *vb = (const uint8_t *)y.values + rb * w;
      for (int i = 0; i < w; i++) if (va[i] != vb[i]) return 0;
      if (x.repr == TFGPU_R_TIME && (x.nanos ? x.nanos[ra] : 0) != (y.nanos ? y.nanos[rb] : 0)) return 0;
    }
  }
  if (!need_text) return 1;
  if constexpr (TEXT) return key_text_equal(p, a, b) ? 1 : 0;
  else return 2;
}
template <bool TEXT> __device__ __forceinline__ void intern_entry(const HashParams &p, int64_t j, uint32_t *owner, uint32_t mask, uint32_t *__restrict__ keyid) {
  const uint64_t *__restrict__ h = p.h;
  const uint64_t a = h[2 * j], b = h[2 * j + 1];
  if (a == 0) { keyid[j] = NOKEY; return; }
  uint32_t slot = (uint32_t)(b ^ (a >> 17)) & mask;
  uint32_t probes = 0;
  for (;;) {
    probes++;
    uint32_t o = __hip_atomic_load(&owner[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (o == NOKEY) o = atomicCAS(&owner[slot], NOKEY, (uint32_t)j);
    if (o == NOKEY || o == (uint32_t)j) { keyid[j] = slot; break; }
    if (h[2 * (int64_t)o] == a && h[2 * (int64_t)o + 1] == b) {
      const int eq = keys_equal<TEXT>(p, j, (int64_t)o);
      if (eq == 1) { keyid[j] = slot; break; }
      if (eq == 2) { p.defer_list[atomicAdd(p.defer_n, 1u)] = (uint32_t)j; break; }  // it owns nothing yet: collapse_intern_text files it as a late comer
    }
    slot = (slot + 1) & mask;  // another key (or, once in 2^128, another string with this hash): probe on
  }
  if (p.dbg) { atomicMax(&p.dbg[0], probes); atomicAdd(&p.dbg[1], probes); }
}
__global__ void __launch_bounds__(256) collapse_intern(HashParams p, int64_t n2, uint32_t *owner, uint32_t mask, uint32_t *__restrict__ keyid) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n2) intern_entry<false>(p, j, owner, mask, keyid);
}
// the entries only the key TEXTS can place (usually none: the launch finds an empty list and leaves)
__global__ void __launch_bounds__(256) collapse_intern_text(HashParams p, uint32_t *owner, uint32_t mask, uint32_t *__restrict__ keyid) {
  const uint32_t n = *p.defer_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) intern_entry<true>(p, (int64_t)p.defer_list[i], owner, mask, keyid);
}

// ---- 3. components -----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) collapse_init(uint32_t *parent, int32_t *rows_m, int32_t *del_i, uint32_t *owner, int64_t cap) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cap) return;
  parent[k] = (uint32_t)k; rows_m[k] = -1; del_i[k] = -1; owner[k] = NOKEY;
}
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x) {
  for (;;) {
    const uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == x) return x;
    x = p;
  }
}
__global__ void __launch_bounds__(256) collapse_link(const uint32_t *__restrict__ keyid, const uint8_t *__restrict__ kind, int64_t n, uint32_t *parent) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  uint32_t a = keyid[2 * r], b = keyid[2 * r + 1];
  if (b == NOKEY || a == b) return;
  // A Delete that has OldKeys is named by them and its ColumnValues are never read (change_item_collapse.go:108-118): tying its
  // current key in would only merge components — sources whose deletes carry no columns (pg, Debezium) give every such row the
  // SAME current key ("null"), and one lane would then replay the whole batch.
  if (kind && kind[r] == TFGPU_K_DELETE) return;
  for (;;) {  // the larger root is hung under the smaller one: roots only ever decrease, no cycles
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { const uint32_t t = a; a = b; b = t; }
    if (atomicCAS(&parent[a], a, b) == a) return;
  }
}
__global__ void __launch_bounds__(256) collapse_roots(const uint32_t *__restrict__ keyid, const uint8_t *__restrict__ kind, int64_t n, uint32_t *parent, uint32_t *__restrict__ root, uint32_t *__restrict__ idx) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint32_t ko = keyid[2 * r + 1];
  root[r] = uf_find(parent, (kind && kind[r] == TFGPU_K_DELETE && ko != NOKEY) ? ko : keyid[2 * r]);  // the key that names the row
  idx[r] = (uint32_t)r;
}

// ---- 5. the reference's loop, one lane per component ------------------------------------------------------------
struct WalkParams {
  int64_t n;
  const uint32_t *sroot, *sidx;  // rows sorted by component, input order inside
  const uint32_t *keyid;         // [2n]: K_cur, K_old (NOKEY = no OldKeys)
  const uint8_t *kind;
  int32_t *rows_m, *rows_v, *k2idx, *del_i, *del_o;  // per key id
  uint32_t *ak;                  // per row: the key the row was filed under
  int32_t *prev;                 // per row, only for batches with ABSENT cells: the item this Update merged into (-1: it starts a chain)
  // only when a PRIMARY-KEY column can be absent: a merged row's current key is no longer its last writer's — an Update that lists
  // none of the key columns leaves the chain's key as it was (TestCollapse "multiple PK, toast")
  const uint8_t *keyform;        // per row: 0 = lists every key column, 1 = lists none, 2 = some (refused)
  uint32_t *rows_kc;             // per key id: CurrentKeysString of the merged row filed there
  uint32_t *bad;
};
__global__ void __launch_bounds__(256) collapse_walk(WalkParams p) {
  const int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q0 >= p.n) return;
  const uint32_t root = p.sroot[q0];
  if (q0 > 0 && p.sroot[q0 - 1] == root) return;  // not the first row of its component
  for (int64_t q = q0; q < p.n && p.sroot[q] == root; q++) {
    const int32_t i = (int32_t)p.sidx[q];
    const uint32_t kind = p.kind[i];
    const uint32_t kc = p.keyid[2 * (int64_t)i], kro = p.keyid[2 * (int64_t)i + 1];
    // OldOrCurrentKeysString: OldKeys name the row only for an Update / Delete that has them
    const uint32_t ko = ((kind == TFGPU_K_UPDATE || kind == TFGPU_K_DELETE) && kro != NOKEY) ? kro : kc;
    if (kind == TFGPU_K_INSERT) {
      p.del_i[ko] = -1;
      p.rows_m[ko] = i; p.rows_v[ko] = i; p.k2idx[ko] = i; p.ak[i] = ko;
      if (p.prev) p.prev[i] = -1;
      if (p.rows_kc) p.rows_kc[ko] = kc;
    } else if (kind == TFGPU_K_UPDATE) {
      p.del_i[ko] = -1;
      const int32_t m = p.rows_m[ko];
      if (m < 0) {  // nothing to merge into: filed under its CURRENT key
        p.rows_m[kc] = i; p.rows_v[kc] = i; p.k2idx[kc] = i; p.ak[i] = kc;
        if (p.prev) p.prev[i] = -1;
        if (p.rows_kc) p.rows_kc[kc] = kc;
      } else {      // current.ColumnValues = c.ColumnValues; the row keeps the first item's Kind and OldKeys
        if (p.prev) p.prev[i] = p.rows_v[ko];  // (compareColumns: columns this item leaves out keep the value of the chain before it)
        const uint32_t km = p.kind[m], krom = p.keyid[2 * (int64_t)m + 1];
        uint32_t cur = kc;  // CurrentKeysString of the merged row: the key columns come from the last item that lists them
        if (p.rows_kc) {
          const uint32_t form = p.keyform[i];
          if (form == 1) cur = p.rows_kc[ko];
          else if (form == 2) *p.bad = 1;
        }
        const uint32_t newk = ((km == TFGPU_K_UPDATE || km == TFGPU_K_DELETE) && krom != NOKEY) ? krom : cur;
        if (newk != ko) p.rows_m[ko] = -1;
        p.rows_m[newk] = m; p.rows_v[newk] = i; p.k2idx[newk] = i; p.ak[i] = newk;
        if (p.rows_kc) p.rows_kc[newk] = cur;
      }
    } else if (kind == TFGPU_K_DELETE) {
      const int32_t m = p.rows_m[ko];
      p.rows_m[ko] = -1;
      uint32_t k = ko; int32_t o = i;
      if (m >= 0) { const uint32_t krom = p.keyid[2 * (int64_t)m + 1]; if (krom != NOKEY) { k = krom; o = m; } }  // c.OldKeys = current.OldKeys
      p.del_i[k] = i; p.del_o[k] = o; p.ak[i] = k;
    }
  }
}

// ---- 6. result order: non-row items, rows by hashKToIdx, deletes (by position) ----------------------------------
__global__ void __launch_bounds__(256) collapse_flags(int64_t n, const uint8_t *__restrict__ kind, const uint32_t *__restrict__ ak, const int32_t *__restrict__ rows_m,
                                                      const int32_t *__restrict__ k2idx, const int32_t *__restrict__ del_i, uint32_t *__restrict__ f) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t kd = kind[i];
  uint32_t f0 = 0, f1 = 0, f2 = 0;
  if (kd == TFGPU_K_INSERT || kd == TFGPU_K_UPDATE) { const uint32_t k = ak[i]; f1 = rows_m[k] >= 0 && k2idx[k] == (int32_t)i; }
  else if (kd == TFGPU_K_DELETE) f2 = del_i[ak[i]] == (int32_t)i;
  else f0 = 1;
  f[i] = f0; f[n + i] = f1; f[2 * n + i] = f2;
}
__global__ void __launch_bounds__(256) collapse_select(int64_t n, const uint8_t *__restrict__ kind, const uint32_t *__restrict__ ak, const int32_t *__restrict__ rows_m,
                                                       const int32_t *__restrict__ rows_v, const int32_t *__restrict__ del_o, const uint32_t *__restrict__ pos,
                                                       int32_t *__restrict__ sel_meta, int32_t *__restrict__ sel_val, int32_t *__restrict__ sel_old) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t kd = kind[i];
  const int64_t c = (kd == TFGPU_K_INSERT || kd == TFGPU_K_UPDATE) ? 1 : kd == TFGPU_K_DELETE ? 2 : 0;
  const uint32_t a = pos[c * n + i], b = pos[c * n + i + 1];
  if (a == b) return;
  int32_t m = (int32_t)i, v = (int32_t)i, o = (int32_t)i;
  if (c == 1) { const uint32_t k = ak[i]; m = rows_m[k]; v = rows_v[k]; o = m; }
  else if (c == 2) o = del_o[ak[i]];
  sel_meta[a] = m; sel_val[a] = v; sel_old[a] = o;
}
// which of its PRIMARY-KEY columns a row lists (only launched when one of them can be absent)
__global__ void __launch_bounds__(256) collapse_keyform(int64_t n, int nk, const uint8_t *const *kabs, uint8_t *form) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int out = 0;
  for (int k = 0; k < nk; k++) if (kabs[k] && ((kabs[k][r >> 3] >> (r & 7)) & 1u)) out++;
  form[r] = out == 0 ? 0 : out == nk ? 1 : 2;
}
// ---- the compareColumns merge over ABSENT cells (change_item_collapse.go:7-35, :86-100) --------------------------
// Output row k = the chain sel_meta[k] (first item) … sel_val[k] (last writer), linked backwards by prev[].  Column j of the row
// comes from the last item of the chain that lists it: sel_col[j][k]; none does → the last writer's (absent) cell, and the row
// does not list the column.  first_at[j][k] = how far behind the last writer the EARLIEST item listing the column sits (the
// reference's `total` orders names by first appearance); NEVER for a column no item lists.
struct MergeParams {
  int64_t m, n;
  const int32_t *sel_meta, *sel_val, *prev;
  const uint8_t *const *absent;  // [na] bitmaps of the columns that can be absent
  int32_t *sel_col;              // [na][m]
  uint32_t *first_at;            // [na][m]
};
static constexpr uint32_t NEVER = 0xFFFFFFFFu;
__global__ void __launch_bounds__(256) collapse_merge_cols(MergeParams p) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= p.m) return;
  const int j = blockIdx.y;
  const uint8_t *ab = p.absent[j];
  const int32_t head = p.sel_meta[k], tail = p.sel_val[k];
  int32_t r = tail, src = -1;
  uint32_t d = 0, first = NEVER;
  for (int64_t guard = 0; guard <= p.n; guard++) {
    if (!((ab[r >> 3] >> (r & 7)) & 1u)) { if (src < 0) src = r; first = d; }
    if (r == head) break;
    const int32_t q = p.prev[r];
    if (q < 0) break;
    r = q; d++;
  }
  p.sel_col[(int64_t)j * p.m + k] = src < 0 ? tail : src;
  p.first_at[(int64_t)j * p.m + k] = first;
}
// `total` = the first item's names, then what every later item brings, in that item's order: batch order exactly when, walking the
// batch's columns, the first appearance never moves EARLIER in the chain (a column every row lists appears with the first item).
__global__ void __launch_bounds__(256) collapse_merge_order(MergeParams p, const int32_t *__restrict__ acol /* [ncols]: index among the absent-capable columns or -1 */, int ncols, uint32_t *bad) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= p.m) return;
  const int32_t head = p.sel_meta[k];
  int32_t r = p.sel_val[k];
  uint32_t len = 0;  // distance of the first item from the last writer
  for (int64_t guard = 0; guard <= p.n && r != head; guard++) { const int32_t q = p.prev[r]; if (q < 0) break; r = q; len++; }
  if (!len) return;  // one item: its own names
  uint32_t last = len;
  for (int c = 0; c < ncols; c++) {
    const uint32_t f = acol[c] < 0 ? len : p.first_at[(int64_t)acol[c] * p.m + k];
    if (f == NEVER) continue;
    if (f > last) { *bad = 1; return; }
    last = f;
  }
}
// Every output row's ColumnNames as column indices: the columns it lists by first appearance in its chain (earliest item first, an
// item's own columns in batch order — compareColumns' `total`, change_item_collapse.go:19-33), then the ones it does not list.
__global__ void __launch_bounds__(256) collapse_merge_names(MergeParams p, const int32_t *__restrict__ acol, int ncols, uint16_t *__restrict__ order) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= p.m) return;
  const int32_t head = p.sel_meta[k];
  int32_t r = p.sel_val[k];
  uint32_t len = 0;
  for (int64_t guard = 0; guard <= p.n && r != head; guard++) { const int32_t q = p.prev[r]; if (q < 0) break; r = q; len++; }
  auto first_of = [&](int c) { return acol[c] < 0 ? len : p.first_at[(int64_t)acol[c] * p.m + k]; };
  uint16_t *o = order + k * ncols;
  int pos = 0;
  uint32_t cur = len + 1;  // first appearances still to place are < cur
  while (pos < ncols) {
    uint32_t next = NEVER; bool any = false;
    for (int c = 0; c < ncols; c++) { const uint32_t f = first_of(c); if (f != NEVER && f < cur && (!any || f > next)) { next = f; any = true; } }
    if (!any) break;
    for (int c = 0; c < ncols; c++) if (first_of(c) == next) o[pos++] = (uint16_t)c;
    cur = next;
  }
  for (int c = 0; c < ncols; c++) if (first_of(c) == NEVER) o[pos++] = (uint16_t)c;  // not listed: behind the listed ones
}
// does a bitmap of n bits hold a set one?  (bits past n in its last byte do not count)
__global__ void __launch_bounds__(256) collapse_any_bit(const uint8_t *bits, int64_t n, uint32_t *flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nb = (n + 7) / 8;
  if (i >= nb) return;
  uint32_t v = bits[i];
  if (i == nb - 1 && (n & 7)) v &= (1u << (n & 7)) - 1u;
  if (v) *flag = 1u;
}
__global__ void __launch_bounds__(256) collapse_any_non_insert(const uint8_t *kind, int64_t n, uint32_t *flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && kind[i] != TFGPU_K_INSERT && kind[i] != TFGPU_K_SYNCHRONIZE) *flag = 1;  // InsertsOnly :37-44
}

// ---- ChangeItem.KeysChanged (change_item.go:237-286): reflect.DeepEqual of the old and the new value of every PK column ----
struct KeysChangedParams { const KCol *keys; int32_t nkeys; int64_t n; const uint8_t *kind, *old_present; int32_t has_old_keys; uint8_t *out; uint32_t *count; };
__device__ bool cell_deep_equal(const DCol &a, int64_t r, const DCol &b) {  // both non-nil
  if (a.repr != b.repr) return false;  // different Go dynamic types are never deeply equal
  if (a.offsets) {
    const uint32_t a0 = a.offsets[r], an = a.offsets[r + 1] - a0, b0 = b.offsets[r], bn = b.offsets[r + 1] - b0;
    if (an != bn) return false;
    for (uint32_t i = 0; i < an; i++) if (a.data[a0 + i] != b.data[b0 + i]) return false;
    return true;
  }
  switch (a.repr) {
    case TFGPU_R_FLOAT32: return ((const float *)a.values)[r] == ((const float *)b.values)[r];    // NaN != NaN, -0 == 0
    case TFGPU_R_FLOAT64: return ((const double *)a.values)[r] == ((const double *)b.values)[r];
    case TFGPU_R_TIME:
      return ((const int64_t *)a.values)[r] == ((const int64_t *)b.values)[r] && (a.nanos ? a.nanos[r] : 0) == (b.nanos ? b.nanos[r] : 0);
    default: break;
  }
  switch (repr_width_dev(a.repr)) {
    case 1: return ((const uint8_t *)a.values)[r] == ((const uint8_t *)b.values)[r];
    case 2: return ((const uint16_t *)a.values)[r] == ((const uint16_t *)b.values)[r];
    case 4: return ((const uint32_t *)a.values)[r] == ((const uint32_t *)b.values)[r];
    default: return ((const uint64_t *)a.values)[r] == ((const uint64_t *)b.values)[r];
  }
}
__global__ void __launch_bounds__(256) keys_changed_kernel(KeysChangedParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.n) return;
  bool changed = false;
  if (p.kind && p.kind[r] == TFGPU_K_UPDATE) {
    const bool row_has_old = p.has_old_keys && (!p.old_present || ((p.old_present[r >> 3] >> (r & 7)) & 1));
    for (int k = 0; k < p.nkeys && !changed; k++) {
      const KCol &kc = p.keys[k];
      const bool ov = row_has_old && kc.has_old && is_valid(kc.old, r), nv = kc.has_cur && is_valid(kc.cur, r);
      if (ov != nv) changed = true;
      else if (ov && !cell_deep_equal(kc.old, r, kc.cur)) changed = true;
    }
  }
  p.out[r] = changed ? 1 : 0;
  if (changed) atomicAdd(p.count, 1u);
}

static const DColumn *find_first(const std::vector<DColumn> &cols, const std::string &name) {
  for (auto &c : cols) if (c.name == name) return &c;  // KeysChanged breaks at the first match
  return nullptr;
}
static const DColumn *find_col(const std::vector<DColumn> &cols, const std::string &name) {
  const DColumn *hit = nullptr;
  for (auto &c : cols) if (c.name == name) hit = &c;  // a later duplicate wins, like the map write in CurrentKeysString
  return hit;
}

std::unique_ptr<tfgpu_dbatch> collapse_rows(const tfgpu_dbatch &in) {
  const int64_t n = in.nrows;
  hipStream_t st = ctx().stream;
  auto same = [&] { return std::make_unique<tfgpu_dbatch>(in); };
  if (n < 2 || !in.kind) return same();  // len(input) < 2; no Kind array = inserts only
  if (n >= (int64_t)1 << 29) throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_collapse: more than 2^29 rows in one batch");
  if (in.col_order) throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_collapse: the batch's rows carry their own ColumnNames order (it is a collapsed batch already)");
  {  // InsertsOnly
    Buf flag = dalloc_zero(4);
    collapse_any_non_insert<<<cgrid(n), 256, 0, st>>>(ptr<uint8_t>(in.kind), n, ptr<uint32_t>(flag));
    const uint32_t *h = d2h_u32(flag->p);
    sync();
    if (!*h) return same();
  }
  // keyCols := input[0].MakeMapKeys(), walked in sorted order (util.MapKeysInOrder)
  std::vector<std::string> names = in.key_names;
  std::sort(names.begin(), names.end());
  names.erase(std::unique(names.begin(), names.end()), names.end());
  if (names.empty()) return same();

  materialize(in);  // key strings read the packed payload
  std::vector<KCol> kc(names.size());
  for (size_t k = 0; k < names.size(); k++) {
    std::memset(&kc[k], 0, sizeof(KCol));
    if (const DColumn *c = find_col(in.cols, names[k])) { kc[k].cur = dcol_of(*c); kc[k].has_cur = 1; }
    if (const DColumn *c = find_col(in.old_keys, names[k])) { kc[k].old = dcol_of(*c); kc[k].has_old = 1; }
  }
  Buf bkc = upload_small(kc.data(), kc.size() * sizeof(KCol));
  const int64_t n2 = 2 * n;
  int bits = 4;
  while (((int64_t)1 << bits) < 4 * n) bits++;  // load factor <= 1/2 even if every row brings two new keys
  const int64_t cap = (int64_t)1 << bits;
  Buf hashes = dalloc((size_t)n2 * 16), keyid = dalloc((size_t)n2 * 4), bad = dalloc_zero(4);
  Buf owner = dalloc((size_t)cap * 4), parent = dalloc((size_t)cap * 4), rows_m = dalloc((size_t)cap * 4), rows_v = dalloc((size_t)cap * 4),
      k2idx = dalloc((size_t)cap * 4), del_i = dalloc((size_t)cap * 4), del_o = dalloc((size_t)cap * 4);
  std::unique_ptr<KernelTimer> timer = std::make_unique<KernelTimer>("collapse_keys");
  collapse_init<<<cgrid(cap), 256, 0, st>>>(ptr<uint32_t>(parent), ptr<int32_t>(rows_m), ptr<int32_t>(del_i), ptr<uint32_t>(owner), cap);
  HashParams hp{};
  hp.keys = ptr<KCol>(bkc); hp.nkeys = (int32_t)kc.size(); hp.n = n;
  hp.old_present = ptr<uint8_t>(in.old_present); hp.has_old_keys = in.old_keys.empty() ? 0 : 1;
  hp.h = ptr<uint64_t>(hashes); hp.badflag = ptr<uint32_t>(bad);
  static const bool weak = [] { const char *e = std::getenv("TFGPU_COLLAPSE_WEAK_HASH"); return e && e[0] == '1'; }();
  hp.weak = weak ? 1 : 0;
  static const bool dbg_on = [] { const char *e = std::getenv("TFGPU_COLLAPSE_DEBUG"); return e && e[0] == '1'; }();
  Buf dbgb = dbg_on ? dalloc_zero(16) : nullptr;
  hp.dbg = ptr<uint32_t>(dbgb);
  collapse_hash<<<cgrid(n2), 256, 0, st>>>(hp);
  Buf defer = dalloc((size_t)(n2 + 1) * 4);
  TF_HIP(hipMemsetAsync(defer->p, 0, 4, st));
  hp.defer_n = ptr<uint32_t>(defer); hp.defer_list = ptr<uint32_t>(defer) + 1;
  collapse_intern<<<cgrid(n2), 256, 0, st>>>(hp, n2, ptr<uint32_t>(owner), (uint32_t)(cap - 1), ptr<uint32_t>(keyid));
  collapse_intern_text<<<(unsigned)std::min<int64_t>(cgrid(n2), 256), 256, 0, st>>>(hp, ptr<uint32_t>(owner), (uint32_t)(cap - 1), ptr<uint32_t>(keyid));
  if (dbg_on) { const uint32_t *hd = d2h_u32(dbgb->p, 2); sync(); std::fprintf(stderr, "tfgpu collapse: %lld entries, table 2^%d, longest probe sequence %u, probes %u\n", (long long)n2, bits, hd[0], hd[1]); }
  collapse_link<<<cgrid(n), 256, 0, st>>>(ptr<uint32_t>(keyid), ptr<uint8_t>(in.kind), n, ptr<uint32_t>(parent));
  Buf root = dalloc((size_t)n * 4), idx = dalloc((size_t)n * 4), sroot = dalloc((size_t)n * 4), sidx = dalloc((size_t)n * 4);
  collapse_roots<<<cgrid(n), 256, 0, st>>>(ptr<uint32_t>(keyid), ptr<uint8_t>(in.kind), n, ptr<uint32_t>(parent), ptr<uint32_t>(root), ptr<uint32_t>(idx));
  timer.reset(); timer = std::make_unique<KernelTimer>("collapse_sort");
  {
    size_t tmp_bytes = 0;
    TF_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ptr<uint32_t>(root), ptr<uint32_t>(sroot), ptr<uint32_t>(idx), ptr<uint32_t>(sidx), (size_t)n, 0u, (unsigned)bits, st));
    Buf tmp = dalloc(tmp_bytes + 16);
    TF_HIP(rocprim::radix_sort_pairs(tmp->p, tmp_bytes, ptr<uint32_t>(root), ptr<uint32_t>(sroot), ptr<uint32_t>(idx), ptr<uint32_t>(sidx), (size_t)n, 0u, (unsigned)bits, st));
  }
  timer.reset(); timer = std::make_unique<KernelTimer>("collapse_walk");
  Buf ak = dalloc((size_t)n * 4);
  std::vector<size_t> acols;  // the columns some row leaves out of its ColumnNames
  for (size_t c = 0; c < in.cols.size(); c++) if (in.cols[c].absent) acols.push_back(c);
  Buf prev = acols.empty() ? nullptr : dalloc((size_t)n * 4);
  WalkParams wp{};
  wp.n = n; wp.sroot = ptr<uint32_t>(sroot); wp.sidx = ptr<uint32_t>(sidx); wp.keyid = ptr<uint32_t>(keyid); wp.kind = ptr<uint8_t>(in.kind);
  wp.rows_m = ptr<int32_t>(rows_m); wp.rows_v = ptr<int32_t>(rows_v); wp.k2idx = ptr<int32_t>(k2idx); wp.del_i = ptr<int32_t>(del_i); wp.del_o = ptr<int32_t>(del_o);
  wp.ak = ptr<uint32_t>(ak); wp.prev = ptr<int32_t>(prev);
  Buf keyform, rows_kc, badform;
  {
    std::vector<const uint8_t *> kabs;  // the key columns the batch holds, with their absent bitmaps
    bool any = false;
    for (size_t k = 0; k < names.size(); k++) if (const DColumn *c = find_col(in.cols, names[k])) { kabs.push_back(ptr<uint8_t>(c->absent)); any |= (bool)c->absent; }
    if (any) {
      Buf bk = upload_small(kabs.data(), kabs.size() * sizeof(uint8_t *));
      keyform = dalloc((size_t)n); rows_kc = dalloc((size_t)cap * 4); badform = dalloc_zero(4);
      collapse_keyform<<<cgrid(n), 256, 0, st>>>(n, (int)kabs.size(), ptr<const uint8_t *>(bk), ptr<uint8_t>(keyform));
      wp.keyform = ptr<uint8_t>(keyform); wp.rows_kc = ptr<uint32_t>(rows_kc); wp.bad = ptr<uint32_t>(badform);
    }
  }
  collapse_walk<<<cgrid(n), 256, 0, st>>>(wp);
  timer.reset(); timer = std::make_unique<KernelTimer>("collapse_select");
  Buf f = dalloc((size_t)(3 * n + 1) * 4 + 16);
  collapse_flags<<<cgrid(n), 256, 0, st>>>(n, ptr<uint8_t>(in.kind), ptr<uint32_t>(ak), ptr<int32_t>(rows_m), ptr<int32_t>(k2idx), ptr<int32_t>(del_i), ptr<uint32_t>(f));
  exclusive_scan_u32(ptr<uint32_t>(f), ptr<uint32_t>(f), 3 * n, true);
  const uint32_t *hm = d2h_u32(ptr<uint32_t>(f) + 3 * n);
  const uint32_t *hbad = d2h_u32(bad->p);
  const uint32_t *hbadform = badform ? d2h_u32(badform->p) : nullptr;
  timer.reset();
  sync();
  if (hbadform && *hbadform) throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_collapse: an Update that merges into an earlier row lists some of the primary-key columns and leaves others out "
                                                                "(the merged row's key would mix two items' values)");
  if (*hbad) throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_collapse: NaN / Inf in a primary-key column (json.Marshal fails on the key; the reference files every such row under the empty key)");
  const int64_t m = *hm;
  Buf sel_meta = dalloc((size_t)m * 4 + 4), sel_val = dalloc((size_t)m * 4 + 4), sel_old = dalloc((size_t)m * 4 + 4);
  timer = std::make_unique<KernelTimer>("collapse_select");
  collapse_select<<<cgrid(n), 256, 0, st>>>(n, ptr<uint8_t>(in.kind), ptr<uint32_t>(ak), ptr<int32_t>(rows_m), ptr<int32_t>(rows_v), ptr<int32_t>(del_o), ptr<uint32_t>(f),
                                            ptr<int32_t>(sel_meta), ptr<int32_t>(sel_val), ptr<int32_t>(sel_old));
  timer.reset();
  // ColumnValues from the last writer; Kind / PartID / position from the first item; OldKeys from the first item (or,
  // for a Delete, from the row it removed)
  tfgpu_dbatch vals_in = in;
  vals_in.old_keys.clear(); vals_in.old_present = nullptr; vals_in.kind = nullptr; vals_in.part_id = nullptr; vals_in.src_row = nullptr;
  std::unique_ptr<tfgpu_dbatch> out;
  if (acols.empty() || !m) out = gather_rows(vals_in, sel_val, m);
  else {  // compareColumns: every column that can be absent through its own selection (the last item of the chain that lists it)
    const size_t na = acols.size();
    std::vector<const uint8_t *> abs(na);
    std::vector<int32_t> acol(in.cols.size(), -1);
    for (size_t a = 0; a < na; a++) { abs[a] = ptr<uint8_t>(in.cols[acols[a]].absent); acol[acols[a]] = (int32_t)a; }
    Buf babs = upload_small(abs.data(), na * sizeof(uint8_t *)), bacol = upload_small(acol.data(), acol.size() * 4), border = dalloc_zero(4);
    Buf sel_col = dalloc(na * (size_t)m * 4), first_at = dalloc(na * (size_t)m * 4);
    MergeParams mp{};
    mp.m = m; mp.n = n; mp.sel_meta = ptr<int32_t>(sel_meta); mp.sel_val = ptr<int32_t>(sel_val); mp.prev = ptr<int32_t>(prev);
    mp.absent = ptr<const uint8_t *>(babs); mp.sel_col = ptr<int32_t>(sel_col); mp.first_at = ptr<uint32_t>(first_at);
    {
      KernelTimer tm("collapse_merge");
      collapse_merge_cols<<<dim3(cgrid(m), (unsigned)na), 256, 0, st>>>(mp);
      collapse_merge_order<<<cgrid(m), 256, 0, st>>>(mp, ptr<int32_t>(bacol), (int)acol.size(), ptr<uint32_t>(border));
    }
    const uint32_t *hbo = d2h_u32(border->p);
    sync();
    Buf order;
    if (*hbo) {  // an Update brought a column its row did not list, in front of one it did: the merged names are not in batch order
      if (in.cols.size() > 0xFFFFu) throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_collapse: merged ColumnNames out of batch order over more than 65535 columns");
      order = dalloc((size_t)m * in.cols.size() * 2 + 16);
      collapse_merge_names<<<cgrid(m), 256, 0, st>>>(mp, ptr<int32_t>(bacol), (int)acol.size(), ptr<uint16_t>(order));
    }
    tfgpu_dbatch rest = vals_in;
    rest.cols.clear();
    for (size_t c = 0; c < in.cols.size(); c++) if (acol[c] < 0) rest.cols.push_back(in.cols[c]);
    out = gather_rows(rest, sel_val, m);
    std::vector<DColumn> merged(in.cols.size());
    size_t ri = 0;
    for (size_t c = 0; c < in.cols.size(); c++) {
      if (acol[c] < 0) { merged[c] = std::move(out->cols[ri++]); continue; }
      tfgpu_dbatch one;
      one.nrows = n; one.cols.push_back(in.cols[c]);
      auto g = gather_rows(one, subbuf(sel_col, (size_t)acol[c] * (size_t)m * 4, (size_t)m * 4), m);  // (its absent bitmap travels with it: still absent = no item listed it)
      merged[c] = std::move(g->cols[0]);
    }
    out->cols = std::move(merged);
    out->col_order = order;
    // a merge that filled every cell of a column leaves a bitmap of zeros: dropped, so that a batch which is uniform AFTER the collapse is not
    // taken for a ragged one by the entries that refuse those.  One flag word a column, read back together.
    std::vector<std::pair<size_t, const uint32_t *>> chk;
    for (size_t c = 0; c < out->cols.size(); c++) if (out->cols[c].absent && m > 0) {
      Buf flag = dalloc_zero(4);
      collapse_any_bit<<<cgrid((m + 7) / 8), 256, 0, st>>>(ptr<uint8_t>(out->cols[c].absent), m, ptr<uint32_t>(flag));
      chk.push_back({c, d2h_u32(flag->p, 1)});
    }
    if (!chk.empty()) { sync(); for (auto &k : chk) if (!*k.second) out->cols[k.first].absent = nullptr; }
  }
  tfgpu_dbatch meta_in;
  meta_in.nrows = n; meta_in.kind = in.kind; meta_in.part_id = in.part_id; meta_in.src_row = in.src_row;
  auto meta = gather_rows(meta_in, sel_meta, m);
  out->kind = meta->kind; out->part_id = meta->part_id; out->src_row = meta->src_row;
  if (!in.old_keys.empty()) {
    tfgpu_dbatch old_in;
    old_in.nrows = n; old_in.old_keys = in.old_keys; old_in.old_present = in.old_present;
    auto old = gather_rows(old_in, sel_old, m);
    out->old_keys = std::move(old->old_keys); out->old_present = old->old_present;
  }
  out->key_names = in.key_names; out->schema = in.schema; out->ns = in.ns; out->table = in.table;
  return out;
}

// the flags on the device (n bytes, 1 = the row's primary key changed) and, optionally, their count behind them (a uint32 at flags + align4(n));
// a null Buf when the schema has no PrimaryKey column (nothing can change)
Buf keys_changed_device(const tfgpu_dbatch &in, Buf *count_out) {
  const int64_t n = in.nrows;
  hipStream_t st = ctx().stream;
  materialize(in);
  std::vector<KCol> kc;
  for (auto &name : in.key_names) {  // TableSchema order; a PK column listed twice is compared twice, harmlessly
    KCol k; std::memset(&k, 0, sizeof k);
    if (const DColumn *c = find_first(in.cols, name)) { k.cur = dcol_of(*c); k.has_cur = 1; }
    if (const DColumn *c = find_first(in.old_keys, name)) { k.old = dcol_of(*c); k.has_old = 1; }
    kc.push_back(k);
  }
  if (kc.empty() || n == 0) return nullptr;
  Buf bkc = upload_small(kc.data(), kc.size() * sizeof(KCol));
  Buf flags = dalloc((size_t)n + 16), count = dalloc_zero(4);
  KeysChangedParams p{};
  p.keys = ptr<KCol>(bkc); p.nkeys = (int32_t)kc.size(); p.n = n; p.kind = ptr<uint8_t>(in.kind); p.old_present = ptr<uint8_t>(in.old_present);
  p.has_old_keys = in.old_keys.empty() ? 0 : 1; p.out = ptr<uint8_t>(flags); p.count = ptr<uint32_t>(count);
  { KernelTimer t("keys_changed"); keys_changed_kernel<<<cgrid(n), 256, 0, st>>>(p); }
  if (count_out) *count_out = count;
  return flags;
}
// flags to a host array; returns how many rows changed their primary key
int64_t keys_changed_rows(const tfgpu_dbatch &in, uint8_t *host_flags) {
  const int64_t n = in.nrows;
  if (n == 0) return 0;
  Buf count;
  Buf flags = keys_changed_device(in, &count);
  if (!flags) { std::memset(host_flags, 0, (size_t)n); return 0; }  // no PrimaryKey column: nothing can change
  const uint32_t *h = d2h_u32(count->p);
  d2h(host_flags, flags->p, (size_t)n);
  sync();
  return *h;
}

}  // namespace tf

extern "C" int tfgpu_keys_changed(const tfgpu_dbatch *in, uint8_t *changed, int64_t *nchanged) {
  try {
  tf::dense(in, true);  // its rows may still be a selection (tfgpu_dbatch::pending); ABSENT cells are read here
    if (!in || !changed) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_keys_changed: null argument");
    std::lock_guard<std::mutex> lk(tf::ctx().mu);
    const int64_t c = tf::keys_changed_rows(*in, changed);
    if (nchanged) *nchanged = c;
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}

extern "C" int tfgpu_collapse(const tfgpu_dbatch *in, tfgpu_dbatch **out) {
  try {
  tf::dense(in, true);  // its rows may still be a selection (tfgpu_dbatch::pending); ABSENT cells are read here
    if (!in || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_collapse: null argument");
    std::lock_guard<std::mutex> lk(tf::ctx().mu);
    *out = tf::collapse_rows(*in).release();
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}
