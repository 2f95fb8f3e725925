// tf_srjson.hip — Confluent Schema Registry ingest, JSON schemas (the source format of configs[2]):
//
//   ConfluentSrImpl.DoBatch / Do / DoBuf / DoOne   pkg/parsers/registry/confluentschemaregistry/engine/parser.go:108-152
//   makeChangeItemsFromMessageWithJSON             engine/format_json.go:15-67
//   processPayload / convertTypes                  engine/utils_json.go:27-128, types_json.go:25-32
//   jsonx.NewDefaultDecoder(...).Decode(&map)      encoding/json Decoder + UseNumber (scanner.go, stream.go, decode.go)
//
// Messages in, columns out, in lane-per-item passes over HBM-resident bytes:
//   sr_count_frames / sr_fill_frames   one lane per Kafka message walks its frames (0x00 | BE schema id | payload up to the
//                                      next 0x00) — DoBuf's loop; a short tail or a wrong magic byte ends the message
//   sr_parse_frames                    one lane per frame: encoding/json's grammar over the payload with an explicit
//                                      container stack, top-level keys matched against the sorted property names (last
//                                      duplicate wins), then processPayload's per-property rules → the frame's fate
//   sr_message_rule                    one lane per message: the first `_unparsed` frame drops the rest of its message
//   sr_cell_values                     one lane per (property, row): bool / int64 values, text lengths
//   sr_cell_text                       one lane per (text column, row): unquoted strings (escapes, surrogate pairs,
//                                      invalid UTF-8 → U+FFFD), json.Number text, json.Marshal text of `any` values
// First form of the path: correct and HBM-resident, latency-bound (every lane walks its own payload), like the per-line
// generic JSON parser before its tile form.  Algorithmic bytes: B_json in + B_bin out per frame.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "tf_jsonscan.hpp"
#include "tf_segcopy.hpp"
#include "tf_jsontile.hpp"
#include "tf_jsonquick.hpp"

namespace tf {
namespace sr {


struct Prop { uint32_t name_off, name_len; int32_t json_type, required; };
struct Params {
  const uint8_t *data;
  const uint32_t *ms; int64_t nmsg;           // message starts [nmsg + 1]
  uint32_t *fcount;                            // [nmsg + 1] frames per message → first frame of each message
  tfgpu_sr_frame *frames; int64_t nframes;
  uint32_t schema_id; int32_t report_frame_errors;
  const Prop *props; int32_t nprops; const uint8_t *names;
  uint64_t *vrec;                              // [nprops][nframes]: where the property's value sits in the payload, one word per
                                               // cell — start (32) | length (24) | VT_* (8): one store / one load instead of three
  uint8_t *status;                             // [nframes]
  uint32_t *keep;                              // [nframes + 1] → row index
  uint32_t *row_frame; int64_t nrows;          // [nrows]
  uint32_t *nerr;
  uint16_t *guess;                             // [GUESS_N]: property the m-th member of a payload matched last time (a hint, verified)
  const uint8_t *nozero;                       // [nmsg] or null: 1 = no 0x00 byte after the 5-byte prefix (sr_find_zero): ONE frame to the end
  const int64_t *ival;                         // [nprops][nframes] or null: values of `integer` cells whose span word carries VT_IVAL (sr_parse_quick)
};
constexpr uint32_t GUESS_N = 1024;


// ---- frames: DoBuf / DoOne + the payload end of format_json.go:34-38 --------------------------------------------------
// Calls emit(frame) for every frame of message m; returns their number.
template <class F> __device__ uint32_t walk_frames(const Params &p, int64_t m, F emit) {
  MemBytes rd(p.data);
  uint64_t a = p.ms[m]; const uint64_t z = p.ms[m + 1];
  uint32_t k = 0;
  while (a < z) {
    tfgpu_sr_frame f; f.msg = m; f.index = (int32_t)k; f.schema_id = 0; f.start = a; f.len = (uint32_t)(z - a);
    if (z - a < 5) { f.code = TFGPU_ROW_SR_SHORT; emit(k, f); return k + 1; }
    if (rd.at(a) != 0) { f.code = TFGPU_ROW_SR_MAGIC; emit(k, f); return k + 1; }
    f.schema_id = (rd.at(a + 1) << 24) | (rd.at(a + 2) << 16) | (rd.at(a + 3) << 8) | rd.at(a + 4);
    a += 5;
    uint64_t e = a;  // bytes.Index(buf, []byte{0}): eight bytes per step (the lowest zero-byte flag of the SWAR test is exact)
    bool hit = false;
    if (k == 0 && p.nozero && p.nozero[m]) { e = z; hit = true; }  // the cooperative pre-pass saw no terminator: the payload runs to the end
    while (!hit && e + 8 <= z) {
      const uint64_t w = rd.word(e), t = (w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull;
      if (t) { e += (uint32_t)(__ffsll((long long)t) - 1) >> 3; hit = true; break; }
      e += 8;
    }
    if (!hit) while (e < z && rd.at(e) != 0) e++;
    f.start = a; f.len = (uint32_t)(e - a); f.code = TFGPU_ROW_OK;
    emit(k, f);
    k++; a = e;
  }
  return k;
}
// bytes.Index(payload, 0x00) for the usual message — ONE frame that runs to the end — without a lane walking 2 KB serially:
// a wave per message, 64 lanes x 8 bytes per step (a memchr at HBM speed); messages that do hold a 0x00 take the walk.
__global__ void __launch_bounds__(256) sr_find_zero(Params p, uint8_t *nozero) {
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.nmsg) return;
  const int lane = threadIdx.x & 63;
  const uint64_t a = (uint64_t)p.ms[m] + 5, z = p.ms[m + 1];
  bool zero = false;
  if (z > a) {
    const uint64_t n = z - a, nw = n >> 3;
    for (uint64_t w = lane; w < nw; w += 64) {
      const uint64_t pos = a + 8 * w; const uint32_t sh = (uint32_t)(pos & 7) * 8;
      const uint64_t *q = reinterpret_cast<const uint64_t *>(p.data + (pos & ~7ull));
      const uint64_t x = sh ? (q[0] >> sh) | (q[1] << (64 - sh)) : q[0];
      zero |= ((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull) != 0;
    }
    if (lane < (int)(n & 7)) zero |= p.data[a + 8 * nw + lane] == 0;
  }
  const bool any = __any(zero);
  if (lane == 0) nozero[m] = (z >= a && !any) ? 1 : 0;
}
// The count pass keeps every message's FIRST frame: the usual message holds exactly one, and the fill pass then copies it
// instead of walking the bytes a second time.
__global__ void __launch_bounds__(256) sr_count_frames(Params p, tfgpu_sr_frame *first) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.nmsg) return;
  tfgpu_sr_frame *slot = first + m;
  p.fcount[m] = walk_frames(p, m, [slot](uint32_t k, const tfgpu_sr_frame &f) { if (k == 0) *slot = f; });
}
__global__ void __launch_bounds__(256) sr_fill_frames(Params p, const tfgpu_sr_frame *first) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.nmsg) return;
  const uint32_t f0 = p.fcount[m], n = p.fcount[m + 1] - f0;
  if (n == 0) return;
  tfgpu_sr_frame *out = p.frames + f0;
  if (n == 1) { out[0] = first[m]; return; }
  walk_frames(p, m, [out](uint32_t k, const tfgpu_sr_frame &f) { out[k] = f; });
}

// ---- one frame: Decode into a map, then processPayload's rules -----------------------------------------------------------
__device__ uint32_t parse_frame(const Params &p, const int64_t f) {
  const tfgpu_sr_frame &fr = p.frames[f];
  MemBytes rd(p.data);
  uint32_t pos = (uint32_t)fr.start; const uint32_t end = pos + fr.len;
  auto skip_ws = [&]() { while (pos < end && is_ws(rd.at(pos))) pos++; };
  skip_ws();
  if (pos >= end) return TFGPU_ROW_JSON_SYNTAX;  // io.EOF
  bool nil_map = false;
  if (rd.at(pos) == 'n') {  // a top-level null leaves the map nil; a scalar must be followed by white space or the end
    uint32_t t;
    if (!scan_literal(rd, pos, end, t) || (pos < end && !is_ws(rd.at(pos)))) return TFGPU_ROW_JSON_SYNTAX;
    nil_map = true;
  } else if (rd.at(pos) != '{') return TFGPU_ROW_JSON_SYNTAX;  // syntax error or UnmarshalTypeError: not a map either way
  if (!nil_map) {
    pos++;
    skip_ws();
    if (pos >= end) return TFGPU_ROW_JSON_SYNTAX;
    uint32_t member = 0;
    if (rd.at(pos) == '}') pos++;
    else for (;;) {
      skip_ws();
      if (pos >= end || rd.at(pos) != '"') return TFGPU_ROW_JSON_SYNTAX;
      const uint32_t ks = pos;
      bool kplain = false;
      if (!scan_string(rd, pos, end, &kplain)) return TFGPU_ROW_JSON_SYNTAX;
      const uint32_t ke = pos;
      skip_ws();
      if (pos >= end || rd.at(pos) != ':') return TFGPU_ROW_JSON_SYNTAX;
      pos++;
      skip_ws();
      const uint32_t vs = pos;
      uint32_t vt = 0;
      const int rc = skip_value(rd, pos, end, vt);
      if (rc == 1) return TFGPU_ROW_JSON_SYNTAX;
      if (rc == 2) return TFGPU_ROW_HOST_FALLBACK;
      // m[key] = value: the last duplicate wins.  Producers write their keys in one order, so the property the m-th
      // member matched in some other payload is tried first; otherwise binary search over the sorted property names.
      // A plain key (ASCII, no escapes) is its own decoding and compares as raw bytes; any other key by decoded runes.
      auto compare_to = [&](int j) -> int {
        const Prop &pr = p.props[j];
        if (!kplain) { RuneIter a{&rd, nullptr, ks + 1, ke - 1}, b{nullptr, p.names + pr.name_off, 0, pr.name_len}; return rune_compare(a, b); }
        const uint32_t kn = ke - ks - 2, m = kn < pr.name_len ? kn : pr.name_len;
        const uint8_t *nm = p.names + pr.name_off;  // 8-byte aligned, zero-padded to whole words
        for (uint32_t q = 0; q < m; q += 8) {  // eight bytes per compare; the first differing byte decides the order
          const uint32_t nb = m - q < 8 ? m - q : 8u;
          const uint64_t mask = nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1;
          const uint64_t x = rd.word(ks + 1 + q) & mask, y = *reinterpret_cast<const uint64_t *>(nm + q) & mask;
          if (x != y) { const uint32_t sh = ((uint32_t)(__ffsll((long long)(x ^ y)) - 1) >> 3) * 8; return ((x >> sh) & 0xFF) < ((y >> sh) & 0xFF) ? -1 : 1; }
        }
        return kn == pr.name_len ? 0 : kn < pr.name_len ? -1 : 1;
      };
      int found = -1;
      const uint32_t g = member < GUESS_N ? p.guess[member] : 0xFFFFu;
      if (g < (uint32_t)p.nprops && compare_to((int)g) == 0) found = (int)g;
      else {
        int lo = 0, hi = p.nprops - 1;
        while (lo <= hi) {
          const int mid = (lo + hi) >> 1;
          const int c = compare_to(mid);
          if (c == 0) { found = mid; break; }
          if (c < 0) hi = mid - 1; else lo = mid + 1;
        }
        if (found >= 0 && member < GUESS_N) p.guess[member] = (uint16_t)found;  // racing writers all store a valid hint
      }
      if (pos - vs >= (1u << 24)) return TFGPU_ROW_HOST_FALLBACK;  // a 16 MiB value does not fit the span word
      if (found >= 0) p.vrec[(int64_t)found * p.nframes + f] = (uint64_t)vs | (uint64_t)(pos - vs) << 32 | (uint64_t)(vt & 0xFFu) << 56;
      member++;
      skip_ws();
      if (pos >= end) return TFGPU_ROW_JSON_SYNTAX;
      const uint32_t d = rd.at(pos);
      if (d == ',') { pos++; continue; }
      if (d == '}') { pos++; break; }
      return TFGPU_ROW_JSON_SYNTAX;
    }
  }
  // processPayload (utils_json.go:45-66), properties in name order: the first failing one names the error
  bool fallback = false;
  for (int j = 0; j < p.nprops; j++) {
    const int64_t i = (int64_t)j * p.nframes + f;
    const uint64_t rec = p.vrec[i];
    const uint32_t vt = (uint32_t)(rec >> 56) & VT_MASK;
    const Prop &pr = p.props[j];
    if (vt == VT_ABSENT) { if (pr.required) return TFGPU_ROW_SR_REQUIRED; continue; }
    if (vt == VT_NULL && !pr.required) continue;
    switch (pr.json_type) {
      case TFGPU_SRT_BOOLEAN: if (vt != VT_TRUE && vt != VT_FALSE) return TFGPU_ROW_SR_TYPE; break;
      case TFGPU_SRT_INTEGER: { int64_t x; if (vt != VT_NUM || !number_int64(rd, (uint32_t)rec, (uint32_t)(rec >> 32) & 0xFFFFFFu, &x)) return TFGPU_ROW_SR_TYPE; break; }
      case TFGPU_SRT_NUMBER: if (vt != VT_NUM) return TFGPU_ROW_SR_TYPE; break;
      case TFGPU_SRT_STRING: if (vt != VT_STR) return TFGPU_ROW_SR_TYPE; break;
      default:
        if (vt == VT_OBJ || vt == VT_ARR) {
          const int ord = any_keys_order(rd, (uint32_t)rec, (uint32_t)(rec >> 32) & 0xFFFFFFu);
          if (ord == 1) p.vrec[i] = rec | (uint64_t)VT_CANON << 56;
          else if (ord == 2) fallback = true;
        }
    }
  }
  return fallback ? (uint32_t)TFGPU_ROW_HOST_FALLBACK : (uint32_t)ST_OK;
}
// ---------------------------------------------------------------------------
// tile path (tf_jsontile.hpp): consecutive payloads staged in LDS, classified and indexed once; a cell is a (member, payload)
// pair and only has to say where the member's value lies and what it is — the span word the value kernels read — and
// whether the property takes it.  The member → property map is found once (binary search over the sorted names) and kept
// while the payloads keep spelling the same keys.  Compact or blank-separated flat objects whose values are strings without
// escapes, literals or numbers are decided here; every other payload goes to parse_frame through sr_parse_listed.
// ---------------------------------------------------------------------------
// encoding/json's number: -? (0 | [1-9][0-9]*) (. [0-9]+)? ([eE] [+-]? [0-9]+)?
__device__ __forceinline__ bool st_json_number(const uint8_t *sb, uint32_t a, uint32_t b) {
  uint32_t i = a;
  auto dg = [&](uint32_t k) { return k < b && sb[k] >= '0' && sb[k] <= '9'; };
  if (i < b && sb[i] == '-') i++;
  if (!dg(i)) return false;
  if (sb[i] == '0') i++; else while (dg(i)) i++;
  if (i < b && sb[i] == '.') { i++; if (!dg(i)) return false; while (dg(i)) i++; }
  if (i < b && (sb[i] == 'e' || sb[i] == 'E')) { i++; if (i < b && (sb[i] == '+' || sb[i] == '-')) i++; if (!dg(i)) return false; while (dg(i)) i++; }
  return i == b;
}
// bytes [a, b): bit 0 = some byte >= 0x80, bit 1 = some byte < 0x20
__device__ __forceinline__ uint32_t st_byte_classes(const uint8_t *sb, uint32_t a, uint32_t b) {
  const uint64_t ONES = 0x0101010101010101ull, HI = 0x8080808080808080ull;
  uint32_t f = 0;
  for (uint32_t k = a; k < b; k += 8) {
    const uint32_t nb = b - k < 8 ? b - k : 8u;
    uint64_t w = jt_word(sb, k);
    if (nb < 8) { const uint64_t m = (1ull << (8 * nb)) - 1; w = (w & m) | (ONES * 0x20 & ~m); }  // the bytes past b read as blanks
    if (w & HI) f |= 1u;
    if ((w - ONES * 0x20) & ~w & HI) f |= 2u;
  }
  return f;
}
// the escapes of a string body [a, b) as encoding/json's scanner takes them: \" \\ \/ \b \f \n \r \t \uXXXX
__device__ __forceinline__ bool st_escapes_ok(const uint8_t *sb, uint32_t a, uint32_t b) {
  for (uint32_t i = a; i < b; i++) {
    if (sb[i] != '\\') continue;
    if (i + 1 >= b) return false;
    const uint32_t d = sb[i + 1];
    i++;
    if (d == 'u') {
      if (b - (i + 1) < 4) return false;
      for (uint32_t k = 1; k <= 4; k++) { const uint32_t h = sb[i + k]; if (!((h >= '0' && h <= '9') || ((h | 0x20u) >= 'a' && (h | 0x20u) <= 'f'))) return false; }
      i += 4;
    } else if (!(d == '"' || d == '\\' || d == '/' || d == 'b' || d == 'f' || d == 'n' || d == 'r' || d == 't')) return false;
  }
  return true;
}
// raw key bytes against a property name (8-byte aligned, zero-padded): <0, 0, >0 as bytes.Compare
__device__ __forceinline__ int st_key_compare(const uint8_t *sb, uint32_t ks, uint32_t kn, const uint8_t *nm, uint32_t nlen) {
  const uint32_t m = kn < nlen ? kn : nlen;
  for (uint32_t q = 0; q < m; q += 8) {
    const uint32_t nb = m - q < 8 ? m - q : 8u;
    const uint64_t mask = nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1;
    const uint64_t x = jt_word(sb, ks + q) & mask, y = *reinterpret_cast<const uint64_t *>(nm + q) & mask;
    if (x != y) { const uint32_t sh = ((uint32_t)(__ffsll((long long)(x ^ y)) - 1) >> 3) * 8; return ((x >> sh) & 0xFF) < ((y >> sh) & 0xFF) ? -1 : 1; }
  }
  return kn == nlen ? 0 : kn < nlen ? -1 : 1;
}

__global__ void __launch_bounds__(JT_THREADS, 4) sr_parse_tiles(Params p, int32_t frames_per_tile, uint32_t *slow_n, uint32_t *slow_f, int ablate) {
  __shared__ JtLds L;
  uint8_t *const sb = L.sbuf + 16;
  uint16_t *const spos = L.spos, *const lstart = L.lstart, *const lend = L.lend, *const lbase = L.lbase, *const lK = L.lK;
  uint32_t *const misc = L.misc;
  uint8_t *const lslow = L.lslow;
  __shared__ uint32_t lerr[JT_LINES];  // the first failing property of the payload in name order: prop << 8 | code
  __shared__ int16_t mprop[JT_MEM];
  __shared__ uint16_t mks[JT_MEM], mkn[JT_MEM], mko[JT_MEM];
  __shared__ uint8_t mtype[JT_MEM], mreq[JT_MEM], perm[JT_MEM];
  __shared__ uint32_t cfirst[3];  // members whose property is a string come last: a wave's cells are all strings or all scalars
  __shared__ uint16_t owner[JT_OWN];
  __shared__ __attribute__((aligned(16))) uint8_t kref[JT_KREF + 32];
  __shared__ uint32_t mapst[4];  // 0: members, 1: valid, 2: payloads of this tile that spell other keys, 3: first required property no member gives
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < 4) mapst[tid] = 0;
  const int64_t ntiles = (p.nframes + frames_per_tile - 1) / frames_per_tile;
  // a workgroup takes a contiguous run of tiles: the short column segments two neighbouring tiles write share cache lines,
  // and written one after the other by the same CU they leave its L2 as whole lines
  const int64_t tiles_each = (ntiles + gridDim.x - 1) / gridDim.x;
  for (int64_t tile = (int64_t)blockIdx.x * tiles_each; tile < min(ntiles, ((int64_t)blockIdx.x + 1) * tiles_each); tile++) {
  __syncthreads();
  const int64_t f0 = tile * frames_per_tile;
  int nl = (int)min<int64_t>(frames_per_tile, p.nframes - f0);
  auto hand_over = [&](int64_t f) { const uint32_t k = atomicAdd(slow_n, 1u); slow_f[k] = (uint32_t)f; };
  uint32_t first, last, g0;
  bool ours_here = false;
  {
    uint32_t ln = 0, ls_ = 0;
    if (lane < nl) {
      const tfgpu_sr_frame fr = p.frames[f0 + lane];
      ours_here = fr.code == 0 && fr.schema_id == p.schema_id;
      if (!ours_here) { if (wv == 0) p.status[f0 + lane] = fr.code ? (uint8_t)fr.code : (uint8_t)ST_OTHER; }
      else if (fr.len == 0 || fr.start + fr.len > 0xFFFFFFFFull) { if (wv == 0) hand_over(f0 + lane); ours_here = false; }
      else { ln = fr.len; ls_ = (uint32_t)fr.start; }
    }
    const uint64_t nonempty = __ballot(ln != 0);
    if (!nonempty) continue;
    first = (uint32_t)__shfl((int)ls_, __ffsll((long long)nonempty) - 1, 64);
    g0 = first & ~15u;
    const uint64_t fits = __ballot(ln != 0 && ls_ + ln - g0 <= (uint32_t)JT_BYTES);
    if (!fits || p.nprops > JT_OWN) { if (wv == 0 && ln != 0) hand_over(f0 + lane); continue; }
    const int hi = 63 - __clzll((long long)fits);
    last = (uint32_t)__shfl((int)(ls_ + ln), hi, 64);
    if (wv == 0 && lane > hi && ln != 0) hand_over(f0 + lane);
    nl = hi + 1;
    if (wv == 0 && lane < nl) {
      lstart[lane] = (uint16_t)(ln ? ls_ - g0 : 0); lend[lane] = (uint16_t)(ln ? ls_ - g0 + ln : 0);
      lslow[lane] = ln ? 0 : 2; lbase[lane] = 0xFFFFu; lerr[lane] = 0xFFFFFFFFu;
    }
  }
  auto all_slow = [&]() { if (tid < nl && lslow[tid] != 2) hand_over(f0 + tid); };
  for (int i = tid; i < JT_OWN; i += JT_THREADS) owner[i] = 0xFFFFu;
  if (!jt_front(L, p.data, first, last, g0, nl, ablate)) { all_slow(); continue; }
  if (ablate == 4) { all_slow(); continue; }
  JtTile t;
  t.sb = sb; t.spos = spos; t.qmask = L.qmask; t.qpre = L.qpre; t.bpre = L.bpre; t.g0 = g0;
  int jref = 0;
  while (jref < nl && lslow[jref] != 0) jref++;  // the first line that is framed: the one whose keys are looked up when there is no map
  const bool build_map = mapst[1] == 0;  // uniform
  if (build_map && (jref >= nl || lslow[jref])) { all_slow(); continue; }
  const uint32_t bref = build_map ? lbase[jref] : 0u;
  const uint32_t K = build_map ? lK[jref] : mapst[0];
  // ---- member map: which property reads the m-th member (binary search over the sorted names), the key text kept ----
  if (build_map) {
    uint32_t mbad = 0;
    for (uint32_t m = (uint32_t)tid; m < K; m += JT_THREADS) {
      const uint32_t pp = spos[bref + 2 * m], pc = spos[bref + 1 + 2 * m];
      uint32_t kq = pp + 1, ke = pc;
      while (kq < pc && jt_ws(sb[kq])) kq++;
      while (ke > kq && jt_ws(sb[ke - 1])) ke--;
      // a key without escapes: plain ASCII is its own decoding; with other bytes it is decoded rune by rune (invalid UTF-8
      // becomes U+FFFD), so only byte equality with a property name decides anything — no match: parse_frame
      const bool framed = ke >= kq + 2 && sb[kq] == '"' && sb[ke - 1] == '"' && sb[pc] == ':' && jt_quotes_in(t, pp + 1, pc) == 2 && jt_backslashes_in(L, kq, ke) == 0;
      const uint32_t ks = kq + 1, kn = framed ? ke - 1 - ks : 0;
      const uint32_t kcl = framed ? st_byte_classes(sb, ks, ks + kn) : 2u;
      int found = -1;
      if (!framed || (kcl & 2u)) mbad = 1;
      else {
        int lo = 0, hi = p.nprops - 1;
        while (lo <= hi) {
          const int mid = (lo + hi) >> 1;
          const Prop pr = p.props[mid];
          const int c = st_key_compare(sb, ks, kn, p.names + pr.name_off, pr.name_len);
          if (c == 0) { found = mid; break; }
          if (c < 0) hi = mid - 1; else lo = mid + 1;
        }
        if (found < 0 && (kcl & 1u)) mbad = 1;
      }
      mprop[m] = (int16_t)found; mks[m] = (uint16_t)ks; mkn[m] = (uint16_t)kn;
      mtype[m] = found >= 0 ? (uint8_t)p.props[found].json_type : 0; mreq[m] = found >= 0 ? (uint8_t)(p.props[found].required != 0) : 0;
    }
    if (mbad) misc[0] = 1u;
  }
  __syncthreads();
  if (build_map) {
    for (uint32_t m = (uint32_t)tid; m < K; m += JT_THREADS) if (mprop[m] >= 0) owner[mprop[m]] = (uint16_t)m;
    if (tid == 0) {
      uint32_t o = 0;
      for (uint32_t m = 0; m < K; m++) { mko[m] = (uint16_t)o; o += ((uint32_t)mkn[m] + 7u) & ~7u; if (o > (uint32_t)JT_KREF) { misc[0] = 1u; break; } }
    }
  }
  __syncthreads();
  if (build_map && !misc[0]) {
    for (uint32_t m = (uint32_t)tid; m < K; m += JT_THREADS) {
      if (mprop[m] >= 0 && owner[mprop[m]] != (uint16_t)m) misc[0] = 1u;  // a key read twice: the last one wins → parse_frame
      for (uint32_t k = 0; k < mkn[m]; k += 8) *reinterpret_cast<uint64_t *>(kref + mko[m] + k) = jt_word(sb, (uint32_t)mks[m] + k);
    }
    if (wv == 0) {  // string properties behind the others (their cells walk the string's bytes: long loops that scalars should not wait for)
      uint32_t nxt = 0;
      for (uint32_t c = 0; c < 2; c++) {
        if (lane == 0) cfirst[c] = nxt;
        for (uint32_t r0 = 0; r0 < K; r0 += 64) {
          const uint32_t m = r0 + (uint32_t)lane;
          const bool mine = m < K && (uint32_t)(mtype[m] == TFGPU_SRT_STRING && mprop[m] >= 0) == c;
          const uint64_t bal = __ballot(mine);
          if (mine) perm[nxt + lanes_below(bal)] = (uint8_t)m;
          nxt += (uint32_t)__popcll(bal);
        }
      }
      if (lane == 0) cfirst[2] = nxt;
    }
    if (tid == 0) {  // processPayload walks the properties in name order: the first required one no member gives
      uint32_t miss = 0xFFFFFFFFu;
      for (int j = 0; j < p.nprops; j++) if (p.props[j].required && owner[j] == 0xFFFFu) { miss = ((uint32_t)j << 8) | (uint32_t)TFGPU_ROW_SR_REQUIRED; break; }
      mapst[3] = miss;
    }
  }
  __syncthreads();
  if (misc[0]) { all_slow(); continue; }  // uniform
  if (build_map && tid == 0) { mapst[0] = K; mapst[1] = 1u; }
  if (tid == 0) mapst[2] = 0;
  __syncthreads();

  if (ablate == 5) { all_slow(); continue; }
  // ---- cells ----
  for (uint32_t cls = 0; cls < 2; cls++) {
    const uint32_t unl = (uint32_t)nl, f0c = cfirst[cls], items = (cfirst[cls + 1] - f0c) * unl;
    const float inv = __uint_as_float(__float_as_uint(1.0f / (float)unl) - 2u);  // it / unl by a reciprocal a hair too small + one correction
    for (uint32_t it = (uint32_t)tid; it < items; it += JT_THREADS) {
      uint32_t oi = (uint32_t)(__uint2float_rz(it) * inv);
      uint32_t j = it - oi * unl;
      if (j >= unl) { oi++; j -= unl; }
      const uint32_t m = perm[f0c + oi];
      if (lslow[j]) continue;
      if (lK[j] != K) { if (lslow[j] == 0) { lslow[j] = 1; atomicAdd(&mapst[2], 1u); } continue; }
      const uint32_t b = lbase[j];
      const uint32_t pp = spos[b + 2 * m], pc = spos[b + 1 + 2 * m], pn = spos[b + 2 + 2 * m];
      const uint32_t kn = mkn[m];
      uint32_t kq = pp + 1, ke = pc;
      while (kq < pc && jt_ws(sb[kq])) kq++;
      while (ke > kq && jt_ws(sb[ke - 1])) ke--;
      bool ok = ke == kq + 2 + kn && sb[pc] == ':' && sb[kq] == '"' && sb[ke - 1] == '"' && sb[pn] == (m + 1 == K ? '}' : ',');
      if (ok && !jt_same2(sb, kq + 1, kref, mko[m], kn)) { ok = false; if (lslow[j] == 0) atomicAdd(&mapst[2], 1u); }
      uint32_t vs = pc + 1, ve = pn;
      while (vs < pn && jt_ws(sb[vs])) vs++;
      while (ve > vs && jt_ws(sb[ve - 1])) ve--;
      const uint32_t n = ve - vs;
      ok = ok && n > 0;
      uint32_t vt = VT_ABSENT;
      bool isint = false, inrange = false;
      if (ok) {
        const uint32_t c0 = sb[vs];
        if (c0 == '"') {
          ok = n >= 2 && sb[ve - 1] == '"' && jt_quotes_in(t, vs, ve) == 2;
          const uint32_t cl = ok ? st_byte_classes(sb, vs + 1, ve - 1) : 0u;
          if (cl & 2u) ok = false;  // a control byte inside a string literal is a syntax error
          const bool esc = ok && jt_backslashes_in(L, vs, ve) != 0;
          if (esc) ok = st_escapes_ok(sb, vs + 1, ve - 1);  // (decoded by the value kernels, as for every string that is not plain)
          vt = VT_STR | (((cl & 1u) || esc) ? 0u : (uint32_t)VT_PLAIN);
        } else if (jt_lit(sb, vs, n, 0x6C6C756Eu, 4)) vt = VT_NULL;
        else if (jt_lit(sb, vs, n, 0x65757274u, 4)) vt = VT_TRUE;
        else if (jt_lit(sb, vs, n, 0x736C6166u, 5)) vt = VT_FALSE;
        else {
          bool ng; uint64_t mag; uint32_t nd;
          if (jt_quotes_in(t, vs, ve) != 0) ok = false;
          else if (jt_int_token(sb, vs, ve, &ng, &mag, &nd)) {
            if (nd > 1 && sb[ve - nd] == '0') ok = false;  // a leading zero: "invalid character after top-level value"
            isint = true; inrange = mag <= (ng ? (1ull << 63) : (1ull << 63) - 1);
            vt = VT_NUM;
          } else { ok = st_json_number(sb, vs, ve); vt = VT_NUM; }
        }
      }
      if (!ok) { lslow[j] = 1; continue; }
      const int32_t pj = mprop[m];
      if (pj < 0) continue;  // no property of the schema: json.Unmarshal keeps it, processPayload never asks
      p.vrec[(int64_t)pj * p.nframes + (f0 + j)] = (uint64_t)(g0 + vs) | (uint64_t)n << 32 | (uint64_t)(vt & 0xFFu) << 56;
      const uint32_t base = vt & VT_MASK;
      if (base == VT_NULL && !mreq[m]) continue;
      bool good = true;
      switch (mtype[m]) {
        case TFGPU_SRT_BOOLEAN: good = base == VT_TRUE || base == VT_FALSE; break;
        case TFGPU_SRT_INTEGER: good = base == VT_NUM && isint && inrange; break;
        case TFGPU_SRT_NUMBER: good = base == VT_NUM; break;
        case TFGPU_SRT_STRING: good = base == VT_STR; break;
        default: break;
      }
      if (!good) atomicMin(&lerr[j], ((uint32_t)pj << 8) | (uint32_t)TFGPU_ROW_SR_TYPE);
    }
  }
  __syncthreads();
  if (tid < nl && lslow[tid] != 2) {
    if (lslow[tid]) hand_over(f0 + tid);
    else {
      const uint32_t e = min(lerr[tid], mapst[3]);
      p.status[f0 + tid] = e == 0xFFFFFFFFu ? (uint8_t)ST_OK : (uint8_t)(e & 0xFFu);
    }
  }
  if (tid == 0 && mapst[2] * 2 > (uint32_t)nl) mapst[1] = 0;
  }  // tiles
}
#include "tf_srquick.inc"

// the payloads the tile path hands over
__global__ void __launch_bounds__(128) sr_parse_listed(Params p, const uint32_t *slow_n, const uint32_t *slow_f) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= *slow_n) return;
  const int64_t f = slow_f[k];
  p.status[f] = (uint8_t)parse_frame(p, f);
}

__global__ void __launch_bounds__(128) sr_parse_frames(Params p) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= p.nframes) return;
  const tfgpu_sr_frame &fr = p.frames[f];
  uint32_t st;
  if (fr.code) st = (uint32_t)fr.code;
  else if (fr.schema_id != p.schema_id) st = ST_OTHER;
  else st = parse_frame(p, f);
  p.status[f] = (uint8_t)st;
}
// DoBuf: an `_unparsed` item ends its message (doWithSchema returns nil); frames of other schema ids are not ours to judge
__global__ void __launch_bounds__(256) sr_message_rule(Params p) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.nmsg) return;
  bool dead = false;
  for (uint32_t f = p.fcount[m]; f < p.fcount[m + 1]; f++) {
    uint32_t st = p.status[f];
    if (dead) st = ST_DROPPED;
    else if (st != ST_OK && st != ST_OTHER) {
      dead = true;
      const bool frame_err = st == TFGPU_ROW_SR_SHORT || st == TFGPU_ROW_SR_MAGIC;
      if (frame_err && !p.report_frame_errors) st = ST_DROPPED; else atomicAdd(p.nerr, 1u);
    }
    p.status[f] = (uint8_t)st;
    p.keep[f] = st == ST_OK ? 1u : 0u;
  }
}
__global__ void __launch_bounds__(256) sr_row_frames(Params p) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= p.nframes || p.status[f] != ST_OK) return;
  p.row_frame[p.keep[f]] = (uint32_t)f;
}

// ---- cells ------------------------------------------------------------------------------------------------------------
struct OutCol {
  int32_t json_type;
  void *values;        // bool (u8) / int64
  uint32_t *lens;      // text columns: lengths, then offsets [nrows + 1]
  uint8_t *data;       // text payload (second pass)
  uint8_t *valid8;     // one byte per row → packed by sr_pack_validity
  uint8_t *validity;   // bitmap
};
// CANON == false: every cell but the `any` values that need the sorting emitter; CANON == true: only those (a second launch
// when the schema has `any` properties) — the kernel every cell runs through does not carry the emitter's frame stack.
template <bool CANON>
// (grid: x = 256-row blocks, y = property: the property is a scalar — its descriptor comes by scalar loads, the switch on its JSON type is a scalar branch)
__device__ __forceinline__ void sr_cell_value_one(const Params &p, const OutCol *cols, int32_t *src_row, uint32_t *part_id, const int j, const int64_t r) {
  if (r >= p.nrows) return;
  const uint32_t f = p.row_frame[r];
  const OutCol &c = cols[j];
  const int64_t i = (int64_t)j * p.nframes + f;
  const uint64_t rec = p.vrec[i];
  const uint32_t vtr = (uint32_t)(rec >> 56), vt = vtr & VT_MASK, vs = (uint32_t)rec, vl = (uint32_t)(rec >> 32) & 0xFFFFFFu;
  MemBytes rd(p.data);
  if constexpr (CANON) {
    if (!(vtr & VT_CANON)) return;
    CountSink s;
    emit_any_canon(s, rd, vs, vl);
    c.lens[r] = s.n;
    return;
  }
  if (j == 0) { src_row[r] = (int32_t)f; part_id[r] = (uint32_t)p.frames[f].msg; }
  const bool nil = vt == VT_ABSENT || vt == VT_NULL;
  c.valid8[r] = nil ? 0 : 1;
  switch (c.json_type) {
    case TFGPU_SRT_BOOLEAN: ((uint8_t *)c.values)[r] = vt == VT_TRUE ? 1 : 0; break;
    case TFGPU_SRT_INTEGER: {
      int64_t x = 0;
      if (!nil) { if ((vtr & 0x80u) && p.ival) x = p.ival[i]; else number_int64(rd, vs, vl, &x); }  // VT_IVAL: converted while the payload was in LDS
      ((int64_t *)c.values)[r] = x;
      break;
    }
    case TFGPU_SRT_NUMBER: c.lens[r] = nil ? 0u : vl; break;
    case TFGPU_SRT_STRING:
      if (nil) c.lens[r] = 0;
      else if (vtr & VT_PLAIN) c.lens[r] = vl - 2;  // nothing to decode
      else { CountSink s; emit_unquoted(s, rd, vs, vl); c.lens[r] = s.n; }
      break;
    default: if (!(vtr & VT_CANON)) { CountSink s; if (!nil) emit_any(s, rd, vs, vl); c.lens[r] = s.n; }  // flagged cells: the CANON launch
  }
}
// TF_SR_VALUES_RPT rows per lane.  1: measured on the MI355X (gpurun r07r, 2^18 rows x 105 properties) four rows per lane take 0.65 ms
// against 0.36 for one — the cell's dependent loads (row → frame → value record → converted value) do not overlap across the unrolled
// rows the way they do across lanes; the launch is not dispatch-bound, unlike the marking kernels
#ifndef TF_SR_VALUES_RPT
#define TF_SR_VALUES_RPT 1
#endif
template <bool CANON>
__global__ void __launch_bounds__(256) sr_cell_values(Params p, const OutCol *cols, int32_t *src_row, uint32_t *part_id) {
  const int j = (int)blockIdx.y;
#pragma unroll
  for (int k = 0; k < TF_SR_VALUES_RPT; k++) sr_cell_value_one<CANON>(p, cols, src_row, part_id, j, ((int64_t)blockIdx.x * TF_SR_VALUES_RPT + k) * 256 + threadIdx.x);
}
// The text cells that are a plain byte range of the message — number tokens, strings without escapes: nearly all of them — are
// packed destination-centrically (tf_segcopy.hpp: a lane owns aligned 8-byte words of the column's payload and pulls their bytes
// from the messages; a wave stores 512 contiguous bytes); the other cells are zero-filled here and written by sr_cell_text, which
// is launched for the columns sr_mark_special found such cells in.  (One lane copying one cell's bytes was 0.59 ms for the 2^18
// hits messages' text; the same bytes through json_copy_words take 0.21.)
__device__ __forceinline__ uint32_t sr_plain_src(const Params &p, const OutCol &c, int j, int64_t r, bool *special) {
  const uint64_t rec = p.vrec[(int64_t)j * p.nframes + p.row_frame[r]];
  const uint32_t vtr = (uint32_t)(rec >> 56), vt = vtr & VT_MASK, vs = (uint32_t)rec;
  if (vt == VT_ABSENT || vt == VT_NULL) return SEG_NONE;
  if (!(vtr & VT_CANON)) {
    if (c.json_type == TFGPU_SRT_NUMBER) return vs;
    if (c.json_type == TFGPU_SRT_STRING && (vtr & VT_PLAIN)) return vs + 1;
    if (special) *special = true;
  }
  return SEG_NONE;
}
__global__ void __launch_bounds__(256) sr_mark_special(Params p, const OutCol *cols, const int32_t *text_cols, uint32_t *spec) {
  constexpr int RPT = 16;  // rows per lane (a lane that looks at one cell and leaves makes the launch dispatch-bound)
  const int t = (int)blockIdx.y;
  bool special = false;
  for (int k = 0; k < RPT; k++) {
    const int64_t r = ((int64_t)blockIdx.x * RPT + k) * 256 + threadIdx.x;
    if (r < p.nrows) sr_plain_src(p, cols[text_cols[t]], text_cols[t], r, &special);
  }
  if (__any(special) && (threadIdx.x & 63) == 0 && !__atomic_load_n(&spec[t], __ATOMIC_RELAXED)) atomicOr(&spec[t], 1u);  // (a flag that is up is seen by a plain L2 read: no atomic per wave)
}
__global__ void __launch_bounds__(256) sr_copy_words(Params p, const OutCol *cols, const int32_t *text_cols) {
  __shared__ uint32_t doff[256 + 1];
  __shared__ uint32_t soff[256];
  const int j = text_cols[blockIdx.y];
  const OutCol c = cols[j];
  auto so = [&](int64_t r) { return sr_plain_src(p, c, j, r, nullptr); };
  segcopy_run<1>(c.lens, p.nrows, (int64_t)blockIdx.x * 256, p.data, c.data, so, doff, soff);
}
template <bool CANON>
__global__ void __launch_bounds__(256) sr_cell_text(Params p, const OutCol *cols, const int32_t *text_cols, int32_t ntext, int plain_done) {
  const int t = (int)blockIdx.y; const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  const int j = text_cols[t];
  const OutCol &c = cols[j];
  const uint32_t f = p.row_frame[r];
  const int64_t i = (int64_t)j * p.nframes + f;
  const uint64_t rec = p.vrec[i];
  const uint32_t vtr = (uint32_t)(rec >> 56), vt = vtr & VT_MASK, vs = (uint32_t)rec, vl = (uint32_t)(rec >> 32) & 0xFFFFFFu;
  if (vt == VT_ABSENT || vt == VT_NULL) return;
  if (((vtr & VT_CANON) != 0) != CANON) return;
  if (!CANON && plain_done && (c.json_type == TFGPU_SRT_NUMBER || (c.json_type == TFGPU_SRT_STRING && (vtr & VT_PLAIN)))) return;  // sr_copy_words moved it
  MemBytes rd(p.data);
  ByteSink s{c.data + c.lens[r]};
  if constexpr (CANON) { emit_any_canon(s, rd, vs, vl); s.flush(); return; }
  auto copy_raw = [&](uint32_t a, uint32_t n) {  // eight bytes per step
    uint32_t k = 0;
    for (; k + 8 <= n; k += 8) s.put_word(rd.word(a + k), 8);
    if (k < n) s.put_word(rd.word(a + k) & ((1ull << (8 * (n - k))) - 1), n - k);
  };
  if (c.json_type == TFGPU_SRT_NUMBER) copy_raw(vs, vl);
  else if (c.json_type == TFGPU_SRT_STRING && (vtr & VT_PLAIN)) copy_raw(vs + 1, vl - 2);
  else if (c.json_type == TFGPU_SRT_STRING) emit_unquoted(s, rd, vs, vl);
  else emit_any(s, rd, vs, vl);
  s.flush();
}
__global__ void __launch_bounds__(256) sr_pack_validity(const OutCol *cols, int32_t nprops, int64_t nrows) {
  const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nb = (nrows + 7) / 8;
  if (it >= (int64_t)nprops * nb) return;
  const int j = (int)(it / nb); const int64_t b = it - (int64_t)j * nb;
  uint32_t v = 0;
  for (int k = 0; k < 8; k++) { const int64_t r = b * 8 + k; if (r < nrows && cols[j].valid8[r]) v |= 1u << k; }
  cols[j].validity[b] = (uint8_t)v;
}

// isGenerateUpdates (format_json.go:44-47, utils_json.go:57-63): an optional field the payload does not hold is not among the item's ColumnNames —
// the ABSENT bitmap of its column (DColumn::absent), one output byte a thread, and a flag per property that has any
__global__ void __launch_bounds__(256) sr_pack_absent(Params p, const int32_t *opt, int nopt, uint8_t *const *bits, uint32_t *colflag) {
  const int64_t nb = (p.nrows + 7) / 8;
  const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= (int64_t)nopt * nb) return;
  const int k = (int)(it / nb); const int64_t b = it - (int64_t)k * nb;
  const int j = opt[k];
  uint32_t v = 0;
  for (int q = 0; q < 8; q++) {
    const int64_t r = b * 8 + q;
    if (r >= p.nrows) break;
    const uint64_t rec = p.vrec[(int64_t)j * p.nframes + p.row_frame[r]];
    if (((uint32_t)(rec >> 56) & VT_MASK) == VT_ABSENT) v |= 1u << q;
  }
  bits[k][b] = (uint8_t)v;
  if (v) atomicOr(&colflag[k], 1u);
}

static inline unsigned nblk(int64_t n, int t) { return (unsigned)std::max<int64_t>(1, (n + t - 1) / t); }

// input bytes + message starts in HBM, frames counted and listed; returns the frame count
struct Staged {
  Buf bytes, ms, fcount, frames, first, nozero;
  Params p{};
  int64_t nframes = 0;
};
static void stage_frames(Staged &s, const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs) {
  Context &cx = ctx();
  hipStream_t st = cx.stream;
  if (len >= 0xFFFFFFF0ull) throw Error(TFGPU_ERR_UNSUPPORTED, "confluent SR: batch must be < 4 GiB (32-bit offsets)");
  if (mem == TFGPU_MEM_HOST) {
    s.bytes = dalloc(len + 64);
    h2d(s.bytes->p, bytes, len);
    TF_HIP(hipMemsetAsync((char *)s.bytes->p + len, 0, 64, st));
    s.p.data = ptr<uint8_t>(s.bytes);
  } else {
    s.p.data = (const uint8_t *)bytes;
    if (reinterpret_cast<uintptr_t>(s.p.data) & 15) throw Error(TFGPU_ERR_INVALID, "confluent SR: device buffer must be 16-byte aligned");
  }
  const int64_t nmsg = msgs ? msgs->nmsg : 1;
  if (nmsg < 0 || (msgs && nmsg > 0 && !msgs->start)) throw Error(TFGPU_ERR_INVALID, "confluent SR: bad message batch");
  std::vector<uint32_t> ms((size_t)nmsg + 1);
  if (msgs) {
    for (int64_t m = 0; m <= nmsg; m++) {
      if (msgs->start[m] > len || (m && msgs->start[m] < msgs->start[m - 1])) throw Error(TFGPU_ERR_INVALID, "confluent SR: message offsets must be ascending and inside the buffer");
      ms[(size_t)m] = (uint32_t)msgs->start[m];
    }
  } else { ms[0] = 0; ms[1] = (uint32_t)len; }
  s.ms = dalloc(ms.size() * 4 + 16);
  h2d(s.ms->p, ms.data(), ms.size() * 4);
  s.fcount = dalloc_zero((size_t)(nmsg + 1) * 4 + 16);
  s.p.ms = ptr<uint32_t>(s.ms); s.p.nmsg = nmsg; s.p.fcount = ptr<uint32_t>(s.fcount);
  s.first = dalloc((size_t)std::max<int64_t>(nmsg, 1) * sizeof(tfgpu_sr_frame));
  const tfgpu_sr_frame *first = reinterpret_cast<const tfgpu_sr_frame *>(s.first->p);
  s.nozero = dalloc_zero((size_t)std::max<int64_t>(nmsg, 1) + 16);
  if (nmsg) {
    KernelTimer t("sr_frames");
    sr_find_zero<<<(unsigned)((nmsg + 3) / 4), 256, 0, st>>>(s.p, ptr<uint8_t>(s.nozero));
    s.p.nozero = ptr<uint8_t>(s.nozero);
    sr_count_frames<<<nblk(nmsg, 256), 256, 0, st>>>(s.p, reinterpret_cast<tfgpu_sr_frame *>(s.first->p));
  }
  exclusive_scan_u32(s.p.fcount, s.p.fcount, nmsg, true);
  const uint32_t *h = d2h_u32(s.p.fcount + nmsg);
  tf::sync();  // also fences the pageable sources (bytes, ms)
  s.nframes = *h;
  s.frames = dalloc((size_t)std::max<int64_t>(s.nframes, 1) * sizeof(tfgpu_sr_frame));
  s.p.frames = reinterpret_cast<tfgpu_sr_frame *>(s.frames->p); s.p.nframes = s.nframes;
  if (nmsg && s.nframes) { KernelTimer t("sr_frames"); sr_fill_frames<<<nblk(nmsg, 256), 256, 0, st>>>(s.p, first); }
}

struct LastFrames { const tfgpu_sr_frame *host = nullptr; int64_t n = -1; const void *bytes = nullptr; Buf dev; };
static thread_local LastFrames g_last;

}  // namespace sr
}  // namespace tf

using namespace tf;

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

extern "C" int tfgpu_sr_frames(const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs, tfgpu_sr_frame *frames, int64_t cap, int64_t *nframes) {
  TF_API_BEGIN
  if (!nframes || (len && !bytes)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_sr_frames: null argument");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  sr::Staged s;
  sr::stage_frames(s, bytes, len, mem, msgs);
  *nframes = s.nframes;
  if (s.nframes > cap) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_sr_frames: more frames than the output holds (see *nframes)");
  if (s.nframes) { d2h(frames, s.frames->p, (size_t)s.nframes * sizeof(tfgpu_sr_frame)); tf::sync(); }
  sr::g_last = sr::LastFrames{frames, s.nframes, bytes, s.frames};
  return TFGPU_OK;
  TF_API_END
}
// the list tfgpu_sr_frames just wrote, still in HBM — for a caller in this library that hands the untouched host copy back (the Debezium
// receiver's registry form, tf_debezium.hip): 32 bytes per event are neither checked nor uploaded again
namespace tf { namespace sr { Buf last_frames_device(const tfgpu_sr_frame *host, int64_t n, const void *bytes) {
  return (g_last.host == host && g_last.n == n && g_last.bytes == bytes) ? g_last.dev : Buf();
} } }

extern "C" int tfgpu_sr_json_parse(const tfgpu_sr_json_options *o, const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs,
                                   tfgpu_dbatch **out, tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs) {
  TF_API_BEGIN
  if (!o || !out || (len && !bytes) || (o->nprops && !o->props)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_sr_json_parse: null argument");
  const int np = o->nprops;
  std::string names;
  std::vector<sr::Prop> props((size_t)np);
  for (int j = 0; j < np; j++) {
    const tfgpu_sr_property &pr = o->props[j];
    if (!pr.name || pr.json_type < TFGPU_SRT_BOOLEAN || pr.json_type > TFGPU_SRT_ANY) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_sr_json_parse: bad property");
    if (j && std::string(o->props[j - 1].name) >= pr.name) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_sr_json_parse: properties must be sorted by name and unique (util.MapKeysInOrder)");
    props[(size_t)j] = sr::Prop{(uint32_t)names.size(), (uint32_t)std::strlen(pr.name), pr.json_type, pr.required ? 1 : 0};
    names += pr.name;
    names.append((8 - names.size() % 8) % 8, '\0');  // whole 8-byte words: the device compares a word at a time
  }
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  hipStream_t st = cx.stream;
  sr::Staged s;
  sr::stage_frames(s, bytes, len, mem, msgs);
  sr::Params &p = s.p;
  const int64_t nf = s.nframes, nfa = std::max<int64_t>(nf, 1);
  Buf bprops = upload_small(props.data(), std::max<size_t>(props.size(), 1) * sizeof(sr::Prop)), bnames = upload_small(names.data(), names.size());
  Buf vrec = dalloc_zero((size_t)std::max(np, 1) * (size_t)nfa * 8 + 16);  // VT_ABSENT = 0
  Buf status = dalloc_zero((size_t)nfa + 16), keep = dalloc_zero((size_t)(nfa + 1) * 4 + 16), nerr = dalloc_zero(16);
  p.schema_id = o->schema_id; p.report_frame_errors = o->report_frame_errors;
  p.props = ptr<sr::Prop>(bprops); p.nprops = np; p.names = ptr<uint8_t>(bnames);
  p.vrec = ptr<uint64_t>(vrec);
  p.status = ptr<uint8_t>(status); p.keep = ptr<uint32_t>(keep); p.nerr = ptr<uint32_t>(nerr);
  Buf bmap, bival;  // sr_parse_quick's member map and the integer cells it converts in LDS
  Buf guess = dalloc(sr::GUESS_N * 2 + 16);
  TF_HIP(hipMemsetAsync(guess->p, 0xFF, sr::GUESS_N * 2, st));
  p.guess = ptr<uint16_t>(guess);
  if (nf) {
    static const bool tilepath = [] { const char *e = std::getenv("TFGPU_SR_TILES"); return !(e && e[0] == '0'); }();  // 0: A/B runs
    if (tilepath) {
      const uint64_t avg = std::max<uint64_t>(len / (uint64_t)nfa, 1);
      const int32_t per_tile = (int32_t)std::min<uint64_t>(std::max<uint64_t>((uint64_t)(JT_BYTES - 16) * 8 / (avg * 9), 1), (uint64_t)JT_LINES);
      Buf slow = dalloc((size_t)(nf + 1) * 4 + 16);
      TF_HIP(hipMemsetAsync(slow->p, 0, 4, st));
      static const int ablate = [] { const char *e = std::getenv("TFGPU_JT_ABLATE"); return e ? std::atoi(e) : 0; }();
      static const bool quick_off = [] { const char *e = std::getenv("TFGPU_SR_QUICK"); return e && e[0] == '0'; }();  // 0: the round-2 tile kernel (A/B runs, cross-check)
      int32_t used_per_tile = per_tile;
      if (!quick_off && np <= JQ_MEM) {
        const int32_t qper = (int32_t)std::min<uint64_t>(std::max<uint64_t>((uint64_t)(JQ_BYTES - 16) * 8 / (avg * 9), 1), (uint64_t)JQ_LINES);
        used_per_tile = qper;
        bmap = dalloc(sizeof(sr::SrqMap));
        bival = dalloc((size_t)std::max(np, 1) * (size_t)nfa * 8 + 16);
        p.ival = ptr<int64_t>(bival);
        { KernelTimer t("sr_quick_map"); sr::sr_quick_map<<<1, JQ_THREADS, 0, st>>>(p, ptr<sr::SrqMap>(bmap)); }
        { KernelTimer t("sr_parse_quick"); sr::sr_parse_quick<<<sr::nblk(nf, qper), JQ_THREADS, 0, st>>>(p, ptr<sr::SrqMap>(bmap), qper, ptr<int64_t>(bival), ptr<uint32_t>(slow), ptr<uint32_t>(slow) + 1, ablate); }
      } else {
      const unsigned ntile = sr::nblk(nf, per_tile), nb = (unsigned)std::min<int64_t>((int64_t)ntile, (int64_t)cx.num_cus * 2);
      { KernelTimer t("sr_parse_tiles");
        sr::sr_parse_tiles<<<nb, JT_THREADS, 0, st>>>(p, per_tile, ptr<uint32_t>(slow), ptr<uint32_t>(slow) + 1, ablate); }
      }
      { KernelTimer t("sr_parse_frames"); sr::sr_parse_listed<<<sr::nblk(nf, 128), 128, 0, st>>>(p, ptr<uint32_t>(slow), ptr<uint32_t>(slow) + 1); }
      static const bool dbg = [] { const char *e = std::getenv("TFGPU_JSON_TILE_DEBUG"); return e && e[0] == '1'; }();
      if (dbg) {
        const uint32_t *a = d2h_u32(slow->p);
        tf::sync();
        std::fprintf(stderr, "[tfgpu] sr tiles: %lld frames, %d per tile, %u to parse_frame\n", (long long)nf, (int)used_per_tile, *a);
      }
    } else
    { KernelTimer t("sr_parse_frames"); sr::sr_parse_frames<<<sr::nblk(nf, 128), 128, 0, st>>>(p); }
    { KernelTimer t("sr_message_rule"); sr::sr_message_rule<<<sr::nblk(p.nmsg, 256), 256, 0, st>>>(p); }
  }
  exclusive_scan_u32(p.keep, p.keep, nf, true);
  const uint32_t *hrows = d2h_u32(p.keep + nf), *hnerr = d2h_u32(p.nerr);
  tf::sync();
  const int64_t nrows = *hrows, nra = std::max<int64_t>(nrows, 1);
  const uint32_t nerr_total = *hnerr;

  auto db = std::make_unique<tfgpu_dbatch>();
  db->nrows = nrows;
  db->ns = o->table_ns ? o->table_ns : ""; db->table = o->table_name ? o->table_name : "";
  db->src_row = dalloc((size_t)nra * 4); db->part_id = dalloc((size_t)nra * 4);
  Buf row_frame = dalloc((size_t)nra * 4);
  p.row_frame = ptr<uint32_t>(row_frame); p.nrows = nrows;
  std::vector<sr::OutCol> oc((size_t)np);
  std::vector<int32_t> text_cols;
  const int64_t seg_stride = ((nrows + 1 + 3) / 4) * 4;
  int ntext = 0;
  bool has_any = false;
  for (int j = 0; j < np; j++) { if (o->props[j].json_type >= TFGPU_SRT_NUMBER) ntext++; if (o->props[j].json_type == TFGPU_SRT_ANY) has_any = true; }
  Buf lens_all = dalloc_zero((size_t)std::max(ntext, 1) * (size_t)seg_stride * 4 + 16);
  Buf valid8 = dalloc((size_t)std::max(np, 1) * (size_t)nra);
  int ti = 0;
  for (int j = 0; j < np; j++) {
    const tfgpu_sr_property &pr = o->props[j];
    DColumn d;
    d.name = pr.name;
    sr::OutCol &c = oc[(size_t)j];
    std::memset(&c, 0, sizeof c);
    c.json_type = pr.json_type;
    switch (pr.json_type) {
      // (sr_cell_values writes every row of every property and sr_pack_validity every byte of every bitmap: no zero fills — they were ~200
      //  hipMemsetAsync launches per batch, a millisecond of host time)
      case TFGPU_SRT_BOOLEAN: d.dtype = TFGPU_T_BOOLEAN; d.repr = TFGPU_R_BOOL; d.values = dalloc((size_t)nra); c.values = d.values->p; break;
      case TFGPU_SRT_INTEGER: d.dtype = TFGPU_T_INT64; d.repr = TFGPU_R_INT64; d.values = dalloc((size_t)nra * 8); c.values = d.values->p; break;
      case TFGPU_SRT_NUMBER: d.dtype = TFGPU_T_FLOAT64; d.repr = TFGPU_R_JSONNUM; break;
      case TFGPU_SRT_STRING: d.dtype = TFGPU_T_UTF8; d.repr = TFGPU_R_STRING; break;
      default: d.dtype = TFGPU_T_ANY; d.repr = TFGPU_R_JSON;
    }
    if (pr.json_type >= TFGPU_SRT_NUMBER) {
      d.offsets = subbuf(lens_all, (size_t)ti * (size_t)seg_stride * 4, (size_t)(nrows + 1) * 4);
      c.lens = ptr<uint32_t>(d.offsets);
      text_cols.push_back(j);
      ti++;
    }
    c.valid8 = ptr<uint8_t>(valid8) + (size_t)j * (size_t)nra;
    d.validity = dalloc((size_t)((nra + 7) / 8) + 8);
    c.validity = ptr<uint8_t>(d.validity);
    db->schema.push_back({d.name, d.dtype});
    db->cols.push_back(std::move(d));
  }
  Buf boc = upload_small(oc.data(), std::max<size_t>(oc.size(), 1) * sizeof(sr::OutCol));
  if (nrows) {
    sr::sr_row_frames<<<sr::nblk(nf, 256), 256, 0, st>>>(p);
    if (np) {
      KernelTimer t("sr_cell_values");
      sr::sr_cell_values<false><<<dim3(sr::nblk(nrows, 256 * TF_SR_VALUES_RPT), (unsigned)np), 256, 0, st>>>(p, ptr<sr::OutCol>(boc), ptr<int32_t>(db->src_row), ptr<uint32_t>(db->part_id));
      if (has_any) sr::sr_cell_values<true><<<dim3(sr::nblk(nrows, 256 * TF_SR_VALUES_RPT), (unsigned)np), 256, 0, st>>>(p, ptr<sr::OutCol>(boc), ptr<int32_t>(db->src_row), ptr<uint32_t>(db->part_id));
    }
    if (np) sr::sr_pack_validity<<<sr::nblk((int64_t)np * ((nrows + 7) / 8), 256), 256, 0, st>>>(ptr<sr::OutCol>(boc), np, nrows);
  }
  if (np == 0 && nrows) return tf::fail(TFGPU_ERR_UNSUPPORTED, "confluent SR json: a schema without properties");
  // isGenerateUpdates: every item is an Update, and lists the optional fields its payload holds
  std::vector<int32_t> optp;
  std::vector<Buf> optbits;
  const uint32_t *habs = nullptr;
  if (o->is_generate_updates && nrows) {
    db->kind = dalloc((size_t)nra);
    TF_HIP(hipMemsetAsync(db->kind->p, TFGPU_K_UPDATE, (size_t)nra, st));
    for (int j = 0; j < np; j++) if (!o->props[j].required) optp.push_back(j);
    if (!optp.empty()) {
      std::vector<uint8_t *> bp;
      for (size_t k = 0; k < optp.size(); k++) { optbits.push_back(dalloc((size_t)((nra + 7) / 8) + 8)); bp.push_back(ptr<uint8_t>(optbits.back())); }
      Buf bopt = upload_small(optp.data(), optp.size() * 4), bbp = upload_small(bp.data(), bp.size() * sizeof(uint8_t *)), colflag = dalloc_zero(optp.size() * 4);
      sr::sr_pack_absent<<<sr::nblk((int64_t)optp.size() * ((nrows + 7) / 8), 256), 256, 0, st>>>(p, ptr<int32_t>(bopt), (int)optp.size(), ptr<uint8_t *>(bbp), ptr<uint32_t>(colflag));
      habs = d2h_u32(colflag->p, optp.size());  // read at the next sync: a column no row leaves out carries no bitmap
    }
  }
  if (ntext) {
    exclusive_scan_u32_segments(ptr<uint32_t>(lens_all), nrows, ntext, seg_stride);  // offsets in place, the total at [nrows]
    const uint32_t *tot = segment_totals_to_host(ptr<uint32_t>(lens_all), nrows, ntext, seg_stride);
    // OFF by default: measured on the MI355X (gpurun r07b, 2^18 hits messages) sr_cell_text is 0.60 ms one lane per cell and 0.65 ms this way —
    // a third of the text (Cyrillic titles, strings with escapes) is not VT_PLAIN and still takes the walkers, and the marking pass re-reads the value records
    static const bool words = [] { const char *e = std::getenv("TFGPU_SR_COPY_WORDS"); return e && e[0] == '1'; }();
    Buf spec = dalloc_zero((size_t)ntext * 4), btc0 = upload_small(text_cols.data(), text_cols.size() * 4);
    if (nrows && words) sr::sr_mark_special<<<dim3(sr::nblk(nrows, 256 * 16), (unsigned)ntext), 256, 0, st>>>(p, ptr<sr::OutCol>(boc), ptr<int32_t>(btc0), ptr<uint32_t>(spec));
    const uint32_t *hspec = d2h_u32(spec->p, (size_t)ntext);
    tf::sync();
    for (int t = 0; t < ntext; t++) {
      DColumn &d = db->cols[(size_t)text_cols[(size_t)t]];
      d.data_len = tot[t];
      d.data = dalloc(d.data_len + 8);
      oc[(size_t)text_cols[(size_t)t]].data = ptr<uint8_t>(d.data);
    }
    boc = upload_small(oc.data(), oc.size() * sizeof(sr::OutCol));
    Buf btc = upload_small(text_cols.data(), text_cols.size() * 4);
    if (nrows) {
      KernelTimer t("sr_cell_text");
      if (words) {
        sr::sr_copy_words<<<dim3(sr::nblk(nrows, 256), (unsigned)ntext), 256, 0, st>>>(p, ptr<sr::OutCol>(boc), ptr<int32_t>(btc));
        std::vector<int32_t> sp;  // the columns that hold cells the walkers write
        for (int t2 = 0; t2 < ntext; t2++) if (hspec[t2]) sp.push_back(text_cols[(size_t)t2]);
        if (!sp.empty()) { Buf bsp = upload_small(sp.data(), sp.size() * 4); sr::sr_cell_text<false><<<dim3(sr::nblk(nrows, 256), (unsigned)sp.size()), 256, 0, st>>>(p, ptr<sr::OutCol>(boc), ptr<int32_t>(bsp), (int32_t)sp.size(), 1); }
      } else sr::sr_cell_text<false><<<dim3(sr::nblk(nrows, 256), (unsigned)ntext), 256, 0, st>>>(p, ptr<sr::OutCol>(boc), ptr<int32_t>(btc), ntext, 0);
      if (has_any) sr::sr_cell_text<true><<<dim3(sr::nblk(nrows, 256), (unsigned)ntext), 256, 0, st>>>(p, ptr<sr::OutCol>(boc), ptr<int32_t>(btc), ntext, 0);
    }
  }
  // ---- frames the reference turns into `_unparsed` items / rows for the stock path ----
  int64_t ne = 0;
  if (nerr_total) {
    std::vector<uint8_t> hst((size_t)nf);
    std::vector<tfgpu_sr_frame> hfr((size_t)nf);
    d2h(hst.data(), status->p, (size_t)nf); d2h(hfr.data(), s.frames->p, (size_t)nf * sizeof(tfgpu_sr_frame));
    tf::sync();
    for (int64_t f = 0; f < nf; f++) {
      const int c = hst[(size_t)f];
      if (c == sr::ST_OK || c == sr::ST_OTHER || c == sr::ST_DROPPED) continue;
      if (errs && ne < errs_cap) errs[ne] = tfgpu_row_error{f, c, (int32_t)hfr[(size_t)f].msg, -1};
      ne++;
    }
  }
  tf::sync();
  for (size_t k = 0; habs && k < optp.size(); k++) if (habs[k]) db->cols[(size_t)optp[k]].absent = optbits[k];
  if (nerrs) *nerrs = ne;
  *out = db.release();
  return TFGPU_OK;
  TF_API_END
}
