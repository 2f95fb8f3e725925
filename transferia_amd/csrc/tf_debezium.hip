// tf_debezium.hip — Debezium ingest with inline schemas (the source format of BASELINE.json configs[4]):
//
//   DebeziumImpl.DoBatch / Do / DoBuf / DoOne     pkg/parsers/registry/debezium/engine/parser.go:33-130
//   IncludeSchema.Unpack                          pkg/debezium/unpacker/include_schema.go:13-25
//   Receiver.receive / add                        pkg/debezium/receiver.go:150-232, 98-121
//   receiveField / extractVal / convertVal        pkg/debezium/receiver_engine.go:148-371
//   the default receivers                         pkg/debezium/common/field_receiver_default.go:40-355
//   Payload / Source                              pkg/debezium/common/debezium_schema.go:31-56
//   Base64ToNumeric                               pkg/debezium/typeutil/helpers.go:966-996
//
// A Debezium event is {"schema": <Kafka Connect schema, ~12 KB>, "payload": {"before", "after", "source", "op", …}}: one
// message, one ChangeItem.  The schema is data, so the work splits: the shim compiles each distinct schema once (the
// reference caches it by hash too) into a list of field receivers, and the device does everything that is per message —
//   dbz_unpack          one lane per message: the whole message validated as one JSON value (encoding/json's grammar),
//                       the raw "schema" / "payload" members located, the schema bytes hashed for grouping;
//   dbz_parse           one lane per message of the schema at hand: the payload decoded like Decoder(UseNumber).Decode(
//                       &Payload) — member types checked field by field, op → kind, source.{lsn, ts_ms, txId, schema,
//                       table} read — then the before / after members matched against the (sorted) field names and every
//                       value checked against its receiver in schema order: the first failure names the message's fate;
//   dbz_table_rule      rows of another table than the first good row's go to the host (a batch has one TableID);
//   dbz_cell_values     one lane per (field, row): bool / intN / float64 values, text lengths (strings unquoted, base64
//                       decoded, decimals rendered, points formatted);
//   dbz_cell_text       one lane per (text field, row): the bytes.
// OldKeys cost nothing extra: Receiver.add stores the SAME converted value under ColumnValues and OldKeys, so the OldKeys
// columns share the value buffers of the key columns and differ only in their validity (Update / Delete rows).
// Latency-bound lane-per-message form (every lane walks ~13 KB of JSON); algorithmic bytes: message bytes in + column
// bytes out.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>

#include "tf_jsonscan.hpp"
#include "tf_segcopy.hpp"
#include "tf_wave.hpp"
#include "tf_swar.hpp"
#include "tf_jsonquick.hpp"

namespace tf {
namespace dbz {

using namespace tf::sr;

enum : uint8_t { ST_SKIP = 255 };  // a message of another schema; else TFGPU_ROW_OK or a tfgpu_rowerr
constexpr int DEC_MAX = 64;        // decimals wider than this many bytes go to the host

struct FRecv { uint32_t name_off, name_len; int32_t op, optional, scale; };
struct Params {
  const uint8_t *data;
  const uint32_t *ms; int64_t nmsg;
  tfgpu_dbz_frame *frames;                     // [nmsg]
  uint64_t h0, h1;
  int32_t schema_code;                         // TFGPU_ROW_OK, or the fate the host decided for every message of this schema
  const FRecv *fields; int32_t nfields; const uint8_t *names;
  const uint16_t *sorted;                      // field indices in name order (binary search)
  uint32_t *vstart, *vlen; uint8_t *vtype;     // [nfields][nmsg]
  uint8_t *status, *kind;                      // [nmsg]
  uint64_t *lsn, *ts, *tabhash; uint32_t *txid;  // [nmsg]
  uint32_t *tab_s, *tab_n;                     // [2][nmsg] spans of source.schema / source.table string literals
  uint32_t *keep;                              // [nmsg + 1]
  uint32_t *row_msg; int64_t nrows;
  uint32_t *nerr;
  uint32_t *toast;                             // set when some optional field holds `__debezium_unavailable_value`: the row does not list that column
  unsigned long long *first_ok;                // lowest good message (atomicMin)
  const double *p10; const uint64_t *p128;
  // the fast path of dbz_unpack: messages that start with the bytes message `ref` holds in front of its payload value
  int64_t ref; uint32_t plen; uint8_t *same;   // same[m] = 1: the first plen bytes equal the reference's
  const uint8_t *pref;                         // the reference prefix when it is not message `ref` of this batch (a cached one): ref = -1
  tfgpu_dbz_frame pframe;                      // … and its frame, offsets relative to the message start
  // tentative frames (the receiver's two calls only, see FrameCache): tent[m] = 1 — message m shares the reference's prefix and ends
  // in '}' blank* '}' blank*; its payload span runs from the prefix to that first '}' and NOBODY HAS WALKED IT YET.  dbz_parse_quick
  // proves the span a JSON object when it takes the message (every byte is compared or validated); for a message it does not take,
  // dbz_parse_listed / dbz_parse run IncludeSchema.Unpack's full walk first.
  uint8_t *tent;
};

__device__ __forceinline__ uint64_t mix64(uint64_t h, uint64_t w) { h ^= w; h *= 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); }

// A decoded key against an ASCII literal: 1 equal, 2 equal only under ASCII case folding (encoding/json's field match), 0 no
__device__ int key_is(MemBytes &rd, uint32_t ks, uint32_t ke, const char *lit, uint32_t n) {
  // a decoded key is never longer than its literal, and equal length means no escapes: most candidates fall out on the length
  const uint32_t raw = ke - ks - 2;
  if (raw < n) return 0;
  if (raw == n) {
    bool exact = true;
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t c = rd.at(ks + 1 + i), l = (uint8_t)lit[i];
      if (c == l) continue;
      if ((c | 0x20u) == (l | 0x20u) && (c | 0x20u) >= 'a' && (c | 0x20u) <= 'z') { exact = false; continue; }
      return 0;
    }
    return exact ? 1 : 2;
  }
  RuneIter it{&rd, nullptr, ks + 1, ke - 1};
  bool exact = true;
  for (uint32_t i = 0; i < n; i++) {
    const int r = it.next();
    if (r < 0) return 0;
    if (r == (int)(uint8_t)lit[i]) continue;
    if (r < 0x80 && ((uint32_t)r | 0x20u) == ((uint32_t)(uint8_t)lit[i] | 0x20u) && (((uint32_t)r | 0x20u) >= 'a' && ((uint32_t)r | 0x20u) <= 'z')) { exact = false; continue; }
    return 0;
  }
  if (it.next() >= 0) return 0;
  return exact ? 1 : 2;
}
// the raw key bytes against a field name of the same length (no escapes can be involved then)
__device__ __forceinline__ bool key_eq_name(MemBytes &rd, uint32_t ks, uint32_t ke, const uint8_t *name, uint32_t n) {
  if (ke - ks - 2 != n) return false;
  for (uint32_t i = 0; i < n; i++) if (rd.at(ks + 1 + i) != name[i]) return false;
  return true;
}

// ---- IncludeSchema.Unpack ---------------------------------------------------------------------------------------------
__device__ bool unpack_fast(const Params &p, int64_t m);
__device__ void unpack_message(const Params &p, int64_t m) {
  tfgpu_dbz_frame fr{};
  MemBytes rd(p.data);
  uint32_t pos = p.ms[m]; const uint32_t end = p.ms[m + 1];
  auto done = [&](int code) { fr.code = code; p.frames[m] = fr; };
  if (pos >= end) return done(TFGPU_ROW_DBZ_UNPACK);                       // "debezium parser received empty message"
  if (rd.at(pos) == 0) return done(end - pos < 5 ? TFGPU_ROW_HOST_FALLBACK /* buf[5:] panics */ : TFGPU_ROW_DBZ_UNPACK);
  auto skip_ws = [&]() { while (pos < end && is_ws(rd.at(pos))) pos++; };
  skip_ws();
  if (pos >= end) return done(TFGPU_ROW_DBZ_UNPACK);
  bool host = false;
  if (rd.at(pos) != '{') {  // null leaves both RawMessages nil; any other valid value is an UnmarshalTypeError
    uint32_t vt = 0;
    const int rc = skip_value(rd, pos, end, vt);
    skip_ws();
    if (rc == 2) return done(TFGPU_ROW_HOST_FALLBACK);
    if (rc || pos != end) return done(TFGPU_ROW_DBZ_UNPACK);
    return done((vt & VT_MASK) == VT_NULL ? TFGPU_ROW_OK : TFGPU_ROW_DBZ_UNPACK);
  }
  pos++;
  skip_ws();
  if (pos >= end) return done(TFGPU_ROW_DBZ_UNPACK);
  if (rd.at(pos) == '}') pos++;
  else for (;;) {
    skip_ws();
    if (pos >= end || rd.at(pos) != '"') return done(TFGPU_ROW_DBZ_UNPACK);
    const uint32_t ks = pos;
    if (!scan_string(rd, pos, end)) return done(TFGPU_ROW_DBZ_UNPACK);
    const uint32_t ke = pos;
    skip_ws();
    if (pos >= end || rd.at(pos) != ':') return done(TFGPU_ROW_DBZ_UNPACK);
    pos++;
    skip_ws();
    const uint32_t vs = pos;
    uint32_t vt = 0;
    const int rc = skip_value(rd, pos, end, vt);
    if (rc == 1) return done(TFGPU_ROW_DBZ_UNPACK);
    if (rc == 2) return done(TFGPU_ROW_HOST_FALLBACK);
    const int a = key_is(rd, ks, ke, "schema", 6), b = a ? 0 : key_is(rd, ks, ke, "payload", 7);
    if (a == 2 || b == 2) host = true;
    if (a) { fr.schema_start = vs; fr.schema_len = pos - vs; }
    if (b) { fr.payload_start = vs; fr.payload_len = pos - vs; }
    skip_ws();
    if (pos >= end) return done(TFGPU_ROW_DBZ_UNPACK);
    const uint32_t d = rd.at(pos);
    if (d == ',') { pos++; continue; }
    if (d == '}') { pos++; break; }
    return done(TFGPU_ROW_DBZ_UNPACK);
  }
  skip_ws();
  if (pos != end) return done(TFGPU_ROW_DBZ_UNPACK);  // "invalid character after top-level value"
  if (host) return done(TFGPU_ROW_HOST_FALLBACK);
  // equal schema bytes ⇔ equal hash: two 64-bit multiplicative lanes over 8-byte words, the length folded in
  uint64_t h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull;
  const uint32_t s = (uint32_t)fr.schema_start, n = fr.schema_len;
  uint32_t k = 0;
  for (; k + 8 <= n; k += 8) { const uint64_t w = rd.word(s + k); h0 = mix64(h0, w); h1 = mix64(h1 + 0x9E3779B97F4A7C15ull, w ^ (w >> 31)); }
  if (k < n) { const uint64_t w = rd.word(s + k) & ((1ull << (8 * (n - k))) - 1); h0 = mix64(h0, w); h1 = mix64(h1 + 1, w); }
  fr.schema_hash[0] = mix64(h0, n); fr.schema_hash[1] = mix64(h1, ~(uint64_t)n);
  done(TFGPU_ROW_OK);
}
__global__ void __launch_bounds__(128) dbz_unpack(Params p) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m < p.nmsg) unpack_message(p, m);
}
__global__ void __launch_bounds__(128) dbz_unpack_rest(Params p) {   // every message but the reference, which is done
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.nmsg || m == p.ref) return;   // ref = -1 with a cached prefix: every message is walked here
  if (p.same[m] && unpack_fast(p, m)) return;
  unpack_message(p, m);
}

// The reference caches the compiled schema by a hash of its bytes (receiver.go:61-66) because a topic repeats ONE schema in
// every message.  Made literal: the message that opens the batch is walked in full; every other message whose first `plen`
// bytes — everything in front of the payload value: `{"schema":{…12 KB…},"payload":` — equal the reference's inherits its
// schema span and hash, and only its payload value and the closing brace are validated.  A wave compares one message
// (64 lanes x 8 bytes per step: a memcmp at HBM speed instead of a 12 KB serial walk per lane).
__device__ __forceinline__ uint64_t read8u(const uint8_t *base, uint64_t a) {  // 8 bytes at any alignment (buffers carry >= 64 bytes of slack)
  const uint32_t sh = (uint32_t)(a & 7) * 8;
  const uint64_t *q = reinterpret_cast<const uint64_t *>(base + (a & ~7ull));
  const uint64_t x = q[0];
  return sh ? (x >> sh) | (q[1] << (64 - sh)) : x;
}
__global__ void __launch_bounds__(256) dbz_prefix_same(Params p) {
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.nmsg) return;
  const int lane = threadIdx.x & 63;
  const uint64_t a = p.ms[m];
  const uint8_t *rb = p.pref ? p.pref : p.data;
  const uint64_t r = p.pref ? 0 : p.ms[p.ref];
  bool diff = (uint64_t)p.ms[m + 1] - a < (uint64_t)p.plen + 2;   // room for a value and the closing brace
  if (!diff) {
    // The MESSAGE is what streams from HBM: it is read in its own 16-byte lines (one aligned global_load_dwordx4 a lane, 1 KiB a wave
    // step); the reference's bytes for the same positions sit at another alignment and come out of the cache (two shifted 8-byte reads
    // each).  What lies in front of the first line and behind the last one (< 16 bytes each) is compared by lanes 0 and 1.
    const uint64_t a1 = a + ((16u - (uint32_t)(reinterpret_cast<uintptr_t>(p.data + a) & 15u)) & 15u);   // by ADDRESS: a caller's device buffer need not start on a 16-byte line
    const uint32_t pre = min((uint32_t)(a1 - a), p.plen), nint = (p.plen - pre) >> 4;
    for (uint32_t c = lane; c < nint; c += 64) {
      const uint4 x = *reinterpret_cast<const uint4 *>(p.data + a1 + 16ull * c);
      const uint64_t o = r + pre + 16ull * c;
      diff |= ((uint64_t)x.x | ((uint64_t)x.y << 32)) != read8u(rb, o) || ((uint64_t)x.z | ((uint64_t)x.w << 32)) != read8u(rb, o + 8);
    }
    auto edge = [&](uint32_t at, uint32_t n) {  // n < 16 bytes at offset `at` of both
      for (uint32_t k = 0; k < n; k += 8) {
        const uint32_t nb = min(n - k, 8u);
        const uint64_t mask = nb == 8 ? ~0ull : (1ull << (8 * nb)) - 1;
        diff |= ((read8u(p.data, a + at + k) ^ read8u(rb, r + at + k)) & mask) != 0;
      }
    };
    if (lane == 0) edge(0, pre);
    if (lane == 1) edge(pre + 16u * nint, p.plen - pre - 16u * nint);
  }
  const bool any = __any(diff);
  if (lane == 0) p.same[m] = any ? 0 : 1;
}
// the remainder of a message that shares the reference's prefix: the payload value, white space, '}', white space, the end
__device__ bool unpack_fast(const Params &p, int64_t m) {
  const tfgpu_dbz_frame rf = p.pref ? p.pframe : p.frames[p.ref];
  const uint64_t rstart = p.pref ? 0 : p.ms[p.ref];
  MemBytes rd(p.data);
  const uint32_t a = p.ms[m], end = p.ms[m + 1];
  if (p.tent) {  // from the end: blank* '}' blank* then the payload's own '}' — the span is a claim, not a finding
    uint32_t e = end;
    const uint32_t ps = a + p.plen;
    while (e > ps && is_ws(rd.at(e - 1))) e--;
    if (e > ps + 2 && rd.at(e - 1) == '}') {
      e--;
      while (e > ps && is_ws(rd.at(e - 1))) e--;
      if (e >= ps + 2 && rd.at(ps) == '{' && rd.at(e - 1) == '}') {
        tfgpu_dbz_frame fr = rf;
        fr.schema_start = rf.schema_start - rstart + a;
        fr.payload_start = ps; fr.payload_len = e - ps;
        fr.code = TFGPU_ROW_OK;
        p.frames[m] = fr;
        p.tent[m] = 1;
        return true;
      }
    }
  }
  uint32_t pos = a + p.plen;
  auto skip_ws = [&]() { while (pos < end && is_ws(rd.at(pos))) pos++; };
  const uint32_t vs = pos;
  uint32_t vt = 0;
  if (pos >= end || skip_value(rd, pos, end, vt) != 0) return false;
  const uint32_t ve = pos;
  skip_ws();
  if (pos >= end || rd.at(pos) != '}') return false;   // more members (or an error): the full walk decides
  pos++;
  skip_ws();
  if (pos != end) return false;
  tfgpu_dbz_frame fr = rf;
  fr.schema_start = rf.schema_start - rstart + a;
  fr.payload_start = vs; fr.payload_len = ve - vs;
  fr.code = TFGPU_ROW_OK;
  p.frames[m] = fr;
  return true;
}

// ---- values -------------------------------------------------------------------------------------------------------------
// strconv.ParseUint(literal, 10, bits) as encoding/json applies it to an unsigned struct field
__device__ bool lit_uint(MemBytes &rd, uint32_t s, uint32_t n, int bits, uint64_t *out) {
  uint64_t v = 0;
  if (!n) return false;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t c = rd.at(s + i);
    if (c < '0' || c > '9') return false;
    const uint64_t d = c - '0';
    if (v > (~0ull - d) / 10) return false;
    v = v * 10 + d;
  }
  if (bits < 64 && (v >> bits)) return false;
  *out = v;
  return true;
}
// The text a string / number value contributes (extractVal: a string, or a json.Number's text) as a rune stream
struct TextIter {
  RuneIter it; bool raw; MemBytes *rd; uint32_t p, e;
  __device__ TextIter(MemBytes &r, uint32_t vt, uint32_t vs, uint32_t vl) : it{&r, nullptr, vs + 1, vs + vl - 1}, raw((vt & VT_MASK) == VT_NUM), rd(&r), p(vs), e(vs + vl) {}
  __device__ int next() { if (raw) return p < e ? (int)rd->at(p++) : -1; return it.next(); }
};
__device__ __forceinline__ int b64v(int c) {
  if (c >= 'A' && c <= 'Z') return c - 'A';
  if (c >= 'a' && c <= 'z') return c - 'a' + 26;
  if (c >= '0' && c <= '9') return c - '0' + 52;
  return c == '+' ? 62 : c == '/' ? 63 : -1;
}
// base64.StdEncoding.DecodeString over the decoded text: '\r' / '\n' skipped, padding required.  Bytes go to sink(b).
// Returns the decoded length, or -1 on a CorruptInputError.
template <class S> __device__ int b64_decode(TextIter t, S sink) {
  int q0 = 0, q1 = 0, q2 = 0, nq = 0, pad = 0, n = 0;
  for (int c; (c = t.next()) >= 0;) {
    if (c == '\r' || c == '\n') continue;
    if (c == '=') { if (++pad > 2) return -1; continue; }
    if (pad) return -1;
    const int v = c < 0x80 ? b64v(c) : -1;
    if (v < 0) return -1;
    if (nq == 0) q0 = v; else if (nq == 1) q1 = v; else if (nq == 2) q2 = v;
    else { sink((uint32_t)(q0 << 2 | q1 >> 4) & 0xFF); sink((uint32_t)(q1 << 4 | q2 >> 2) & 0xFF); sink((uint32_t)(q2 << 6 | v) & 0xFF); n += 3; nq = -1; }
    nq++;
  }
  if (nq == 1 || (nq == 0 && pad) || (nq == 2 && pad != 2) || (nq == 3 && pad != 1)) return -1;
  if (nq >= 2) { sink((uint32_t)(q0 << 2 | q1 >> 4) & 0xFF); n++; }
  if (nq == 3) { sink((uint32_t)(q1 << 4 | q2 >> 2) & 0xFF); n++; }
  return n;
}
// typeutil.Base64ToNumeric into sink (text).  0 ok, 1 error (base64), 2 host (the reference panics, or wider than DEC_MAX)
template <class S> __device__ int base64_to_numeric(TextIter t, int scale, S &o) {
  uint8_t buf[DEC_MAX];
  int bn = 0; bool over = false;
  const int n = b64_decode(t, [&](uint32_t b) { if (bn < DEC_MAX) buf[bn++] = (uint8_t)b; else over = true; });
  if (n < 0) return 1;
  if (n == 0 || over) return 2;                                   // isHighestBitSet(in[0]) on an empty slice panics
  const bool neg = buf[0] & 0x80;
  if (neg) {                                                       // makeNegativeNum
    for (int i = 0; i < bn; i++) buf[i] = (uint8_t)~buf[i];
    for (int i = bn - 1; i >= 0; i--) if (++buf[i]) break;
  }
  char dig[DEC_MAX * 3 - 36];                                      // 64 bytes < 10^155
  int nd = 0, first = 0;
  while (first < bn && buf[first] == 0) first++;
  while (first < bn) {
    uint32_t rem = 0;
    for (int i = first; i < bn; i++) { const uint32_t cur = rem * 256 + buf[i]; buf[i] = (uint8_t)(cur / 10); rem = cur % 10; }
    dig[nd++] = (char)('0' + rem);
    while (first < bn && buf[first] == 0) first++;
  }
  if (nd == 0) { o.put('0'); return 0; }                           // "0": no sign, no scale
  if (scale < 0) return 2;                                         // the slicing panics
  if (neg) o.put('-');
  if (scale == 0) { for (int i = nd - 1; i >= 0; i--) o.put((uint32_t)dig[i]); return 0; }
  int len = nd, zeros = 0;
  if (scale > len) { zeros = scale - len + 1; len += zeros; }
  // resultStr[0:len-scale] + "." + resultStr[len-scale:]
  for (int i = 0; i < len; i++) {
    if (i == len - scale) o.put('.');
    o.put(i < zeros ? (uint32_t)'0' : (uint32_t)dig[nd - 1 - (i - zeros)]);
  }
  return 0;
}
// A member of a small struct value (Point, VariableScaleDecimal): the LAST member whose decoded key equals `lit`.
__device__ bool struct_member(MemBytes &rd, uint32_t vs, uint32_t vl, const char *lit, uint32_t n, uint32_t *ms, uint32_t *ml, uint32_t *mt) {
  uint32_t pos = vs + 1; const uint32_t end = vs + vl;
  bool found = false;
  auto skip_ws = [&]() { while (pos < end && is_ws(rd.at(pos))) pos++; };
  skip_ws();
  if (pos < end && rd.at(pos) == '}') return false;
  for (;;) {
    skip_ws();
    const uint32_t ks = pos;
    scan_string(rd, pos, end);
    const uint32_t ke = pos;
    skip_ws(); pos++; skip_ws();
    const uint32_t s = pos; uint32_t vt = 0;
    skip_value(rd, pos, end, vt);
    if (key_is(rd, ks, ke, lit, n) == 1) { *ms = s; *ml = pos - s; *mt = vt; found = true; }
    skip_ws();
    if (pos >= end || rd.at(pos) != ',') break;
    pos++;
  }
  return found;
}
// fmt %v of one Point coordinate: a json.Number's text, a string, <nil>, true / false.  false: a container (host)
template <class S> __device__ bool point_part(S &o, MemBytes &rd, uint32_t s, uint32_t l, uint32_t vt) {
  switch (vt & VT_MASK) {
    case VT_NUM: for (uint32_t i = 0; i < l; i++) o.put(rd.at(s + i)); return true;
    case VT_STR: emit_unquoted(o, rd, s, l); return true;
    case VT_NULL: { const char *t = "<nil>"; for (int i = 0; i < 5; i++) o.put((uint32_t)t[i]); return true; }
    case VT_TRUE: { const char *t = "true"; for (int i = 0; i < 4; i++) o.put((uint32_t)t[i]); return true; }
    case VT_FALSE: { const char *t = "false"; for (int i = 0; i < 5; i++) o.put((uint32_t)t[i]); return true; }
  }
  return false;
}
// The text of a DECIMAL / POINT / VSD value into o.  0 ok, 1 error, 2 host.
template <class S> __device__ int render_text(S &o, MemBytes &rd, const FRecv &f, uint32_t vtr, uint32_t vs, uint32_t vl) {
  const uint32_t vt = vtr & VT_MASK;
  if (f.op == TFGPU_DBZ_DECIMAL) {
    if (vt != VT_STR && vt != VT_NUM) return 1;
    if (f.scale == INT32_MIN) return 1;                            // strconv.Atoi(parameters.scale) fails
    return base64_to_numeric(TextIter(rd, vt, vs, vl), f.scale, o);
  }
  if (vt != VT_OBJ) return 2;                                      // in.(map[string]interface{}) panics
  if (f.op == TFGPU_DBZ_POINT) {
    uint32_t xs, xl, xt, ys, yl, yt;
    if (!struct_member(rd, vs, vl, "x", 1, &xs, &xl, &xt) || !struct_member(rd, vs, vl, "y", 1, &ys, &yl, &yt)) return 1;
    o.put('(');
    if (!point_part(o, rd, xs, xl, xt)) return 2;
    o.put(',');
    if (!point_part(o, rd, ys, yl, yt)) return 2;
    o.put(')');
    return 0;
  }
  uint32_t as, al, at, ss, sl, stt;
  if (!struct_member(rd, vs, vl, "value", 5, &as, &al, &at)) return 1;
  if ((at & VT_MASK) != VT_STR) return 2;                          // .(string) panics
  int scale = 0;
  if (struct_member(rd, vs, vl, "scale", 5, &ss, &sl, &stt)) {
    if ((stt & VT_MASK) != VT_NUM) return 2;                       // .(json.Number) panics
    int64_t x;
    if (!number_int64(rd, ss, sl, &x)) return 1;
    if (x > 2147483647ll || x < -2147483648ll) return 2;
    scale = (int)x;
  }
  return base64_to_numeric(TextIter(rd, at, as, al), scale, o);
}
struct NullSink { __device__ __forceinline__ void put(uint32_t) {} };
__device__ bool is_unavailable(MemBytes &rd, uint32_t vs, uint32_t vl) {
  const char *w = "__debezium_unavailable_value";
  RuneIter it{&rd, nullptr, vs + 1, vs + vl - 1};
  for (int i = 0; i < 28; i++) if (it.next() != (int)w[i]) return false;
  return it.next() < 0;
}
// receiveField for one present value: TFGPU_ROW_OK, TFGPU_ROW_DBZ_FIELD or TFGPU_ROW_HOST_FALLBACK
__device__ int check_value(const Params &p, MemBytes &rd, const FRecv &f, uint32_t vtr, uint32_t vs, uint32_t vl) {
  const uint32_t vt = vtr & VT_MASK;
  if (vt == VT_NULL) return TFGPU_ROW_OK;
  if (vt == VT_STR && vl >= 30 && is_unavailable(rd, vs, vl)) return TFGPU_ROW_HOST_FALLBACK;  // absent: ragged ColumnNames
  switch (f.op) {
    case TFGPU_DBZ_INT8: case TFGPU_DBZ_INT16: case TFGPU_DBZ_INT32: case TFGPU_DBZ_INT64: {
      int64_t x;
      return vt == VT_NUM && number_int64(rd, vs, vl, &x) ? TFGPU_ROW_OK : TFGPU_ROW_DBZ_FIELD;
    }
    case TFGPU_DBZ_BOOLEAN: return (vt == VT_TRUE || vt == VT_FALSE) ? TFGPU_ROW_OK : TFGPU_ROW_DBZ_FIELD;
    case TFGPU_DBZ_FLOAT64: {
      if (vt != VT_NUM) return TFGPU_ROW_DBZ_FIELD;
      tf::Field fl{&rd, vs, vl};
      double d;
      const int rc = parse_float_go(fl, 0, vl, p.p10, p.p128, &d);
      return rc == 0 ? TFGPU_ROW_OK : rc == 3 ? TFGPU_ROW_HOST_FALLBACK : TFGPU_ROW_DBZ_FIELD;
    }
    case TFGPU_DBZ_STRING: return (vt == VT_STR || vt == VT_NUM) ? TFGPU_ROW_OK : TFGPU_ROW_DBZ_FIELD;
    case TFGPU_DBZ_BYTES:
      if (vt != VT_STR && vt != VT_NUM) return TFGPU_ROW_DBZ_FIELD;
      return b64_decode(TextIter(rd, vt, vs, vl), [](uint32_t) {}) >= 0 ? TFGPU_ROW_OK : TFGPU_ROW_DBZ_FIELD;
    case TFGPU_DBZ_DECIMAL: case TFGPU_DBZ_POINT: case TFGPU_DBZ_VSD: {
      NullSink s;
      const int rc = render_text(s, rd, f, vtr, vs, vl);
      return rc == 0 ? TFGPU_ROW_OK : rc == 1 ? TFGPU_ROW_DBZ_FIELD : TFGPU_ROW_HOST_FALLBACK;
    }
  }
  return TFGPU_ROW_HOST_FALLBACK;
}

// ---- Receiver.receive for one message ---------------------------------------------------------------------------------
__device__ int parse_message(const Params &p, int64_t m) {
  const tfgpu_dbz_frame &fr = p.frames[m];
  if (fr.payload_len == 0) return TFGPU_ROW_DBZ_PAYLOAD;             // Decode of no bytes: EOF
  MemBytes rd(p.data);
  uint32_t pos = (uint32_t)fr.payload_start; const uint32_t end = pos + fr.payload_len;
  auto skip_ws = [&]() { while (pos < end && is_ws(rd.at(pos))) pos++; };
  const uint32_t c0 = rd.at(pos);
  if (c0 == 'n') return TFGPU_ROW_DBZ_OP;                            // null: the zero Payload, Op == ""
  if (c0 != '{') return TFGPU_ROW_DBZ_PAYLOAD;                       // UnmarshalTypeError
  // members of interest: the last occurrence wins
  enum { M_AFTER, M_BEFORE, M_OP, M_SOURCE, M_TS, M_N };
  uint32_t ms_[M_N], ml_[M_N], mt_[M_N];
  for (int i = 0; i < M_N; i++) { ms_[i] = 0; ml_[i] = 0; mt_[i] = VT_ABSENT; }
  bool fold = false, dup = false;  // dup: a struct key repeats — encoding/json decodes every occurrence into the same field (maps and structs merge): host
  // walks the members of the object at pos ('{'); calls on(ks, ke, vs, vl, vt) for each
  auto members = [&](auto on) {
    pos++;
    skip_ws();
    if (rd.at(pos) == '}') { pos++; return; }
    for (;;) {
      skip_ws();
      const uint32_t ks = pos;
      scan_string(rd, pos, end);
      const uint32_t ke = pos;
      skip_ws(); pos++; skip_ws();
      const uint32_t vs = pos; uint32_t vt = 0;
      skip_value(rd, pos, end, vt);
      on(ks, ke, vs, pos - vs, vt);
      skip_ws();
      if (rd.at(pos) == ',') { pos++; continue; }
      pos++;  // '}'
      return;
    }
  };
  bool seen_tx = false;
  static const char *const TOP[M_N] = {"after", "before", "op", "source", "ts_ms"};
  const uint32_t TOPN[M_N] = {5, 6, 2, 6, 5};
  members([&](uint32_t ks, uint32_t ke, uint32_t vs, uint32_t vl, uint32_t vt) {
    for (int i = 0; i < M_N; i++) {
      const int k = key_is(rd, ks, ke, TOP[i], TOPN[i]);
      if (k) { if ((mt_[i] & VT_MASK) != VT_ABSENT) dup = true; ms_[i] = vs; ml_[i] = vl; mt_[i] = vt; if (k == 2) fold = true; return; }
    }
    if (const int k = key_is(rd, ks, ke, "transaction", 11)) { if (k == 2) fold = true; if (seen_tx) dup = true; seen_tx = true; }
  });
  bool bad = false;
  auto vt_of = [&](int i) { return mt_[i] & VT_MASK; };
  if (vt_of(M_AFTER) != VT_ABSENT && vt_of(M_AFTER) != VT_OBJ && vt_of(M_AFTER) != VT_NULL) bad = true;
  if (vt_of(M_BEFORE) != VT_ABSENT && vt_of(M_BEFORE) != VT_OBJ && vt_of(M_BEFORE) != VT_NULL) bad = true;
  if (vt_of(M_OP) != VT_ABSENT && vt_of(M_OP) != VT_STR && vt_of(M_OP) != VT_NULL) bad = true;
  if (vt_of(M_TS) == VT_NUM) { uint64_t x; if (!lit_uint(rd, ms_[M_TS], ml_[M_TS], 64, &x)) bad = true; }
  else if (vt_of(M_TS) != VT_ABSENT && vt_of(M_TS) != VT_NULL) bad = true;
  uint64_t lsn = 0, ts = 0, tx = 0, th = 0x6A09E667F3BCC908ull;
  uint32_t sch_s = 0, sch_n = 0, tab_s = 0, tab_n = 0;
  if (vt_of(M_SOURCE) == VT_OBJ) {
    const uint32_t save = pos;
    pos = ms_[M_SOURCE];
    uint32_t seen = 0;
    members([&](uint32_t ks, uint32_t ke, uint32_t vs, uint32_t vl, uint32_t vtr) {
      const uint32_t vt = vtr & VT_MASK;
      static const char *const STRS[8] = {"connector", "db", "name", "sequence", "snapshot", "version", "schema", "table"};
      const uint32_t STRN[8] = {9, 2, 4, 8, 8, 7, 6, 5};
      for (int i = 0; i < 8; i++) {
        const int k = key_is(rd, ks, ke, STRS[i], STRN[i]);
        if (!k) continue;
        if (k == 2) fold = true;
        if (seen & (1u << i)) dup = true;
        seen |= 1u << i;
        if (vt == VT_STR) { if (i == 6) { sch_s = vs; sch_n = vl; } if (i == 7) { tab_s = vs; tab_n = vl; } }
        else if (vt != VT_NULL) bad = true;
        return;
      }
      int k;
      if ((k = key_is(rd, ks, ke, "lsn", 3))) { if (seen & 0x100u) dup = true; seen |= 0x100u; if (k == 2) fold = true; if (vt == VT_NUM) { if (!lit_uint(rd, vs, vl, 64, &lsn)) bad = true; } else if (vt != VT_NULL) bad = true; return; }
      if ((k = key_is(rd, ks, ke, "ts_ms", 5))) { if (seen & 0x200u) dup = true; seen |= 0x200u; if (k == 2) fold = true; if (vt == VT_NUM) { if (!lit_uint(rd, vs, vl, 64, &ts)) bad = true; } else if (vt != VT_NULL) bad = true; return; }
      if ((k = key_is(rd, ks, ke, "txId", 4))) { if (seen & 0x400u) dup = true; seen |= 0x400u; if (k == 2) fold = true; if (vt == VT_NUM) { if (!lit_uint(rd, vs, vl, 32, &tx)) bad = true; } else if (vt != VT_NULL) bad = true; return; }
      if ((k = key_is(rd, ks, ke, "xmin", 4))) { if (seen & 0x800u) dup = true; seen |= 0x800u; if (k == 2) fold = true; int64_t x; if (vt == VT_NUM) { if (!number_int64(rd, vs, vl, &x)) bad = true; } else if (vt != VT_NULL) bad = true; return; }
    });
    pos = save;
  } else if (vt_of(M_SOURCE) != VT_ABSENT && vt_of(M_SOURCE) != VT_NULL) bad = true;
  if (dup) return TFGPU_ROW_HOST_FALLBACK;
  if (bad) return TFGPU_ROW_DBZ_PAYLOAD;
  if (fold) return TFGPU_ROW_HOST_FALLBACK;
  // opToKind
  int kind = -1;
  if (vt_of(M_OP) == VT_STR) {
    RuneIter it{&rd, nullptr, ms_[M_OP] + 1, ms_[M_OP] + ml_[M_OP] - 1};
    const int r = it.next();
    if (r >= 0 && it.next() < 0) kind = (r == 'c' || r == 'r') ? TFGPU_K_INSERT : r == 'u' ? TFGPU_K_UPDATE : r == 'd' ? TFGPU_K_DELETE : -1;
  }
  if (kind < 0) return TFGPU_ROW_DBZ_OP;
  if (p.schema_code != TFGPU_ROW_OK) return p.schema_code;        // receiveSchema's fate, decided by the host for the whole schema
  p.kind[m] = (uint8_t)kind;
  p.lsn[m] = lsn; p.ts[m] = ts; p.txid[m] = (uint32_t)tx;
  {  // TableID of the item: the decoded source.schema / source.table
    RuneIter a{&rd, nullptr, sch_s + 1, sch_s + (sch_n ? sch_n - 1 : 1)};
    if (sch_n) for (int r; (r = a.next()) >= 0;) th = mix64(th, (uint64_t)r);
    th = mix64(th, 0xFFFFFFFFull);
    RuneIter b{&rd, nullptr, tab_s + 1, tab_s + (tab_n ? tab_n - 1 : 1)};
    if (tab_n) for (int r; (r = b.next()) >= 0;) th = mix64(th, (uint64_t)r);
    p.tabhash[m] = th;
    p.tab_s[m] = sch_s; p.tab_n[m] = sch_n; p.tab_s[p.nmsg + m] = tab_s; p.tab_n[p.nmsg + m] = tab_n;
  }
  // the values map: `before` for Delete, else `after`; every schema field must be in it
  const int vi = kind == TFGPU_K_DELETE ? M_BEFORE : M_AFTER;
  if (vt_of(vi) == VT_OBJ) {
    const uint32_t save = pos;
    pos = ms_[vi];
    int ordinal = 0;  // producers write the members in schema order: the field of the same ordinal is tried first (verified, a hint only)
    members([&](uint32_t ks, uint32_t ke, uint32_t vs, uint32_t vl, uint32_t vt) {
      const int guess = ordinal++;
      if (guess < p.nfields) {
        const FRecv &g = p.fields[guess];
        if (key_eq_name(rd, ks, ke, p.names + g.name_off, g.name_len)) { const int64_t i = (int64_t)guess * p.nmsg + m; p.vstart[i] = vs; p.vlen[i] = vl; p.vtype[i] = (uint8_t)vt; return; }
      }
      int lo = 0, hi = p.nfields - 1;
      while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const FRecv &f = p.fields[p.sorted[mid]];
        RuneIter a{&rd, nullptr, ks + 1, ke - 1}, b{nullptr, p.names + f.name_off, 0, f.name_len};
        const int c = rune_compare(a, b);
        if (c == 0) { const int64_t i = (int64_t)p.sorted[mid] * p.nmsg + m; p.vstart[i] = vs; p.vlen[i] = vl; p.vtype[i] = (uint8_t)vt; return; }
        if (c < 0) hi = mid - 1; else lo = mid + 1;
      }
    });
    pos = save;
  }
  for (int j = 0; j < p.nfields; j++) {
    const int64_t i = (int64_t)j * p.nmsg + m;
    const uint32_t vtr = p.vtype[i];
    if ((vtr & VT_MASK) == VT_ABSENT) return TFGPU_ROW_DBZ_FIELD;  // "unable to get field %s from 'after'"
    if ((vtr & VT_MASK) == VT_STR && p.vlen[i] >= 30 && p.fields[j].optional && is_unavailable(rd, p.vstart[i], p.vlen[i])) {
      // receiveField's isAbsent (receiver_engine.go:143-148, receiver.go:98-105): the item does not list the column — an ABSENT cell
      // (DColumn::absent); to the cell kernels it reads as a nil.  (Under a key field the OldKeys turn ragged too: check_value hands that to the host.)
      p.vtype[i] = (uint8_t)(VT_NULL | VT_CANON);
      *p.toast = 1u;
      continue;
    }
    const int rc = check_value(p, rd, p.fields[j], vtr, p.vstart[i], p.vlen[i]);
    if (rc != TFGPU_ROW_OK) return rc;
  }
  return TFGPU_ROW_OK;
}
__device__ void registry_walk(const Params &p, int64_t e);  // (schema-registry framed events, below)
__global__ void __launch_bounds__(128) dbz_parse(Params p) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.nmsg) return;
  const bool claimed = p.tent && p.tent[m];
  if (claimed) { if (p.tent[m] == 2) registry_walk(p, m); else unpack_message(p, m); p.tent[m] = 0; }  // a claimed span: IncludeSchema.Unpack's walk (or the registry form's) first
  const tfgpu_dbz_frame &fr = p.frames[m];
  int st;
  if (claimed && fr.code != TFGPU_ROW_OK) st = fr.code;   // (the host saw code OK for the claim: the walk's verdict is reported here)
  else if (fr.code != TFGPU_ROW_OK || fr.schema_hash[0] != p.h0 || fr.schema_hash[1] != p.h1) st = ST_SKIP;
  else st = parse_message(p, m);
  p.status[m] = (uint8_t)st;
  if (st == TFGPU_ROW_OK) atomicMin(p.first_ok, (unsigned long long)m);
}
#include "tf_dbzquick.inc"

// The segment map of dbz_parse_quick from the group's first payload (host code: 1-2 KB of text, once per call).  false: the
// payload is not of the shape the quick kernel reads (blanks, arrays, nested values, escaped or repeated keys, `after` in front of
// `before`, a values map that is not the schema's fields in schema order): the walker takes every message, as before.
static bool build_quick_map(const std::string &s, const std::vector<std::string> &names, const std::vector<int> &ops, DqMap &M) {
  std::memset(&M, 0, sizeof M);
  M.slot_ent[0] = M.slot_ent[1] = -1; M.opent = ~0u;
  const int F = (int)names.size();
  if (F < 1 || F > DQ_F) return false;
  size_t kr = 0;
  auto add_text = [&](const std::string &t, uint16_t &ko, uint16_t &kn) {
    const size_t room = (t.size() + 7) / 8 * 8;
    if (kr + room + 8 > (size_t)DQ_KREF || t.size() > 0x7FFF) return false;
    ko = (uint16_t)kr; kn = (uint16_t)t.size();
    std::memcpy(M.kref + kr, t.data(), t.size());
    kr += room;
    return true;
  };
  std::vector<std::string> fkey((size_t)F);
  for (int f = 0; f < F; f++) {
    for (unsigned char c : names[(size_t)f]) if (c < 0x20 || c >= 0x7F || c == '"' || c == '\\') return false;  // (a producer may spell such a key with or without escapes)
    fkey[(size_t)f] = "\"" + names[(size_t)f] + "\":";
    M.fkn[f] = (uint16_t)fkey[(size_t)f].size();
  }
  const size_t n = s.size();
  size_t i = 0;
  auto lower = [](std::string k) { for (auto &c : k) if (c >= 'A' && c <= 'Z') c = (char)(c + 32); return k; };
  // a key literal at i: plain ASCII, no escapes; leaves i behind the ':'
  auto key = [&](std::string &k) {
    if (i >= n || s[i] != '"') return false;
    size_t e = i + 1;
    while (e < n && s[e] != '"') { const unsigned char c = (unsigned char)s[e]; if (c < 0x20 || c >= 0x7F || c == '\\') return false; e++; }
    if (e + 1 >= n || s[e + 1] != ':') return false;
    k = s.substr(i + 1, e - i - 1);
    i = e + 2;
    return true;
  };
  // a scalar at i (the payload is valid JSON: its extent is all that is asked); leaves i behind it
  auto scalar = [&]() {
    if (i >= n) return false;
    const char c = s[i];
    if (c == '"') { size_t e = i + 1; while (e < n && s[e] != '"') e += s[e] == '\\' ? 2 : 1; if (e >= n) return false; i = e + 1; return true; }
    if (c == '{' || c == '[') return false;
    size_t e = i;
    while (e < n && (std::isalnum((unsigned char)s[e]) || s[e] == '-' || s[e] == '+' || s[e] == '.')) e++;
    if (e == i) return false;
    i = e;
    return true;
  };
  auto add_ent = [&](const std::string &prefix, uint8_t role, char sep, uint8_t slot) {
    if (M.nent >= (uint32_t)DQ_ENT || (sep != ',' && sep != '}')) return false;
    DqEnt &e = M.ent[M.nent];
    if (!add_text(prefix, e.ko, e.kn)) return false;
    e.role = role; e.sep = (uint8_t)sep; e.slot = slot;
    M.nent++;
    return true;
  };
  // a values map at i ('{'): the schema's fields in schema order, scalars or — Point, VariableScaleDecimal — objects of scalars
  struct RowDesc { std::vector<DqRow> row; std::vector<std::string> text; std::vector<uint8_t> gk, gfirst; };
  auto row_object = [&](RowDesc &d) {
    i++;
    for (int f = 0; f < F; f++) {
      if (s.compare(i, fkey[(size_t)f].size(), fkey[(size_t)f])) return false;
      i += fkey[(size_t)f].size();
      if (i >= n) return false;
      const bool lastf = f + 1 == F;
      if (s[i] == '{') {
        if (ops[(size_t)f] != TFGPU_DBZ_POINT && ops[(size_t)f] != TFGPU_DBZ_VSD) return false;
        if (d.gk.size() >= (size_t)DQ_G) return false;
        const uint8_t g = (uint8_t)d.gk.size();
        d.gfirst.push_back((uint8_t)d.row.size());
        i++;
        if (i < n && s[i] == '}') return false;
        std::string pre = fkey[(size_t)f] + "{";
        int k = 0; bool has_x = false, has_y = false, has_value = false;
        for (;;) {
          const size_t k0 = i;
          std::string k2;
          if (!key(k2)) return false;
          uint8_t role = IR_ANY;
          if (ops[(size_t)f] == TFGPU_DBZ_VSD) { if (k2 == "value") { if (has_value) return false; has_value = true; role = IR_B64; } else if (k2 == "scale") role = IR_SCALE; }
          else { if (k2 == "x") has_x = true; if (k2 == "y") has_y = true; }
          for (const char *nm : {"value", "scale", "x", "y"}) if (lower(k2) == nm && k2 != nm) return false;
          pre += s.substr(k0, i - k0);
          if (!scalar() || i >= n) return false;
          const char c = s[i];
          if (c != ',' && c != '}') return false;
          DqRow r{}; r.kind = k == 0 ? RK_HEAD : RK_INNER; r.field = (uint8_t)f; r.grp = g; r.role = role; r.gb = g; r.last = c == '}' ? 1 : 0;
          d.row.push_back(r); d.text.push_back(pre); pre.clear();
          k++; i++;
          if (c == '}') break;
        }
        if (k > 255 || (ops[(size_t)f] == TFGPU_DBZ_VSD ? !has_value : !(has_x && has_y))) return false;
        // (the walker takes the LAST member of a name; a repeated one sends the map away)
        for (size_t a = d.row.size() - (size_t)k; a < d.row.size(); a++) for (size_t b = a + 1; b < d.row.size(); b++) {
          auto keyof = [&](size_t q) { const std::string &t = d.text[q]; const size_t e = t.rfind("\":"); const size_t b0 = t.rfind('"', e - 1); return t.substr(b0 + 1, e - b0 - 1); };
          if (keyof(a) == keyof(b)) return false;
        }
        d.gk.push_back((uint8_t)k);
        d.row[d.gfirst[g]].last = lastf ? 1 : 0;   // RK_HEAD: what follows it when it is null; whether its object is one member long is gk's to say
        DqRow t{}; t.kind = RK_TAIL; t.field = (uint8_t)f; t.grp = g; t.gb = g; t.last = lastf ? 1 : 0;
        d.row.push_back(t); d.text.push_back("");
        if (i >= n || s[i] != (lastf ? '}' : ',')) return false;
        i++;
      } else {
        if (!scalar() || i >= n || s[i] != (lastf ? '}' : ',')) return false;
        DqRow r{}; r.kind = RK_FIELD; r.field = (uint8_t)f; r.gb = (uint8_t)d.gk.size(); r.last = lastf ? 1 : 0;
        d.row.push_back(r); d.text.push_back(fkey[(size_t)f]);
        i++;
      }
    }
    return true;
  };
  static const char *const TOP[] = {"before", "after", "source", "op", "ts_ms", "transaction"};
  static const char *const SRC[] = {"connector", "db", "name", "sequence", "snapshot", "version", "schema", "table", "lsn", "ts_ms", "txid", "xmin"};
  static const char *const SRC_EXACT[] = {"connector", "db", "name", "sequence", "snapshot", "version", "schema", "table", "lsn", "ts_ms", "txId", "xmin"};
  static const uint8_t SRC_ROLE[] = {DR_SSTR, DR_SSTR, DR_SSTR, DR_SSTR, DR_SSTR, DR_SSTR, DR_SCHEMA, DR_TABLE, DR_LSN, DR_STS, DR_TXID, DR_XMIN};
  if (n < 2 || s[0] != '{') return false;
  i = 1;
  size_t seg = 0;  // where the current segment starts
  uint32_t seen_top = 0;
  RowDesc best;
  for (;;) {
    std::string k;
    if (!key(k)) return false;
    int top = -1;
    for (int t = 0; t < 6; t++) if (lower(k) == TOP[t]) { if (k != TOP[t]) return false; top = t; }   // (a key Go would bind by case folding: the walker sends those to the host)
    if (top >= 0) { if (seen_top & (1u << top)) return false; seen_top |= 1u << top; }
    if (i >= n) return false;
    char sep;
    if (top == 0 || top == 1) {  // a slot: null, or a values map
      if (top == 0 && (seen_top & 2u)) return false;  // `after` in front of `before`
      const std::string prefix = s.substr(seg, i - seg);
      if (s[i] == 'n') { if (s.compare(i, 4, "null")) return false; i += 4; }
      else if (s[i] == '{') {
        RowDesc d;
        if (!row_object(d)) return false;
        if (d.row.size() > best.row.size()) best = std::move(d);   // (with both maps present: the one with more struct fields spelled out)
      } else return false;
      if (i >= n) return false;
      sep = s[i];
      M.slot_ent[top] = (int32_t)M.nent;
      if (!add_ent(prefix, DR_SLOT, sep, (uint8_t)top)) return false;
    } else if (s[i] == '{') {   // source, transaction, anything else: its members flattened, scalars only
      i++;
      if (i < n && s[i] == '}') return false;
      uint32_t seen_src = 0;
      for (;;) {
        std::string k2;
        if (!key(k2)) return false;
        uint8_t role = DR_ANY;
        if (top == 2) for (int t = 0; t < 12; t++) if (lower(k2) == SRC[t]) {
          if (k2 != SRC_EXACT[t] || (seen_src & (1u << t))) return false;
          seen_src |= 1u << t; role = SRC_ROLE[t];
        }
        const std::string prefix = s.substr(seg, i - seg);
        if (!scalar() || i >= n) return false;
        const char c = s[i];
        if (!add_ent(prefix, role, c, 0)) return false;
        i++;
        seg = i;
        if (c == '}') break;
      }
      if (i >= n) return false;
      sep = s[i];
      if (!add_ent("", DR_EMPTY, sep, 0)) return false;  // what lies between that '}' and the separator behind it: nothing
    } else {
      const std::string prefix = s.substr(seg, i - seg);
      const bool is_null = !s.compare(i, 4, "null");
      if (!scalar() || i >= n) return false;
      sep = s[i];
      const uint8_t role = top == 2 ? DR_NULL : top == 3 ? DR_OP : top == 4 ? DR_TS : DR_ANY;
      if (top == 2 && !is_null) return false;
      if (top == 3) M.opent = M.nent;
      if (!add_ent(prefix, role, sep, 0)) return false;
    }
    i++;
    seg = i;
    if (sep == '}') break;
  }
  if (i != n || M.opent == ~0u || best.row.empty() || best.row.size() > (size_t)DQ_ROW) return false;
  for (size_t r = 0; r < best.row.size(); r++) {
    DqRow &R = best.row[r];
    if (!add_text(best.text[r], R.ko, R.kn)) return false;
    M.row[r] = R;
  }
  M.nrow = (uint32_t)best.row.size(); M.ngrp = (uint32_t)best.gk.size();
  for (size_t g = 0; g < best.gk.size(); g++) { M.gk[g] = best.gk[g]; M.gfirst[g] = best.gfirst[g]; }
  M.nf = (uint32_t)F;
  M.valid = 1;
  return true;
}

// a batch carries ONE TableID: good rows of another table than the first good row's are the stock code's
__global__ void __launch_bounds__(256) dbz_table_rule(Params p) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.nmsg) return;
  uint32_t st = p.status[m];
  const unsigned long long f = *p.first_ok;
  if (st == TFGPU_ROW_OK && f < (unsigned long long)p.nmsg && p.tabhash[m] != p.tabhash[f]) { st = TFGPU_ROW_HOST_FALLBACK; p.status[m] = (uint8_t)st; }
  p.keep[m] = st == TFGPU_ROW_OK ? 1u : 0u;
  if (st != TFGPU_ROW_OK && st != ST_SKIP) atomicAdd(p.nerr, 1u);
}
__global__ void __launch_bounds__(256) dbz_row_msgs(Params p) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.nmsg || p.status[m] != TFGPU_ROW_OK) return;
  p.row_msg[p.keep[m]] = (uint32_t)m;
}
// decoded source.schema / source.table of the first good message: [len u32][bytes ≤ 252] twice
__global__ void dbz_table_name(Params p, uint8_t *out) {
  const unsigned long long f = *p.first_ok;
  if (threadIdx.x || f >= (unsigned long long)p.nmsg) return;
  MemBytes rd(p.data);
  for (int k = 0; k < 2; k++) {
    const uint32_t s = p.tab_s[(int64_t)k * p.nmsg + f], n = p.tab_n[(int64_t)k * p.nmsg + f];
    uint8_t *o = out + k * 256;
    uint32_t len = 0;
    if (n) {
      RuneIter it{&rd, nullptr, s + 1, s + n - 1};
      struct Cap { uint8_t *p; uint32_t n; __device__ void put(uint32_t c) { if (n < 252) p[4 + n] = (uint8_t)c; n++; } } cap{o, 0};
      for (int r; (r = it.next()) >= 0;) put_utf8(cap, (uint32_t)r);
      len = cap.n;
    }
    o[0] = (uint8_t)len; o[1] = (uint8_t)(len >> 8); o[2] = (uint8_t)(len >> 16); o[3] = (uint8_t)(len >> 24);
  }
}

// ---- cells --------------------------------------------------------------------------------------------------------------
struct OutCol {
  int32_t op;
  void *values;
  uint32_t *lens;      // text columns: lengths, then offsets [nrows + 1]
  uint8_t *data;
  uint8_t *valid8, *old8;      // one byte per row: ColumnValues holds a non-nil value / OldKeys does
  uint8_t *validity, *old_validity;
};
struct DbzRow { int64_t msg; uint64_t lsn, commit_time; uint32_t id; uint8_t names_form, pad[3]; };
static_assert(sizeof(DbzRow) == sizeof(tfgpu_dbz_row), "row meta layout");

// Three kernels by what a field's receiver costs (the fields of a class are the grid's y, through `flist`): CLS 0 — booleans, integers,
// strings, base64 lengths: a few registers; CLS 1 — float64 (the Eisel-Lemire parse and its 128-bit table); CLS 2 — the rendered logical
// types (decimals as big integers, points, variable-scale decimals).  One kernel for all of them ran every integer cell at the rendered
// types' 120 VGPRs and 160 bytes of scratch (four waves a SIMD for a kernel of dependent random reads).
enum { DC_LIGHT = 0, DC_FLOAT = 1, DC_RENDER = 2 };
__host__ __device__ inline int dbz_field_class(int op) { return op == TFGPU_DBZ_FLOAT64 ? DC_FLOAT : (op >= TFGPU_DBZ_BOOLEAN && op <= TFGPU_DBZ_BYTES) ? DC_LIGHT : DC_RENDER; }
template <int CLS>
__global__ void __launch_bounds__(256) dbz_cell_values(Params p, const OutCol *cols, const int32_t *flist, int32_t *src_row, uint8_t *kinds, uint8_t *old_present8, DbzRow *rows) {
  const int j = flist[blockIdx.y]; const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (the field is the grid's y: a scalar)
  if (r >= p.nrows) return;
  const uint32_t m = p.row_msg[r];
  const OutCol &c = cols[j];
  const FRecv &f = p.fields[j];
  const int64_t i = (int64_t)j * p.nmsg + m;
  const uint32_t vtr = p.vtype[i], vt = vtr & VT_MASK, vs = p.vstart[i], vl = p.vlen[i];
  const int kind = p.kind[m];
  MemBytes rd(p.data);
  if (j == 0) {
    src_row[r] = (int32_t)m; kinds[r] = (uint8_t)kind; old_present8[r] = kind != TFGPU_K_INSERT;
    rows[r] = DbzRow{(int64_t)m, p.lsn[m], p.ts[m] * 1000000ull, p.txid[m], (uint8_t)(kind == TFGPU_K_DELETE), {0, 0, 0}};
  }
  const bool nil = vt == VT_NULL;
  c.valid8[r] = (!nil && kind != TFGPU_K_DELETE) ? 1 : 0;         // a Delete keeps ColumnNames / ColumnValues nil
  if (c.old8) c.old8[r] = (!nil && kind != TFGPU_K_INSERT) ? 1 : 0;
  if constexpr (CLS == DC_FLOAT) {
    double d = 0;
    if (!nil) { tf::Field fl{&rd, vs, vl}; parse_float_go(fl, 0, vl, p.p10, p.p128, &d); }
    ((double *)c.values)[r] = d;
    return;
  } else if constexpr (CLS == DC_RENDER) {
    CountSink s;
    if (!nil) render_text(s, rd, f, vtr, vs, vl);
    c.lens[r] = s.n;
    return;
  } else
  switch (f.op) {
    case TFGPU_DBZ_BOOLEAN: ((uint8_t *)c.values)[r] = vt == VT_TRUE; break;
    case TFGPU_DBZ_INT8: case TFGPU_DBZ_INT16: case TFGPU_DBZ_INT32: case TFGPU_DBZ_INT64: {
      int64_t x = 0;
      if (!nil) number_int64(rd, vs, vl, &x);
      if (f.op == TFGPU_DBZ_INT8) ((int8_t *)c.values)[r] = (int8_t)x;        // Go's truncating conversions
      else if (f.op == TFGPU_DBZ_INT16) ((int16_t *)c.values)[r] = (int16_t)x;
      else if (f.op == TFGPU_DBZ_INT32) ((int32_t *)c.values)[r] = (int32_t)x;
      else ((int64_t *)c.values)[r] = x;
      break;
    }
    case TFGPU_DBZ_STRING:
      if (nil) c.lens[r] = 0;
      else if (vt == VT_NUM) c.lens[r] = vl;
      else if (vtr & VT_PLAIN) c.lens[r] = vl - 2;
      else { CountSink s; emit_unquoted(s, rd, vs, vl); c.lens[r] = s.n; }
      break;
    case TFGPU_DBZ_BYTES: { int n = 0; if (!nil) n = b64_decode(TextIter(rd, vt, vs, vl), [](uint32_t) {}); c.lens[r] = n > 0 ? (uint32_t)n : 0u; break; }
    default: break;  // (a rendered type: CLS 2)
  }
}
// The text cells that are a plain byte range of the message (a string field's string without escapes, or its number token) go
// through tf_segcopy.hpp's destination-centric packing, like the CSV / JSON / SR text columns; the others (escapes, base64,
// rendered logical types) are zero-filled there and written by dbz_cell_text, launched for the columns that hold such cells.
__device__ __forceinline__ uint32_t dbz_plain_src(const Params &p, int j, int64_t r, bool *special) {
  const int64_t i = (int64_t)j * p.nmsg + p.row_msg[r];
  const uint32_t vtr = p.vtype[i], vt = vtr & VT_MASK;
  if (vt == VT_NULL) return SEG_NONE;
  if (p.fields[j].op == TFGPU_DBZ_STRING) {
    if (vt == VT_NUM) return p.vstart[i];
    if (vtr & VT_PLAIN) return p.vstart[i] + 1;
  }
  if (special) *special = true;
  return SEG_NONE;
}
__global__ void __launch_bounds__(256) dbz_mark_special(Params p, const int32_t *text_cols, uint32_t *spec) {
  constexpr int RPT = 16;  // rows per lane (a lane that looks at one cell and leaves makes the launch dispatch-bound)
  const int t = (int)blockIdx.y;
  bool special = false;
  for (int k = 0; k < RPT; k++) {
    const int64_t r = ((int64_t)blockIdx.x * RPT + k) * 256 + threadIdx.x;
    if (r < p.nrows) dbz_plain_src(p, text_cols[t], r, &special);
  }
  if (__any(special) && (threadIdx.x & 63) == 0 && !__atomic_load_n(&spec[t], __ATOMIC_RELAXED)) atomicOr(&spec[t], 1u);  // (a flag that is up is seen by a plain L2 read: no atomic per wave)
}
__global__ void __launch_bounds__(256) dbz_copy_words(Params p, const OutCol *cols, const int32_t *text_cols) {
  __shared__ uint32_t doff[256 + 1];
  __shared__ uint32_t soff[256];
  const int j = text_cols[blockIdx.y];
  const OutCol c = cols[j];
  auto so = [&](int64_t r) { return dbz_plain_src(p, j, r, nullptr); };
  segcopy_run<1>(c.lens, p.nrows, (int64_t)blockIdx.x * 256, p.data, c.data, so, doff, soff);
}
// RENDER == false: strings and base64 (the columns whose receiver is TFGPU_DBZ_STRING / _BYTES); true: the rendered logical types
template <bool RENDER>
__global__ void __launch_bounds__(256) dbz_cell_text(Params p, const OutCol *cols, const int32_t *text_cols, int32_t ntext, int plain_done) {
  const int t = (int)blockIdx.y; const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  const int j = text_cols[t];
  const OutCol &c = cols[j];
  const FRecv &f = p.fields[j];
  const uint32_t m = p.row_msg[r];
  const int64_t i = (int64_t)j * p.nmsg + m;
  const uint32_t vtr = p.vtype[i], vt = vtr & VT_MASK, vs = p.vstart[i], vl = p.vlen[i];
  if (vt == VT_NULL) return;
  if (plain_done && f.op == TFGPU_DBZ_STRING && (vt == VT_NUM || (vtr & VT_PLAIN))) return;  // dbz_copy_words moved it
  MemBytes rd(p.data);
  ByteSink s{c.data + c.lens[r]};
  auto copy_raw = [&](uint32_t a, uint32_t n) {
    uint32_t k = 0;
    for (; k + 8 <= n; k += 8) s.put_word(rd.word(a + k), 8);
    if (k < n) s.put_word(rd.word(a + k) & ((1ull << (8 * (n - k))) - 1), n - k);
  };
  if constexpr (RENDER) render_text(s, rd, f, vtr, vs, vl);
  else if (f.op == TFGPU_DBZ_STRING) {
    if (vt == VT_NUM) copy_raw(vs, vl);
    else if (vtr & VT_PLAIN) copy_raw(vs + 1, vl - 2);
    else emit_unquoted(s, rd, vs, vl);
  } else b64_decode(TextIter(rd, vt, vs, vl), [&](uint32_t b) { s.put(b); });
  s.flush();
}
// rows that do not list a column (see the walker's isAbsent note): one flag byte per (field, row) and a flag per field that has any
__global__ void __launch_bounds__(256) dbz_absent_cells(Params p, int64_t nra, uint8_t *abs8, uint32_t *colflag) {
  const int j = (int)blockIdx.y; const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool a = false;
  if (r < p.nrows) {
    const uint32_t m = p.row_msg[r];
    const uint32_t v = p.vtype[(int64_t)j * p.nmsg + m];
    a = v == (VT_NULL | VT_CANON) && p.kind[m] != TFGPU_K_DELETE;  // (a Delete lists nothing at all)
    abs8[(int64_t)j * nra + r] = a ? 1 : 0;
  }
  if (__any(a) && (threadIdx.x & 63) == 0) atomicOr(&colflag[j], 1u);
}
struct PackJob { const uint8_t *bytes; uint8_t *bits; };
// every byte-per-row flag array of a batch → its bitmap, in one launch (grid y = the array)
__global__ void __launch_bounds__(256) dbz_pack_bits_all(const PackJob *jobs, int64_t n) {
  const PackJob j = jobs[blockIdx.y];
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b * 8 >= n) return;
  uint32_t v = 0;
  for (int k = 0; k < 8 && b * 8 + k < n; k++) v |= (uint32_t)(j.bytes[b * 8 + k] & 1u) << k;
  j.bits[b] = (uint8_t)v;
}

__global__ void dbz_gather_totals(const uint32_t *lens_all, int64_t seg_stride, int64_t nrows, int ntext, uint32_t *out) {
  for (int t = threadIdx.x; t < ntext; t += blockDim.x) out[t] = lens_all[(int64_t)t * seg_stride + nrows];
}

// ---- schema-registry framed events (NewReceiver with a registry client) ---------------------------------------------------------
// DebeziumImpl.DoOne (pkg/parsers/registry/debezium/engine/parser.go:33-57) cuts a Kafka message into events 0x00 | schema id | payload
// exactly as the Confluent-SR parser cuts its frames (tfgpu_sr_frames); SchemaRegistry.Unpack (pkg/debezium/unpacker/schema_registry.go:
// 18-34) hands Receiver.receive the bytes behind the five-byte prefix, and UnmarshalPayload (common/debezium_schema.go:62-68) runs a
// json.Decoder over them: white space, ONE value — an object or array ends at its bracket whatever follows, any other value must be
// followed by white space or the end — then the struct binding parse_message restates.  This kernel is that first half: the span of
// the value for tfgpu_debezium_parse, or the event's fate.  A message too short for buf[5:] is the reference's panic (host).
// the walk over one event's bytes behind the prefix: [start, start + len)
__device__ void registry_payload(const Params &p, tfgpu_dbz_frame &fr, uint32_t start, uint32_t len) {
  MemBytes rd(p.data);
  fr.payload_start = start; fr.payload_len = 0; fr.code = TFGPU_ROW_OK;
  uint32_t pos = start; const uint32_t end = start + len;
  while (pos < end && is_ws(rd.at(pos))) pos++;
  if (pos >= end) return;  // (nothing but white space: Decode's io.EOF — payload_len stays 0, TFGPU_ROW_DBZ_PAYLOAD in the parse)
  const uint32_t vs = pos;
  uint32_t vt = 0;
  const int rc = skip_value(rd, pos, end, vt);
  const uint32_t kind = vt & VT_MASK;
  if (rc == 2) fr.code = TFGPU_ROW_HOST_FALLBACK;
  else if (rc != 0) fr.code = TFGPU_ROW_DBZ_PAYLOAD;
  else if (kind != VT_OBJ && kind != VT_ARR && pos < end && !is_ws(rd.at(pos))) fr.code = TFGPU_ROW_DBZ_PAYLOAD;  // "invalid character after top-level value"
  else { fr.payload_start = vs; fr.payload_len = pos - vs; }
}
// a claimed span nobody proved (Params::tent[e] == 2): the walk after all.  The event's bytes are its slot behind the prefix.
__device__ void registry_walk(const Params &p, int64_t e) {
  tfgpu_dbz_frame fr = p.frames[e];
  const uint32_t a = p.ms[e] + 5u, z = p.ms[e + 1];
  registry_payload(p, fr, a, z - a);
  p.frames[e] = fr;
}
__global__ void __launch_bounds__(128) dbz_registry_frames(Params p, const tfgpu_sr_frame *ev) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.nmsg) return;
  const tfgpu_sr_frame f = ev[e];
  tfgpu_dbz_frame fr;
  fr.schema_start = 0; fr.schema_len = 0; fr.payload_start = f.start; fr.payload_len = 0; fr.reserved = 0;
  fr.schema_hash[0] = f.schema_id; fr.schema_hash[1] = TFGPU_DBZ_REGISTRY_HASH;
  fr.code = TFGPU_ROW_OK;
  MemBytes rd(p.data);
  if (f.code) {  // DoOne looks at the magic byte first (parser.go:37-39), then slices buf[5:]
    fr.code = f.code;
    if (f.code == TFGPU_ROW_SR_SHORT) fr.code = f.len && rd.at((uint32_t)f.start) != 0 ? TFGPU_ROW_SR_MAGIC : TFGPU_ROW_HOST_FALLBACK;
    p.frames[e] = fr;
    return;
  }
  if (p.tent) {  // blank* '{' … '}' blank*: the span between the braces is a CLAIM — dbz_parse_quick proves it an object when it takes the
                 // event (every byte compared or validated); for an event it does not take the walker runs registry_walk first
    uint32_t a = (uint32_t)f.start, z = a + f.len;
    while (a < z && is_ws(rd.at(a))) a++;
    while (z > a && is_ws(rd.at(z - 1))) z--;
    if (z >= a + 2 && rd.at(a) == '{' && rd.at(z - 1) == '}') {
      fr.payload_start = a; fr.payload_len = z - a;
      p.frames[e] = fr;
      p.tent[e] = 2;
      return;
    }
  }
  registry_payload(p, fr, (uint32_t)f.start, f.len);
  p.frames[e] = fr;
}

static inline unsigned nblk(int64_t n, int t) { return (unsigned)std::max<int64_t>(1, (n + t - 1) / t); }

struct Staged { Buf bytes, ms; Params p{}; };
static void stage(Staged &s, const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs) {
  Context &cx = ctx();
  if (len >= 0xFFFFFFF0ull) throw Error(TFGPU_ERR_UNSUPPORTED, "debezium: batch must be < 4 GiB (32-bit offsets)");
  if (mem == TFGPU_MEM_HOST) {
    s.bytes = dalloc(len + 64);
    h2d(s.bytes->p, bytes, len);
    TF_HIP(hipMemsetAsync((char *)s.bytes->p + len, 0, 64, cx.stream));
    s.p.data = ptr<uint8_t>(s.bytes);
  } else {
    s.p.data = (const uint8_t *)bytes;
    if (reinterpret_cast<uintptr_t>(s.p.data) & 15) throw Error(TFGPU_ERR_INVALID, "debezium: device buffer must be 16-byte aligned");
  }
  const int64_t nmsg = msgs ? msgs->nmsg : 1;
  if (nmsg < 0 || (msgs && nmsg > 0 && !msgs->start)) throw Error(TFGPU_ERR_INVALID, "debezium: bad message batch");
  std::vector<uint32_t> ms((size_t)nmsg + 1);
  if (msgs) {
    for (int64_t m = 0; m <= nmsg; m++) {
      if (msgs->start[m] > len || (m && msgs->start[m] < msgs->start[m - 1])) throw Error(TFGPU_ERR_INVALID, "debezium: message offsets must be ascending and inside the buffer");
      ms[(size_t)m] = (uint32_t)msgs->start[m];
    }
  } else { ms[0] = 0; ms[1] = (uint32_t)len; }
  s.ms = dalloc(ms.size() * 4 + 16);
  h2d(s.ms->p, ms.data(), ms.size() * 4);
  s.p.ms = ptr<uint32_t>(s.ms); s.p.nmsg = nmsg;
}

}  // namespace dbz
}  // namespace tf

namespace tf { namespace sr { Buf last_frames_device(const tfgpu_sr_frame *host, int64_t n, const void *bytes); } }  // tf_srjson.hip
using namespace tf;

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

// The frames tfgpu_debezium_unpack just computed, still in HBM: the receiver (tf_dbzrecv.cpp) hands the host copy straight back to
// tfgpu_debezium_parse, unmodified — it says so (dbz_trust_frames) and the 5 MB of frames per 2^17 messages are neither checked
// nor uploaded again.  Any other caller's frames are checked and uploaded as before.
namespace tf { namespace dbz {
// `uniform`: every frame is OK, carries the known schema hash and a payload the tile parser can stage — the receiver asked for
// that answer instead of the frames (dbz_lazy_frames): only frame 0 and its payload's bytes came down, with the widest payload in chunks
struct FrameCache { const tfgpu_dbz_frame *host = nullptr; int64_t nmsg = -1; const void *bytes = nullptr; Buf dev, tent; bool uniform = false; uint32_t max_chunks = 0; std::string pay0; };
static thread_local FrameCache g_frames;
static thread_local bool g_trust = false, g_tentative = false, g_last_quick = false, g_lazy = false;
void dbz_trust_frames(bool on) { g_trust = on; }
// the receiver, around its unpack call: when the batch is uniform (see FrameCache) `frames` receives frame 0 only — 48 bytes instead
// of 5 MB per 2^17 messages, and none of the three host loops over them runs
static thread_local uint64_t g_lazy_h0 = 0, g_lazy_h1 = 0;
// (h0, h1): the schema hash every frame must carry — tfgpu_debezium_registry_frames has no other way to know it; the cached-prefix call uses its prefix's
void dbz_lazy_frames(bool on, uint64_t h0, uint64_t h1) { g_lazy = on; g_lazy_h0 = h0; g_lazy_h1 = h1; }
bool dbz_last_unpack_uniform() { return g_frames.uniform; }
struct FrameStats { uint32_t n_odd, max_chunks, pay_len, pad; tfgpu_dbz_frame f0; uint8_t pay[JQ_BYTES]; };
// one thread per frame: how many are not (OK, hash (h0, h1), payload of 2 .. JQ_BYTES - 32 bytes), the widest payload in the 16-byte
// chunks the tile parser stages; workgroup 0 also leaves frame 0 and its payload's bytes
__global__ void __launch_bounds__(256) dbz_frame_stats(const tfgpu_dbz_frame *fr, int64_t nmsg, uint64_t h0, uint64_t h1, const uint8_t *data, FrameStats *out) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t odd = 0, chunks = 0;
  if (m < nmsg) {
    const tfgpu_dbz_frame f = fr[m];
    const bool ok = f.code == TFGPU_ROW_OK && f.schema_hash[0] == h0 && f.schema_hash[1] == h1 && f.payload_len >= 2 && f.payload_len <= (uint32_t)JQ_BYTES - 32u;
    odd = ok ? 0u : 1u;
    const uint32_t ps = (uint32_t)f.payload_start, pl = f.payload_len;
    if (ok) chunks = (((ps + pl + 15u) & ~15u) - (ps & ~15u)) >> 4;
  }
  const uint64_t any_odd = __ballot(odd != 0);
  for (int d = 32; d; d >>= 1) chunks = max(chunks, (uint32_t)__shfl_xor((int)chunks, d, 64));
  if ((threadIdx.x & 63) == 0) { if (any_odd) atomicAdd(&out->n_odd, (uint32_t)__popcll(any_odd)); atomicMax(&out->max_chunks, chunks); }
  if (blockIdx.x == 0 && nmsg > 0) {
    const tfgpu_dbz_frame f = fr[0];
    if (threadIdx.x == 0) { out->f0 = f; out->pay_len = f.code == TFGPU_ROW_OK && f.payload_len <= (uint32_t)JQ_BYTES ? f.payload_len : 0u; }
    if (f.code == TFGPU_ROW_OK && f.payload_len <= (uint32_t)JQ_BYTES)
      for (uint32_t i = threadIdx.x; i < f.payload_len; i += 256) out->pay[i] = data[f.payload_start + i];
  }
}
// the receiver, around its unpack call: spans may be claimed from the messages' ends (Params::tent) — it will hand the frames to
// tfgpu_debezium_parse under dbz_trust_frames, where the claims are proven or the full walk is run
void dbz_tentative_frames(bool on) { g_tentative = on; }
bool dbz_last_parse_was_quick() { return g_last_quick; }
} }

// TFGPU_DBZ_HOSTTIME=1 (profiling only): wall time of the host sections of the two calls, to stderr; every mark synchronizes
struct HostClock {
  bool on; std::chrono::steady_clock::time_point t0; const char *what;
  explicit HostClock(const char *w) : on([] { const char *e = std::getenv("TFGPU_DBZ_HOSTTIME"); return e && e[0] == '1'; }()), t0(std::chrono::steady_clock::now()), what(w) {}
  void mark(const char *name) {
    if (!on) return;
    tf::sync();
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "tfgpu hosttime %s: %-28s %.3f ms\n", what, name, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};


// The lazy form of the two frame calls (dbz_lazy_frames): count the frames that are not (OK, hash (h0, h1), a payload the tile parser stages);
// none → frame 0, its payload's bytes and the widest payload come down instead of the frames, the cache says `uniform`, true is returned.
static bool frames_stay_down(const dbz::Params &p, int64_t n, uint64_t h0, uint64_t h1, tfgpu_dbz_frame *frames, const void *bytes, const Buf &fr, const Buf &tent) {
  static const bool lazy_off = [] { const char *e = std::getenv("TFGPU_DBZ_LAZY_FRAMES"); return e && e[0] == '0'; }();  // A/B runs
  if (!dbz::g_lazy || lazy_off || n <= 0) return false;
  Context &cx = ctx();
  Buf bst = dalloc(sizeof(dbz::FrameStats));
  TF_HIP(hipMemsetAsync(bst->p, 0, 16, cx.stream));
  { KernelTimer t("dbz_frame_stats"); dbz::dbz_frame_stats<<<dbz::nblk(n, 256), 256, 0, cx.stream>>>(p.frames, n, h0, h1, p.data, reinterpret_cast<dbz::FrameStats *>(bst->p)); }
  const dbz::FrameStats *hs = reinterpret_cast<const dbz::FrameStats *>(d2h_u32(bst->p, sizeof(dbz::FrameStats) / 4));
  tf::sync();
  if (hs->n_odd) return false;
  frames[0] = hs->f0;
  dbz::g_frames = dbz::FrameCache{frames, n, bytes, fr, tent, true, std::max<uint32_t>(hs->max_chunks, 1u), std::string(reinterpret_cast<const char *>(hs->pay), hs->pay_len)};
  return true;
}

static int debezium_unpack_impl(const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs, tfgpu_dbz_frame *frames, const tfgpu_dbz_prefix *known) {
  TF_API_BEGIN
  if ((len && !bytes) || !frames) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_unpack: null argument");
  if (known && (!known->bytes || known->len < 2 || known->schema_off + (uint64_t)known->schema_len > known->len))
    return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_unpack_cached: bad prefix");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  HostClock hc("unpack");
  dbz::Staged s;
  dbz::stage(s, bytes, len, mem, msgs);
  hc.mark("stage (offsets up)");
  const int64_t nmsg = s.p.nmsg;
  Buf fr = dalloc((size_t)std::max<int64_t>(nmsg, 1) * sizeof(tfgpu_dbz_frame));
  s.p.frames = reinterpret_cast<tfgpu_dbz_frame *>(fr->p);
  static const bool no_fast = [] { const char *e = std::getenv("TFGPU_DBZ_FULL_WALK"); return e && *e == '1'; }();  // A/B runs
  if (known && nmsg && !no_fast) {
    // a prefix the shim validated in an earlier batch (its schema cache): no message of this batch is walked in full unless it differs
    Buf pb = dalloc((size_t)known->len + 64), same = dalloc_zero((size_t)nmsg + 16);
    h2d(pb->p, known->bytes, known->len);
    TF_HIP(hipMemsetAsync((char *)pb->p + known->len, 0, 64, cx.stream));
    s.p.ref = -1; s.p.plen = known->len; s.p.same = ptr<uint8_t>(same); s.p.pref = ptr<uint8_t>(pb);
    Buf tent;
    static const bool tent_off = [] { const char *e = std::getenv("TFGPU_DBZ_TENTATIVE"); return e && e[0] == '0'; }();
    if (dbz::g_tentative && !tent_off) { tent = dalloc_zero((size_t)nmsg + 16); s.p.tent = ptr<uint8_t>(tent); }
    std::memset(&s.p.pframe, 0, sizeof s.p.pframe);
    s.p.pframe.schema_start = known->schema_off; s.p.pframe.schema_len = known->schema_len;
    s.p.pframe.schema_hash[0] = known->schema_hash[0]; s.p.pframe.schema_hash[1] = known->schema_hash[1];
    { KernelTimer t("dbz_prefix_same"); dbz::dbz_prefix_same<<<(unsigned)((nmsg + 3) / 4), 256, 0, cx.stream>>>(s.p); }
    { KernelTimer t("dbz_unpack"); dbz::dbz_unpack_rest<<<dbz::nblk(nmsg, 128), 128, 0, cx.stream>>>(s.p); }
    hc.mark(tent ? "kernels (claimed spans)" : "kernels");
    if (frames_stay_down(s.p, nmsg, known->schema_hash[0], known->schema_hash[1], frames, bytes, fr, tent)) { hc.mark("frame stats down"); return TFGPU_OK; }
    d2h(frames, fr->p, (size_t)nmsg * sizeof(tfgpu_dbz_frame));
    tf::sync();
    hc.mark("frames down");
    dbz::g_frames = dbz::FrameCache{frames, nmsg, bytes, fr, tent};
    return TFGPU_OK;
  }
  if (nmsg >= 64 && !no_fast) {
    // the opening message in full, alone; then every message that shares its bytes up to the payload value walks only the rest
    dbz::Params one = s.p;
    one.nmsg = 1;
    { KernelTimer t("dbz_unpack"); dbz::dbz_unpack<<<1, 128, 0, cx.stream>>>(one); }
    tfgpu_dbz_frame f0;
    d2h(&f0, fr->p, sizeof f0);
    std::vector<uint32_t> ms01(2);
    d2h(ms01.data(), s.ms->p, 8);
    tf::sync();
    const uint64_t msg_end = ms01[1];
    // usable when the payload is the LAST member: the message is prefix + payload value + '}' (+ white space)
    const bool usable = f0.code == TFGPU_ROW_OK && f0.schema_len && f0.payload_len && f0.payload_start > f0.schema_start &&
                        f0.payload_start + f0.payload_len < msg_end;
    if (usable) {
      Buf same = dalloc_zero((size_t)nmsg + 16);
      s.p.ref = 0; s.p.plen = (uint32_t)(f0.payload_start - ms01[0]); s.p.same = ptr<uint8_t>(same);
      { KernelTimer t("dbz_prefix_same"); dbz::dbz_prefix_same<<<(unsigned)((nmsg + 3) / 4), 256, 0, cx.stream>>>(s.p); }
      { KernelTimer t("dbz_unpack"); dbz::dbz_unpack_rest<<<dbz::nblk(nmsg, 128), 128, 0, cx.stream>>>(s.p); }
    } else {
      KernelTimer t("dbz_unpack"); dbz::dbz_unpack<<<dbz::nblk(nmsg, 128), 128, 0, cx.stream>>>(s.p);
    }
  } else if (nmsg) { KernelTimer t("dbz_unpack"); dbz::dbz_unpack<<<dbz::nblk(nmsg, 128), 128, 0, cx.stream>>>(s.p); }
  if (nmsg) d2h(frames, fr->p, (size_t)nmsg * sizeof(tfgpu_dbz_frame));
  tf::sync();
  dbz::g_frames = dbz::FrameCache{frames, nmsg, bytes, fr, nullptr};
  return TFGPU_OK;
  TF_API_END
}

extern "C" int tfgpu_debezium_unpack(const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs, tfgpu_dbz_frame *frames) {
  return debezium_unpack_impl(bytes, len, mem, msgs, frames, nullptr);
}
extern "C" int tfgpu_debezium_unpack_cached(const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs, const tfgpu_dbz_prefix *known, tfgpu_dbz_frame *frames) {
  if (!known) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_unpack_cached: null prefix");
  return debezium_unpack_impl(bytes, len, mem, msgs, frames, known);
}

extern "C" int tfgpu_debezium_registry_frames(const void *bytes, uint64_t len, int mem, const tfgpu_messages *event_msgs, const tfgpu_sr_frame *events, tfgpu_dbz_frame *frames) {
  TF_API_BEGIN
  if ((len && !bytes) || !frames || !event_msgs || (event_msgs->nmsg && !events)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_registry_frames: null argument");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  dbz::Staged s;
  dbz::stage(s, bytes, len, mem, event_msgs);
  const int64_t n = s.p.nmsg;
  // the list tfgpu_sr_frames wrote, handed back untouched by the receiver (it says so: dbz_trust_frames): its device copy serves
  Buf ev = dbz::g_trust ? sr::last_frames_device(events, n, bytes) : Buf();
  const bool ev_trusted = (bool)ev;
  for (int64_t e = 0; e < n && !ev_trusted; e++) {  // the events come back from the host: a span that leaves its slot would send a lane outside the buffer
    const tfgpu_sr_frame &f = events[e];
    if (f.start < event_msgs->start[e] || f.start + (uint64_t)f.len > event_msgs->start[e + 1])
      return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_registry_frames: event " + std::to_string(e) + " does not lie inside its slot (events must come from tfgpu_sr_frames over the same bytes)");
  }
  Buf fr = dalloc((size_t)std::max<int64_t>(n, 1) * sizeof(tfgpu_dbz_frame));
  if (!ev_trusted) ev = dalloc((size_t)std::max<int64_t>(n, 1) * sizeof(tfgpu_sr_frame));
  s.p.frames = reinterpret_cast<tfgpu_dbz_frame *>(fr->p);
  Buf tent;
  static const bool tent_off = [] { const char *e = std::getenv("TFGPU_DBZ_TENTATIVE"); return e && e[0] == '0'; }();
  if (dbz::g_tentative && !tent_off && n) { tent = dalloc_zero((size_t)n + 16); s.p.tent = ptr<uint8_t>(tent); }
  if (n) {
    if (!ev_trusted) h2d(ev->p, events, (size_t)n * sizeof(tfgpu_sr_frame));
    { KernelTimer t("dbz_registry_frames"); dbz::dbz_registry_frames<<<dbz::nblk(n, 128), 128, 0, cx.stream>>>(s.p, reinterpret_cast<const tfgpu_sr_frame *>(ev->p)); }
    if (frames_stay_down(s.p, n, dbz::g_lazy_h0, dbz::g_lazy_h1, frames, bytes, fr, tent)) return TFGPU_OK;
    d2h(frames, fr->p, (size_t)n * sizeof(tfgpu_dbz_frame));
    tf::sync();
  }
  dbz::g_frames = dbz::FrameCache{frames, n, bytes, fr, tent};
  return TFGPU_OK;
  TF_API_END
}

extern "C" int tfgpu_debezium_parse(const tfgpu_dbz_options *o, const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs,
                                    const tfgpu_dbz_frame *frames, tfgpu_dbatch **out, tfgpu_dbz_row *rows, int64_t rows_cap,
                                    tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs) {
  TF_API_BEGIN
  if (!o || !out || !frames || (len && !bytes) || (o->nfields && !o->fields)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_parse: null argument");
  const int nf = o->nfields;
  if (nf > 4096) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_debezium_parse: more than 4096 fields");
  std::string names;
  std::vector<dbz::FRecv> fields((size_t)std::max(nf, 1));
  int schema_code = o->schema_code;
  if (schema_code != TFGPU_ROW_OK && schema_code != TFGPU_ROW_DBZ_SCHEMA && schema_code != TFGPU_ROW_HOST_FALLBACK) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_parse: bad schema_code");
  for (int j = 0; j < nf; j++) {
    const tfgpu_dbz_field &f = o->fields[j];
    if (!f.name || f.op < TFGPU_DBZ_BOOLEAN || f.op > TFGPU_DBZ_HOST) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_parse: bad field");
    if (f.op == TFGPU_DBZ_HOST && schema_code == TFGPU_ROW_OK) schema_code = TFGPU_ROW_HOST_FALLBACK;
    fields[(size_t)j] = dbz::FRecv{(uint32_t)names.size(), (uint32_t)std::strlen(f.name), f.op, f.optional ? 1 : 0, f.scale};
    names += f.name;
    names.append((8 - names.size() % 8) % 8, '\0');
  }
  std::vector<uint16_t> sorted((size_t)std::max(nf, 1));
  for (int j = 0; j < nf; j++) sorted[(size_t)j] = (uint16_t)j;
  std::sort(sorted.begin(), sorted.begin() + nf, [&](uint16_t a, uint16_t b) { return std::strcmp(o->fields[a].name, o->fields[b].name) < 0; });
  for (int j = 1; j < nf; j++) if (!std::strcmp(o->fields[sorted[(size_t)j - 1]].name, o->fields[sorted[(size_t)j]].name))
    return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_debezium_parse: the schema names a field twice (rows of the reference would repeat a column): host");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  hipStream_t st = cx.stream;
  HostClock hc("parse");
  dbz::Staged s;
  dbz::stage(s, bytes, len, mem, msgs);
  hc.mark("stage (offsets up)");
  dbz::Params &p = s.p;
  const int64_t nmsg = p.nmsg, nma = std::max<int64_t>(nmsg, 1);
  // the frames come back from the host: spans that leave their message would send lanes outside the buffer
  const bool trusted = dbz::g_trust && dbz::g_frames.host == frames && dbz::g_frames.nmsg == nmsg && dbz::g_frames.bytes == bytes && dbz::g_frames.dev;
  for (int64_t m = 0; m < nmsg && !trusted; m++) {
    const tfgpu_dbz_frame &f = frames[m];
    if (f.code != TFGPU_ROW_OK) continue;
    const uint64_t a = msgs ? msgs->start[m] : 0, z = msgs ? msgs->start[m + 1] : len;
    const bool s_ok = f.schema_len == 0 || (f.schema_start >= a && f.schema_start + f.schema_len <= z);
    const bool p_ok = f.payload_len == 0 || (f.payload_start >= a && f.payload_start + f.payload_len <= z);
    if (!s_ok || !p_ok) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_parse: frame " + std::to_string(m) + " does not lie inside its message (frames must come from tfgpu_debezium_unpack over the same bytes)");
  }
  if (!trusted && dbz::g_frames.host == frames && dbz::g_frames.tent) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_parse: these frames hold claimed payload spans (an unpack under dbz_tentative_frames) and must be parsed under dbz_trust_frames");
  p.tent = trusted && dbz::g_frames.tent ? ptr<uint8_t>(dbz::g_frames.tent) : nullptr;
  Buf bfr = trusted ? dbz::g_frames.dev : dalloc((size_t)nma * sizeof(tfgpu_dbz_frame));
  if (nmsg && !trusted) h2d(bfr->p, frames, (size_t)nmsg * sizeof(tfgpu_dbz_frame));
  p.frames = reinterpret_cast<tfgpu_dbz_frame *>(bfr->p);
  p.h0 = o->schema_hash[0]; p.h1 = o->schema_hash[1]; p.schema_code = schema_code;
  Buf bfields = upload_small(fields.data(), fields.size() * sizeof(dbz::FRecv)), bnames = upload_small(names.data(), std::max<size_t>(names.size(), 8)),
      bsorted = upload_small(sorted.data(), sorted.size() * 2);
  p.fields = ptr<dbz::FRecv>(bfields); p.nfields = nf; p.names = ptr<uint8_t>(bnames); p.sorted = ptr<uint16_t>(bsorted);
  const size_t cells = (size_t)std::max(nf, 1) * (size_t)nma;
  Buf vstart = dalloc(cells * 4), vlen = dalloc(cells * 4);
  // everything that starts as z0block, as views of one block cleared by one fill (twelve fills of their own were 60 us of every batch)
  auto z0a64 = [](size_t n) { return (n + 63) & ~(size_t)63; };
  const size_t z0sizes[] = {cells + 16, (size_t)nma + 16, (size_t)nma + 16, (size_t)(nma + 1) * 4 + 16, 32, (size_t)nma * 8, (size_t)nma * 8, (size_t)nma * 8, (size_t)nma * 4, (size_t)nma * 8, (size_t)nma * 8, 512};
  size_t z0total = 0;
  for (size_t z : z0sizes) z0total += z0a64(z);
  Buf z0block = dalloc_zero(z0total);
  size_t z0off = 0; int z0i = 0;
  auto zview = [&]() { Buf b = subbuf(z0block, z0off, z0sizes[z0i]); z0off += z0a64(z0sizes[z0i]); z0i++; return b; };
  Buf vtype = zview(), status = zview(), kind = zview(), keep = zview(), misc = zview();
  Buf lsn = zview(), ts = zview(), tabhash = zview(), txid = zview(), tab_s = zview(), tab_n = zview();
  p.vstart = ptr<uint32_t>(vstart); p.vlen = ptr<uint32_t>(vlen); p.vtype = ptr<uint8_t>(vtype);
  p.status = ptr<uint8_t>(status); p.kind = ptr<uint8_t>(kind); p.keep = ptr<uint32_t>(keep);
  p.lsn = ptr<uint64_t>(lsn); p.ts = ptr<uint64_t>(ts); p.tabhash = ptr<uint64_t>(tabhash); p.txid = ptr<uint32_t>(txid);
  p.tab_s = ptr<uint32_t>(tab_s); p.tab_n = ptr<uint32_t>(tab_n);
  p.nerr = ptr<uint32_t>(misc); p.first_ok = ptr<unsigned long long>(misc) + 1; p.toast = ptr<uint32_t>(misc) + 4;
  TF_HIP(hipMemsetAsync(p.first_ok, 0xFF, 8, st));
  p.p10 = pow10_table(); p.p128 = reinterpret_cast<const uint64_t *>(p.p10 + 632);
  Buf tabname = zview();
  hc.mark("frames check + up, buffers");
  // ---- dbz_parse_quick for the messages that spell the group's first payload's members (tf_dbzquick.inc); TFGPU_DBZ_QUICK=0: the walker for all ----
  static const bool quick_off = [] { const char *e = std::getenv("TFGPU_DBZ_QUICK"); return e && e[0] == '0'; }();
  bool quick = !quick_off && schema_code == TFGPU_ROW_OK && nf >= 1 && nf <= dbz::DQ_F && nmsg >= 64;
  for (int j = 0; quick && j < nf; j++) {
    const int op = o->fields[j].op;
    if (op == TFGPU_DBZ_HOST) quick = false;
  }
  auto eligible = [&](int64_t m) {
    const tfgpu_dbz_frame &f = frames[m];
    return f.code == TFGPU_ROW_OK && f.schema_hash[0] == p.h0 && f.schema_hash[1] == p.h1 && f.payload_len >= 2 && f.payload_len <= (uint32_t)JQ_BYTES - 32u;
  };
  Buf bmap, bslow;
  int64_t qtiles = 0;
  uint32_t lpt = 0;
  // (a uniform batch, FrameCache: the unpack call answered all of this on the device; `frames` holds frame 0 only)
  const bool uni = trusted && dbz::g_frames.uniform && frames[0].schema_hash[0] == p.h0 && frames[0].schema_hash[1] == p.h1 && dbz::g_frames.pay0.size() == frames[0].payload_len;
  if (trusted && dbz::g_frames.uniform && !uni) {  // the caller's schema is not the one every frame carries: no frame is eligible, nothing else to look at
    quick = false;
  }
  if (quick) {
    int64_t m0 = -1;
    uint32_t maxchunks = 1;   // the longest payload the tile kernel could take, in 16-byte chunks as it stages them
    if (uni) { m0 = 0; maxchunks = dbz::g_frames.max_chunks; }
    else for (int64_t m = 0; m < nmsg; m++) {
      if (!eligible(m)) continue;
      if (m0 < 0) m0 = m;
      const uint32_t ps = (uint32_t)frames[m].payload_start, pl = frames[m].payload_len;
      maxchunks = std::max(maxchunks, (((ps + pl + 15u) & ~15u) - (ps & ~15u)) >> 4);
    }
    dbz::DqMap map;
    quick = false;
    if (m0 >= 0) {
      std::string pay((size_t)frames[m0].payload_len, '\0');
      if (uni) pay = dbz::g_frames.pay0;
      else if (mem == TFGPU_MEM_HOST) std::memcpy(&pay[0], (const uint8_t *)bytes + frames[m0].payload_start, pay.size());
      else { d2h(&pay[0], p.data + frames[m0].payload_start, pay.size()); tf::sync(); }
      std::vector<std::string> fnames;
      std::vector<int> fops;
      for (int j = 0; j < nf; j++) { fnames.push_back(o->fields[j].name); fops.push_back(o->fields[j].op); }
      quick = dbz::build_quick_map(pay, fnames, fops, map);
    }
    if (quick) {
      lpt = std::min<uint32_t>((uint32_t)JQ_LINES, std::max<uint32_t>(1u, (uint32_t)(JQ_BYTES / 16) / maxchunks));
      qtiles = (nmsg + lpt - 1) / lpt;
      bmap = upload_const(&map, sizeof map);
      bslow = dalloc((size_t)(nmsg + 1) * 4 + 16);
      TF_HIP(hipMemsetAsync(bslow->p, 0, 4, st));
    }
  }
  hc.mark("quick map + tiles");
  dbz::g_last_quick = quick;
  if (nmsg) {
    if (quick) {
      Buf taken = dalloc_zero((size_t)nmsg + 16);
      dbz::QParams qp{p, reinterpret_cast<const dbz::DqMap *>(bmap->p), lpt, ptr<uint32_t>(bslow), ptr<uint32_t>(bslow) + 1, ptr<uint8_t>(taken)};
      { KernelTimer t("dbz_parse_quick"); dbz::dbz_parse_quick<<<(unsigned)qtiles, JQ_THREADS, 0, st>>>(qp); }
      std::vector<uint16_t> ff;
      for (int j = 0; j < nf; j++) if (o->fields[j].op == TFGPU_DBZ_FLOAT64) ff.push_back((uint16_t)j);
      if (!ff.empty()) {
        Buf bff = upload_small(ff.data(), ff.size() * 2);
        KernelTimer t("dbz_quick_floats");
        dbz::dbz_quick_floats<<<dbz::nblk(nmsg, 128), 128, 0, st>>>(qp, ptr<uint16_t>(bff), (int)ff.size());
      }
      { KernelTimer t("dbz_parse"); dbz::dbz_parse_listed<<<dbz::nblk(nmsg, 128), 128, 0, st>>>(p, ptr<uint32_t>(bslow) + 1, ptr<uint32_t>(bslow)); }
      if (const char *e = std::getenv("TFGPU_DBZ_DEBUG"); e && e[0] == '1') {  // how the messages were routed (costs a sync)
        uint32_t ns = 0;
        d2h(&ns, bslow->p, 4); tf::sync();
        std::fprintf(stderr, "tfgpu dbz quick: %lld messages, %lld tiles, %u to the walker\n", (long long)nmsg, (long long)qtiles, ns);
      }
      dbz::dbz_first_ok<<<dbz::nblk(nmsg, 256), 256, 0, st>>>(p);
    } else { KernelTimer t("dbz_parse"); dbz::dbz_parse<<<dbz::nblk(nmsg, 128), 128, 0, st>>>(p); }
    dbz::dbz_table_rule<<<dbz::nblk(nmsg, 256), 256, 0, st>>>(p);
    dbz::dbz_table_name<<<1, 64, 0, st>>>(p, ptr<uint8_t>(tabname));
  }
  exclusive_scan_u32(p.keep, p.keep, nmsg, true);
  const uint32_t *hrows = d2h_u32(p.keep + nmsg), *hnerr = d2h_u32(p.nerr, 5);  // [0] errors, [4] some row leaves a column out
  uint8_t htab[512];
  d2h(htab, tabname->p, 512);
  tf::sync();
  hc.mark("parse kernels + row count");
  const int64_t nrows = *hrows, nra = std::max<int64_t>(nrows, 1);
  const uint32_t nerr_total = hnerr[0];
  const bool any_toast = hnerr[4] != 0;
  if (nrows > rows_cap && rows) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_parse: more rows than `rows` holds");

  auto db = std::make_unique<tfgpu_dbatch>();
  db->nrows = nrows;
  {
    uint32_t l0, l1;
    std::memcpy(&l0, htab, 4); std::memcpy(&l1, htab + 256, 4);
    if (l0 > 252 || l1 > 252) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_debezium_parse: source.schema / source.table longer than 252 bytes");
    db->ns.assign((const char *)htab + 4, l0); db->table.assign((const char *)htab + 260, l1);
  }
  db->src_row = dalloc((size_t)nra * 4); db->kind = dalloc_zero((size_t)nra + 16);
  Buf row_msg = dalloc((size_t)nra * 4), drows = dalloc((size_t)nra * sizeof(tfgpu_dbz_row)), old8 = dalloc_zero((size_t)nra);
  p.row_msg = ptr<uint32_t>(row_msg); p.nrows = nrows;
  std::vector<dbz::OutCol> oc((size_t)std::max(nf, 1));
  std::vector<int32_t> text_cols;
  const int64_t seg_stride = ((nrows + 1 + 3) / 4) * 4;
  int ntext = 0, nkeys = 0;
  for (int j = 0; j < nf; j++) { if (o->fields[j].op >= TFGPU_DBZ_STRING) ntext++; if (!o->fields[j].optional) nkeys++; }
  // everything the cell kernels expect zeroed, in ONE block and one memset (a 61-column table made ~200 allocations and as many
  // memset launches here: 0.36 ms of host time per batch): text lengths, the byte-per-row flags, the fixed-width values, the bitmaps
  auto a256 = [](size_t n) { return (n + 255) & ~(size_t)255; };
  const size_t bm_bytes = a256((size_t)((nra + 7) / 8) + 8);
  size_t zero_bytes = a256((size_t)std::max(ntext, 1) * (size_t)seg_stride * 4 + 16) + a256((size_t)std::max(nf, 1) * (size_t)nra) + a256((size_t)std::max(nkeys, 1) * (size_t)nra);
  for (int j = 0; j < nf; j++) {
    const int op = o->fields[j].op;
    const size_t w = (op == TFGPU_DBZ_BOOLEAN || op == TFGPU_DBZ_INT8) ? 1 : op == TFGPU_DBZ_INT16 ? 2 : op == TFGPU_DBZ_INT32 ? 4 : (op == TFGPU_DBZ_INT64 || op == TFGPU_DBZ_FLOAT64) ? 8 : 0;
    zero_bytes += a256((size_t)nra * w) + bm_bytes + (o->fields[j].optional ? 0 : bm_bytes);
  }
  zero_bytes += bm_bytes;  // OldKeys' own bitmap
  Buf zarena = dalloc_zero(zero_bytes);
  size_t zoff = 0;
  auto zcarve = [&](size_t bytes) { Buf b = subbuf(zarena, zoff, bytes); zoff += a256(bytes); return b; };
  Buf lens_all = zcarve((size_t)std::max(ntext, 1) * (size_t)seg_stride * 4 + 16);
  Buf valid8 = zcarve((size_t)std::max(nf, 1) * (size_t)nra), oldv8 = zcarve((size_t)std::max(nkeys, 1) * (size_t)nra);
  int ti = 0, ki = 0;
  for (int j = 0; j < nf; j++) {
    const tfgpu_dbz_field &f = o->fields[j];
    DColumn d;
    d.name = f.name;
    dbz::OutCol &c = oc[(size_t)j];
    std::memset(&c, 0, sizeof c);
    c.op = f.op;
    auto fixed = [&](int dtype, int repr, size_t w) { d.dtype = dtype; d.repr = repr; d.values = zcarve((size_t)nra * w); c.values = d.values->p; };
    switch (f.op) {
      case TFGPU_DBZ_BOOLEAN: fixed(TFGPU_T_BOOLEAN, TFGPU_R_BOOL, 1); break;
      case TFGPU_DBZ_INT8: fixed(TFGPU_T_INT8, TFGPU_R_INT8, 1); break;
      case TFGPU_DBZ_INT16: fixed(TFGPU_T_INT16, TFGPU_R_INT16, 2); break;
      case TFGPU_DBZ_INT32: fixed(TFGPU_T_INT32, TFGPU_R_INT32, 4); break;
      case TFGPU_DBZ_INT64: fixed(TFGPU_T_INT64, TFGPU_R_INT64, 8); break;
      case TFGPU_DBZ_FLOAT64: fixed(TFGPU_T_FLOAT64, TFGPU_R_FLOAT64, 8); break;
      case TFGPU_DBZ_BYTES: d.dtype = TFGPU_T_BYTES; d.repr = TFGPU_R_BYTES; break;
      case TFGPU_DBZ_VSD: d.dtype = TFGPU_T_FLOAT64; d.repr = TFGPU_R_JSONNUM; break;
      case TFGPU_DBZ_HOST: d.dtype = TFGPU_T_ANY; d.repr = TFGPU_R_JSON; break;
      default: d.dtype = TFGPU_T_UTF8; d.repr = TFGPU_R_STRING;
    }
    if (f.op >= TFGPU_DBZ_STRING) {
      d.offsets = subbuf(lens_all, (size_t)ti * (size_t)seg_stride * 4, (size_t)(nrows + 1) * 4);
      c.lens = ptr<uint32_t>(d.offsets);
      text_cols.push_back(j);
      ti++;
    }
    c.valid8 = ptr<uint8_t>(valid8) + (size_t)j * (size_t)nra;
    d.validity = zcarve((size_t)((nra + 7) / 8) + 8);
    c.validity = ptr<uint8_t>(d.validity);
    if (!f.optional) {
      c.old8 = ptr<uint8_t>(oldv8) + (size_t)ki * (size_t)nra;
      ki++;
      db->key_names.push_back(f.name);
    }
    db->schema.push_back({d.name, d.dtype});
    db->cols.push_back(std::move(d));
  }
  Buf boc = upload_small(oc.data(), oc.size() * sizeof(dbz::OutCol));
  if (nrows) {
    dbz::dbz_row_msgs<<<dbz::nblk(nmsg, 256), 256, 0, st>>>(p);
    if (nf) {
      KernelTimer t("dbz_cell_values");
      std::vector<int32_t> fl[3];
      for (int j = 0; j < nf; j++) fl[dbz::dbz_field_class(o->fields[j].op)].push_back(j);
      auto launch = [&](auto kernel, const std::vector<int32_t> &l) {
        if (l.empty()) return;
        Buf bl = upload_small(l.data(), l.size() * 4);
        kernel<<<dim3(dbz::nblk(nrows, 256), (unsigned)l.size()), 256, 0, st>>>(p, ptr<dbz::OutCol>(boc), ptr<int32_t>(bl), ptr<int32_t>(db->src_row), ptr<uint8_t>(db->kind), ptr<uint8_t>(old8),
                                                                                reinterpret_cast<dbz::DbzRow *>(drows->p));
      };
      launch(dbz::dbz_cell_values<dbz::DC_LIGHT>, fl[0]);
      launch(dbz::dbz_cell_values<dbz::DC_FLOAT>, fl[1]);
      launch(dbz::dbz_cell_values<dbz::DC_RENDER>, fl[2]);
    }
  }
  hc.mark("columns + cell values");
  if (nf == 0 && nrows) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_debezium_parse: a table schema without fields");
  if (ntext) {
    exclusive_scan_u32_segments(ptr<uint32_t>(lens_all), nrows, ntext, seg_stride);
    Buf totals = dalloc((size_t)ntext * 4 + 16);
    dbz::dbz_gather_totals<<<1, 64, 0, st>>>(ptr<uint32_t>(lens_all), seg_stride, nrows, ntext, ptr<uint32_t>(totals));
    const uint32_t *tot = d2h_u32(totals->p, (size_t)ntext);   // one read-back for all the text columns' sizes, one block for their bytes
    static const bool words = [] { const char *e = std::getenv("TFGPU_DBZ_COPY_WORDS"); return !(e && e[0] == '0'); }();  // 0: A/B runs
    Buf spec = dalloc_zero((size_t)ntext * 4), btc0 = upload_small(text_cols.data(), text_cols.size() * 4);
    if (nrows && words) dbz::dbz_mark_special<<<dim3(dbz::nblk(nrows, 256 * 16), (unsigned)ntext), 256, 0, st>>>(p, ptr<int32_t>(btc0), ptr<uint32_t>(spec));
    const uint32_t *hspec = d2h_u32(spec->p, (size_t)ntext);
    tf::sync();
    size_t text_bytes = 0;
    for (int t = 0; t < ntext; t++) text_bytes += a256((size_t)tot[t] + 8);
    Buf tarena = dalloc(std::max<size_t>(text_bytes, 256));
    size_t toff = 0;
    for (int t = 0; t < ntext; t++) {
      DColumn &d = db->cols[(size_t)text_cols[(size_t)t]];
      d.data_len = tot[t];
      d.data = subbuf(tarena, toff, (size_t)d.data_len + 8);
      toff += a256((size_t)d.data_len + 8);
      oc[(size_t)text_cols[(size_t)t]].data = ptr<uint8_t>(d.data);
    }
    boc = upload_small(oc.data(), oc.size() * sizeof(dbz::OutCol));
    Buf btc = upload_small(text_cols.data(), text_cols.size() * 4);
    auto launch_text = [&](const std::vector<int32_t> &colsl, int plain_done) {  // the listed text columns, by receiver class
      std::vector<int32_t> lt, hv;
      for (int32_t j : colsl) (dbz::dbz_field_class(o->fields[j].op) == dbz::DC_RENDER ? hv : lt).push_back(j);
      if (!lt.empty()) { Buf b = upload_small(lt.data(), lt.size() * 4); dbz::dbz_cell_text<false><<<dim3(dbz::nblk(nrows, 256), (unsigned)lt.size()), 256, 0, st>>>(p, ptr<dbz::OutCol>(boc), ptr<int32_t>(b), (int32_t)lt.size(), plain_done); }
      if (!hv.empty()) { Buf b = upload_small(hv.data(), hv.size() * 4); dbz::dbz_cell_text<true><<<dim3(dbz::nblk(nrows, 256), (unsigned)hv.size()), 256, 0, st>>>(p, ptr<dbz::OutCol>(boc), ptr<int32_t>(b), (int32_t)hv.size(), plain_done); }
    };
    if (nrows) {
      KernelTimer t("dbz_cell_text");
      if (words) {
        dbz::dbz_copy_words<<<dim3(dbz::nblk(nrows, 256), (unsigned)ntext), 256, 0, st>>>(p, ptr<dbz::OutCol>(boc), ptr<int32_t>(btc));
        std::vector<int32_t> sp;
        for (int t2 = 0; t2 < ntext; t2++) if (hspec[t2]) sp.push_back(text_cols[(size_t)t2]);
        launch_text(sp, 1);
      } else launch_text(text_cols, 0);
    }
  }
  // bitmaps: ColumnValues validity per column; OldKeys = the key columns' buffers under their own validity
  const unsigned gb = dbz::nblk((nrows + 7) / 8, 256);
  std::vector<dbz::PackJob> jobs;  // one launch for all of them (a 61-column table made ~70 launches here)
  for (int j = 0; j < nf; j++) jobs.push_back(dbz::PackJob{oc[(size_t)j].valid8, oc[(size_t)j].validity});
  for (int j = 0; j < nf; j++) {
    if (o->fields[j].optional) continue;
    DColumn k = db->cols[(size_t)j];  // shares values / offsets / data
    k.validity = zcarve((size_t)((nra + 7) / 8) + 8);
    jobs.push_back(dbz::PackJob{oc[(size_t)j].old8, ptr<uint8_t>(k.validity)});
    db->old_keys.push_back(std::move(k));
  }
  if (!db->old_keys.empty()) {
    db->old_present = zcarve((size_t)((nra + 7) / 8) + 8);
    jobs.push_back(dbz::PackJob{ptr<uint8_t>(old8), ptr<uint8_t>(db->old_present)});
  }
  Buf abs8;
  if (any_toast && nrows && nf) {  // `__debezium_unavailable_value`: the columns those rows do not list (rare: its own launch and read-back)
    abs8 = dalloc((size_t)nf * (size_t)nra);
    Buf colflag = dalloc_zero((size_t)nf * 4);
    dbz::dbz_absent_cells<<<dim3(dbz::nblk(nrows, 256), (unsigned)nf), 256, 0, st>>>(p, nra, ptr<uint8_t>(abs8), ptr<uint32_t>(colflag));
    const uint32_t *hcf = d2h_u32(colflag->p, (size_t)nf);
    tf::sync();
    for (int j = 0; j < nf; j++) if (hcf[j]) {
      DColumn &d = db->cols[(size_t)j];
      d.absent = dalloc_zero((size_t)((nra + 7) / 8) + 8);
      jobs.push_back(dbz::PackJob{ptr<uint8_t>(abs8) + (size_t)j * (size_t)nra, ptr<uint8_t>(d.absent)});
    }
  }
  if (nrows && !jobs.empty()) {
    Buf bj = upload_small(jobs.data(), jobs.size() * sizeof(dbz::PackJob));
    dbz::dbz_pack_bits_all<<<dim3(gb, (unsigned)jobs.size()), 256, 0, st>>>(reinterpret_cast<const dbz::PackJob *>(bj->p), nrows);
  }
  hc.mark("text + bitmaps");
  if (rows && nrows) d2h(rows, drows->p, (size_t)nrows * sizeof(tfgpu_dbz_row));
  int64_t ne = 0;
  if (nerr_total) {
    std::vector<uint8_t> hst((size_t)nmsg);
    d2h(hst.data(), status->p, (size_t)nmsg);
    tf::sync();
    for (int64_t m = 0; m < nmsg; m++) {
      const int c = hst[(size_t)m];
      if (c == TFGPU_ROW_OK || c == dbz::ST_SKIP) continue;
      if (errs && ne < errs_cap) errs[ne] = tfgpu_row_error{m, c, (int32_t)m, -1};
      ne++;
    }
  }
  tf::sync();
  hc.mark("rows down");
  if (nerrs) *nerrs = ne;
  *out = db.release();
  return TFGPU_OK;
  TF_API_END
}
