// tf_protobuf.hip — the per-message half of the Confluent-SR parser's PROTOBUF branch (the descriptor comes from tf_protoschema.cpp).
//
//   ConfluentSrImpl.DoOne / doWithSchema          pkg/parsers/registry/confluentschemaregistry/engine/parser.go:108-120, 30-59
//   makeChangeItemsFromMessageWithProtobuf        engine/format_protobuf.go:16-90 (the index byte :31-43, dynamicMessage.Unmarshal :50-53)
//   unpackProtobufDynamicMessage, unpackVal       engine/utils_protobuf.go:87-112, types_protobuf.go:36-149
//
// One Kafka message is one protobuf message (doWithSchema consumes the whole rest).  Kernels:
//   pb_decode       lane = message: prefix (length, magic byte, schema id), the message-index byte, then the wire format — tag, value,
//                   tag, value … — keeping for every schema field its LAST occurrence (a scalar's raw 64 bits, or the span of a
//                   length-delimited one); a message-typed field's bytes are walked too, so that a row is only made of bytes that
//                   unmarshal.  HBM-bound in principle (each byte read once); one lane per message is latency-bound for long ones —
//                   the JSON paths' tile staging would apply if this format becomes a bench line.
//   pb_cells<ANY>   lane = (row, field): the Go value — truncation / zigzag / bit casts — into the column; text lengths (the `any`
//                   columns, whose lengths need the JSON emitter, in their own launch: the emitter's registers would otherwise set
//                   the occupancy of every cheap cell)
//   pb_text_copy / pb_text_any   lane = (row, text field): string / bytes copied; message fields marshalled (all members, keys sorted, absent
//                   members as their zero values: unpackNotRepeatedVal walks GetKnownFields), repeated fields as the array of their
//                   elements (unpackRepeatedVal; packed runs unrolled)
// HBM layout: rec[field][message] (8 bytes: raw value, or start | len << 32) + present[field][message] — a message's fields are
// scattered over 8 * nfields bytes per message, coalesced across messages in the cell kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "tf_common.hpp"
#include "tf_devfloat.hpp"
#include "tf_devparse.hpp"
#include "tf_emit.hpp"

struct tfgpu_pb_schema;

namespace tf {
namespace pbd {

enum : uint8_t { ST_OK = 0, ST_SKIP = 250 };  // other values: TFGPU_ROW_* codes
struct DField { int32_t number, ptype, mem_off, nmem; uint32_t name_off, name_len; int32_t repeated, oneof; };  // members / names: offsets into the tables below
struct DMember { int32_t number, ptype; uint32_t name_off, name_len; };

struct Params {
  const uint8_t *data; const uint32_t *ms; int64_t nmsg;
  uint32_t schema_id; int32_t schema_code, report_frame_errors;
  const DField *fields; int32_t nfields; const DMember *members; const uint8_t *names;
  const int16_t *lut; int32_t lutn;    // field number → field index (-1: unknown) for numbers below lutn; larger numbers are searched
  uint64_t *rec; uint8_t *present;     // [nfields][nmsg]
  uint8_t *status; uint32_t *keep;     // [nmsg], [nmsg + 1]
  uint32_t *nerr;
  uint32_t *row_msg; int64_t nrows;
};

constexpr uint32_t MAP_MAX_ENTRIES = 32;   // a map field with more entries than this is the stock code's (TFGPU_ROW_HOST_FALLBACK): the emitter below is quadratic in them
__device__ __forceinline__ int want_wt(int t) {
  switch (t) {
    case TFGPU_PB_DOUBLE: case TFGPU_PB_FIXED64: case TFGPU_PB_SFIXED64: return 1;
    case TFGPU_PB_FLOAT: case TFGPU_PB_FIXED32: case TFGPU_PB_SFIXED32: return 5;
    case TFGPU_PB_STRING: case TFGPU_PB_BYTES: case TFGPU_PB_MESSAGE: return 2;
    default: return 0;
  }
}
// a base-128 varint at [i, end): false = truncated / longer than ten bytes (proto.DecodeVarint's errors).  Bytes come through MemBytes'
// 8-byte window: one load per eight bytes of a message instead of one per byte.
__device__ __forceinline__ bool varint(MemBytes &rd, uint32_t &i, uint32_t end, uint64_t *v) {
  uint64_t x = 0;
  for (uint32_t s = 0; s < 70; s += 7) {
    if (i >= end) return false;
    const uint32_t c = rd.at(i++);
    x |= (uint64_t)(c & 0x7Fu) << s;   // (bits beyond 64 fall off: Go's uint64 shift does the same)
    if (!(c & 0x80u)) { *v = x; return true; }
  }
  return false;
}
__device__ __forceinline__ uint64_t le(MemBytes &rd, uint32_t i, int n) { return rd.bytes(i, n); }  // never past the word holding the value's last byte

// One message's fields over [a, z): calls on(number, wire type, raw value or start, len) for every field; 0 ok, 1 does not unmarshal, 2 host
template <class F> __device__ int walk(MemBytes &d, uint32_t a, uint32_t z, F on) {
  uint32_t i = a;
  while (i < z) {
    uint64_t tag;
    if (!varint(d, i, z, &tag)) return 1;
    const uint32_t wt = (uint32_t)tag & 7u;
    const uint64_t num = tag >> 3;
    if (num == 0 || num > 536870911ull) return 2;   // field number 0 / beyond 2^29 - 1: what the reference's decoder makes of it is not pinned
    uint64_t raw = 0; uint32_t len = 0;
    if (wt == 0) { if (!varint(d, i, z, &raw)) return 1; }
    else if (wt == 1) { if (z - i < 8) return 1; raw = le(d, i, 8); i += 8; }
    else if (wt == 5) { if (z - i < 4) return 1; raw = le(d, i, 4); i += 4; }
    else if (wt == 2) { uint64_t n; if (!varint(d, i, z, &n)) return 1; if (n > (uint64_t)(z - i)) return 1; raw = i; len = (uint32_t)n; i += (uint32_t)n; }
    else return 2;  // groups (and the reserved wire types: an error in one library, skipped by another — the stock code decides)
    const int r = on((uint32_t)num, wt, raw, len);
    if (r) return r;
  }
  return 0;
}

__global__ void __launch_bounds__(128) pb_decode(Params p) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.nmsg) return;
  MemBytes d(p.data);
  const uint32_t a = p.ms[m], z = p.ms[m + 1];
  auto done = [&](int st) {
    p.status[m] = (uint8_t)st;
    p.keep[m] = st == ST_OK ? 1u : 0u;
    if (st != ST_OK && st != ST_SKIP) atomicAdd(p.nerr, 1u);
  };
  if (z == a) return done(ST_SKIP);                                              // DoBuf: an empty message yields nothing
  if (z - a < 5) return done(p.report_frame_errors ? TFGPU_ROW_SR_SHORT : ST_SKIP);
  if (d.at(a) != 0) return done(p.report_frame_errors ? TFGPU_ROW_SR_MAGIC : ST_SKIP);
  const uint32_t id = (d.at(a + 1) << 24) | (d.at(a + 2) << 16) | (d.at(a + 3) << 8) | d.at(a + 4);
  if (id != p.schema_id) return done(ST_SKIP);
  if (z - a == 5) return done(TFGPU_ROW_HOST_FALLBACK);                            // buf[0] of an empty slice: the reference panics
  if (d.at(a + 5) != 0) return done(TFGPU_ROW_HOST_FALLBACK);                         // message indexes: another message of the file
  if (p.schema_code != TFGPU_ROW_OK) return done(p.schema_code);
  for (int f = 0; f < p.nfields; f++) p.present[(int64_t)f * p.nmsg + m] = 0;
  bool twice = false;
  const int rc = walk(d, a + 6, z, [&](uint32_t num, uint32_t wt, uint64_t raw, uint32_t len) {
    int f = -1;
    if (num < (uint32_t)p.lutn) f = p.lut[num];
    else for (int k = 0; k < p.nfields; k++) if ((uint32_t)p.fields[k].number == num) { f = k; break; }
    if (f >= 0) {
      const DField &fd = p.fields[f];
      const int64_t i = (int64_t)f * p.nmsg + m;
      if (fd.oneof) {  // a oneof member: setting it clears the others of its group (the dynamic message does so while it unmarshals): the LAST member on the wire stays
        for (int k = 0; k < p.nfields; k++) if (k != f && p.fields[k].oneof == fd.oneof) p.present[(int64_t)k * p.nmsg + m] = 0;
      }
      if (fd.repeated) {  // every occurrence is an element; numeric kinds also arrive packed: a length-delimited run of them
        const int ew = want_wt(fd.ptype);
        if (fd.ptype == TFGPU_PB_MESSAGE) {
          // a repeated message field's element, or a map field's entry {key = 1, value = 2}: a nested message that unmarshals eagerly — its known members
          // with their wire types (anything else: the stock code), a map's entries counted (the device orders them by key with a scan a key: small maps)
          if (wt != 2) return 2;
          const int r2 = walk(d, (uint32_t)raw, (uint32_t)raw + len, [&](uint32_t n2, uint32_t w2, uint64_t, uint32_t) {
            for (int k = 0; k < fd.nmem; k++) if ((uint32_t)p.members[fd.mem_off + k].number == n2 && (int)w2 != want_wt(p.members[fd.mem_off + k].ptype)) return 2;
            return 0;
          });
          if (r2) return r2;
          if (fd.repeated == 2) { const uint64_t seen = p.present[i] ? p.rec[i] + 1 : 1; p.rec[i] = seen; if (seen > MAP_MAX_ENTRIES) return 2; }
          p.present[i] = 1;
          return 0;
        }
        if ((int)wt != ew) {
          if (wt != 2) return 2;
          uint32_t q = (uint32_t)raw; const uint32_t qe = q + len;
          while (q < qe) {
            if (ew == 0) { uint64_t v; if (!varint(d, q, qe, &v)) return 1; }
            else { const uint32_t w = ew == 1 ? 8u : 4u; if (qe - q < w) return 1; q += w; }
          }
        }
        p.present[i] = 1;
        return 0;
      }
      if ((int)wt != want_wt(fd.ptype)) return 2;                                  // a known field with another wire type
      if (fd.ptype == TFGPU_PB_MESSAGE) {
        if (p.present[i]) twice = true;                                            // protobuf merges the occurrences
        const int r2 = walk(d, (uint32_t)raw, (uint32_t)raw + len, [&](uint32_t n2, uint32_t w2, uint64_t, uint32_t) {
          for (int k = 0; k < fd.nmem; k++) if ((uint32_t)p.members[fd.mem_off + k].number == n2 && (int)w2 != want_wt(p.members[fd.mem_off + k].ptype)) return 2;
          return 0;
        });
        if (r2) return r2;
      }
      p.rec[i] = wt == 2 ? (raw | ((uint64_t)len << 32)) : raw;
      p.present[i] = 1;
      return 0;
    }
    return 0;  // an unknown field: kept aside by the dynamic message, no column reads it
  });
  if (rc == 1) return done(TFGPU_ROW_SR_PROTO);
  if (rc == 2 || twice) return done(TFGPU_ROW_HOST_FALLBACK);
  done(ST_OK);
}
__global__ void __launch_bounds__(256) pb_row_msgs(Params p) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m < p.nmsg && p.status[m] == ST_OK) p.row_msg[p.keep[m]] = (uint32_t)m;
}

__global__ void __launch_bounds__(256) pb_src_rows(Params p, int32_t *src_row, uint32_t *part_id) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < p.nrows) { src_row[r] = (int32_t)p.row_msg[r]; part_id[r] = p.row_msg[r]; }
}

struct OutCol {
  int32_t ptype;
  void *values;        // fixed-width kinds
  uint32_t *lens;      // text kinds: lengths, then offsets [nrows + 1]
  uint8_t *data;
  uint8_t *validity;   // bitmap, written 64 rows (one wave) at a time
};

// the Go value of a scalar occurrence (unpackNotRepeatedVal's type assertions: int32 / int64 / uint32 / uint64 / float32 / float64 / bool)
__device__ __forceinline__ int64_t as_i64(int t, uint64_t raw) {
  switch (t) {
    case TFGPU_PB_INT32: case TFGPU_PB_ENUM: case TFGPU_PB_SFIXED32: return (int64_t)(int32_t)(uint32_t)raw;
    case TFGPU_PB_SINT32: { const uint32_t v = (uint32_t)raw; return (int64_t)(int32_t)((v >> 1) ^ (0u - (v & 1u))); }
    case TFGPU_PB_SINT64: return (int64_t)((raw >> 1) ^ (0ull - (raw & 1ull)));
    case TFGPU_PB_UINT32: case TFGPU_PB_FIXED32: return (int64_t)(uint32_t)raw;
    case TFGPU_PB_BOOL: return raw != 0;
    default: return (int64_t)raw;  // int64, sfixed64, uint64, fixed64: the bits
  }
}

// json.Marshal of one member's value.  false: NaN / Inf (json.Marshal fails: host)
template <class S> __device__ bool emit_member(S &s, const uint8_t *d, int t, bool present, uint64_t raw) {
  switch (t) {
    case TFGPU_PB_STRING: emit_json_string(s, d + (present ? (uint32_t)raw : 0u), present ? (uint32_t)(raw >> 32) : 0u, false); return true;
    case TFGPU_PB_BYTES: s.put('"'); emit_base64(s, d + (present ? (uint32_t)raw : 0u), present ? (uint32_t)(raw >> 32) : 0u); s.put('"'); return true;
    case TFGPU_PB_BOOL: { const char *w = present && raw ? "true" : "false"; while (*w) s.put((uint8_t)*w++); return true; }
    case TFGPU_PB_DOUBLE: case TFGPU_PB_FLOAT: {
      double v = 0;
      if (present) { if (t == TFGPU_PB_DOUBLE) v = __longlong_as_double((long long)raw); else v = (double)__uint_as_float((uint32_t)raw); }
      if (v != v || v == INFINITY || v == -INFINITY) return false;
      dev::fmt_json_float(s, v, t == TFGPU_PB_DOUBLE ? 64 : 32);
      return true;
    }
    case TFGPU_PB_UINT64: case TFGPU_PB_FIXED64: emit_u64(s, present ? raw : 0ull); return true;
    default: emit_i64(s, present ? as_i64(t, raw) : 0); return true;
  }
}
// a message field's map: {"name":value,…} over ALL its members in name order (the host sorted them), the last occurrence of each.
// (One walk of the field's bytes per member.  Collecting up to eight members' values in ONE walk — values in registers, unrolled loops —
//  was measured: pb_text 0.68 -> 1.08 ms on the reference's 60-column message, gpurun r08l; the walks are short and the registers dear.)
template <class S> __device__ bool emit_message(S &s, const Params &p, const DField &fd, uint32_t a, uint32_t len) {
  const uint8_t *d = p.data;
  MemBytes rd(p.data);
  s.put('{');
  for (int k = 0; k < fd.nmem; k++) {
    const DMember &mb = p.members[fd.mem_off + k];
    bool present = false; uint64_t raw = 0;
    walk(rd, a, a + len, [&](uint32_t n2, uint32_t w2, uint64_t r2, uint32_t l2) { if ((uint32_t)mb.number == n2) { present = true; raw = w2 == 2 ? (r2 | ((uint64_t)l2 << 32)) : r2; } return 0; });
    if (k) s.put(',');
    emit_json_string(s, p.names + mb.name_off, mb.name_len, false);
    s.put(':');
    if (!emit_member(s, d, mb.ptype, present, raw)) return false;
  }
  s.put('}');
  return true;
}

// a repeated field's []interface{}: the elements of every occurrence in wire order, packed runs unrolled.  false: a NaN / Inf element
// MSG: the instantiation that takes element MESSAGES too (the wide kernels below)
template <bool MSG, class S> __device__ bool emit_array(S &s, const Params &p, const DField &fd, uint32_t a, uint32_t z) {
  const uint8_t *d = p.data;
  MemBytes rd(p.data);
  const int ew = want_wt(fd.ptype);
  bool first = true, ok = true;
  s.put('[');
  walk(rd, a, z, [&](uint32_t num, uint32_t wt, uint64_t raw, uint32_t len) {
    if (num != (uint32_t)fd.number) return 0;
    if constexpr (MSG) if (fd.ptype == TFGPU_PB_MESSAGE) {  // an element message: its map, as a singular message field's (unpackRepeatedVal → unpackNotRepeatedVal over *dynamic.Message)
      if (!first) s.put(',');
      first = false;
      if (!emit_message(s, p, fd, (uint32_t)raw, len)) ok = false;
      return 0;
    }
    auto one = [&](uint64_t r) { if (!first) s.put(','); first = false; if (!emit_member(s, d, fd.ptype, true, r)) ok = false; };
    if ((int)wt == ew) one(wt == 2 ? (raw | ((uint64_t)len << 32)) : raw);
    else {  // packed (pb_decode let nothing else through)
      uint32_t q = (uint32_t)raw; const uint32_t qe = q + len;
      while (q < qe) {
        uint64_t v = 0;
        if (ew == 0) { if (!varint(rd, q, qe, &v)) break; }
        else { const int w = ew == 1 ? 8 : 4; v = le(rd, q, w); q += (uint32_t)w; }
        one(v);
      }
    }
    return 0;
  });
  s.put(']');
  return ok;
}

// a map<string, V> field's map[string]interface{} (types_protobuf.go:57-71) as json.Marshal writes it: keys in byte order, the LAST entry of a key
// (protobuf's map semantics), an entry without a key / value member under "" / the zero value.  No scratch list of the entries: the smallest key
// above the one just written is found by one scan of the field's occurrences a key (<= MAP_MAX_ENTRIES entries: pb_decode counted them).
struct MapKey { uint32_t at, len; };
__device__ __forceinline__ int key_cmp(const uint8_t *d, MapKey a, MapKey b) {
  const uint32_t n = min(a.len, b.len);
  for (uint32_t i = 0; i < n; i++) { const int x = (int)d[a.at + i] - (int)d[b.at + i]; if (x) return x; }
  return (a.len > b.len) - (a.len < b.len);
}
template <class S> __device__ bool emit_map(S &s, const Params &p, const DField &fd, uint32_t a, uint32_t z) {
  const uint8_t *d = p.data;
  MemBytes rd(p.data);
  const int vt = p.members[fd.mem_off + 1].ptype;
  bool ok = true, have_prev = false, first = true;
  MapKey prev{0, 0};
  s.put('{');
  for (uint32_t round = 0; round <= MAP_MAX_ENTRIES; round++) {
    bool found = false; MapKey best{0, 0}; bool vpresent = false; uint64_t vraw = 0;
    walk(rd, a, z, [&](uint32_t num, uint32_t wt, uint64_t raw, uint32_t len) {
      if (num != (uint32_t)fd.number || wt != 2) return 0;
      MapKey k{0, 0}; bool vp = false; uint64_t vr = 0;
      walk(rd, (uint32_t)raw, (uint32_t)raw + len, [&](uint32_t n2, uint32_t w2, uint64_t r2, uint32_t l2) {
        if (n2 == 1 && w2 == 2) k = MapKey{(uint32_t)r2, l2};
        else if (n2 == 2) { vp = true; vr = w2 == 2 ? (r2 | ((uint64_t)l2 << 32)) : r2; }
        return 0;
      });
      if (have_prev && key_cmp(d, k, prev) <= 0) return 0;          // written already (or a duplicate of one that was)
      const int c = found ? key_cmp(d, k, best) : -1;
      if (c <= 0) { found = true; best = k; vpresent = vp; vraw = vr; }  // a smaller key, or a LATER entry of the same one: the last one wins
      return 0;
    });
    if (!found) break;
    if (!first) s.put(',');
    first = false;
    emit_json_string(s, d + best.at, best.len, false);
    s.put(':');
    if (!emit_member(s, d, vt, vpresent, vraw)) ok = false;
    prev = best; have_prev = true;
  }
  s.put('}');
  return ok;
}

// ANY = false: the columns whose cells are a few instructions each (fixed-width values, string / bytes lengths) — a kernel of a dozen
// registers at full occupancy; ANY = true: the `any` columns, whose lengths come from the JSON emitter (float formatting, base64, the
// string escaper: 120 VGPRs).  One kernel for both ran every cheap cell at the emitter's occupancy.  WIDE (with ANY): map fields and repeated
// MESSAGE fields — the key-order scan and a message emitter inside the element walk took the `any` kernels from 124 VGPRs and no scratch to 180 and
// 184 bytes (sr_proto 159 -> 136 M messages/s on a schema that holds neither: profiles/r25b_*); their columns have their own instantiation now.
template <bool ANY, bool WIDE = false> __global__ void __launch_bounds__(256) pb_cells(Params p, const OutCol *cols, const int32_t *list, int32_t *src_row, uint32_t *part_id, uint32_t *host_rows) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = list[blockIdx.y];
  const bool in = r < p.nrows;
  const OutCol &c = cols[j];
  const DField &fd = p.fields[j];
  bool valid = in;
  if (in) {
    const uint32_t m = p.row_msg[r];
    if (blockIdx.y == 0 && !ANY) { src_row[r] = (int32_t)m; part_id[r] = m; }
    const int64_t i = (int64_t)j * p.nmsg + m;
    const bool present = p.present[i] != 0;
    const uint64_t raw = present ? p.rec[i] : 0ull;
    if constexpr (ANY) {
      if (fd.repeated) {  // an absent repeated field is the empty slice: []
        CountSink s;
        bool ok;
        if constexpr (WIDE) ok = fd.repeated == 2 ? emit_map(s, p, fd, p.ms[m] + 6u, p.ms[m + 1]) : emit_array<true>(s, p, fd, p.ms[m] + 6u, p.ms[m + 1]);   // (an absent map field is the empty map: {})
        else ok = emit_array<false>(s, p, fd, p.ms[m] + 6u, p.ms[m + 1]);
        if (!ok) { host_rows[r] = 1; s.n = 0; }
        c.lens[r] = s.n;
      } else if (!present) { c.lens[r] = 0; valid = false; }   // a nil *dynamic.Message: nil
      else {
        CountSink s;
        if (!emit_message(s, p, fd, (uint32_t)raw, (uint32_t)(raw >> 32))) { host_rows[r] = 1; s.n = 0; }
        c.lens[r] = s.n;
      }
    } else switch (fd.ptype) {
      case TFGPU_PB_STRING: case TFGPU_PB_BYTES: c.lens[r] = present ? (uint32_t)(raw >> 32) : 0u; break;
      case TFGPU_PB_DOUBLE: case TFGPU_PB_INT64: case TFGPU_PB_UINT64: case TFGPU_PB_FIXED64: case TFGPU_PB_SFIXED64: case TFGPU_PB_SINT64:
        ((uint64_t *)c.values)[r] = (uint64_t)as_i64(fd.ptype, raw); break;
      case TFGPU_PB_BOOL: ((uint8_t *)c.values)[r] = raw != 0; break;
      default: ((uint32_t *)c.values)[r] = fd.ptype == TFGPU_PB_FLOAT ? (uint32_t)raw : (uint32_t)as_i64(fd.ptype, raw);
    }
  }
  // the validity bitmap a wave at a time: 64 consecutive rows are one aligned 8-byte word (one store instead of 64 atomics on two words)
  const uint64_t bal = __ballot(valid);
  if ((threadIdx.x & 63) == 0 && in) reinterpret_cast<uint64_t *>(c.validity)[r >> 6] = bal;
}
// string / bytes cells: their bytes, eight at a time (the light half of the text pass)
__global__ void __launch_bounds__(256) pb_text_copy(Params p, const OutCol *cols, const int32_t *list) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  const int j = list[blockIdx.y];
  const OutCol &c = cols[j];
  const int64_t i = (int64_t)j * p.nmsg + p.row_msg[r];
  if (!p.present[i]) return;
  const uint64_t raw = p.rec[i];
  WriteSink s{c.data + c.lens[r]};  // (offsets by now)
  put_bytes(s, p.data + (uint32_t)raw, (uint32_t)(raw >> 32));
  s.flush();
}
// `any` cells: message fields and repeated fields marshalled
template <bool WIDE> __global__ void __launch_bounds__(256) pb_text_any(Params p, const OutCol *cols, const int32_t *list) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  const int j = list[blockIdx.y];
  const OutCol &c = cols[j];
  const DField &fd = p.fields[j];
  const uint32_t m = p.row_msg[r];
  const int64_t i = (int64_t)j * p.nmsg + m;
  if (c.lens[r + 1] == c.lens[r]) return;  // nil, or a row for the host: nothing was counted
  WriteSink s{c.data + c.lens[r]};
  if (WIDE && fd.repeated == 2) emit_map(s, p, fd, p.ms[m] + 6u, p.ms[m + 1]);
  else if (fd.repeated) emit_array<WIDE>(s, p, fd, p.ms[m] + 6u, p.ms[m + 1]);
  else { const uint64_t raw = p.rec[i]; emit_message(s, p, fd, (uint32_t)raw, (uint32_t)(raw >> 32)); }
  s.flush();
}

static inline unsigned nblk(int64_t n, int t) { return (unsigned)std::max<int64_t>(1, (n + t - 1) / t); }

}  // namespace pbd
std::unique_ptr<tfgpu_dbatch> compact_rows(const tfgpu_dbatch &in, Buf keep);  // tf_transform.hip
}  // namespace tf

// 32-bit offsets hold less than 4 GiB per column.  TFGPU_TEST_TEXT_LIMIT (tests only) lowers the bound so that the refusal can be
// exercised without a 4 GiB batch.
static bool text_total_exceeds(uint64_t bytes) {
  static const uint64_t limit = [] { const char *e = std::getenv("TFGPU_TEST_TEXT_LIMIT"); return e && *e ? (uint64_t)std::strtoull(e, nullptr, 10) : 0xFFFFFFF0ull; }();
  return bytes >= limit;
}

using namespace tf;

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

extern "C" int tfgpu_pb_schema_info(const tfgpu_pb_schema *s, int32_t *code, const tfgpu_pb_field **fields, int32_t *nfields, const char **table_ns, const char **table_name, const char **record, const char **why);

extern "C" int tfgpu_sr_proto_parse(const tfgpu_pb_schema *sch, uint32_t schema_id, int32_t report_frame_errors, const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs,
                                    tfgpu_dbatch **out, tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs) {
  TF_API_BEGIN
  if (!sch || !out || (len && !bytes)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_sr_proto_parse: null argument");
  int32_t code = 0, nf = 0;
  const tfgpu_pb_field *fl = nullptr;
  const char *ns = "", *table = "";
  tfgpu_pb_schema_info(sch, &code, &fl, &nf, &ns, &table, nullptr, nullptr);
  if (len >= 0xFFFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "confluent SR protobuf: batch must be < 4 GiB (32-bit offsets)");
  if (nf > 4096) return tf::fail(TFGPU_ERR_UNSUPPORTED, "confluent SR protobuf: more than 4096 fields");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  hipStream_t st = cx.stream;
  pbd::Params p{};
  Buf staged;
  if (mem == TFGPU_MEM_HOST) {
    staged = dalloc(len + 64);
    h2d(staged->p, bytes, len);
    TF_HIP(hipMemsetAsync((char *)staged->p + len, 0, 64, st));
    p.data = ptr<uint8_t>(staged);
  } else {
    p.data = (const uint8_t *)bytes;
    if (reinterpret_cast<uintptr_t>(p.data) & 15) return tf::fail(TFGPU_ERR_INVALID, "confluent SR protobuf: device buffer must be 16-byte aligned");
  }
  const int64_t nmsg = msgs ? msgs->nmsg : 1, nma = std::max<int64_t>(nmsg, 1);
  if (nmsg < 0 || (msgs && nmsg > 0 && !msgs->start)) return tf::fail(TFGPU_ERR_INVALID, "confluent SR protobuf: bad message batch");
  std::vector<uint32_t> ms((size_t)nmsg + 1);
  if (msgs) {
    for (int64_t m = 0; m <= nmsg; m++) {
      if (msgs->start[m] > len || (m && msgs->start[m] < msgs->start[m - 1])) return tf::fail(TFGPU_ERR_INVALID, "confluent SR protobuf: message offsets must be ascending and inside the buffer");
      ms[(size_t)m] = (uint32_t)msgs->start[m];
    }
  } else { ms[0] = 0; ms[1] = (uint32_t)len; }
  Buf bms = dalloc(ms.size() * 4 + 16);
  h2d(bms->p, ms.data(), ms.size() * 4);
  // the descriptor tables
  std::vector<pbd::DField> fields((size_t)std::max(nf, 1));
  std::vector<pbd::DMember> members(1);
  std::string names;
  members.clear();
  for (int j = 0; j < nf; j++) {
    pbd::DField &d = fields[(size_t)j];
    d.number = fl[j].number; d.ptype = fl[j].ptype; d.mem_off = (int32_t)members.size(); d.nmem = fl[j].nmembers; d.repeated = fl[j].repeated; d.oneof = fl[j].oneof;
    d.name_off = (uint32_t)names.size(); d.name_len = (uint32_t)std::strlen(fl[j].name); names += fl[j].name;
    for (int k = 0; k < fl[j].nmembers; k++) {
      const tfgpu_pb_member &mb = fl[j].members[k];
      members.push_back(pbd::DMember{mb.number, mb.ptype, (uint32_t)names.size(), (uint32_t)std::strlen(mb.name)});
      names += mb.name;
    }
  }
  names.append(16, '\0');
  if (members.empty()) members.push_back(pbd::DMember{0, 0, 0, 0});
  std::vector<int16_t> lut(1024, (int16_t)-1);
  for (int j = 0; j < nf; j++) if (fl[j].number < (int32_t)lut.size()) lut[(size_t)fl[j].number] = (int16_t)j;
  Buf blut = upload_small(lut.data(), lut.size() * 2);
  Buf bfields = upload_small(fields.data(), fields.size() * sizeof(pbd::DField)), bmembers = upload_small(members.data(), members.size() * sizeof(pbd::DMember)), bnames = upload_small(names.data(), names.size());
  Buf rec = dalloc((size_t)std::max(nf, 1) * (size_t)nma * 8 + 16), present = dalloc((size_t)std::max(nf, 1) * (size_t)nma + 16);
  Buf status = dalloc_zero((size_t)nma + 16), keep = dalloc_zero((size_t)(nma + 1) * 4 + 16), nerr = dalloc_zero(16);
  p.ms = ptr<uint32_t>(bms); p.nmsg = nmsg; p.schema_id = schema_id; p.schema_code = code; p.report_frame_errors = report_frame_errors;
  p.fields = ptr<pbd::DField>(bfields); p.nfields = code == TFGPU_ROW_OK ? nf : 0; p.members = ptr<pbd::DMember>(bmembers); p.names = ptr<uint8_t>(bnames);
  p.lut = ptr<int16_t>(blut); p.lutn = (int32_t)lut.size();
  p.rec = ptr<uint64_t>(rec); p.present = ptr<uint8_t>(present); p.status = ptr<uint8_t>(status); p.keep = ptr<uint32_t>(keep); p.nerr = ptr<uint32_t>(nerr);
  if (nmsg) { KernelTimer t("pb_decode"); pbd::pb_decode<<<pbd::nblk(nmsg, 128), 128, 0, st>>>(p); }
  exclusive_scan_u32(p.keep, p.keep, nmsg, true);
  const uint32_t *hrows = d2h_u32(p.keep + nmsg), *hnerr = d2h_u32(p.nerr);
  tf::sync();
  const int64_t nrows = *hrows, nra = std::max<int64_t>(nrows, 1);
  uint32_t nerr_total = *hnerr;
  if (nf == 0 && nrows) return tf::fail(TFGPU_ERR_UNSUPPORTED, "confluent SR protobuf: a message without fields");

  auto db = std::make_unique<tfgpu_dbatch>();
  db->nrows = nrows;
  db->ns = ns ? ns : ""; db->table = table ? table : "";
  db->src_row = dalloc((size_t)nra * 4); db->part_id = dalloc((size_t)nra * 4);
  Buf row_msg = dalloc((size_t)nra * 4), host_rows = dalloc_zero((size_t)(nra + 1) * 4 + 16);
  p.row_msg = ptr<uint32_t>(row_msg); p.nrows = nrows;
  std::vector<pbd::OutCol> oc((size_t)std::max(nf, 1));
  std::vector<int32_t> text_cols;
  const int64_t seg_stride = ((nrows + 1 + 3) / 4) * 4;
  int ntext = 0;
  bool has_msg = false;
  for (int j = 0; j < nf; j++) { const int t = fl[j].ptype; if (fl[j].repeated || t == TFGPU_PB_STRING || t == TFGPU_PB_BYTES || t == TFGPU_PB_MESSAGE) ntext++; if (fl[j].repeated || t == TFGPU_PB_MESSAGE) has_msg = true; }
  Buf lens_all = dalloc_zero((size_t)std::max(ntext, 1) * (size_t)seg_stride * 4 + 16);
  // the columns' validity bitmaps as views of one block cleared by one fill (a fill per column was two dispatches each: 0.6 ms of a 60-column batch)
  const size_t vbytes = ((size_t)((nra + 63) / 64) * 8 + 8 + 63) & ~(size_t)63;
  Buf valid_all = dalloc_zero((size_t)std::max(nf, 1) * vbytes);
  int ti = 0;
  for (int j = 0; j < nf; j++) {
    DColumn d;
    d.name = fl[j].name;
    pbd::OutCol &c = oc[(size_t)j];
    std::memset(&c, 0, sizeof c);
    c.ptype = fl[j].ptype;
    auto fixed = [&](int dtype, int repr, size_t w) { d.dtype = dtype; d.repr = repr; d.values = dalloc((size_t)nra * w + 8); c.values = d.values->p; };
    if (fl[j].repeated) { d.dtype = TFGPU_T_ANY; d.repr = TFGPU_R_JSON; }   // handleField: a repeated field is `any`
    else switch (fl[j].ptype) {  // protoSchemaTypes (types_protobuf.go:16-35) and the Go value unpackNotRepeatedVal asserts
      case TFGPU_PB_DOUBLE: fixed(TFGPU_T_FLOAT64, TFGPU_R_FLOAT64, 8); break;
      case TFGPU_PB_FLOAT: fixed(TFGPU_T_FLOAT32, TFGPU_R_FLOAT32, 4); break;
      case TFGPU_PB_INT64: case TFGPU_PB_SFIXED64: case TFGPU_PB_SINT64: fixed(TFGPU_T_INT64, TFGPU_R_INT64, 8); break;
      case TFGPU_PB_UINT64: case TFGPU_PB_FIXED64: fixed(TFGPU_T_UINT64, TFGPU_R_UINT64, 8); break;
      case TFGPU_PB_INT32: case TFGPU_PB_SFIXED32: case TFGPU_PB_SINT32: fixed(TFGPU_T_INT32, TFGPU_R_INT32, 4); break;
      case TFGPU_PB_UINT32: case TFGPU_PB_FIXED32: fixed(TFGPU_T_UINT32, TFGPU_R_UINT32, 4); break;
      case TFGPU_PB_BOOL: fixed(TFGPU_T_BOOLEAN, TFGPU_R_BOOL, 1); break;
      case TFGPU_PB_ENUM: fixed(TFGPU_T_UTF8, TFGPU_R_INT32, 4); break;   // ytschema.TypeString holding val.(int32)
      case TFGPU_PB_STRING: d.dtype = TFGPU_T_UTF8; d.repr = TFGPU_R_STRING; break;
      case TFGPU_PB_BYTES: d.dtype = TFGPU_T_BYTES; d.repr = TFGPU_R_BYTES; break;
      case TFGPU_PB_MESSAGE: d.dtype = TFGPU_T_ANY; d.repr = TFGPU_R_JSON; break;
      default: return tf::fail(TFGPU_ERR_INVALID, "tfgpu_sr_proto_parse: bad field type");
    }
    if (!d.values) {
      d.offsets = subbuf(lens_all, (size_t)ti * (size_t)seg_stride * 4, (size_t)(nrows + 1) * 4);
      c.lens = ptr<uint32_t>(d.offsets);
      text_cols.push_back(j);
      ti++;
    }
    d.validity = subbuf(valid_all, (size_t)j * vbytes, (size_t)((nra + 63) / 64) * 8 + 8);
    c.validity = ptr<uint8_t>(d.validity);
    db->schema.push_back({d.name, d.dtype});
    db->cols.push_back(std::move(d));
  }
  std::vector<int32_t> light_cols, any_cols, wide_cols;
  auto wide = [&](int j) { return fl[j].repeated == 2 || (fl[j].repeated && fl[j].ptype == TFGPU_PB_MESSAGE); };   // maps, repeated messages
  for (int j = 0; j < nf; j++) (wide(j) ? wide_cols : (fl[j].repeated || fl[j].ptype == TFGPU_PB_MESSAGE) ? any_cols : light_cols).push_back(j);
  Buf boc = upload_small(oc.data(), oc.size() * sizeof(pbd::OutCol));
  if (nrows) {
    pbd::pb_row_msgs<<<pbd::nblk(nmsg, 256), 256, 0, st>>>(p);
    KernelTimer t("pb_cells");
    const int32_t none = 0;
    Buf bl = light_cols.empty() ? upload_small(&none, 4) : upload_small(light_cols.data(), light_cols.size() * 4), ba = any_cols.empty() ? upload_small(&none, 4) : upload_small(any_cols.data(), any_cols.size() * 4);
    if (!light_cols.empty()) pbd::pb_cells<false><<<dim3(pbd::nblk(nrows, 256), (unsigned)light_cols.size()), 256, 0, st>>>(p, ptr<pbd::OutCol>(boc), ptr<int32_t>(bl), ptr<int32_t>(db->src_row), ptr<uint32_t>(db->part_id), ptr<uint32_t>(host_rows));
    else pbd::pb_src_rows<<<pbd::nblk(nrows, 256), 256, 0, st>>>(p, ptr<int32_t>(db->src_row), ptr<uint32_t>(db->part_id));
    if (!any_cols.empty()) pbd::pb_cells<true><<<dim3(pbd::nblk(nrows, 256), (unsigned)any_cols.size()), 256, 0, st>>>(p, ptr<pbd::OutCol>(boc), ptr<int32_t>(ba), ptr<int32_t>(db->src_row), ptr<uint32_t>(db->part_id), ptr<uint32_t>(host_rows));
    if (!wide_cols.empty()) { Buf bw = upload_small(wide_cols.data(), wide_cols.size() * 4); pbd::pb_cells<true, true><<<dim3(pbd::nblk(nrows, 256), (unsigned)wide_cols.size()), 256, 0, st>>>(p, ptr<pbd::OutCol>(boc), ptr<int32_t>(bw), ptr<int32_t>(db->src_row), ptr<uint32_t>(db->part_id), ptr<uint32_t>(host_rows)); }
  }
  if (ntext) {
    // The `any` marshalling expands its input (absent members as "name":0, \u00XX, base64): a column's bytes are summed in 64 bits
    // BEFORE the 32-bit scan, and a column of 4 GiB or more is refused — the scan would wrap and the text kernels write at wrapped offsets.
    Buf tot64 = dalloc_zero((size_t)ntext * 8 + 16);
    sum_u32_segments_u64(ptr<uint32_t>(lens_all), nrows, ntext, seg_stride, ptr<unsigned long long>(tot64));
    exclusive_scan_u32_segments(ptr<uint32_t>(lens_all), nrows, ntext, seg_stride);
    const uint32_t *h64 = d2h_u32(tot64->p, (size_t)ntext * 2);
    tf::sync();
    for (int t = 0; t < ntext; t++) {
      const uint64_t bytes = (uint64_t)h64[2 * t] | (uint64_t)h64[2 * t + 1] << 32;
      if (text_total_exceeds(bytes)) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_sr_proto_parse: column " + db->cols[(size_t)text_cols[(size_t)t]].name + " holds 4 GiB of text or more: split the batch");
    }
    for (int t = 0; t < ntext; t++) {
      DColumn &d = db->cols[(size_t)text_cols[(size_t)t]];
      d.data_len = (uint64_t)h64[2 * t] | (uint64_t)h64[2 * t + 1] << 32;  // (below 4 GiB, checked above: what the 32-bit scan left at [nrows])
      d.data = dalloc(d.data_len + 16);
      oc[(size_t)text_cols[(size_t)t]].data = ptr<uint8_t>(d.data);
    }
    boc = upload_small(oc.data(), oc.size() * sizeof(pbd::OutCol));
    if (nrows) {
      KernelTimer t("pb_text");
      std::vector<int32_t> copy_cols, any_text, wide_text;
      for (int32_t j : text_cols) (wide(j) ? wide_text : (fl[j].repeated || fl[j].ptype == TFGPU_PB_MESSAGE) ? any_text : copy_cols).push_back(j);
      if (!copy_cols.empty()) { Buf bc = upload_small(copy_cols.data(), copy_cols.size() * 4); pbd::pb_text_copy<<<dim3(pbd::nblk(nrows, 256), (unsigned)copy_cols.size()), 256, 0, st>>>(p, ptr<pbd::OutCol>(boc), ptr<int32_t>(bc)); }
      if (!any_text.empty()) { Buf bc = upload_small(any_text.data(), any_text.size() * 4); pbd::pb_text_any<false><<<dim3(pbd::nblk(nrows, 256), (unsigned)any_text.size()), 256, 0, st>>>(p, ptr<pbd::OutCol>(boc), ptr<int32_t>(bc)); }
      if (!wide_text.empty()) { Buf bc = upload_small(wide_text.data(), wide_text.size() * 4); pbd::pb_text_any<true><<<dim3(pbd::nblk(nrows, 256), (unsigned)wide_text.size()), 256, 0, st>>>(p, ptr<pbd::OutCol>(boc), ptr<int32_t>(bc)); }
    }
  }
  // rows whose `any` value holds a NaN / Inf: the reference keeps the Go float inside the map / slice, the column's JSON text cannot
  std::vector<uint32_t> hostm;
  std::unique_ptr<tfgpu_dbatch> result = std::move(db);
  bool some_host_rows = false;
  if (has_msg && nrows) {  // almost never: one flag word says so, the per-row marks come down only when it is set
    const uint32_t *hflag = any_nonzero_to_host(ptr<uint32_t>(host_rows), nrows);
    tf::sync();
    some_host_rows = *hflag != 0;
  }
  if (some_host_rows) {
    std::vector<uint32_t> hr((size_t)nrows), rm((size_t)nrows);
    d2h(hr.data(), host_rows->p, (size_t)nrows * 4); d2h(rm.data(), row_msg->p, (size_t)nrows * 4);
    tf::sync();
    bool any = false;
    std::vector<uint32_t> keepv((size_t)nrows + 1, 0u);
    for (int64_t r = 0; r < nrows; r++) { if (hr[(size_t)r]) { any = true; hostm.push_back(rm[(size_t)r]); } else keepv[(size_t)r] = 1u; }
    if (any) {
      Buf dk = dalloc(keepv.size() * 4 + 16);
      h2d(dk->p, keepv.data(), keepv.size() * 4);
      result = tf::compact_rows(*result, dk);
    }
  }
  int64_t ne = 0;
  if (nerr_total || !hostm.empty()) {
    std::vector<uint8_t> hst((size_t)nma);
    d2h(hst.data(), status->p, (size_t)nmsg);
    tf::sync();
    size_t hi = 0;
    for (int64_t m = 0; m < nmsg; m++) {
      int c = hst[(size_t)m];
      if (hi < hostm.size() && hostm[hi] == (uint32_t)m) { c = TFGPU_ROW_HOST_FALLBACK; hi++; }
      if (c == pbd::ST_OK || c == pbd::ST_SKIP) continue;
      if (errs && ne < errs_cap) errs[ne] = tfgpu_row_error{m, c, (int32_t)m, -1};
      ne++;
    }
  }
  if (nerrs) *nerrs = ne;
  *out = result.release();
  return TFGPU_OK;
  TF_API_END
}
