// tf_dbzemit.hip — tfgpu_debezium_emit: the Debezium EMITTER of the queue sinks, from device columns.
//
// Reference: queue.DebeziumSerializer.serialize (pkg/serializer/queue/debezium_serializer.go:26-43) → Emitter.EmitKV
// (pkg/debezium/emitter_value_converter.go:574-690): 0..3 (key, value) messages per ChangeItem — one for an insert or a plain
// update, (delete, tombstone) for a delete, (delete, tombstone, insert) for an update that changed its primary key.  Every
// message is PackerIncludeSchema.Pack (packer/packer_include_schema.go:14-40; PackerSkipSchema's bare payload when the settings
// disable schemas): {"payload":P,"schema":S} with S rendered once per
// table (ToKafkaSchemaKey / ToKafkaSchemaVal, :384-448; getFieldDescr, fields_descr.go:19-69; buildSourceSchemaDescr,
// fields_descr_source.go:7-107) and P = valPayload (:456-527) / makeKey (:255-271), all Go maps marshalled by
// util.JSONMarshalUnescape: members in byte order of their names, no HTML escaping.
//
// Here: the host renders S and a CELL LIST per stream (keys, values) — constants, converted column values (AddPg,
// pkg/debezium/pg/emitter.go:262-629) and the op / source / ts_ms tail, each with the predicate of the events it belongs to —
// and the kernels run it over the event list: a length pass and a write pass in which one lane walks one event's cells (descriptors are
// scalar: every lane is at the same cell), a scan between them; constants longer than 64 bytes (S is kilobytes) are filled by a
// coalesced pass of their own.
// HBM-bound byte work: the schema constant dominates the bytes of a message, as it does in the reference's output.
//
// Device-resident original types (everything else is refused BY NAME with TFGPU_ERR_UNSUPPORTED and stays with the stock
// emitter): pg:boolean, bit(1), smallint, integer, bigint, oid, real, double precision, text / character* / uuid / cidr /
// macaddr / citext / int4range / int8range / daterange, inet, bytea, date, timestamp[(p)] with / without time zone (time.Time
// values), time[(p)] with / without time zone, json / jsonb, hstore (a map), xml, numeric[(p,s)] (precise: up to 38 digits;
// string), money, bit(n) / bit varying(n), point, interval, tsrange, numrange and tstzrange (two plain bounds).  Texts whose reading
// belongs to jackc/pgtype's parsers (infinity / empty / unbounded ranges; odd clock shapes) are left to the host, value by value.
// The ydb: (ydb/emitter.go) and mysql: (mysql/emitter.go) families are resident as well, with their `source` blocks.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "tf_common.hpp"
#include "tf_emit.hpp"
#include "tf_devparse.hpp"

namespace tf {
namespace dbz {

enum { DK_BOOL = 1, DK_BIT1, DK_SMALLINT, DK_INTEGER, DK_BIGINT, DK_OID, DK_REAL, DK_DOUBLE, DK_STRING, DK_INET, DK_BYTEA, DK_DATE, DK_TS, DK_TSTZ,
       DK_JSON, DK_NUMERIC, DK_NUMERIC_TEXT, DK_BITS, DK_TIME, DK_TIMETZ, DK_MONEY, DK_XML, DK_POINT, DK_TSRANGE, DK_NUMRANGE, DK_TSTZRANGE, DK_INTERVAL, DK_MARSHAL, DK_YDB_UINT64, DK_YDB_DATE, DK_INT_REPRS, DK_MY_TINYINT1, DK_MY_FLOAT, DK_MY_BINARY, DK_MY_BIT1, DK_MY_BITS, DK_MY_TIMESTAMP, DK_MY_DATETIME, DK_MY_TIME, DK_MY_DECIMAL, DK_MY_YEAR, DK_ARR_INT, DK_ARR_STRING, DK_ARR_COPY, DK_HSTORE, DK_ARR_ELEM, DK_TO_STRING, DK_WRONG_TYPE = 100 /* "unknown type of value" */, DK_HOST_TYPE = 101 };
enum { EC_CONST = 0, EC_VALUE = 1, EC_TAIL = 2 };
// which events a cell belongs to
enum { EA_VALUE = 0 /* every event that has a value */, EA_AFTER, EA_AFTER_NULL, EA_BEFORE_NULL, EA_BEFORE_D, EA_BEFORE_U,
       EA_KEY = 8 /* every event */, EA_KEY_NEW, EA_KEY_OLD };
enum { EV_REGULAR = 0, EV_DELETE = 1, EV_TOMBSTONE = 2, EV_INSERT = 3 };  // emitType (emitter_value_converter.go:76-97)
enum { TN_NONE = 0, TN_LSN, TN_TS, TN_ID, TN_STEP, TN_TXID, TN_FILE6, TN_POS };
constexpr uint32_t CONST_INLINE = 64;

struct ECell {
  DCol c;
  uint32_t pre_off, pre_len;  // constant bytes in front of the value (EC_CONST: the constant itself)
  uint32_t kind, apply, dk;
  uint32_t arg;               // DK_TS: divider; DK_NUMERIC: schema scale | put-scale << 16
  uint32_t from_old;          // the column is an OldKeys column: nil in rows without OldKeys
  uint32_t edk;               // DK_ARR_ELEM: the converter of one element
  DCol alt; uint32_t dk_alt, has_alt;  // … unless the row's own value stands in (a MySQL delete's `before`: ColumnValues under the OldKeys)
  const uint8_t *absent;      // `after` members of a batch whose rows list different columns (DColumn::absent): a row that leaves the column out
  uint32_t ph_off, ph_len;    // … writes the TOAST placeholder here (buildKV, emitter_value_converter.go:311-323), whatever the column's type
};
struct EParams {
  const ECell *cells; int32_t ncells;
  int64_t nev;
  const int32_t *ev_row; const uint8_t *ev_type;  // null: event e is row e, regular
  const uint8_t *kind; const uint8_t *old_present; const int32_t *src_row;
  int32_t has_old, has_prev, snapshot;
  const uint32_t *m_id; const uint64_t *m_lsn, *m_commit;
  const uint32_t *m_tx_off; const uint8_t *m_tx;  // ChangeItem.TxID (dt.source.type = ydb)
  const uint8_t *blob;
  uint32_t tseg_off[6], tseg_len[6]; int32_t tnum[6]; int32_t ntseg;  // EC_TAIL: ,"op":" <op> then segment k followed by number k
  const double *p10; const uint64_t *p128;
  uint32_t *cell;      // [ncells][nev]
  uint32_t *ev_len;    // [nev+1]
  unsigned long long *total64, *err;  // err[0]: first (event << 16 | cell) the reference fails on, err[1]: first one left to the host
  uint8_t *out;
};

struct PtrView { const uint8_t *p; __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return p[i]; } };

__device__ __forceinline__ int64_t trunc_div(int64_t a, int64_t b) { return a / b; }  // Go and C++ both truncate toward zero

__device__ __forceinline__ uint32_t event_op(const EParams &p, int64_t e, int64_t r) {
  const uint32_t k = p.kind ? p.kind[r] : (uint32_t)TFGPU_K_INSERT;
  const uint32_t t = p.ev_type ? p.ev_type[e] : (uint32_t)EV_REGULAR;
  if (k == TFGPU_K_INSERT) return p.snapshot ? 'r' : 'c';
  if (k == TFGPU_K_UPDATE) return t == EV_REGULAR ? 'u' : t == EV_DELETE ? 'd' : 'c';
  return 'd';
}
__device__ __forceinline__ bool row_has_old(const EParams &p, int64_t r) { return p.has_old && (!p.old_present || ((p.old_present[r >> 3] >> (r & 7)) & 1)); }
__device__ __forceinline__ bool applies(const EParams &p, uint32_t apply, int64_t e, int64_t r) {
  const uint32_t t = p.ev_type ? p.ev_type[e] : (uint32_t)EV_REGULAR;
  if (apply >= EA_KEY) {
    if (apply == EA_KEY) return true;
    const bool from_new = t == EV_INSERT || !row_has_old(p, r);  // makeKey: useAfter || len(OldKeys.KeyNames) == 0
    return (apply == EA_KEY_NEW) == from_new;
  }
  if (t == EV_TOMBSTONE) return false;
  const uint32_t op = event_op(p, e, r);
  switch (apply) {
    case EA_AFTER: return op != 'd';
    case EA_AFTER_NULL: return op == 'd';
    case EA_BEFORE_D: return op == 'd';
    case EA_BEFORE_U: return op == 'u' && p.has_prev && row_has_old(p, r);
    case EA_BEFORE_NULL: return op != 'd' && !(op == 'u' && p.has_prev && row_has_old(p, r));
    default: return true;
  }
}

// a JSON string whose content is itself written through emit_json_string: only '"' and '\' are left to escape
template <class S> struct JsonEscSink {
  S &s;
  __device__ __forceinline__ void put(uint32_t c) { if (c == '"' || c == '\\') s.put('\\'); s.put(c); }
};

template <class S, class G> __device__ __forceinline__ void emit_base64_gen(S &s, uint32_t n, const G &get) {
  const char *T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  uint32_t i = 0;
  for (; i + 3 <= n; i += 3) {
    const uint32_t v = get(i) << 16 | get(i + 1) << 8 | get(i + 2);
    s.put(T[v >> 18]); s.put(T[(v >> 12) & 63]); s.put(T[(v >> 6) & 63]); s.put(T[v & 63]);
  }
  if (n - i == 1) { const uint32_t v = get(i) << 16; s.put(T[v >> 18]); s.put(T[(v >> 12) & 63]); s.put('='); s.put('='); }
  else if (n - i == 2) { const uint32_t v = get(i) << 16 | get(i + 1) << 8; s.put(T[v >> 18]); s.put(T[(v >> 12) & 63]); s.put(T[(v >> 6) & 63]); s.put('='); }
}

typedef unsigned __int128 u128;
__device__ __forceinline__ bool mul10(u128 &c, uint32_t add) {
  const u128 lim = (~(u128)0 - 9) / 10;
  if (c > lim) return false;
  c = c * 10 + add;
  return true;
}
// DecimalToDebeziumHandlingModePrecise (typeutil/helpers.go:269-300) over the value's text: ExponentialFloatFormToNumeric (:340-365,
// shopspring's String()), StringFixed(schema scale) when the type has one, DecimalToDebeziumPrimitivesImpl (:388-419) — whose two's
// complement is computed from big.Int.Bytes() of the inverted magnitude (leading zero bytes lost): restated as written.
// 0 ok, 1 the reference fails, 2 not decided here (more than 38 digits, an exponent past ±64)
template <class S> __device__ int emit_numeric(S &s, const uint8_t *t, uint32_t n, uint32_t schema_scale, bool put_scale, bool verbatim = false) {
  if (n == 0 && !verbatim) return 1;  // "empty string as an input is not supported" (DecimalToDebeziumPrimitives alone takes "" as zero)
  bool has_e = false, only0 = true;
  int32_t first_dot = -1;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t c = t[i];
    if (c == 'e' || c == 'E') has_e = true;
    if (c == '.' && first_dot < 0) { first_dot = (int32_t)i; continue; }
    if (c != '0' && c != '-') only0 = false;
  }
  if (verbatim && has_e) return 1;  // no ExponentialFloatFormToNumeric in front: big.Int.SetString refuses the letter
  bool neg = false; u128 C = 0; int32_t sc = 0;
  bool zero_text = false;
  if (!has_e && schema_scale == 0 && only0) { zero_text = true; sc = first_dot < 0 ? 0 : (int32_t)n - 1 - first_dot; }  // containsOnly(decimalInt, '0', '-')
  else {
    uint32_t i = 0;
    bool lead_neg = false;
    if (has_e && t[0] == '-') { lead_neg = true; i = 1; }  // the sign is split off before shopspring sees the rest
    if (i < n && (t[i] == '+' || t[i] == '-')) { if (lead_neg) return 2; neg = t[i] == '-'; i++; }
    if (lead_neg) neg = true;
    uint32_t nd = 0; int32_t F = 0; bool dot = false;
    for (; i < n; i++) {
      const uint32_t c = t[i];
      if (c >= '0' && c <= '9') { if (!mul10(C, c - '0')) return 2; nd++; if (dot) F++; }
      else if (c == '.' && !dot) dot = true;
      else break;
    }
    if (nd == 0) return 1;
    int32_t E = 0;
    if (i < n) {
      if (t[i] != 'e' && t[i] != 'E') return 1;
      i++;
      bool eneg = false;
      if (i < n && (t[i] == '+' || t[i] == '-')) { eneg = t[i] == '-'; i++; }
      if (i >= n) return 1;
      for (; i < n; i++) { const uint32_t c = t[i]; if (c < '0' || c > '9') return 1; E = E * 10 + (int32_t)(c - '0'); if (E > 64) return 2; }
      if (eneg) E = -E;
    }
    sc = F - E;
    if (has_e) {  // Decimal.String(): no exponent, trailing fractional zeros trimmed
      if (C == 0) sc = 0;
      while (sc > 0 && C % 10 == 0) { C /= 10; sc--; }
      while (sc < 0) { if (!mul10(C, 0)) return 2; sc++; }
    }
    if (schema_scale > 0) {  // StringFixed: half away from zero
      while (sc > (int32_t)schema_scale + 1) { C /= 10; sc--; }
      if (sc == (int32_t)schema_scale + 1) { const uint32_t d = (uint32_t)(C % 10); C /= 10; if (d >= 5) C += 1; sc--; }
      while (sc < (int32_t)schema_scale) { if (!mul10(C, 0)) return 2; sc++; }
    }
    if (sc < 0) return 1;
    if (C == 0 && !has_e && schema_scale == 0) return 2;  // a zero spelt with a '+' ("+0.0"): the reference indexes an empty byte slice (isHighestBitSet, helpers.go:966-968)
  }
  uint8_t buf[17]; uint32_t bn = 0;
  if (zero_text || C == 0) { buf[0] = 0; bn = 1; }
  else {
    uint32_t L = 0; { u128 x = C; while (x) { L++; x >>= 8; } }
    if (neg) {
      const u128 mask = L == 16 ? ~(u128)0 : (((u128)1 << (8 * L)) - 1);
      const u128 v = ((~C) & mask) + 1;   // big.Int.SetBytes(^bytes) + 1: C >= 1, so no carry out of L bytes
      uint32_t L2 = 0; { u128 x = v; while (x) { L2++; x >>= 8; } }
      const bool top = (uint32_t)(v >> (8 * (L2 - 1))) & 0x80u;
      if (!top) buf[bn++] = 0xFF;
      for (uint32_t k = 0; k < L2; k++) buf[bn++] = (uint8_t)(v >> (8 * (L2 - 1 - k)));
    } else {
      const bool top = (uint32_t)(C >> (8 * (L - 1))) & 0x80u;
      if (top) buf[bn++] = 0;
      for (uint32_t k = 0; k < L; k++) buf[bn++] = (uint8_t)(C >> (8 * (L - 1 - k)));
    }
  }
  if (put_scale) { put_lit(s, "{\"scale\":"); emit_u64(s, (uint64_t)sc); put_lit(s, ",\"value\":"); }
  s.put('"'); emit_base64_gen(s, bn, [&](uint32_t i) { return (uint32_t)buf[i]; }); s.put('"');
  if (put_scale) s.put('}');
  return 0;
}

template <class S> __device__ __forceinline__ int emit_int_text(S &s, const uint8_t *t, uint32_t n) {  // json.Number.Int64()
  int64_t v; PtrView f{t};
  if (parse_int64(f, 0, n, false, &v) != 0) return 1;
  emit_i64(s, v);
  return 0;
}


// the text between a JSON string's quotes: emit_json_string through a sink that drops its first and its last byte
template <class S> struct BodySink {
  S &s; uint32_t state = 0, held = 0;
  __device__ __forceinline__ void put(uint32_t c) { if (state == 0) { state = 1; return; } if (state == 2) s.put(held); held = c; state = 2; }
};
template <class S> __device__ __forceinline__ void json_body(S &s, const uint8_t *p, uint32_t n) { BodySink<S> b{s}; emit_json_string(b, p, n, false); }
__device__ __forceinline__ bool dg2(const uint8_t *p, uint32_t &v) { if (p[0] < '0' || p[0] > '9' || p[1] < '0' || p[1] > '9') return false; v = (p[0] - '0') * 10u + (p[1] - '0'); return true; }
// "HH:MM:SS[.f{1,6}]" → microseconds of the day, *end = where the clock text stops; false = not this shape
__device__ __forceinline__ bool clock_micros(const uint8_t *t, uint32_t n, int64_t *us, uint32_t *end, bool comma_ok = false) {
  uint32_t h, m, sec;
  if (n < 8 || !dg2(t, h) || t[2] != ':' || !dg2(t + 3, m) || t[5] != ':' || !dg2(t + 6, sec)) return false;
  if (!comma_ok && (h > 23 || m > 59 || sec > 59)) return false;  // what pgtype's readers make of such a field is theirs to say (the callers leave the value to the host)
  int64_t v = ((int64_t)h * 3600 + m * 60 + sec) * 1000000;
  uint32_t i = 8;
  if (i < n && (t[i] == '.' || (comma_ok && t[i] == ','))) {  // (time.Parse takes a comma for the period of a fractional second; pgtype's strconv-based readers do not)
    uint32_t k = 0, f = 0; i++;
    while (i < n && t[i] >= '0' && t[i] <= '9' && k < 6) { f = f * 10 + (t[i] - '0'); i++; k++; }
    if (k == 0) return false;
    for (; k < 6; k++) f *= 10;
    v += f;
  }
  *us = v; *end = i;
  return true;
}
// a zone "Z" | "±hh[:mm[:ss]]" at [i, n): seconds east, false = not this shape
__device__ __forceinline__ bool zone_seconds(const uint8_t *t, uint32_t i, uint32_t n, int32_t *off) {
  if (i + 1 == n && t[i] == 'Z') { *off = 0; return true; }
  if (i >= n || (t[i] != '+' && t[i] != '-')) return false;
  const bool neg = t[i] == '-'; i++;
  uint32_t h = 0, m = 0, sec = 0;
  if (i + 2 > n || !dg2(t + i, h)) return false; i += 2;
  if (i < n) { if (t[i] != ':' || i + 3 > n || !dg2(t + i + 1, m)) return false; i += 3; }
  if (i < n) { if (t[i] != ':' || i + 3 > n || !dg2(t + i + 1, sec)) return false; i += 3; }
  if (i != n) return false;
  const int32_t v = (int32_t)(h * 3600 + m * 60 + sec);
  *off = neg ? -v : v;
  return true;
}
template <class S> __device__ __forceinline__ void emit_u128(S &s, u128 v) {
  uint8_t d[40]; int k = 0;
  do { d[k++] = (uint8_t)('0' + (uint32_t)(v % 10)); v /= 10; } while (v);
  while (k) s.put(d[--k]);
}
// ExponentialFloatFormToNumeric (typeutil/helpers.go:340-365) of one bound: the text as it is without an exponent, shopspring's String() with one.
// 0 ok, 1 the reference fails, 2 host
template <class S> __device__ int emit_exp_numeric(S &s, const uint8_t *t, uint32_t n) {
  if (n == 0) return 2;  // an unbounded side: pgtype's business
  bool has_e = false;
  for (uint32_t i = 0; i < n; i++) if (t[i] == 'e' || t[i] == 'E') has_e = true;
  if (!has_e) { json_body(s, t, n); return 0; }
  uint32_t i = 0; bool neg = false;
  if (t[0] == '-') { neg = true; i = 1; }
  if (i < n && t[i] == '+' && !neg) i++;  // shopspring takes a '+'
  else if (i < n && (t[i] == '+' || t[i] == '-')) return 2;
  u128 C = 0; uint32_t nd = 0; int32_t F = 0, E = 0; bool dot = false;
  for (; i < n; i++) { const uint32_t c = t[i]; if (c >= '0' && c <= '9') { if (!mul10(C, c - '0')) return 2; nd++; if (dot) F++; } else if (c == '.' && !dot) dot = true; else break; }
  if (nd == 0 || i >= n || (t[i] != 'e' && t[i] != 'E')) return 1;
  i++;
  bool eneg = false;
  if (i < n && (t[i] == '+' || t[i] == '-')) { eneg = t[i] == '-'; i++; }
  if (i >= n) return 1;
  for (; i < n; i++) { const uint32_t c = t[i]; if (c < '0' || c > '9') return 1; E = E * 10 + (int32_t)(c - '0'); if (E > 64) return 2; }
  int32_t sc = F - (eneg ? -E : E);
  if (C == 0) sc = 0;
  while (sc > 0 && C % 10 == 0) { C /= 10; sc--; }
  while (sc < 0) { if (!mul10(C, 0)) return 2; sc++; }
  if (neg) s.put('-');
  if (sc == 0) { emit_u128(s, C); return 0; }
  u128 pw = 1; for (int32_t k = 0; k < sc; k++) pw *= 10;   // sc <= 38 + 64: C < 10^38 keeps C / pw meaningful only while pw fits
  if (sc > 38) {  // 0.000…digits
    put_lit(s, "0."); uint32_t dl = 0; { u128 x = C; while (x) { dl++; x /= 10; } }
    for (int32_t k = 0; k < sc - (int32_t)dl; k++) s.put('0');
    emit_u128(s, C); return 0;
  }
  emit_u128(s, C / pw); s.put('.');
  const u128 fr = C % pw; uint32_t dl = 0; { u128 x = fr; while (x) { dl++; x /= 10; } }
  for (int32_t k = 0; k < sc - (int32_t)dl; k++) s.put('0');
  if (fr) emit_u128(s, fr);
  return 0;
}
// "YYYY-MM-DD HH:MM:SS[.f]zone" (pgtype.Timestamptz.DecodeText's shapes) → unix seconds; false = not this shape
__device__ __forceinline__ bool pg_tstz_seconds(const uint8_t *t, uint32_t n, int64_t *out) {
  if (n < 20) return false;
  for (int k = 0; k < 4; k++) if (t[k] < '0' || t[k] > '9') return false;
  uint32_t mo, d; int64_t us; uint32_t end; int32_t off;
  if (t[4] != '-' || !dg2(t + 5, mo) || t[7] != '-' || !dg2(t + 8, d) || t[10] != ' ') return false;
  if (!clock_micros(t + 11, n - 11, &us, &end)) return false;
  {  // any count of fractional digits: the zone starts after them
    uint32_t i = 11 + 8;
    if (i < n && t[i] == '.') { i++; while (i < n && t[i] >= '0' && t[i] <= '9') i++; }
    if (!zone_seconds(t, i, n, &off)) return false;
  }
  const int64_t y = (t[0] - '0') * 1000 + (t[1] - '0') * 100 + (t[2] - '0') * 10 + (t[3] - '0');
  if (mo < 1 || mo > 12 || d < 1 || d > (uint32_t)dev::days_in_month((int)mo, y)) return false;
  *out = dev::days_from_civil(y, (int)mo, (int)d) * 86400 + us / 1000000 - off;
  return true;
}

// ParsePgDateTimeWithTimezone (typeutil/helpers.go:446-467): the layout is picked by in[10] and the last byte — "2006-01-02T15:04:05Z", "2006-01-02 15:04:05Z",
// "2006-01-02T15:04:05-07:00", "2006-01-02 15:04:05-07" — and time.Parse takes a fractional second after the seconds.  0 ok, 1 the reference fails, 2 host
__device__ __forceinline__ int pg_datetime_tz(const uint8_t *t, uint32_t n, int64_t *sec, int32_t *nsec) {
  if (n < 11) return 2;  // in[10] panics
  const bool tee = t[10] == 'T', zed = t[n - 1] == 'Z';
  for (int k = 0; k < 4; k++) if (t[k] < '0' || t[k] > '9') return 1;
  uint32_t mo, d, h, m, sc;
  if (n < 19 || t[4] != '-' || !dg2(t + 5, mo) || t[7] != '-' || !dg2(t + 8, d) || t[10] != (tee ? 'T' : ' ') || !dg2(t + 11, h) || t[13] != ':' || !dg2(t + 14, m) || t[16] != ':' || !dg2(t + 17, sc)) return 1;
  uint32_t i = 19; int32_t ns = 0;
  if (i < n && (t[i] == '.' || t[i] == ',') && i + 1 < n && t[i + 1] >= '0' && t[i + 1] <= '9') {
    uint32_t k = 0; i++;
    while (i < n && t[i] >= '0' && t[i] <= '9') { if (k < 9) { ns = ns * 10 + (int32_t)(t[i] - '0'); k++; } i++; }
    for (; k < 9; k++) ns *= 10;
  }
  int32_t off = 0;
  if (zed) { if (i + 1 != n) return 1; }
  else {
    if (i >= n || (t[i] != '+' && t[i] != '-')) return 1;
    const bool neg = t[i] == '-'; uint32_t zh, zm = 0; i++;
    if (i + 2 > n || !dg2(t + i, zh)) return 1; i += 2;
    if (tee) { if (i + 3 != n || t[i] != ':' || !dg2(t + i + 1, zm)) return 1; }
    else if (i != n) return 1;
    if (zh > 24 || zm > 60) return 1;  // time.Parse's own ranges for a zone offset
    off = (int32_t)(zh * 3600 + zm * 60); if (neg) off = -off;
  }
  const int64_t y = (t[0] - '0') * 1000 + (t[1] - '0') * 100 + (t[2] - '0') * 10 + (t[3] - '0');
  if (mo < 1 || mo > 12 || d < 1 || d > (uint32_t)dev::days_in_month((int)mo, y) || h > 23 || m > 59 || sc > 59) return 1;
  *sec = dev::days_from_civil(y, (int)mo, (int)d) * 86400 + (int64_t)h * 3600 + m * 60 + sc - off; *nsec = ns;
  return 0;
}

// AddPg for one non-nil value: 0 ok, 1 the reference returns an error, 2 left to the host
template <class S> __device__ int emit_scalar(S &s, const EParams &p, const ECell &ec, const DCol &c, uint32_t dk, const CellBits &b) {
  uint32_t vn; const uint8_t *vp = cell_text(c, b, vn);
  switch (dk) {
    case DK_BOOL: put_lit(s, (uint8_t)b.v ? "true" : "false"); return 0;
    case DK_BIT1: put_lit(s, (vn == 1 && vp[0] == '1') ? "true" : "false"); return 0;
    case DK_SMALLINT: case DK_INTEGER: case DK_BIGINT: case DK_OID:
      if (c.repr == TFGPU_R_JSONNUM) return emit_int_text(s, vp, vn);
      if (c.repr == TFGPU_R_FLOAT64) {  // pg:oid, restored snapshot: int64(t)
        const double d = __longlong_as_double((long long)b.v);
        if (!(d > -9.2e18 && d < 9.2e18)) return 2;
        emit_i64(s, (int64_t)d); return 0;
      }
      emit_int(s, c, b); return 0;
    case DK_REAL: {
      double d;
      if (c.repr == TFGPU_R_JSONNUM) { PtrView f{vp}; const int rc = parse_float_go(f, 0, vn, p.p10, p.p128, &d); if (rc == 3) return 2; if (rc) return 1; }
      else d = cell_f64(c, b);
      const float f32 = (float)d;
      if (f32 != f32 || f32 - f32 != 0.f) return 1;  // json: unsupported value
      dev::fmt_json_float(s, (double)f32, 32); return 0;
    }
    case DK_DOUBLE: {
      double d;
      if (c.repr == TFGPU_R_JSONNUM) { PtrView f{vp}; const int rc = parse_float_go(f, 0, vn, p.p10, p.p128, &d); if (rc == 3) return 2; if (rc) return 1; }
      else d = cell_f64(c, b);
      if (d != d) { put_lit(s, "\"NaN\""); return 0; }
      if (d - d != 0) { put_lit(s, d < 0 ? "\"-Infinity\"" : "\"Infinity\""); return 0; }
      dev::fmt_json_float(s, d, 64); return 0;
    }
    case DK_STRING: emit_json_string(s, vp, vn, false); return 0;
    case DK_INET: if (vn >= 3 && vp[vn - 3] == '/' && vp[vn - 2] == '3' && vp[vn - 1] == '2') vn -= 3; emit_json_string(s, vp, vn, false); return 0;
    case DK_BYTEA:
      if (c.repr == TFGPU_R_BYTES) { s.put('"'); emit_base64(s, vp, vn); s.put('"'); }
      else emit_json_string(s, vp, vn, false);
      return 0;
    case DK_DATE: {
      int64_t sec = (int64_t)b.v; int32_t ns;
      if (c.repr == TFGPU_R_STRING) { const int rc = pg_datetime_tz(vp, vn, &sec, &ns); if (rc) return rc; }  // "restored snapshot": ParsePgDateTimeWithTimezone
      emit_i64(s, trunc_div(sec, 86400)); return 0;
    }
    case DK_TS: emit_i64(s, trunc_div((int64_t)b.v * 1000000 + b.ns / 1000, (int64_t)ec.arg)); return 0;
    case DK_TSTZ: {  // SprintfDebeziumTime (helpers.go:1105-1114)
      int64_t sec = (int64_t)b.v; int32_t ns = b.ns;
      if (c.repr == TFGPU_R_STRING) { const int rc = pg_datetime_tz(vp, vn, &sec, &ns); if (rc) return rc; }
      s.put('"'); emit_rfc3339nano(s, sec, ns); s.put('"'); return 0;
    }
    case DK_JSON:
      if (c.repr == TFGPU_R_JSON) { emit_json_string(s, vp, vn, false); return 0; }  // the column holds json.Marshal's text: marshalling it again is the identity
      { s.put('"'); JsonEscSink<S> q{s}; emit_json_string(q, vp, vn, false); s.put('"'); }  // a Go string: its marshal, as a string
      return 0;
    case DK_NUMERIC: return emit_numeric(s, vp, vn, ec.arg & 0xFFFFu, ((ec.arg >> 16) & 1u) != 0);
    case DK_NUMERIC_TEXT: emit_json_string(s, vp, vn, false); return 0;
    case DK_BITS: {  // ChangeItemsBitsToDebeziumHonest (helpers.go:48-78)
      bool any = false;
      for (uint32_t i = 0; i < vn; i++) if (vp[i] == '1') { any = true; break; }
      s.put('"');
      if (any) {
        const uint32_t size = (vn + 7) / 8;
        emit_base64_gen(s, size, [&](uint32_t j) {
          const uint32_t k = size - 1 - j; uint32_t v = 0;
          for (uint32_t q = 0; q < 8; q++) { const uint32_t i = 8 * k + q; if (i < vn && vp[i] == '1') v |= 1u << (7 - q); }
          return v; });
      }
      s.put('"');
      return 0;
    }
    case DK_TIME: {  // pgtype.Time.Scan(string) → Microseconds / divider (pg/emitter.go:541-561)
      int64_t us; uint32_t end;
      if (!clock_micros(vp, vn, &us, &end) || end != vn) return 2;
      { int64_t r = trunc_div(us, (int64_t)(ec.arg & 0xFFFFu)); if (ec.arg >> 16) r -= r % 1000; emit_i64(s, r); }  // into an array: divider 1, milliseconds' accuracy
      return 0;
    }
    case DK_TIMETZ: {  // TimeWithTimeZoneToTime(val).UTC().Format("15:04:05.999999Z") (pg/emitter.go:524-540)
      int64_t us; uint32_t end; int32_t off;
      if (!clock_micros(vp, vn, &us, &end) || end >= vn || vp[end] == 'Z' || !zone_seconds(vp, end, vn, &off)) return 2;
      int64_t sod = (us / 1000000 - off) % 86400; if (sod < 0) sod += 86400;
      s.put('"'); emit_clock(s, sod);
      uint32_t f = ec.arg ? 0u : (uint32_t)(us % 1000000);  // into an array: "15:04:05Z"
      if (f) { uint8_t d[6]; for (int k = 5; k >= 0; k--) { d[k] = (uint8_t)('0' + f % 10); f /= 10; } int last = 5; while (d[last] == '0') last--; s.put('.'); for (int k = 0; k <= last; k++) s.put(d[k]); }
      s.put('Z'); s.put('"'); return 0;
    }
    case DK_MONEY:  // DecimalToDebeziumPrimitives(colVal.(string)[1:]) (pg/emitter.go:477-482)
      if (vn == 0) return 2;  // the reference's slice expression panics
      if (ec.arg) { emit_json_string(s, vp + 1, vn - 1, false); return 0; }  // decimal.handling.mode = string
      return emit_numeric(s, vp + 1, vn - 1, 0, false, true);
    case DK_XML: {  // typeutil.UnescapeUnicode (helpers.go:530-550): \uXXXX becomes the rune byte(XXXX); piecewise escaping is exact (the insertions are whole runes)
      s.put('"');
      uint32_t a = 0, i = 0;
      while (i < vn) {
        if (vp[i] == '\\' && vn - i > 5 && vp[i + 1] == 'u') {
          uint32_t u = 0; bool hex = true;
          for (uint32_t k = 2; k < 6; k++) { const uint32_t c = vp[i + k]; const uint32_t d = c >= '0' && c <= '9' ? c - '0' : (c | 32) >= 'a' && (c | 32) <= 'f' ? (c | 32) - 'a' + 10 : 99; if (d == 99) hex = false; u = u * 16 + d; }
          if (hex) {
            if (i > a) json_body(s, vp + a, i - a);
            uint8_t r[2]; const uint32_t b = u & 0xFF; uint32_t rn = 1;
            if (b < 0x80) r[0] = (uint8_t)b; else { r[0] = (uint8_t)(0xC0 | (b >> 6)); r[1] = (uint8_t)(0x80 | (b & 0x3F)); rn = 2; }
            json_body(s, r, rn);
            i += 6; a = i; continue;
          }
        }
        i++;
      }
      if (vn > a) json_body(s, vp + a, vn - a);
      s.put('"'); return 0;
    }
    case DK_POINT: {  // PointToDebezium (helpers.go:552-572)
      if (vn < 2) return 2;
      uint32_t comma = 0, commas = 0;
      for (uint32_t i = 1; i + 1 < vn; i++) if (vp[i] == ',') { if (!commas) comma = i; commas++; }
      if (commas != 1) return 1;
      double x, y; PtrView f{vp};
      int rc = parse_float_go(f, 1, comma, p.p10, p.p128, &x); if (rc == 3) return 2; if (rc) return 1;
      rc = parse_float_go(f, comma + 1, vn - 1, p.p10, p.p128, &y); if (rc == 3) return 2; if (rc) return 1;
      if (x != x || y != y || x - x != 0 || y - y != 0) return 1;  // json: unsupported value
      put_lit(s, "{\"srid\":null,\"wkb\":\"\",\"x\":"); dev::fmt_json_float(s, x, 64); put_lit(s, ",\"y\":"); dev::fmt_json_float(s, y, 64); s.put('}');
      return 0;
    }
    case DK_TSRANGE: {  // pg/emitter.go:432-442: every comma-separated part unquoted, then quoted
      if (vn < 2 || vp[0] >= 0x80 || vp[vn - 1] >= 0x80) return 2;
      s.put('"'); json_body(s, vp, 1);
      uint32_t a = 1;
      for (uint32_t i = 1; i <= vn - 1; i++) if (i == vn - 1 || vp[i] == ',') {
        uint32_t pa = a, pb = i;
        if (pb - pa == 1 && vp[pa] == '"') return 2;  // UnquoteIfQuoted of a lone quote panics
        if (pb > pa && vp[pa] == '"' && vp[pb - 1] == '"') { pa++; pb--; }
        if (a != 1) s.put(',');
        put_lit(s, "\\\""); json_body(s, vp + pa, pb - pa); put_lit(s, "\\\"");
        a = i + 1;
      }
      json_body(s, vp + vn - 1, 1); s.put('"'); return 0;
    }
    case DK_NUMRANGE: {  // NumRangeToDebezium (helpers.go:574-591), the unquoted two-bound form
      if (vn < 3 || (vp[0] != '[' && vp[0] != '(') || (vp[vn - 1] != ']' && vp[vn - 1] != ')')) return 2;
      uint32_t comma = 0, commas = 0;
      for (uint32_t i = 1; i + 1 < vn; i++) { if (vp[i] == ',') { if (!commas) comma = i; commas++; } if (vp[i] == '"' || vp[i] == '\\') return 2; }
      if (commas != 1) return 2;
      put_lit(s, "\"[");
      int rc = emit_exp_numeric(s, vp + 1, comma - 1); if (rc) return rc;
      s.put(',');
      rc = emit_exp_numeric(s, vp + comma + 1, vn - 1 - (comma + 1)); if (rc) return rc;
      put_lit(s, ")\""); return 0;
    }
    case DK_TSTZRANGE: {  // TstZRangeQuote (helpers.go:604-614): both bounds in UTC as "2006-01-02 15:04:05+00", the input's brackets
      if (vn < 3 || (vp[0] != '[' && vp[0] != '(') || (vp[vn - 1] != ']' && vp[vn - 1] != ')')) return 2;
      uint32_t comma = 0, commas = 0;
      for (uint32_t i = 1; i + 1 < vn; i++) if (vp[i] == ',') { if (!commas) comma = i; commas++; }
      if (commas != 1) return 2;
      uint32_t la = 1, lb = comma, ra = comma + 1, rb = vn - 1;
      if (lb - la >= 2 && vp[la] == '"' && vp[lb - 1] == '"') { la++; lb--; }
      if (rb - ra >= 2 && vp[ra] == '"' && vp[rb - 1] == '"') { ra++; rb--; }
      int64_t l, r;
      if (!pg_tstz_seconds(vp + la, lb - la, &l) || !pg_tstz_seconds(vp + ra, rb - ra, &r)) return 2;
      s.put('"'); s.put(vp[0]);
      put_lit(s, "\\\""); emit_date(s, l); s.put(' '); emit_clock(s, l); put_lit(s, "+00\\\",\\\""); emit_date(s, r); s.put(' '); emit_clock(s, r); put_lit(s, "+00\\\"");
      s.put(vp[vn - 1]); s.put('"'); return 0;
    }
    case DK_INTERVAL: {  // ParsePostgresInterval (typeutil/helpers.go:469-507) over pgtype v1.12.0's Interval.DecodeText: (count, unit) pairs, then [-]H:MM:SS[.f]
      uint32_t parts = 1;
      for (uint32_t i = 0; i < vn; i++) if (vp[i] == ' ') parts++;
      int32_t months = 0, days = 0; int64_t micro = 0;
      PtrView f{vp};
      uint32_t a = 0;  // start of the current part
      auto part_end = [&](uint32_t from) { uint32_t e = from; while (e < vn && vp[e] != ' ') e++; return e; };
      for (uint32_t k = 0; k + 1 < parts; k += 2) {
        const uint32_t e0 = part_end(a), a1 = e0 + 1, e1 = part_end(a1);
        int64_t sc;
        if (parse_int64(f, a, e0, false, &sc) != 0) return 1;  // "bad interval format"
        const uint8_t *u = vp + a1; const uint32_t ul = e1 - a1;
        auto is = [&](const char *w, uint32_t wl) { if (ul != wl) return false; for (uint32_t q = 0; q < wl; q++) if (u[q] != (uint8_t)w[q]) return false; return true; };
        if (is("year", 4) || is("years", 5)) months = (int32_t)((uint32_t)months + (uint32_t)((uint64_t)sc * 12ull));
        else if (is("mon", 3) || is("mons", 4) || is("month", 5) || is("months", 6)) months = (int32_t)((uint32_t)months + (uint32_t)(uint64_t)sc);  // ("months" → "mons", "month" → "mon" first)
        else if (is("day", 3) || is("days", 4)) days = (int32_t)(uint32_t)(uint64_t)sc;
        a = e1 + 1;
      }
      if (parts & 1) {
        uint32_t c1 = vn, c2 = vn;
        for (uint32_t i = a; i < vn; i++) if (vp[i] == ':') { if (c1 == vn) c1 = i; else { c2 = i; break; } }
        if (c2 == vn) return 1;          // fewer than three ':'-separated pieces
        if (c1 == a) return 2;           // an empty hour text: pgtype indexes its first byte
        bool neg = false; uint32_t ha = a;
        if (vp[a] == '-') { neg = true; ha++; }
        int64_t h, m, sec, us = 0;
        if (parse_int64(f, ha, c1, false, &h) != 0 || parse_int64(f, c1 + 1, c2, false, &m) != 0) return 1;
        uint32_t d1 = vn, dots = 0;
        for (uint32_t i = c2 + 1; i < vn; i++) if (vp[i] == '.') { if (!dots) d1 = i; dots++; }
        if (parse_int64(f, c2 + 1, d1, false, &sec) != 0) return 1;
        if (dots == 1) {
          if (parse_int64(f, d1 + 1, vn, false, &us) != 0) return 1;
          for (uint32_t q = vn - d1 - 1; q < 6; q++) us = (int64_t)((uint64_t)us * 10ull);
        }
        uint64_t mu = (uint64_t)h * 3600000000ull + (uint64_t)m * 60000000ull + (uint64_t)sec * 1000000ull + (uint64_t)us;
        if (neg) mu = 0ull - mu;
        micro = (int64_t)mu;
      }
      const int64_t years = months / 12, mrem = months % 12;
      int64_t hours = 0, minutes = 0, seconds = 0; uint64_t usabs = 0;
      if (micro != 0) {
        int64_t rem = micro;
        hours = rem / 3600000000ll; rem %= 3600000000ll;
        minutes = rem / 60000000ll; rem %= 60000000ll;
        seconds = rem / 1000000ll; rem %= 1000000ll;
        usabs = (uint64_t)(rem < 0 ? -rem : rem);  // the array holds |microseconds|: the sign is gone when it is parsed back
      }
      const uint64_t total = (uint64_t)years * 31557600ull + (uint64_t)mrem * 2629800ull + (uint64_t)(int64_t)days * 86400ull + (uint64_t)hours * 3600ull + (uint64_t)minutes * 60ull + (uint64_t)seconds;
      emit_u64(s, total * 1000000ull + usabs); return 0;
    }
    case DK_MARSHAL: {  // v.AddVal(colName, colVal): json.Marshal of the Go value as it is (ydb/emitter.go:130-160)
      if (c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) { const double d = cell_f64(c, b); if (d != d || d - d != 0) return 1; }  // json: unsupported value
      emit_json_cell(s, c, b, 0, false); return 0;
    }
    case DK_YDB_UINT64: emit_i64(s, (int64_t)b.v); return 0;  // int64(t)
    case DK_YDB_DATE: emit_i64(s, (int64_t)(int32_t)(uint32_t)(uint64_t)trunc_div((int64_t)b.v, 86400)); return 0;  // DateToInt32
    case DK_INT_REPRS:  // the MySQL integer families: the Go types the reference's switch names, as they are (a uint64 through int64(t))
      if (c.repr == TFGPU_R_UINT64) { emit_i64(s, (int64_t)b.v); return 0; }
      emit_int(s, c, b); return 0;
    case DK_MY_TINYINT1: put_lit(s, (int8_t)b.v == 1 ? "true" : "false"); return 0;
    case DK_MY_FLOAT: {
      double d;
      if (c.repr == TFGPU_R_JSONNUM) { PtrView f{vp}; const int rc = parse_float_go(f, 0, vn, p.p10, p.p128, &d); if (rc == 3) return 2; if (rc) return 1; }
      else d = cell_f64(c, b);  // a float32 widens: float64(t)
      if (d != d || d - d != 0) return 1;
      dev::fmt_json_float(s, d, 64); return 0;
    }
    case DK_MY_BINARY: {  // MysqlFitBinaryLength (zero padding up to a ONE-digit length) + ParseBytea (helpers.go:1062-1080, 509-528)
      uint32_t total = vn;
      if (ec.arg) { const uint32_t want = ec.arg - 1; if (want < vn) return 2; total = want; }  // make() with a negative length panics
      s.put('"'); emit_base64_gen(s, total, [&](uint32_t i) { return i < vn ? (uint32_t)vp[i] : 0u; }); s.put('"'); return 0;
    }
    case DK_MY_BIT1:
      if (c.repr == TFGPU_R_STRING) { const bool t = (vn == 4 && vp[0] == 'A' && vp[1] == 'Q' && vp[2] == '=' && vp[3] == '=') || (vn == 12 && vp[10] == 'E' && vp[11] == '=' && vp[0] == 'A' && vp[1] == 'A' && vp[2] == 'A' && vp[3] == 'A' && vp[4] == 'A' && vp[5] == 'A' && vp[6] == 'A' && vp[7] == 'A' && vp[8] == 'A' && vp[9] == 'A'); put_lit(s, t ? "true" : "false"); return 0; }
      if (vn == 8) { put_lit(s, vp[7] == 1 ? "true" : "false"); return 0; }
      if (vn != 1) return 1;  // "type mysql:bit has len(t) != 1"
      put_lit(s, vp[0] == 1 ? "true" : "false"); return 0;
    case DK_MY_BITS: {  // ShrinkMysqlBit + ParseMysqlBit: the last ceil(size / 8) bytes, reversed (helpers.go:1040-1060, 900-918)
      const uint32_t div = (ec.arg + 7) / 8;
      if (div > vn) return 2;
      s.put('"'); emit_base64_gen(s, div, [&](uint32_t i) { return (uint32_t)vp[vn - 1 - i]; }); s.put('"'); return 0;
    }
    case DK_MY_TIMESTAMP: {  // FormatTime(t.UTC(), precision) (helpers.go:795-805)
      s.put('"'); emit_date(s, (int64_t)b.v); s.put('T'); emit_clock(s, (int64_t)b.v);
      uint32_t f = (uint32_t)(b.ns / 1000); uint8_t d[6]; for (int k = 5; k >= 0; k--) { d[k] = (uint8_t)('0' + f % 10); f /= 10; }
      int last = (int)ec.arg - 1; while (last >= 0 && d[last] == '0') last--;
      if (last >= 0) { s.put('.'); for (int k = 0; k <= last; k++) s.put(d[k]); }
      s.put('Z'); s.put('"'); return 0;
    }
    case DK_MY_DATETIME: emit_u64(s, (uint64_t)(int64_t)b.v * (uint64_t)ec.arg + (uint64_t)((uint32_t)b.ns / (1000000000u / ec.arg))); return 0;
    case DK_MY_TIME: {  // ParseTimeWithoutTZ: the layout is chosen by the text's length (helpers.go:729-757)
      int64_t us; uint32_t end; uint32_t h, m, sec;
      if (vn != 8 && (vn < 10 || vn > 15)) return 1;
      if (!clock_micros(vp, vn, &us, &end, true) || end != vn) return 1;
      dg2(vp, h); dg2(vp + 3, m); dg2(vp + 6, sec);
      if (h > 23 || m > 59 || sec > 59) return 1;  // time.Parse: out of range
      emit_u64(s, (uint64_t)us); return 0;
    }
    case DK_MY_DECIMAL: {  // DecimalToDebeziumPrimitives of the json.Number's text — of "" for any other Go type (mysql/emitter.go:362-371)
      const bool num = c.repr == TFGPU_R_JSONNUM;
      if (ec.arg) { emit_json_string(s, vp, num ? vn : 0u, false); return 0; }  // decimal.handling.mode = string
      return emit_numeric(s, vp, num ? vn : 0u, 0, false, true);
    }
    case DK_MY_YEAR: {  // strconv.Atoi
      int64_t y; PtrView f{vp};
      if (parse_int64(f, 0, vn, false, &y) != 0) return 1;
      emit_i64(s, y); return 0;
    }
    case DK_ARR_INT: case DK_ARR_STRING: case DK_ARR_COPY: {  // add (emitter_value_converter.go:139-168): every element of a []interface{} through AddPg(intoArr)
      if (vn == 4 && vp[0] == 'n' && vp[1] == 'u' && vp[2] == 'l' && vp[3] == 'l') { put_lit(s, "null"); return 0; }
      if (vn < 2 || vp[0] != '[' || vp[vn - 1] != ']') return 2;  // not a slice: the reference emits an empty array
      if (dk == DK_ARR_COPY) { put_bytes(s, vp, vn); return 0; }   // pg:boolean passes every element through as it is
      s.put('[');
      uint32_t i = 1; bool first = true;
      PtrView f{vp};
      while (i < vn - 1) {
        if (!first) { if (vp[i] != ',') return 2; i++; s.put(','); }
        first = false;
        const uint32_t c0 = vp[i];
        if (c0 == 'n') { if (i + 4 > vn - 1 || vp[i + 1] != 'u' || vp[i + 2] != 'l' || vp[i + 3] != 'l') return 2; put_lit(s, "null"); i += 4; continue; }
        if (dk == DK_ARR_INT) {
          if (c0 == '"' || c0 == '[' || c0 == '{' || c0 == 't' || c0 == 'f') return 1;  // "unknown type of value for pg:integer"
          uint32_t e = i; while (e < vn - 1 && vp[e] != ',') e++;
          int64_t v;
          if (parse_int64(f, i, e, false, &v) != 0) return 1;  // "unable to get int64 from json.Number"
          emit_i64(s, v); i = e;
        } else {
          if (c0 != '"') return 2;  // colVal.(string) panics on anything else
          uint32_t e = i + 1;
          while (e < vn - 1 && vp[e] != '"') e += vp[e] == '\\' ? 2 : 1;
          if (e >= vn - 1) return 2;
          put_bytes(s, vp + i, e + 1 - i); i = e + 1;  // the text is json.Marshal's own: marshalling the string again writes the same bytes
        }
      }
      s.put(']'); return 0;
    }
    case DK_TO_STRING:  // typeutil.UnknownTypeToString (helpers.go:1150-1161): a string as it is, anything else as its JSONMarshalUnescape text
      if (c.repr == TFGPU_R_STRING) { emit_json_string(s, vp, vn, false); return 0; }
      if (c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) { const double d = cell_f64(c, b); if (d != d || d - d != 0) return 1; }
      { s.put('"'); JsonEscSink<S> q{s}; emit_json_cell(q, c, b, 0, false); s.put('"'); }
      return 0;
    case DK_HSTORE:  // a map marshals as pg:json does; text goes through HstoreToJSON (providers/postgres/hstore.go:27-43): "" is {}, a text that opens with '{' is taken as JSON already
      if (c.repr == TFGPU_R_JSON) { emit_json_string(s, vp, vn, false); return 0; }
      if (vn == 0) { put_lit(s, "\"{}\""); return 0; }
      if (vp[0] != '{') return 2;  // HstoreToMap: pgtype's hstore reader
      emit_json_string(s, vp, vn, false); return 0;
    case DK_WRONG_TYPE: return 1;
    default: return 2;
  }
}

// which Go types AddPg takes for an ARRAY ELEMENT (a json.Number, a string, a bool, a nested value): 0 ok, 1 the reference's error, 2 host
__device__ __forceinline__ int elem_ok(uint32_t dk, int repr) {
  const bool str = repr == TFGPU_R_STRING, num = repr == TFGPU_R_JSONNUM;
  switch (dk) {
    case DK_SMALLINT: case DK_INTEGER: case DK_BIGINT: case DK_OID: case DK_REAL: case DK_DOUBLE: return num ? 0 : 1;
    case DK_NUMERIC: case DK_NUMERIC_TEXT: return (str || num) ? 0 : (repr == TFGPU_R_JSON ? 2 : 1);  // (a map {Int, Exp}: host)
    case DK_BYTEA: return str ? 0 : 1;
    case DK_DATE: case DK_TSTZ: return str ? 0 : 1;
    case DK_JSON: return 0;
    case DK_BIT1: return 0;
    case DK_INET: case DK_TIME: case DK_TIMETZ: case DK_MONEY: case DK_XML: case DK_POINT: case DK_TSRANGE: case DK_NUMRANGE: case DK_TSTZRANGE: case DK_INTERVAL: case DK_BITS: case DK_HSTORE: return str ? 0 : 2;
    default: return 2;
  }
}
template <class S> __device__ int emit_value(S &s, const EParams &p, const ECell &ec, const DCol &c, uint32_t dk, const CellBits &b) {
  if (dk != DK_ARR_ELEM) return emit_scalar(s, p, ec, c, dk, b);
  // add (emitter_value_converter.go:139-168): every element of the []interface{} through AddPg(intoArr = true); the column holds json.Marshal's text of the slice
  uint32_t vn; const uint8_t *vp = cell_text(c, b, vn);
  if (vn == 4 && vp[0] == 'n' && vp[1] == 'u' && vp[2] == 'l' && vp[3] == 'l') { put_lit(s, "null"); return 0; }
  if (vn < 2 || vp[0] != '[' || vp[vn - 1] != ']') return 2;
  const uint32_t base = (uint32_t)b.v;  // the cell's first byte inside the column's data
  s.put('[');
  uint32_t i = 1; bool first = true;
  while (i < vn - 1) {
    if (!first) { if (vp[i] != ',') return 2; i++; s.put(','); }
    first = false;
    const uint32_t c0 = vp[i];
    uint32_t e = i; int repr;
    if (c0 == '"') { e = i + 1; bool esc = false; while (e < vn - 1 && vp[e] != '"') { if (vp[e] == '\\') { esc = true; e++; } e++; } if (e >= vn - 1) return 2; e++; repr = TFGPU_R_STRING; if (esc && ec.edk != DK_JSON) return 2; }
    else if (c0 == '[' || c0 == '{') {  // a nested value: skipped as a whole (strings inside may hold brackets)
      int depth = 0; bool in = false;
      for (; e < vn - 1; e++) { const uint32_t ch = vp[e]; if (in) { if (ch == '\\') e++; else if (ch == '"') in = false; continue; } if (ch == '"') in = true; else if (ch == '[' || ch == '{') depth++; else if (ch == ']' || ch == '}') { if (--depth == 0) { e++; break; } } }
      repr = TFGPU_R_JSON;
    } else { while (e < vn - 1 && vp[e] != ',') e++; repr = (c0 == 't' || c0 == 'f') ? TFGPU_R_BOOL : c0 == 'n' ? 0 : TFGPU_R_JSONNUM; }
    if (repr == 0) { if (e - i != 4) return 2; put_lit(s, "null"); i = e; continue; }
    DCol ce = c; ce.repr = repr; ce.validity = nullptr;
    CellBits be; be.ns = 0; be.valid = true;
    int rc;
    if (ec.edk == DK_JSON) { be.v = (uint64_t)(base + i) | ((uint64_t)(base + e) << 32); ce.repr = TFGPU_R_JSON; rc = emit_scalar(s, p, ec, ce, DK_JSON, be); }  // JSONMarshalUnescape of the element, as a string
    else {
      rc = elem_ok(ec.edk, repr);
      if (rc) return rc;
      if (repr == TFGPU_R_STRING) be.v = (uint64_t)(base + i + 1) | ((uint64_t)(base + e - 1) << 32);  // between the quotes: no escapes in there (checked)
      else if (repr == TFGPU_R_BOOL) { be.v = c0 == 't'; ce.offsets = nullptr; }
      else be.v = (uint64_t)(base + i) | ((uint64_t)(base + e) << 32);
      if (ec.edk == DK_BIT1 && repr != TFGPU_R_STRING) { put_lit(s, "false"); rc = 0; }  // colVal == "1" is false for any other Go type
      else rc = emit_scalar(s, p, ec, ce, ec.edk, be);
    }
    if (rc) return rc;
    i = e;
  }
  s.put(']');
  return 0;
}

template <class S> __device__ __forceinline__ void emit_tail(S &s, const EParams &p, int64_t e, int64_t r) {
  const int64_t k = p.src_row ? p.src_row[r] : r;
  const uint64_t commit = p.m_commit ? p.m_commit[k] : 0ull;
  put_lit(s, ",\"op\":\""); s.put(event_op(p, e, r));
  for (int q = 0; q < p.ntseg; q++) {
    put_bytes(s, p.blob + p.tseg_off[q], p.tseg_len[q]);
    switch (p.tnum[q]) {
      case TN_LSN: emit_u64(s, p.m_lsn ? p.m_lsn[k] : 0ull); break;
      case TN_TS: emit_u64(s, commit / 1000000ull); break;  // CommitTime / 1000000; GetPayloadTSMS().UnixNano() / 1000000 is the same number
      case TN_ID: emit_u64(s, p.m_id ? p.m_id[k] : 0u); break;
      case TN_STEP: emit_u64(s, commit); break;
      case TN_FILE6: { const uint64_t f = (p.m_lsn ? p.m_lsn[k] : 0ull) / 1000000000000ull; if (f < 1000000ull) emit_dec_pad(s, (uint32_t)f, 6); else emit_u64(s, f); break; }  // LSNToFileAndPos: "%06d"
      case TN_POS: emit_u64(s, (p.m_lsn ? p.m_lsn[k] : 0ull) % 1000000000000ull); break;
      case TN_TXID:  // *string: nil for an empty TxID (emitter_value_converter.go:370-376)
        if (!p.m_tx_off || p.m_tx_off[k + 1] == p.m_tx_off[k]) put_lit(s, "null");
        else emit_json_string(s, p.m_tx + p.m_tx_off[k], p.m_tx_off[k + 1] - p.m_tx_off[k], false);
        break;
      default: break;
    }
  }
}

// one cell of one event; returns the error class (0 = none)
template <class S> __device__ __forceinline__ int emit_cell(S &s, const EParams &p, const ECell &ec, int64_t e, int64_t r) {
  if (ec.kind == EC_CONST) { put_bytes(s, p.blob + ec.pre_off, ec.pre_len); return 0; }
  if (ec.kind == EC_TAIL) { emit_tail(s, p, e, r); return 0; }
  put_bytes(s, p.blob + ec.pre_off, ec.pre_len);
  if (ec.absent && ((ec.absent[r >> 3] >> (r & 7)) & 1u)) { put_bytes(s, p.blob + ec.ph_off, ec.ph_len); return 0; }  // TOASTed: not among the row's ColumnNames
  if (ec.from_old && !row_has_old(p, r)) {
    if (!ec.has_alt) { put_lit(s, "null"); return 0; }
    const CellBits b = load_cell(ec.alt, r);
    if (!b.valid) { put_lit(s, "null"); return 0; }
    return emit_value(s, p, ec, ec.alt, ec.dk_alt, b);
  }
  const CellBits b = load_cell(ec.c, r);
  if (!b.valid) { put_lit(s, "null"); return 0; }
  return emit_value(s, p, ec, ec.c, ec.dk, b);
}

__global__ void __launch_bounds__(256) dbz_cell_len(EParams p) {
  const int32_t ci = (int32_t)blockIdx.y; const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.nev) return;
  const ECell &ec = p.cells[ci];
  const int64_t r = p.ev_row ? p.ev_row[e] : e;
  uint32_t n = 0;
  if (applies(p, ec.apply, e, r)) {
    if (ec.kind == EC_CONST) n = ec.pre_len;
    else {
      CountSink s;
      const int rc = emit_cell(s, p, ec, e, r);
      if (rc) atomicMin(&p.err[rc - 1], ((unsigned long long)e << 16) | (unsigned long long)ci);
      n = s.n;
    }
  }
  p.cell[(int64_t)ci * p.nev + e] = n;
}
// A stream's running total takes ONE atomic per workgroup: an atomic per wave was ~6 000 of them on one word per 4 x 10^5 events, and the L2 takes them one
// after another at ~9 ns apiece — half of dbz_walk_len's duration.  Every thread of the workgroup calls it (256 threads).
__device__ __forceinline__ void add_total64(unsigned long long *total, unsigned long long sum) {
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t lo = __shfl_down((uint32_t)sum, d, 64), hi = __shfl_down((uint32_t)(sum >> 32), d, 64);
    if ((int)(threadIdx.x & 63) + d < 64) sum += ((unsigned long long)hi << 32) | lo;
  }
  __shared__ unsigned long long wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) { const unsigned long long t = wsum[0] + wsum[1] + wsum[2] + wsum[3]; if (t) atomicAdd(total, t); }
}
__global__ void __launch_bounds__(256) dbz_event_layout(EParams p) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t off = 0;
  if (e < p.nev) {  // (no early return: every wave meets the one barrier inside add_total64 exactly once)
    constexpr int U = 8;
    int32_t c = 0;
    for (; c + U <= p.ncells; c += U) {
      uint32_t n[U];
#pragma unroll
      for (int q = 0; q < U; q++) n[q] = p.cell[(int64_t)(c + q) * p.nev + e];
#pragma unroll
      for (int q = 0; q < U; q++) { p.cell[(int64_t)(c + q) * p.nev + e] = off; off += n[q]; }
    }
    for (; c < p.ncells; c++) { const uint32_t n = p.cell[(int64_t)c * p.nev + e]; p.cell[(int64_t)c * p.nev + e] = off; off += n; }
    p.ev_len[e] = off;
  }
  add_total64(p.total64, (unsigned long long)off);
}
__global__ void __launch_bounds__(256) dbz_cell_write(EParams p) {
  const int32_t ci = (int32_t)blockIdx.y; const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.nev) return;
  const ECell &ec = p.cells[ci];
  if (ec.kind == EC_CONST && ec.pre_len > CONST_INLINE) return;  // dbz_fill_const
  const int64_t r = p.ev_row ? p.ev_row[e] : e;
  if (!applies(p, ec.apply, e, r)) return;
  WriteSink w{p.out + (uint64_t)p.ev_len[e] + p.cell[(int64_t)ci * p.nev + e]};
  emit_cell(w, p, ec, e, r);
  w.flush();
}
// The same two passes as ONE LANE PER EVENT walking its cells with one sink (the default; TFGPU_DBZ_WALK=0 keeps the cell-major kernels above for comparison): a message's
// payload half is a few hundred bytes spread over ~20 cells, and a lane per (cell, event) writes each cell's few bytes at a time no other byte of the same cache lines is
// written — the line is fetched, merged and written back once per cell.  A lane that walks its event writes the payload front to back; the cells are the same for every lane
// (descriptors are scalar loads, the converter switch a scalar branch), only `applies` diverges.  The per-cell length array and the layout kernel are not needed: the length
// pass keeps one running count, the write pass one running pointer, and the place of a long constant is noted for dbz_fill_const as the walk passes it.
__global__ void __launch_bounds__(256) dbz_walk_len(EParams p) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  CountSink s;
  if (e < p.nev) {  // (no early return: every wave meets the one barrier inside add_total64 exactly once)
    const int64_t r = p.ev_row ? p.ev_row[e] : e;
    for (int32_t ci = 0; ci < p.ncells; ci++) {
      const ECell &ec = p.cells[ci];
      if (!applies(p, ec.apply, e, r)) continue;
      if (ec.kind == EC_CONST) { s.n += ec.pre_len; continue; }
      const int rc = emit_cell(s, p, ec, e, r);
      if (rc) atomicMin(&p.err[rc - 1], ((unsigned long long)e << 16) | (unsigned long long)ci);
    }
    p.ev_len[e] = s.n;
  }
  add_total64(p.total64, (unsigned long long)s.n);
}
__global__ void __launch_bounds__(256) dbz_walk_write(EParams p, uint32_t *const_at /* [long constants][nev] */) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.nev) return;
  const int64_t r = p.ev_row ? p.ev_row[e] : e;
  uint8_t *const base = p.out + (uint64_t)p.ev_len[e];
  WriteSink w{base};
  int32_t slot = 0;
  for (int32_t ci = 0; ci < p.ncells; ci++) {
    const ECell &ec = p.cells[ci];
    const bool is_long = ec.kind == EC_CONST && ec.pre_len > CONST_INLINE;
    if (applies(p, ec.apply, e, r)) {
      if (is_long) { w.flush(); const_at[(int64_t)slot * p.nev + e] = (uint32_t)(w.p - base); w.p += ec.pre_len; }  // dbz_fill_const writes it
      else emit_cell(w, p, ec, e, r);
    }
    if (is_long) slot++;
  }
  w.flush();
}

// a long constant (the schema half of a message) into every event that holds it: one WAVE per (event, 2 KiB piece) — a key's schema is a
// few hundred bytes, a value's a few KiB, so a workgroup per event would leave most of its lanes without a byte to move
constexpr uint32_t FILL_PIECE = 2048;
__global__ void __launch_bounds__(256) dbz_fill_const(EParams p, int32_t ci, uint32_t pieces, const uint32_t *at /* [nev]: where the constant sits inside its event */) {
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  const int64_t e = item / pieces; const uint32_t piece = (uint32_t)(item % pieces);
  if (e >= p.nev) return;
  const ECell &ec = p.cells[ci];
  // a wave's stores wait for what says where they go: with 10^5..10^6 waves of one or two stores each, the chain event → row → kind → … is the
  // kernel's duration.  The schema halves belong to every event (keys) or to every event but the tombstones (values): one load, beside the offsets'
  if (ec.apply == EA_VALUE) { if (p.ev_type && p.ev_type[e] == EV_TOMBSTONE) return; }
  else if (ec.apply != EA_KEY) { const int64_t r = p.ev_row ? p.ev_row[e] : e; if (!applies(p, ec.apply, e, r)) return; }
  uint8_t *dst = p.out + (uint64_t)p.ev_len[e] + at[e];
  const uint8_t *src = p.blob + ec.pre_off;
  const uint32_t a = piece * FILL_PIECE, b = min(a + FILL_PIECE, ec.pre_len);
  // the destination's alignment decides the split: bytes up to its next 16-byte boundary, 16-byte stores (one per lane and KiB), a byte tail;
  // the source sits wherever the blob put it: unaligned 16-byte loads of a few cache-resident KiB
  const uint32_t head = min((uint32_t)((16 - ((uintptr_t)(dst + a) & 15)) & 15), b - a);
  if (lane < head) dst[a + lane] = src[a + lane];
  const uint32_t a16 = a + head, quads = (b - a16) / 16;
  struct __attribute__((packed, aligned(1))) U128 { uint32_t x, y, z, w; };
  for (uint32_t q = lane; q < quads; q += 64) { const U128 v = reinterpret_cast<const U128 *>(src + a16)[q]; reinterpret_cast<uint4 *>(dst + a16)[q] = make_uint4(v.x, v.y, v.z, v.w); }
  const uint32_t t0 = a16 + quads * 16;
  if (lane < b - t0) dst[t0 + lane] = src[t0 + lane];
}

// the event list (emitKV, emitter_value_converter.go:629-672): per row its number of messages, then — after a scan — (row, emitType) of each
__device__ __forceinline__ uint32_t row_events(uint32_t kind, bool changed, int skip_tomb) {
  if (kind == TFGPU_K_INSERT) return 1;
  if (kind == TFGPU_K_UPDATE) return changed ? (skip_tomb ? 2u : 3u) : 1u;
  if (kind == TFGPU_K_DELETE) return skip_tomb ? 1u : 2u;
  return 0;  // other kinds emit nothing
}
__global__ void __launch_bounds__(256) dbz_event_count(const uint8_t *kind, const uint8_t *changed, int64_t n, int skip_tomb, uint32_t *cnt) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) cnt[r] = row_events(kind[r], changed && changed[r], skip_tomb);
}
__global__ void __launch_bounds__(256) dbz_event_fill(const uint8_t *kind, const uint8_t *changed, int64_t n, int skip_tomb, const uint32_t *at, int32_t *ev_row, uint8_t *ev_type) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint32_t k = kind[r]; const bool ch = changed && changed[r];
  uint32_t e = at[r];
  auto ev = [&](uint32_t t) { ev_row[e] = (int32_t)r; ev_type[e] = (uint8_t)t; e++; };
  if (k == TFGPU_K_INSERT) ev(EV_REGULAR);
  else if (k == TFGPU_K_UPDATE) { if (ch) { ev(EV_DELETE); if (!skip_tomb) ev(EV_TOMBSTONE); ev(EV_INSERT); } else ev(EV_REGULAR); }
  else if (k == TFGPU_K_DELETE) { ev(EV_DELETE); if (!skip_tomb) ev(EV_TOMBSTONE); }
}

static inline unsigned blocks(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + 255) / 256); }

// ---- host: Go's JSON text -------------------------------------------------------------------------------------------------
static void jstr(std::string &out, const std::string &s) {  // encoding/json appendString, escapeHTML = false
  static const char *HEX = "0123456789abcdef";
  out.push_back('"');
  const size_t n = s.size();
  for (size_t i = 0; i < n;) {
    const unsigned char c = (unsigned char)s[i];
    if (c < 0x80) {
      if (c >= 0x20 && c != '"' && c != '\\') out.push_back((char)c);
      else switch (c) {
        case '"': out += "\\\""; break; case '\\': out += "\\\\"; break; case '\b': out += "\\b"; break; case '\f': out += "\\f"; break;
        case '\n': out += "\\n"; break; case '\r': out += "\\r"; break; case '\t': out += "\\t"; break;
        default: out += "\\u00"; out.push_back(HEX[c >> 4]); out.push_back(HEX[c & 15]);
      }
      i++; continue;
    }
    size_t need = 0; unsigned char lo = 0x80, hi = 0xBF;
    if (c >= 0xC2 && c <= 0xDF) need = 1;
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
    bool ok = need > 0 && i + need < n;
    for (size_t k = 1; ok && k <= need; k++) { const unsigned char d = (unsigned char)s[i + k]; if (d < (k == 1 ? lo : 0x80) || d > (k == 1 ? hi : 0xBF)) ok = false; }
    if (!ok) { out += "\\ufffd"; i++; continue; }
    if (need == 2 && c == 0xE2 && (unsigned char)s[i + 1] == 0x80 && ((unsigned char)s[i + 2] == 0xA8 || (unsigned char)s[i + 2] == 0xA9)) out += (unsigned char)s[i + 2] == 0xA8 ? "\\u2028" : "\\u2029";
    else out.append(s, i, need + 1);
    i += need + 1;
  }
  out.push_back('"');
}
static std::string jstr(const std::string &s) { std::string o; jstr(o, s); return o; }

static bool starts(const std::string &s, const char *p) { return s.compare(0, std::strlen(p), p) == 0; }
// "pg:<what>[(d)] <with|without> time zone" (providers/postgres/type.go:140-237): d = -1 without a parameter, -2 = no match
static int time_family(const std::string &t, const char *what, const char *tz) {
  const std::string head = std::string("pg:") + what, tail = std::string(" ") + tz + " time zone";
  if (!starts(t, head.c_str())) return -2;
  size_t i = head.size();
  int d = -1;
  if (i + 2 < t.size() && t[i] == '(' && t[i + 1] >= '0' && t[i + 1] <= '6' && t[i + 2] == ')') { d = t[i + 1] - '0'; i += 3; }
  return t.compare(i, std::string::npos, tail) == 0 ? d : -2;
}
// typeutil.IsPgNumeric / DecimalGetPrecisionAndScale (helpers.go:1142-1148, 174-195): 0 no, 1 "pg:numeric", 2 "pg:numeric(p,s)"
static int numeric_family(const std::string &t, int *precision, int *scale) {
  if (t == "pg:numeric") return 1;
  if (!starts(t, "pg:numeric(")) return 0;
  size_t i = 11; int p = 0, s = 0; size_t a = i;
  while (i < t.size() && t[i] >= '0' && t[i] <= '9' && p < 100000) p = p * 10 + (t[i++] - '0');
  if (i == a || i >= t.size() || t[i] != ',') return 0;
  a = ++i;
  while (i < t.size() && t[i] >= '0' && t[i] <= '9' && s < 100000) s = s * 10 + (t[i++] - '0');
  if (i == a || i >= t.size() || t[i] != ')') return 0;
  *precision = p; *scale = s;
  return 2;
}

struct Params {
  std::map<std::string, std::string> m;
  const std::string &get(const char *k) const { static const std::string empty; auto it = m.find(k); return it == m.end() ? empty : it->second; }
};
struct ColPlan { uint32_t dk = 0, arg = 0, edk = 0; std::string descr; };  // the converter and the column's field description

[[noreturn]] static void unsupported(const std::string &col, const std::string &what) {
  throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_debezium_emit: column " + col + ": " + what + " stays with the stock emitter");
}
// getFieldDescr (fields_descr.go:19-69) + GetKafkaTypeDescrByPgType (pg/emitter.go:222-260), and the device converter of the type
static std::vector<std::string> enum_values(const char *props, bool *ok);
static ColPlan plan_column(const tfgpu_colschema &cs, const Params &P, bool snapshot, bool into_arr = false, const char *type_override = nullptr) {
  const std::string name = cs.name ? cs.name : "", t = type_override ? type_override : (cs.original_type ? cs.original_type : "");
  ColPlan cp;
  if (!into_arr && starts(t, "pg:") && t.size() > 5 && t.compare(t.size() - 2, 2, "[]") == 0) {  // AddFieldDescr (fields_descr.go:71-96): the element's description under "items"
    const std::string et = t.substr(0, t.size() - 2);
    ColPlan el = plan_column(cs, P, snapshot, true, et.c_str());
    if (el.dk == DK_SMALLINT || el.dk == DK_INTEGER || el.dk == DK_BIGINT) cp.dk = DK_ARR_INT;
    else if (el.dk == DK_STRING) cp.dk = DK_ARR_STRING;
    else if (el.dk == DK_BOOL) cp.dk = DK_ARR_COPY;
    else if (el.dk == DK_TS || el.dk == DK_WRONG_TYPE || el.dk >= DK_MARSHAL) unsupported(name, "an array of " + et);  // timestamp texts are pgtype.Timestamp.Set's
    else { cp.dk = DK_ARR_ELEM; cp.edk = el.dk; cp.arg = el.arg; }
    std::string o = "{";
    if (P.get("dt.add.original.type.info") == "true") o += "\"__dt_original_type_info\":{\"original_type\":" + jstr(t) + "},";
    o += "\"field\":" + jstr(name) + ",\"items\":" + el.descr + ",\"optional\":" + ((cs.flags & TFGPU_COL_KEY) ? "false" : "true") + ",\"type\":\"array\"}";
    cp.descr = o;
    return cp;
  }
  if (t.empty()) throw Error(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: column " + name + ": unknown source type (no OriginalType; emitter_value_converter.go:188-196)");
  std::string kafka, dname, extra;  // extra: further members, already in key order relative to each other ("doc" / "fields" / "parameters")
  bool found = false, unknown_to_string = false;
  if (starts(t, "ydb:")) {  // GetKafkaTypeDescrByYDBType / AddYDB (ydb/emitter.go:15-232)
    static const struct { const char *t; const char *kafka; const char *name; uint32_t dk, arg; } YDB[] = {
      {"ydb:Bool", "boolean", "", DK_MARSHAL, 0}, {"ydb:Int8", "int8", "", DK_MARSHAL, 0}, {"ydb:Int16", "int16", "", DK_MARSHAL, 0}, {"ydb:Int32", "int32", "", DK_MARSHAL, 0},
      {"ydb:Int64", "int64", "", DK_MARSHAL, 0}, {"ydb:Uint8", "int8", "", DK_MARSHAL, 0}, {"ydb:Uint16", "int16", "", DK_MARSHAL, 0}, {"ydb:Uint32", "int32", "", DK_MARSHAL, 0},
      {"ydb:Uint64", "int64", "", DK_YDB_UINT64, 0}, {"ydb:Float", "float", "", DK_MARSHAL, 0}, {"ydb:Double", "double", "", DK_MARSHAL, 0}, {"ydb:String", "bytes", "", DK_MARSHAL, 0},
      {"ydb:Utf8", "string", "", DK_MARSHAL, 0}, {"ydb:Json", "string", "io.debezium.data.Json", DK_JSON, 0}, {"ydb:JsonDocument", "string", "io.debezium.data.Json", DK_JSON, 0},
      {"ydb:Uuid", "string", "", DK_STRING, 0}, {"ydb:Date", "int32", "io.debezium.time.Date", DK_YDB_DATE, 0}, {"ydb:Datetime", "int64", "io.debezium.time.Timestamp", DK_TS, 1000},
      {"ydb:Timestamp", "int64", "io.debezium.time.MicroTimestamp", DK_TS, 1}, {"ydb:Interval", "int64", "io.debezium.time.MicroDuration", DK_MARSHAL, 0}};
    for (auto &e : YDB) if (t == e.t) { kafka = e.kafka; dname = e.name; cp.dk = e.dk; cp.arg = e.arg; found = true; break; }
    if (!found && t == "ydb:Decimal") {
      const std::string &mode = P.get("decimal.handling.mode");
      found = true;
      if (mode == "precise") { kafka = "bytes"; dname = "org.apache.kafka.connect.data.Decimal"; cp.dk = DK_NUMERIC; cp.arg = 9u | (1u << 17);  // numeric(22,9); colVal.(string)
        extra = ",\"parameters\":{\"connect.decimal.precision\":\"22\",\"scale\":\"9\"}"; }
      else if (mode == "string") { kafka = "string"; cp.dk = DK_NUMERIC_TEXT; cp.arg = 1u << 17; }
      else unsupported(name, "decimal.handling.mode=" + mode);
    }
    if (!found && t == "ydb:DyNumber") { found = true; kafka = "struct"; dname = "io.debezium.data.VariableScaleDecimal"; cp.dk = DK_NUMERIC; cp.arg = 1u << 16; }
    if (!found) {
      const std::string &policy = P.get("dt.unknown.types.policy");
      if (policy == "fail") throw Error(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: unable to add field description: unknown ydbType: " + t + " (column " + name + ")");
      if (policy != "to_string" || (cs.flags & TFGPU_COL_KEY) || into_arr) unsupported(name, "dt.unknown.types.policy=" + policy + " for " + t);  // (skip: a member that exists only when its value is nil)
      found = true; unknown_to_string = true; kafka = "string"; cp.dk = DK_TO_STRING;
    }
  }
  else if (starts(t, "mysql:")) {  // GetKafkaTypeDescrByMysqlType / AddMysql (mysql/emitter.go:20-388)
    std::string u = t; for (size_t i = 0; i < t.size(); i++) if (!((t[i] >= 'a' && t[i] <= 'z') || t[i] == ':')) { u = t.substr(0, i); break; }  // abstract.TrimMySQLType
    const bool uns = t.size() > 9 && t.compare(t.size() - 9, 9, " unsigned") == 0;
    auto paren_digit = [&](const char *what) -> int { const std::string h = std::string("mysql:") + what + "("; if (!starts(t, h.c_str()) || t.size() < h.size() + 2 || t[h.size()] < '0' || t[h.size()] > '9' || t[h.size() + 1] != ')') return -1; return t[h.size()] - '0'; };
    auto bit = [](int r) { return 1u << r; };
    found = true;
    if (u == "mysql:int" || u == "mysql:mediumint") { kafka = (u == "mysql:int" && uns && (snapshot || !(cs.flags & TFGPU_COL_KEY))) ? "int64" : "int32"; cp.dk = DK_INT_REPRS; cp.arg = bit(TFGPU_R_INT32) | bit(TFGPU_R_UINT32); }
    else if (u == "mysql:bigint") { kafka = "int64"; cp.dk = DK_INT_REPRS; cp.arg = bit(TFGPU_R_INT64) | bit(TFGPU_R_UINT64); }
    else if (u == "mysql:smallint") { kafka = uns ? "int32" : "int16"; cp.dk = DK_INT_REPRS; cp.arg = bit(TFGPU_R_INT16) | bit(TFGPU_R_UINT16); }
    else if (u == "mysql:tinyint") { if (t == "mysql:tinyint(1)") { kafka = "boolean"; cp.dk = DK_MY_TINYINT1; } else { kafka = "int16"; cp.dk = DK_INT_REPRS; cp.arg = bit(TFGPU_R_INT8) | bit(TFGPU_R_UINT8); } }
    else if (u == "mysql:float" || u == "mysql:double") { kafka = "double"; cp.dk = DK_MY_FLOAT; cp.arg = u == "mysql:float" ? 1u : 0u; }
    else if (t == "mysql:blob" || t == "mysql:longblob" || t == "mysql:mediumblob" || t == "mysql:tinyblob" || starts(t, "mysql:binary(") || starts(t, "mysql:varbinary(")) {
      kafka = "bytes"; cp.dk = DK_MY_BINARY;
      if (!starts(t, "mysql:varbinary")) { const size_t q = t.find('('); if (q != std::string::npos && q + 2 < t.size() && t[q + 1] >= '0' && t[q + 1] <= '9' && t[q + 2] == ')') cp.arg = (uint32_t)(t[q + 1] - '0') + 1u; }  // `^.*\((\d)\).*`
    }
    else if (starts(t, "mysql:bit(")) {
      const size_t z = t.find(')', 10);
      if (z == std::string::npos) throw Error(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: column " + name + ": unsupported pg type, can't find closing bracket: " + t);
      if (t == "mysql:bit(1)") { kafka = "boolean"; cp.dk = DK_MY_BIT1; }
      else { kafka = "bytes"; dname = "io.debezium.data.Bits"; extra = ",\"parameters\":{\"length\":" + jstr(t.substr(10, z - 10)) + "}"; cp.dk = DK_MY_BITS; cp.arg = (uint32_t)std::min<long>(std::atol(t.substr(10, z - 10).c_str()), 1 << 20); }
    }
    else if (t == "mysql:longtext" || t == "mysql:mediumtext" || t == "mysql:text" || t == "mysql:tinytext" || starts(t, "mysql:char(") || starts(t, "mysql:varchar(")) { kafka = "string"; cp.dk = DK_STRING; }
    else if (starts(t, "mysql:enum(") || starts(t, "mysql:set(")) {
      const bool en = starts(t, "mysql:enum(");
      std::string raw = t.substr(en ? 11 : 10, t.size() - (en ? 11 : 10) - 1), allowed;  // UnwrapMysqlEnumsAndSets (helpers.go:1022-1038)
      for (std::string str = raw;;) { if (str.empty()) break; str = str.substr(1); const size_t q = str.find('\''); if (q == std::string::npos) { allowed.clear(); break; } allowed += str.substr(0, q) + ","; if (q + 2 >= str.size()) break; str = str.substr(q + 2); }
      if (!allowed.empty()) allowed.pop_back();
      kafka = "string"; dname = en ? "io.debezium.data.Enum" : "io.debezium.data.EnumSet"; extra = ",\"parameters\":{\"allowed\":" + jstr(allowed) + "}"; cp.dk = DK_STRING;
    }
    else if (t == "mysql:json") { kafka = "string"; dname = "io.debezium.data.Json"; cp.dk = DK_JSON; }
    else if (t == "mysql:date") { kafka = "int32"; dname = "io.debezium.time.Date"; cp.dk = DK_DATE; }
    else if (starts(t, "mysql:datetime")) {
      const int pd = paren_digit("datetime");
      kafka = "int64"; dname = (pd >= 1 && pd <= 3) || pd == -1 ? "io.debezium.time.Timestamp" : "io.debezium.time.MicroTimestamp";  // GetTimeDivider fails without a "(d)": divider 0
      cp.dk = DK_MY_DATETIME; cp.arg = (t == "mysql:datetime" || pd <= 3) ? 1000u : 1000000u;
    }
    else if (starts(t, "mysql:timestamp")) {
      const int pd = t == "mysql:timestamp" ? 0 : paren_digit("timestamp");
      if (pd < 0) unsupported(name, "original type " + t);  // FormatTime with precision -1 panics
      kafka = "string"; dname = "io.debezium.time.ZonedTimestamp"; cp.dk = DK_MY_TIMESTAMP; cp.arg = (uint32_t)pd;
    }
    else if (t == "mysql:time" || starts(t, "mysql:time(")) { kafka = "int64"; dname = "io.debezium.time.MicroTime"; cp.dk = DK_MY_TIME; }
    else if (starts(t, "mysql:decimal(")) {
      const std::string &mode = P.get("decimal.handling.mode");
      int pr = 0, sc = 0; { size_t i = 14; size_t a = i; while (i < t.size() && t[i] >= '0' && t[i] <= '9' && pr < 100000) pr = pr * 10 + (t[i++] - '0'); bool ok = i > a && i < t.size() && t[i] == ','; a = ++i; while (ok && i < t.size() && t[i] >= '0' && t[i] <= '9' && sc < 100000) sc = sc * 10 + (t[i++] - '0'); if (!(ok && i > a && i < t.size() && t[i] == ')')) { pr = 0; sc = 0; } }
      cp.dk = DK_MY_DECIMAL;
      if (mode == "precise") { kafka = "bytes"; dname = "org.apache.kafka.connect.data.Decimal"; extra = ",\"parameters\":{\"connect.decimal.precision\":\"" + std::to_string(pr) + "\",\"scale\":\"" + std::to_string(sc) + "\"}"; }
      else if (mode == "string") { kafka = "string"; cp.arg = 1; }
      else unsupported(name, "decimal.handling.mode=" + mode);
    }
    else if (starts(t, "mysql:year")) { kafka = "int32"; dname = "io.debezium.time.Year"; cp.dk = DK_MY_YEAR; }
    else throw Error(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: unable to add field description: unknown mysqlType: " + t + " (column " + name + ")");
  }
  else if (!starts(t, "pg:")) unsupported(name, "original type " + t);
  else if (t.size() > 2 && t.compare(t.size() - 2, 2, "[]") == 0) unsupported(name, "array type " + t);  // an array of arrays  // extra: further members, already in key order relative to each other ("doc" / "fields" / "parameters")
  static const struct { const char *t; const char *kafka; const char *name; uint32_t dk; } PLAIN[] = {
    {"pg:boolean", "boolean", "", DK_BOOL}, {"pg:bit(1)", "boolean", "", DK_BIT1}, {"pg:smallint", "int16", "", DK_SMALLINT}, {"pg:integer", "int32", "", DK_INTEGER},
    {"pg:bigint", "int64", "", DK_BIGINT}, {"pg:oid", "int64", "", DK_OID}, {"pg:real", "float", "", DK_REAL}, {"pg:double precision", "double", "", DK_DOUBLE},
    {"pg:bytea", "bytes", "", DK_BYTEA}, {"pg:json", "string", "io.debezium.data.Json", DK_JSON}, {"pg:jsonb", "string", "io.debezium.data.Json", DK_JSON},
    {"pg:uuid", "string", "io.debezium.data.Uuid", DK_STRING}, {"pg:inet", "string", "", DK_INET}, {"pg:int4range", "string", "", DK_STRING},
    {"pg:int8range", "string", "", DK_STRING}, {"pg:daterange", "string", "", DK_STRING}, {"pg:text", "string", "", DK_STRING},
    {"pg:date", "int32", "io.debezium.time.Date", DK_DATE}, {"pg:cidr", "string", "", DK_STRING}, {"pg:macaddr", "string", "", DK_STRING},
    {"pg:character", "string", "", DK_STRING}, {"pg:character varying", "string", "", DK_STRING}, {"pg:USER-DEFINED:citext", "string", "", DK_STRING}};
  if (!found) for (auto &e : PLAIN) if (t == e.t) { kafka = e.kafka; dname = e.name; cp.dk = e.dk; found = true; break; }
  int prec = 0, scale = 0, d;
  if (found) {}
  else if (t == "pg:xml") { kafka = "string"; dname = "io.debezium.data.Xml"; cp.dk = DK_XML; }
  else if (t == "pg:USER-DEFINED:hstore") { kafka = "string"; dname = "io.debezium.data.Json"; cp.dk = DK_HSTORE; }
  else if (t == "pg:numrange") { kafka = "string"; cp.dk = DK_NUMRANGE; }
  else if (t == "pg:tsrange") { kafka = "string"; cp.dk = DK_TSRANGE; }
  else if (t == "pg:tstzrange") { kafka = "string"; cp.dk = DK_TSTZRANGE; }
  else if (t == "pg:point") { kafka = "struct"; dname = "io.debezium.data.geometry.Point"; cp.dk = DK_POINT; }
  else if (t == "pg:money") {
    const std::string &mode = P.get("decimal.handling.mode");
    cp.dk = DK_MONEY;
    if (mode == "precise") { kafka = "bytes"; dname = "org.apache.kafka.connect.data.Decimal"; extra = ",\"parameters\":{\"scale\":\"2\"}"; }
    else if (mode == "string") { kafka = "string"; cp.arg = 1; }
    else unsupported(name, "decimal.handling.mode=" + mode);
  }
  else if (starts(t, "pg:interval")) {
    kafka = "int64"; dname = "io.debezium.time.MicroDuration";
    cp.dk = P.get("interval.handling.mode") == "numeric" ? DK_INTERVAL : DK_WRONG_TYPE;  // "unsupported interval.handling.mode" fails the first non-nil value
  }
  else if ((d = time_family(t, "time", "with")) != -2) { kafka = "string"; dname = "io.debezium.time.ZonedTime"; cp.dk = DK_TIMETZ; cp.arg = into_arr ? 1u : 0u; }
  else if ((d = time_family(t, "time", "without")) != -2) {
    const uint32_t divider = (!into_arr && d >= 1 && d <= 3) ? 1000u : 1u;  // GetTimeDivider (helpers.go:106-123); 1 into an array
    kafka = divider == 1 ? "int64" : "int32"; dname = divider == 1 ? "io.debezium.time.MicroTime" : "io.debezium.time.Time"; cp.dk = DK_TIME; cp.arg = divider | (into_arr ? 1u << 16 : 0u);
  }
  else if (starts(t, "pg:bit(") || starts(t, "pg:bit varying(")) {
    const size_t a = starts(t, "pg:bit(") ? 7 : 15, z = t.find(')', a);
    if (z == std::string::npos) throw Error(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: column " + name + ": unsupported pg type, can't find closing bracket: " + t);
    kafka = "bytes"; dname = "io.debezium.data.Bits"; extra = ",\"parameters\":{\"length\":" + jstr(t.substr(a, z - a)) + "}"; cp.dk = DK_BITS;
  } else if (starts(t, "pg:character(") || starts(t, "pg:character varying(")) { kafka = "string"; cp.dk = DK_STRING; }
  else if ((d = time_family(t, "timestamp", "with")) != -2) { kafka = "string"; dname = "io.debezium.time.ZonedTimestamp"; cp.dk = DK_TSTZ; }
  else if ((d = time_family(t, "timestamp", "without")) != -2) {
    const uint32_t divider = (d >= 1 && d <= 3) ? 1000u : 1u;  // GetTimeDivider (helpers.go:106-123)
    kafka = "int64"; dname = divider == 1 ? "io.debezium.time.MicroTimestamp" : "io.debezium.time.Timestamp"; cp.dk = DK_TS; cp.arg = divider;
  } else if ((d = numeric_family(t, &prec, &scale)) != 0) {
    const std::string &mode = P.get("decimal.handling.mode");
    if (mode == "precise") {
      cp.dk = DK_NUMERIC;
      if (d == 1) {
        kafka = "struct"; dname = "io.debezium.data.VariableScaleDecimal"; cp.arg = 1u << 16;
        extra = ",\"doc\":\"Variable scaled decimal\"";  // "doc" < "field": placed by the assembler below
      } else {
        if (scale > 0xFFFF) unsupported(name, "numeric scale beyond 65535");
        kafka = "bytes"; dname = "org.apache.kafka.connect.data.Decimal"; cp.arg = (uint32_t)scale;
        extra = ",\"parameters\":{\"connect.decimal.precision\":\"" + std::to_string(prec) + "\",\"scale\":\"" + std::to_string(scale) + "\"}";
      }
    } else if (mode == "string") { kafka = "string"; cp.dk = DK_NUMERIC_TEXT; }
    else unsupported(name, "decimal.handling.mode=" + mode);
  } else if (cs.properties_json && std::strstr(cs.properties_json, "\"pg:enum_all_values\"")) {  // pgEnum (pg/emitter.go:165-173): Properties[pg:enum_all_values]
    bool ok = false;
    const std::vector<std::string> vals = enum_values(cs.properties_json, &ok);
    if (!ok) unsupported(name, "enum type " + t + " (its pg:enum_all_values property is not a list of strings)");
    std::string joined; for (size_t i = 0; i < vals.size(); i++) { if (i) joined += ","; joined += vals[i]; }
    kafka = "string"; dname = "io.debezium.data.Enum"; extra = ",\"parameters\":{\"allowed\":" + jstr(joined) + "}"; cp.dk = DK_STRING;
  }
  else {
    const std::string &policy = P.get("dt.unknown.types.policy");
    if (policy == "fail") throw Error(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: unable to add field description: unknown pgType: " + t + " (column " + name + ")");
    if (policy != "to_string" || (cs.flags & TFGPU_COL_KEY) || into_arr) unsupported(name, "dt.unknown.types.policy=" + policy + " for " + t);  // (skip: a member that exists only when its value is nil)
    unknown_to_string = true; kafka = "string"; cp.dk = DK_TO_STRING;   // the column is described as a utf8 column without an original type (emitter_value_converter.go:108-117)
  }
  // the description, members in key order: __dt_original_type_info, doc, field, fields, name, optional, parameters, type, version
  std::string o = "{";
  if (P.get("dt.add.original.type.info") == "true") o += "\"__dt_original_type_info\":{\"original_type\":" + jstr(unknown_to_string ? std::string() : t) + "},";
  const bool var_scale = cp.dk == DK_NUMERIC && ((cp.arg >> 16) & 1u);
  if (var_scale) o += "\"doc\":\"Variable scaled decimal\",";
  if (cp.dk == DK_POINT) o += "\"doc\":\"Geometry (POINT)\",";
  std::string sep = "";
  if (!into_arr) { o += "\"field\":" + jstr(name); sep = ","; }
  if (var_scale) o += sep + "\"fields\":[{\"field\":\"scale\",\"optional\":false,\"type\":\"int32\"},{\"field\":\"value\",\"optional\":false,\"type\":\"bytes\"}]";
  if (var_scale) sep = ",";
  if (cp.dk == DK_POINT) { o += sep + "\"fields\":[{\"field\":\"x\",\"optional\":false,\"type\":\"double\"},{\"field\":\"y\",\"optional\":false,\"type\":\"double\"},{\"field\":\"wkb\",\"optional\":true,\"type\":\"bytes\"},"
                            "{\"field\":\"srid\",\"optional\":true,\"type\":\"int32\"}]"; sep = ","; }
  if (!dname.empty()) { o += sep + "\"name\":" + jstr(dname); sep = ","; }
  o += sep + "\"optional\":" + ((cs.flags & TFGPU_COL_KEY) ? "false" : "true");
  if (!var_scale && !extra.empty()) o += extra;
  o += ",\"type\":" + jstr(kafka);
  if (!dname.empty()) o += ",\"version\":1";
  o += "}";
  cp.descr = o;
  return cp;
}

// the list of strings under "pg:enum_all_values" in json.Marshal(ColSchema.Properties)
static std::vector<std::string> enum_values(const char *props, bool *ok) {
  std::vector<std::string> out; *ok = false;
  const char *p = std::strstr(props, "\"pg:enum_all_values\"");
  if (!p) return out;
  p += std::strlen("\"pg:enum_all_values\"");
  while (*p == ' ') p++;
  if (*p != ':') return out;
  p++; while (*p == ' ') p++;
  if (*p != '[') return out;
  p++;
  for (;;) {
    while (*p == ' ' || *p == ',') p++;
    if (*p == ']') { *ok = true; return out; }
    if (*p != '"') return out;
    p++;
    std::string v;
    while (*p && *p != '"') {
      if (*p != '\\') { v.push_back(*p++); continue; }
      p++;
      switch (*p) {
        case 'n': v.push_back('\n'); break; case 't': v.push_back('\t'); break; case 'r': v.push_back('\r'); break; case 'b': v.push_back('\b'); break; case 'f': v.push_back('\f'); break;
        case 'u': {
          unsigned cp = 0; for (int k = 1; k <= 4; k++) { const char c = p[k]; if (!c) return out; cp = cp * 16 + (unsigned)(c >= '0' && c <= '9' ? c - '0' : (c | 32) - 'a' + 10); }
          p += 4;
          if (cp < 0x80) v.push_back((char)cp); else if (cp < 0x800) { v.push_back((char)(0xC0 | (cp >> 6))); v.push_back((char)(0x80 | (cp & 0x3F))); }
          else { v.push_back((char)(0xE0 | (cp >> 12))); v.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); v.push_back((char)(0x80 | (cp & 0x3F))); }  // (surrogate pairs: not in enum labels we decode)
          break;
        }
        default: if (!*p) return out; v.push_back(*p);
      }
      p++;
    }
    if (*p != '"') return out;
    p++;
    out.push_back(v);
  }
}

// which Go dynamic types AddPg takes for the type: 0 ok, 1 the reference's "unknown type of value", 2 host
static int repr_ok(uint32_t dk, int repr, uint32_t arg = 0) {
  auto in = [&](std::initializer_list<int> l) { for (int x : l) if (x == repr) return true; return false; };
  switch (dk) {
    case DK_BOOL: return in({TFGPU_R_BOOL}) ? 0 : 2;
    case DK_BIT1: case DK_STRING: case DK_INET: case DK_BITS: return in({TFGPU_R_STRING}) ? 0 : 2;
    case DK_SMALLINT: return in({TFGPU_R_INT16, TFGPU_R_INT64, TFGPU_R_JSONNUM}) ? 0 : 1;
    case DK_INTEGER: return in({TFGPU_R_INT32, TFGPU_R_INT64, TFGPU_R_JSONNUM}) ? 0 : 1;
    case DK_BIGINT: return in({TFGPU_R_INT64, TFGPU_R_JSONNUM}) ? 0 : 1;
    case DK_OID: return in({TFGPU_R_INT32, TFGPU_R_INT64, TFGPU_R_UINT32, TFGPU_R_FLOAT64, TFGPU_R_JSONNUM}) ? 0 : 1;
    case DK_REAL: return in({TFGPU_R_FLOAT32, TFGPU_R_FLOAT64, TFGPU_R_JSONNUM}) ? 0 : 1;
    case DK_DOUBLE: return in({TFGPU_R_FLOAT64, TFGPU_R_JSONNUM}) ? 0 : 1;
    case DK_BYTEA: return in({TFGPU_R_STRING, TFGPU_R_BYTES}) ? 0 : 1;
    case DK_DATE: return in({TFGPU_R_TIME, TFGPU_R_STRING}) ? 0 : 1;
    case DK_TS: return in({TFGPU_R_TIME}) ? 0 : 2;
    case DK_TSTZ: return in({TFGPU_R_TIME, TFGPU_R_STRING}) ? 0 : 1;
    case DK_JSON: return in({TFGPU_R_JSON, TFGPU_R_STRING}) ? 0 : 2;
    case DK_HSTORE: return in({TFGPU_R_JSON, TFGPU_R_STRING}) ? 0 : 1;
    case DK_TIME: case DK_TIMETZ: case DK_MONEY: case DK_XML: case DK_POINT: case DK_TSRANGE: case DK_NUMRANGE: case DK_TSTZRANGE: case DK_INTERVAL: return in({TFGPU_R_STRING}) ? 0 : 2;
    case DK_WRONG_TYPE: return 0;
    case DK_NUMERIC: case DK_NUMERIC_TEXT: return in({TFGPU_R_STRING}) ? 0 : ((arg >> 17) & 1u) ? 2 : in({TFGPU_R_JSONNUM}) ? 0 : 1;  // ydb:Decimal asserts a string
    case DK_MARSHAL: return 0;
    case DK_INT_REPRS: return repr > 0 && repr < 32 && ((arg >> repr) & 1u) ? 0 : 1;
    case DK_MY_TINYINT1: return in({TFGPU_R_INT8}) ? 0 : 2;
    case DK_MY_FLOAT: return in({TFGPU_R_FLOAT64, TFGPU_R_JSONNUM}) || (arg && in({TFGPU_R_FLOAT32})) ? 0 : 1;
    case DK_MY_BINARY: return in({TFGPU_R_BYTES}) ? 0 : in({TFGPU_R_STRING}) ? 2 : 1;
    case DK_MY_BIT1: return in({TFGPU_R_BYTES, TFGPU_R_STRING}) ? 0 : 2;
    case DK_MY_BITS: return in({TFGPU_R_BYTES}) ? 0 : 2;
    case DK_MY_TIMESTAMP: case DK_MY_DATETIME: return in({TFGPU_R_TIME}) ? 0 : 2;
    case DK_MY_TIME: case DK_MY_YEAR: return in({TFGPU_R_STRING}) ? 0 : 2;
    case DK_MY_DECIMAL: case DK_TO_STRING: return 0;
    case DK_ARR_INT: case DK_ARR_STRING: case DK_ARR_COPY: case DK_ARR_ELEM: return in({TFGPU_R_JSON}) ? 0 : 2;
    case DK_YDB_UINT64: return in({TFGPU_R_UINT64}) ? 0 : 1;
    case DK_YDB_DATE: return in({TFGPU_R_TIME}) ? 0 : 1;
    default: return 2;
  }
}

// one stream's cells
struct CellList {
  std::vector<ECell> cells;
  std::string *blob;
  void push_const(uint32_t apply, const std::string &text) {
    if (text.empty()) return;
    if (!cells.empty() && cells.back().kind == EC_CONST && cells.back().apply == apply && cells.back().pre_off + cells.back().pre_len == blob->size()) {
      *blob += text; cells.back().pre_len += (uint32_t)text.size(); return;
    }
    ECell c{}; c.kind = EC_CONST; c.apply = apply; c.pre_off = (uint32_t)blob->size(); c.pre_len = (uint32_t)text.size();
    *blob += text; cells.push_back(c);
  }
  void push_value(uint32_t apply, const std::string &prefix, const DColumn &col, const ColPlan &cp, bool from_old, const DColumn *alt = nullptr) {
    // a Go type AddPg does not take for this column fails (or leaves to the host) the first NON-NIL value of it, as the reference's type
    // switches do: a column of nils has no type to object to
    const int rk = repr_ok(cp.dk, col.repr, cp.arg);
    ECell c{}; c.kind = EC_VALUE; c.apply = apply; c.pre_off = (uint32_t)blob->size(); c.pre_len = (uint32_t)prefix.size();
    *blob += prefix; c.c = dcol_of(col); c.dk = rk ? (uint32_t)(DK_WRONG_TYPE + rk - 1) : cp.dk; c.arg = cp.arg; c.edk = cp.edk; c.from_old = from_old ? 1u : 0u;
    if (alt) { const int ra = repr_ok(cp.dk, alt->repr, cp.arg); c.alt = dcol_of(*alt); c.dk_alt = ra ? (uint32_t)(DK_WRONG_TYPE + ra - 1) : cp.dk; c.has_alt = 1; }
    cells.push_back(c);
  }
};
struct Member { std::string name; const DColumn *col; const ColPlan *plan; bool from_old; std::string constant; const DColumn *alt = nullptr; };
// one Go map as an object: members in key order, `{` … `}` (or `{}`)
static void push_object(CellList &L, uint32_t apply, std::vector<Member> ms) {
  std::stable_sort(ms.begin(), ms.end(), [](const Member &a, const Member &b) { return a.name < b.name; });
  for (size_t i = 1; i < ms.size(); i++) if (ms[i].name == ms[i - 1].name) throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_debezium_emit: column name " + ms[i].name + " repeats: stays with the stock emitter");
  if (ms.empty()) { L.push_const(apply, "{}"); return; }
  for (size_t i = 0; i < ms.size(); i++) {
    const std::string pre = std::string(i ? "," : "{") + jstr(ms[i].name) + ":";
    if (ms[i].col) {
      L.push_value(apply, pre, *ms[i].col, *ms[i].plan, ms[i].from_old, ms[i].alt);
      if (!ms[i].constant.empty() && ms[i].col->absent) {  // rows that leave the column out write this constant instead (the TOAST placeholder)
        ECell &c = L.cells.back();
        c.absent = ptr<uint8_t>(ms[i].col->absent); c.ph_off = (uint32_t)L.blob->size(); c.ph_len = (uint32_t)ms[i].constant.size();
        *L.blob += ms[i].constant;
      }
    } else L.push_const(apply, pre + ms[i].constant);
  }
  L.push_const(apply, "}");
}

// `sp`: the events' starts on the host — in the lane's page-locked ring when they fit (a pageable vector is a staged copy behind a fresh allocation:
// ~0.1 ms per 4 x 10^5 events and stream), else in `start`.  The ring keeps a call's read-backs apart as long as they total less than its
// size minus the largest of them (Context::pin wraps only in front of an allocation): RING_EACH bounds each, four of them per call.
struct Stream { Buf out; std::vector<uint32_t> start; const uint32_t *sp = nullptr; uint64_t total = 0; };
constexpr size_t RING_EACH = 1792u << 10;

// One stream in two halves, so that the keys' and the values' kernels queue behind one another and the host waits twice per call, not four times:
// measure = cell lengths + per-event layout (+ the read-back of the total and of the first failing cell), write = scan, cells, constants (+ the starts' read-back).
struct StreamRun {
  EParams p; const CellList *L; const std::string *blob; const char *what;
  Buf dcells, dblob, cell, ev_len, acc, err, const_at;
  bool walk = true;
  uint64_t host[3];
  Stream S;
};
static void stream_measure(StreamRun &R, EParams p, const CellList &L, const std::string &blob, hipStream_t st, const char *what) {
  const int64_t nev = p.nev;
  R.L = &L; R.blob = &blob; R.what = what;
  R.dcells = upload_small(L.cells.data(), L.cells.size() * sizeof(ECell));
  std::string padded = blob; padded.append(16, '\0');  // the text readers load whole words
  R.dblob = upload_small(padded.data(), padded.size());
  static const bool walk_default = [] { const char *e = std::getenv("TFGPU_DBZ_WALK"); return !(e && e[0] == '0'); }();  // A/B measurements
  R.walk = walk_default;
  if (!R.walk) R.cell = dalloc((size_t)L.cells.size() * nev * 4 + 16);
  R.ev_len = dalloc((size_t)(nev + 1) * 4 + 16); R.acc = dalloc_zero(8);
  R.err = dalloc(16);
  TF_HIP(hipMemsetAsync(R.err->p, 0xFF, 16, st));
  p.cells = ptr<ECell>(R.dcells); p.ncells = (int32_t)L.cells.size(); p.blob = ptr<uint8_t>(R.dblob);
  p.cell = ptr<uint32_t>(R.cell); p.ev_len = ptr<uint32_t>(R.ev_len); p.total64 = ptr<unsigned long long>(R.acc); p.err = ptr<unsigned long long>(R.err);
  R.p = p;
  if (R.walk) { KernelTimer t("dbz_walk_len"); dbz_walk_len<<<blocks(nev), 256, 0, st>>>(p); }
  else {
    { KernelTimer t("dbz_cell_len"); dbz_cell_len<<<dim3(blocks(nev), (unsigned)L.cells.size()), 256, 0, st>>>(p); }
    { KernelTimer t("dbz_event_layout"); dbz_event_layout<<<blocks(nev), 256, 0, st>>>(p); }
  }
  d2h(R.host, R.acc->p, 8); d2h(R.host + 1, R.err->p, 16);
}
static void stream_check(StreamRun &R) {  // after the sync that follows stream_measure
  const CellList &L = *R.L; const std::string &blob = *R.blob; const char *what = R.what; const uint64_t *host = R.host;
  if (host[1] != ~0ull || host[2] != ~0ull) {
    const bool invalid = host[1] != ~0ull && (host[2] == ~0ull || host[1] <= host[2]);
    const uint64_t key = invalid ? host[1] : host[2];
    const int64_t e = (int64_t)(key >> 16); const uint32_t ci = (uint32_t)(key & 0xFFFF);
    std::string nm = "?";
    {  // the member name is the tail of the cell's prefix: …"name":
      const ECell &ec = L.cells[ci]; std::string pre = blob.substr(ec.pre_off, ec.pre_len);
      const size_t q2 = pre.rfind("\":"), q1 = q2 == std::string::npos ? q2 : pre.rfind('"', q2 - 1);
      if (q1 != std::string::npos) nm = pre.substr(q1 + 1, q2 - q1 - 1);
    }
    if (invalid) throw Error(TFGPU_ERR_INVALID, std::string("tfgpu_debezium_emit: ") + what + " of event " + std::to_string(e) + ": unable to emit value, colName: " + nm);
    throw Error(TFGPU_ERR_UNSUPPORTED, std::string("tfgpu_debezium_emit: ") + what + " of event " + std::to_string(e) + ", column " + nm +
                ": a value the device leaves to the stock emitter (a Go type it does not convert for this column, a numeric beyond 38 digits, a float text Go decides with big arithmetic)");
  }
  R.S.total = host[0];
  if (R.S.total > 0xFFFFFFF0ull) throw Error(TFGPU_ERR_UNSUPPORTED, std::string("tfgpu_debezium_emit: ") + what + "s of one call exceed 4 GiB (" + std::to_string(R.S.total) + " bytes): emit the batch in slices");
}
static void stream_write(StreamRun &R, hipStream_t st) {
  EParams &p = R.p; const CellList &L = *R.L; const int64_t nev = p.nev;
  exclusive_scan_u32(p.ev_len, p.ev_len, nev, true);
  R.S.out = dalloc(R.S.total + 64);
  p.out = ptr<uint8_t>(R.S.out);
  size_t nlong = 0;
  for (auto &c : L.cells) if (c.kind == EC_CONST && c.pre_len > CONST_INLINE) nlong++;
  if (R.walk) {
    R.const_at = dalloc(std::max<size_t>(nlong, 1) * (size_t)nev * 4 + 16);
    KernelTimer t("dbz_walk_write"); dbz_walk_write<<<blocks(nev), 256, 0, st>>>(p, ptr<uint32_t>(R.const_at));
  } else { KernelTimer t("dbz_cell_write"); dbz_cell_write<<<dim3(blocks(nev), (unsigned)L.cells.size()), 256, 0, st>>>(p); }
  size_t slot = 0;
  for (size_t ci = 0; ci < L.cells.size(); ci++) if (L.cells[ci].kind == EC_CONST && L.cells[ci].pre_len > CONST_INLINE) {
    const uint32_t pieces = (L.cells[ci].pre_len + FILL_PIECE - 1) / FILL_PIECE;
    const uint64_t items = (uint64_t)nev * pieces;
    if (items > 0x1FFFFFFFFull) throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_debezium_emit: too many events for one call");
    const uint32_t *at = R.walk ? ptr<uint32_t>(R.const_at) + slot * (size_t)nev : p.cell + ci * (size_t)nev;
    KernelTimer t("dbz_fill_const");
    dbz_fill_const<<<(unsigned)((items + 3) / 4), 256, 0, st>>>(p, (int32_t)ci, pieces, at);
    slot++;
  }
  if ((size_t)(nev + 1) * 4 <= RING_EACH) R.S.sp = d2h_u32(p.ev_len, (size_t)nev + 1);
  else { R.S.start.resize((size_t)nev + 1); d2h(R.S.start.data(), p.ev_len, (size_t)(nev + 1) * 4); R.S.sp = R.S.start.data(); }
}

}  // namespace dbz

Buf keys_changed_device(const tfgpu_dbatch &in, Buf *count_out);  // tf_collapse.hip
template <class T> static const T *dbz_meta(const tfgpu_row_meta *m, const T *p, size_t count, std::vector<Buf> &keep) {
  if (!m || !p) return nullptr;
  if (m->mem == TFGPU_MEM_DEVICE) return p;
  Buf d = dalloc(count * sizeof(T) + 16);
  h2d(d->p, p, count * sizeof(T));
  keep.push_back(d);
  return reinterpret_cast<const T *>(d->p);
}
}  // namespace tf

extern "C" int tfgpu_debezium_emit(const tfgpu_dbz_emit_options *o, const tfgpu_dbatch *b, const tfgpu_row_meta *meta, tfgpu_dbuf **keys, uint64_t *key_start,
                                   tfgpu_dbuf **values, uint64_t *val_start, uint8_t *val_null, int64_t *msg_row, int64_t cap, int64_t *nmsg) {
  using namespace tf; using namespace tf::dbz;
  try {
  tf::dense(b, true);  // its rows may still be a selection (tfgpu_dbatch::pending); rows that leave columns out (ABSENT cells) get the TOAST placeholder
    if (!o || !b || !keys || !values || !nmsg || !o->table_schema || (cap > 0 && (!key_start || !val_start || !val_null || !msg_row)))
      return fail(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: null argument");
    if (o->nparams < 0 || (o->nparams > 0 && (!o->param_keys || !o->param_values))) return fail(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: nparams without the key / value arrays");
    if (o->table_schema->ncols < 0 || (o->table_schema->ncols > 0 && !o->table_schema->cols)) return fail(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: table_schema without columns");
    Context &cx = ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    hipStream_t st = cx.stream;
    const int64_t n = b->nrows;
    static const bool timing = std::getenv("TFGPU_DBZ_TIMING") != nullptr;   // measurement only: host phases to stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (!timing) return; auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "dbz_emit %-10s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count()); t_prev = t; };
    // ---- connector parameters (parameters.go:140-215) ----
    Params P;
    static const char *const DEFAULTS[][2] = {{"database.dbname", ""}, {"topic.prefix", ""}, {"dt.unknown.types.policy", "fail"}, {"dt.add.original.type.info", "false"},
      {"dt.source.type", ""}, {"decimal.handling.mode", "precise"}, {"tombstones.on.delete", "true"}, {"binary.handling.mode", "bytes"},
      {"unavailable.value.placeholder", "__debezium_unavailable_value"}, {"key.converter", "org.apache.kafka.connect.json.JsonConverter"},
      {"value.converter", "org.apache.kafka.connect.json.JsonConverter"}, {"key.converter.schemas.enable", "true"}, {"value.converter.schemas.enable", "true"},
      {"dt.batching.max.size", "0"}, {"interval.handling.mode", "numeric"}};
    for (auto &d : DEFAULTS) P.m[d[0]] = d[1];
    for (int i = 0; i < o->nparams; i++) if (o->param_keys[i]) P.m[o->param_keys[i]] = o->param_values[i] ? o->param_values[i] : "";
    // the packers (packer/factory.go:13-98): a schema registry URL / YSR namespace selects the registry packers (host), schemas.enable=false the
    // payload-only packer, anything else PackerIncludeSchema; the converter class names are not consulted there
    for (const char *k : {"key.converter.schema.registry.url", "value.converter.schema.registry.url", "value.converter.ysr.namespace.id"})
      if (!P.get(k).empty()) return fail(TFGPU_ERR_UNSUPPORTED, std::string("tfgpu_debezium_emit: ") + k + " is set: the schema-registry packers stay with the stock emitter");
    const bool key_schema = P.get("key.converter.schemas.enable") != "false", val_schema = P.get("value.converter.schemas.enable") != "false";
    {  // parameters.Validate (validate.go:5-17): dt.batching.max.size needs a schema registry (or logbroker)
      const std::string &bs = P.get("dt.batching.max.size");
      long long v = 0; bool ok = !bs.empty();   // strconv.Atoi: anything it refuses counts as 0
      { size_t i = (bs[0] == '+' || bs[0] == '-') ? 1 : 0; if (i >= bs.size() || bs.size() - i > 18) ok = false; for (; ok && i < bs.size(); i++) { if (bs[i] < '0' || bs[i] > '9') ok = false; else v = v * 10 + (bs[i] - '0'); } }
      if (ok && v != 0) return fail(TFGPU_ERR_INVALID, o->drop_keys ? "tfgpu_debezium_emit: dt.batching.max.size can be used ONLY with schema-registry for values encoding" : "tfgpu_debezium_emit: dt.batching.max.size can be used only with lb/yds");
    }
    if (P.get("binary.handling.mode") != "bytes") return fail(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: unsupported binary.handling.mode: " + P.get("binary.handling.mode"));
    const std::string &source_type = P.get("dt.source.type");
    if (!source_type.empty() && source_type != "pg" && source_type != "ydb" && source_type != "mysql") return fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_debezium_emit: dt.source.type=" + source_type + " stays with the stock emitter");
    const std::string server = P.get("topic.prefix"), database = P.get("database.dbname"), version = o->version ? o->version : "1.1.2.Final";
    const bool skip_tombstones = P.get("tombstones.on.delete") == "false";

    auto kres = std::make_unique<tfgpu_dbuf>(), vres = std::make_unique<tfgpu_dbuf>();
    if (n == 0) { kres->mem = dalloc(64); vres->mem = dalloc(64); *keys = kres.release(); *values = vres.release(); *nmsg = 0; if (cap >= 0 && key_start) { key_start[0] = 0; val_start[0] = 0; } return TFGPU_OK; }
    if (meta && meta->n < 0) return fail(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: row meta with a negative length");
    materialize(*b);

    // ---- the table's columns ----
    const tfgpu_schema &ts = *o->table_schema;
    std::map<std::string, int> index;
    std::vector<ColPlan> plans((size_t)ts.ncols);
    int npk = 0;
    for (int i = 0; i < ts.ncols; i++) {
      const std::string nm = ts.cols[i].name ? ts.cols[i].name : "";
      index[nm] = i;   // mapColToIndex: the last one of a repeated name
      plans[(size_t)i] = plan_column(ts.cols[i], P, o->snapshot != 0);
      if (ts.cols[i].flags & TFGPU_COL_KEY) npk++;
    }
    auto col_index = [&](const std::string &nm) -> int {
      auto it = index.find(nm);
      if (it == index.end()) throw Error(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: invalid changeItem - column absent in schema: " + nm);
      return it->second;
    };
    const bool has_old = !b->old_keys.empty();
    const bool has_prev = (int)b->old_keys.size() > npk;  // hasPreviousValues (emitter_value_converter.go:274-283)

    // ---- the schema halves (ToKafkaSchemaKey / ToKafkaSchemaVal) ----
    std::string fields_all, fields_key;
    for (int i = 0; i < ts.ncols; i++) {
      if (!fields_all.empty()) fields_all += ",";
      fields_all += plans[(size_t)i].descr;
      if (ts.cols[i].flags & TFGPU_COL_KEY) { if (!fields_key.empty()) fields_key += ","; fields_key += plans[(size_t)i].descr; }
    }
    const std::string record = server + "." + b->ns + "." + b->table;
    const std::string key_schema_json = "{\"fields\":[" + fields_key + "],\"name\":" + jstr(record + ".Key") + ",\"optional\":false,\"type\":\"struct\"}";
    auto side = [&](const char *field) { return "{\"field\":\"" + std::string(field) + "\",\"fields\":[" + fields_all + "],\"name\":" + jstr(record + ".Value") + ",\"optional\":true,\"type\":\"struct\"}"; };
    auto f = [](const char *field, const char *type, bool optional) { return std::string("{\"field\":\"") + field + "\",\"optional\":" + (optional ? "true" : "false") + ",\"type\":\"" + type + "\"}"; };
    std::string src_fields = f("version", "string", false) + "," + f("connector", "string", false) + "," + f("name", "string", false) + "," + f("ts_ms", "int64", false) +
      ",{\"default\":\"false\",\"field\":\"snapshot\",\"name\":\"io.debezium.data.Enum\",\"optional\":true,\"parameters\":{\"allowed\":\"true,last,false\"},\"type\":\"string\",\"version\":1}," +
      f("db", "string", false) + "," + f("table", "string", false);
    std::string src_schema = "{\"field\":\"source\",\"fields\":[";
    if (source_type == "mysql") {   // the `table` field turns optional, seven more fields (fields_descr_source.go:70-103)
      const std::string tbl = f("table", "string", false); src_fields.replace(src_fields.rfind(tbl), tbl.size(), f("table", "string", true));
      src_fields += "," + f("file", "string", false) + "," + f("gtid", "string", true) + "," + f("pos", "int64", false) + "," + f("query", "string", true) + "," + f("row", "int32", false) + "," +
                    f("server_id", "int64", false) + "," + f("thread", "int64", true);
    }
    if (source_type == "pg") src_fields += "," + f("lsn", "int64", true) + "," + f("schema", "string", false) + "," + f("txId", "int64", true) + "," + f("xmin", "int64", true);
    src_schema += src_fields + "],";
    if (source_type == "pg") src_schema += "\"name\":\"io.debezium.connector.postgresql.Source\",";
    if (source_type == "mysql") src_schema += "\"name\":\"io.debezium.connector.mysql.Source\",";
    src_schema += "\"optional\":false,\"type\":\"struct\"}";
    const std::string tx_schema = "{\"field\":\"transaction\",\"fields\":[" + f("id", "string", false) + "," + f("total_order", "int64", false) + "," + f("data_collection_order", "int64", false) + "],\"optional\":true,\"type\":\"struct\"}";
    const std::string val_schema_json = "{\"fields\":[" + side("before") + "," + side("after") + "," + src_schema + "," + f("op", "string", false) + "," + f("ts_ms", "int64", true) + "," + tx_schema +
      "],\"name\":" + jstr(record + ".Envelope") + ",\"optional\":false,\"type\":\"struct\"}";

    lap("plan");
    // ---- events: on the device (a count per row, a scan, a fill); the host learns only how many there are ----
    Buf d_ev_row, d_ev_type;
    bool identity = true;
    int64_t nev = n;
    if (b->kind) {
      identity = false;
      Buf changed;
      {  // ChangeItem.KeysChanged (change_item.go:237-286) over the TableSchema's PrimaryKey columns — the ones `table_schema` flags
        tfgpu_dbatch view;
        view.nrows = n; view.cols = b->cols; view.old_keys = b->old_keys; view.old_present = b->old_present; view.kind = b->kind;
        for (int i = 0; i < ts.ncols; i++) if (ts.cols[i].flags & TFGPU_COL_KEY) view.key_names.push_back(ts.cols[i].name ? ts.cols[i].name : "");
        changed = keys_changed_device(view, nullptr);
      }
      Buf cnt = dalloc((size_t)(n + 1) * 4 + 16);
      dbz_event_count<<<blocks(n), 256, 0, st>>>(ptr<uint8_t>(b->kind), ptr<uint8_t>(changed), n, skip_tombstones ? 1 : 0, ptr<uint32_t>(cnt));
      exclusive_scan_u32(ptr<uint32_t>(cnt), ptr<uint32_t>(cnt), n, true);
      const uint32_t *h = d2h_u32(ptr<uint32_t>(cnt) + n);
      sync();
      nev = *h;
      if (nev > 0) {
        d_ev_row = dalloc((size_t)nev * 4 + 16); d_ev_type = dalloc((size_t)nev + 16);
        dbz_event_fill<<<blocks(n), 256, 0, st>>>(ptr<uint8_t>(b->kind), ptr<uint8_t>(changed), n, skip_tombstones ? 1 : 0, ptr<uint32_t>(cnt), ptr<int32_t>(d_ev_row), ptr<uint8_t>(d_ev_type));
      }
    }
    if (cap >= 0 && nev > cap) return fail(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: " + std::to_string(nev) + " messages, room for " + std::to_string(cap));
    *nmsg = nev;
    if (nev == 0) { kres->mem = dalloc(64); vres->mem = dalloc(64); *keys = kres.release(); *values = vres.release(); if (key_start) { key_start[0] = 0; val_start[0] = 0; } return TFGPU_OK; }

    lap("events");
    std::vector<Buf> keep;
    EParams p{};
    p.nev = nev;
    p.ev_row = ptr<int32_t>(d_ev_row); p.ev_type = ptr<uint8_t>(d_ev_type);
    p.kind = ptr<uint8_t>(b->kind); p.old_present = ptr<uint8_t>(b->old_present); p.src_row = ptr<int32_t>(b->src_row);
    p.has_old = has_old; p.has_prev = has_prev; p.snapshot = o->snapshot ? 1 : 0;
    if (meta) {
      // every row's entry must exist: src_row (or the row index) is below meta->n
      if (!b->src_row && meta->n < n) return fail(TFGPU_ERR_INVALID, "tfgpu_debezium_emit: row meta shorter than the batch");
      p.m_id = dbz_meta(meta, meta->id, (size_t)meta->n, keep); p.m_lsn = dbz_meta(meta, meta->lsn, (size_t)meta->n, keep); p.m_commit = dbz_meta(meta, meta->commit_time, (size_t)meta->n, keep);
      if ((source_type == "ydb" || source_type == "mysql") && meta->tx_id_offsets && meta->n > 0) {
        uint32_t tx_bytes = 0;
        if (meta->mem == TFGPU_MEM_DEVICE) { const uint32_t *h = d2h_u32(meta->tx_id_offsets + meta->n); sync(); tx_bytes = *h; } else tx_bytes = meta->tx_id_offsets[meta->n];
        p.m_tx_off = dbz_meta(meta, meta->tx_id_offsets, (size_t)meta->n + 1, keep); p.m_tx = dbz_meta(meta, meta->tx_id_data, (size_t)tx_bytes, keep);
      }
    }
    p.p10 = pow10_table(); p.p128 = reinterpret_cast<const uint64_t *>(p.p10 + 632);

    // ---- values: {"payload":{"after":A,"before":B,"op":…,"source":{…},"transaction":null,"ts_ms":N},"schema":S} ----
    std::string blob;
    CellList V; V.blob = &blob;
    {
      V.push_const(EA_VALUE, val_schema ? "{\"payload\":{\"after\":" : "{\"after\":");
      V.push_const(EA_AFTER_NULL, "null");
      std::vector<Member> after;
      std::vector<char> in_batch((size_t)ts.ncols, 0);
      for (auto &c : b->cols) { const int i = col_index(c.name); in_batch[(size_t)i] = 1; after.push_back({c.name, &c, &plans[(size_t)i], false, c.absent ? jstr(P.get("unavailable.value.placeholder")) : std::string()}); }
      if (ts.ncols > (int)b->cols.size())  // TOAST (buildKV :311-323)
        for (int i = 0; i < ts.ncols; i++) if (!in_batch[(size_t)i]) after.push_back({ts.cols[i].name ? ts.cols[i].name : "", nullptr, nullptr, false, jstr(P.get("unavailable.value.placeholder"))});
      push_object(V, EA_AFTER, after);
      V.push_const(EA_VALUE, ",\"before\":");
      V.push_const(EA_BEFORE_NULL, "null");
      std::vector<Member> before_d, before_u;
      std::vector<char> in_old((size_t)ts.ncols, 0);
      // a delete's `before`: every column nil, under the row's own values when the source is MySQL, under the OldKeys (valPayload :463-486)
      const bool my_before = source_type == "mysql";
      auto own = [&](const std::string &nm) -> const DColumn * { if (!my_before) return nullptr; for (auto &c : b->cols) if (c.name == nm) return &c; return nullptr; };
      for (auto &c : b->old_keys) { const int i = col_index(c.name); in_old[(size_t)i] = 1; Member m{c.name, &c, &plans[(size_t)i], true, ""}; before_u.push_back(m); m.alt = own(c.name); before_d.push_back(m); }
      for (int i = 0; i < ts.ncols; i++) if (!in_old[(size_t)i] && index[ts.cols[i].name ? ts.cols[i].name : ""] == i) {
        const std::string nm = ts.cols[i].name ? ts.cols[i].name : "";
        if (const DColumn *c = own(nm)) before_d.push_back({nm, c, &plans[(size_t)i], false, ""});
        else before_d.push_back({nm, nullptr, nullptr, false, "null"});
      }
      push_object(V, EA_BEFORE_D, before_d);
      if (has_prev) push_object(V, EA_BEFORE_U, before_u);
      ECell tail{}; tail.kind = EC_TAIL; tail.apply = EA_VALUE; V.cells.push_back(tail);
      if (val_schema) V.push_const(EA_VALUE, ",\"schema\":" + val_schema_json + "}");
      // the tail's segments: source is a map too — connector, db, lsn, name, schema, snapshot, table, ts_ms, txId, version, xmin
      std::vector<std::pair<std::string, int>> segs;
      const std::string snap = o->snapshot ? "true" : "false";
      if (source_type == "pg") {
        segs.push_back({"\",\"source\":{\"connector\":\"postgresql\",\"db\":" + jstr(database) + ",\"lsn\":", TN_LSN});
        segs.push_back({",\"name\":" + jstr(server) + ",\"schema\":" + jstr(b->ns) + ",\"snapshot\":\"" + snap + "\",\"table\":" + jstr(b->table) + ",\"ts_ms\":", TN_TS});
        segs.push_back({",\"txId\":", TN_ID});
        segs.push_back({",\"version\":" + jstr(version) + ",\"xmin\":null},\"transaction\":null,\"ts_ms\":", TN_TS});
      } else if (source_type == "mysql") {   // db = ChangeItem.Schema, file / pos from the LSN, gtid = TxID (emitter_value_converter.go:352-367)
        segs.push_back({"\",\"source\":{\"connector\":\"mysql\",\"db\":" + jstr(b->ns) + ",\"file\":\"mysql-log.", TN_FILE6});
        segs.push_back({"\",\"gtid\":", TN_TXID});
        segs.push_back({",\"name\":" + jstr(server) + ",\"pos\":", TN_POS});
        segs.push_back({",\"query\":null,\"row\":0,\"server_id\":0,\"snapshot\":\"" + snap + "\",\"table\":" + jstr(b->table) + ",\"thread\":null,\"ts_ms\":", TN_TS});
        segs.push_back({",\"version\":" + jstr(version) + "},\"transaction\":null,\"ts_ms\":", TN_TS});
      } else if (source_type == "ydb") {   // + txId (*string) and step = CommitTime (emitter_value_converter.go:368-377)
        segs.push_back({"\",\"source\":{\"db\":" + jstr(database) + ",\"name\":" + jstr(server) + ",\"snapshot\":\"" + snap + "\",\"step\":", TN_STEP});
        segs.push_back({",\"table\":" + jstr(b->table) + ",\"ts_ms\":", TN_TS});
        segs.push_back({",\"txId\":", TN_TXID});
        segs.push_back({",\"version\":" + jstr(version) + "},\"transaction\":null,\"ts_ms\":", TN_TS});
      } else {
        segs.push_back({"\",\"source\":{\"db\":" + jstr(database) + ",\"name\":" + jstr(server) + ",\"snapshot\":\"" + snap + "\",\"table\":" + jstr(b->table) + ",\"ts_ms\":", TN_TS});
        segs.push_back({",\"version\":" + jstr(version) + "},\"transaction\":null,\"ts_ms\":", TN_TS});
      }
      segs.push_back({"}", TN_NONE});
      p.ntseg = (int32_t)segs.size();
      for (size_t q = 0; q < segs.size(); q++) { p.tseg_off[q] = (uint32_t)blob.size(); p.tseg_len[q] = (uint32_t)segs[q].first.size(); p.tnum[q] = segs[q].second; blob += segs[q].first; }
    }
    // ---- keys: {"payload":{pk members},"schema":KS} ----
    CellList K; K.blob = &blob;
    if (!o->drop_keys) {
      if (key_schema) K.push_const(EA_KEY, "{\"payload\":");
      std::vector<Member> from_new, from_old;
      for (auto &c : b->cols) {
        const int i = col_index(c.name);
        if (!(ts.cols[i].flags & TFGPU_COL_KEY)) continue;
        if (c.absent) return fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_debezium_emit: rows leave the primary-key column " + c.name + " out of their ColumnNames (the message key then has fewer members: the stock emitter takes the batch)");
        from_new.push_back({c.name, &c, &plans[(size_t)i], false, ""});
      }
      for (auto &c : b->old_keys) { const int i = col_index(c.name); if (ts.cols[i].flags & TFGPU_COL_KEY) from_old.push_back({c.name, &c, &plans[(size_t)i], true, ""}); }
      push_object(K, EA_KEY_NEW, from_new);
      if (has_old) push_object(K, EA_KEY_OLD, from_old);
      if (key_schema) K.push_const(EA_KEY, ",\"schema\":" + key_schema_json + "}");
    }
    if (V.cells.size() > 0xFFFF || K.cells.size() > 0xFFFF) return fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_debezium_emit: more than 65535 cells per message");

    lap("cells");
    StreamRun RV, RK;
    stream_measure(RV, p, V, blob, st, "value");
    if (!o->drop_keys) stream_measure(RK, p, K, blob, st, "key");
    sync();
    stream_check(RV);
    if (!o->drop_keys) stream_check(RK);
    lap("measure");
    stream_write(RV, st);
    if (!o->drop_keys) stream_write(RK, st);
    std::vector<int32_t> ev_row_own; std::vector<uint8_t> ev_type_own;
    const int32_t *ev_row = nullptr; const uint8_t *ev_type = nullptr;
    if (!identity) {
      if ((size_t)nev * 4 <= RING_EACH) {
        ev_row = reinterpret_cast<const int32_t *>(d2h_u32(d_ev_row->p, (size_t)nev));
        uint8_t *t = cx.pin_n<uint8_t>((size_t)nev);
        if (!t) throw Error(TFGPU_ERR_NOMEM, "pinned read-back arena exhausted");
        d2h(t, d_ev_type->p, (size_t)nev);
        ev_type = t;
      } else {
        ev_row_own.resize((size_t)nev); ev_type_own.resize((size_t)nev);
        d2h(ev_row_own.data(), d_ev_row->p, (size_t)nev * 4); d2h(ev_type_own.data(), d_ev_type->p, (size_t)nev);
        ev_row = ev_row_own.data(); ev_type = ev_type_own.data();
      }
    }
    sync();
    if (o->drop_keys) { RK.S.out = dalloc(64); RK.S.start.assign((size_t)nev + 1, 0u); RK.S.sp = RK.S.start.data(); }
    Stream &SV = RV.S, &SK = RK.S;
    lap("write");
    kres->mem = SK.out; kres->size = SK.total; vres->mem = SV.out; vres->size = SV.total;
    {  // the caller's arrays, one plain widening loop each (they vectorize; the starts interleaved with a branch per event did not)
      const uint32_t *__restrict ks = SK.sp, *__restrict vs = SV.sp;
      uint64_t *__restrict ko = key_start, *__restrict vo = val_start;
      for (int64_t e = 0; e <= nev; e++) ko[e] = ks[e];
      for (int64_t e = 0; e <= nev; e++) vo[e] = vs[e];
      uint8_t *__restrict vn = val_null; int64_t *__restrict mr = msg_row;
      if (identity) { std::memset(vn, 0, (size_t)nev); for (int64_t e = 0; e < nev; e++) mr[e] = e; }
      else {
        const uint8_t *__restrict et = ev_type; const int32_t *__restrict er = ev_row;
        for (int64_t e = 0; e < nev; e++) vn[e] = et[e] == EV_TOMBSTONE ? 1 : 0;
        for (int64_t e = 0; e < nev; e++) mr[e] = er[e];
      }
    }
    *keys = kres.release(); *values = vres.release();
    lap("outputs");
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}
