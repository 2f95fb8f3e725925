// tf_serialize.hip — sink-side marshalling from device columns:
//
//   TFGPU_FMT_CH_JSON_EACH_ROW  httpuploader.MarshalCItoJSON / marshalValue / marshalNumericValue /
//                               marshalTime / writeQuoted (pkg/providers/clickhouse/httpuploader/marshal.go:65-419)
//   TFGPU_FMT_JSON              jsonSerializer (pkg/serializer/json.go:29-83, json_format.go:32-82) under
//                               batchSerializer (batch.go:206-219): encoding/json of a map — keys sorted,
//                               SetEscapeHTML(false)
//   TFGPU_FMT_CSV               csvSerializer (csv.go:22-74, csv_format.go:32-127) + encoding/csv quoting
//
// Columns in, row-major text out: (1) one lane per CELL computes the cell's byte length, (2) one lane per
// ROW turns the lengths into in-row offsets, (3) prefix scan of the row lengths, (4) one lane per cell
// writes key + value at its final position.  Every value is rendered by the same templated emitter in
// passes (1) and (4) — a counting sink, then a writing sink — so lengths and bytes cannot disagree.
// HBM-bound byte kernels: algorithmic traffic B_bin + B_text per row.
#include <algorithm>

#include <unordered_map>

#include "tf_emit.hpp"
#include "tf_f64range.hpp"
#include "tf_wave.hpp"

namespace tf {

struct SCol {
  DCol c;
  uint32_t pre_off, pre_len;  // key / separator bytes emitted before the value
  uint32_t ch_flags;          // TFGPU_CH_*
  uint32_t prec;              // DateTime64 precision
  uint32_t kind;              // queue formats: QC_*
  uint32_t apply;             // queue formats: QA_* (which rows hold this cell)
  const uint8_t *absent;      // native queue format, QA_LISTED cells: the column's ABSENT bitmap (DColumn::absent) or null; `prec` = the column's index
};
// queue formats (tfgpu_queue_serialize): a row is a list of cells — header, constants, values, old-key values, trailer
enum { QC_VALUE = 0, QC_HEADER = 1, QC_CONST = 2, QC_TRAILER = 3, QC_NAME = 4 /* `,"name"`: an entry of a row's own columnnames */ };
enum { QA_ALWAYS = 0, QA_NAMES = 1 /* names_form == 0 */, QA_OLD = 2 /* the row has OldKeys */, QA_ROW_EVENT = 3 /* kind != other */,
       // batches whose rows list different columns (tfgpu_column.absent): a cell of a column the row lists (the first listed one drops its
       // leading comma), a constant of rows that list any column at all (`,"columnvalues":[` … `]`: len(c.ColumnValues) > 0, change_item.go:600)
       QA_LISTED = 4, QA_ANY_LISTED = 5 };
constexpr uint32_t QCONST_INLINE = 64;  // constants up to this many bytes are written by the cell's lane, longer ones by ser_fill_const

struct QueueParams {
  int32_t qformat;               // 0 = not a queue format, else TFGPU_QFMT_*
  int32_t has_old;
  const uint8_t *kind;           // [nrows] or null
  const int32_t *src_row;        // [nrows] or null
  const uint32_t *part_id;       // [nrows] or null
  const uint8_t *old_present;    // bitmap or null
  const uint32_t *m_id; const uint64_t *m_lsn, *m_commit; const int64_t *m_counter;   // row meta, indexed by src_row
  const uint32_t *m_tx_off; const uint8_t *m_tx; const uint32_t *m_q_off; const uint8_t *m_q; const uint8_t *m_form;
  uint32_t hdr_off, hdr_len;     // `,"schema":…,"table":…,"part":` in the blob
  int32_t ngroups;               // group part ids: rows [gstart[g], gstart[g+1]) carry blob[gpart[g] .. gpart[g+1])
  const int64_t *gstart; const uint32_t *gpart;
  const uint8_t *msg_flags;      // [nrows]: bit 0 = first row of its message, bit 1 = last
  const int32_t *first_listed;   // [nrows], batches with ABSENT cells: the first column the row lists, -1 = none
  const uint16_t *col_order;     // [nrows][ord_n] or null: the row's own order of its columns (tfgpu_dbatch::col_order): the name cells
  int32_t ord_name0, ord_val0, ord_n;  // [ord_name0, + ord_n) and the value cells [ord_val0, + ord_n) are laid out in that order
};

struct SerParams {
  const SCol *cols;
  int32_t ncols;
  int64_t nrows;
  int32_t format, any_as_string, closing_newline;
  const uint8_t *blob;     // prefixes
  uint32_t *cell;          // [ncols][nrows]: length, then offset inside the row
  uint32_t *row_len;       // [nrows+1] → row offsets after the scan
  int32_t *last_present;   // CH: index of the last emitted column per row, -1 = none
  unsigned long long *total64;  // sum of the row lengths (the offsets are 32-bit)
  uint8_t *out;
  int32_t ablate;          // TFGPU_SER_ABLATE (measurement only): 1 = the write pass skips text cells, 2 = it skips the others, 3 = the len pass skips text cells
  QueueParams q;
};

__device__ __forceinline__ int64_t pow10_i64(int k) { int64_t r = 1; while (k-- > 0) r *= 10; return r; }

// ---- one value, ClickHouse JSONEachRow (marshalValue).  Returns false when the column is skipped. ----
template <class S> __device__ __forceinline__ bool emit_ch_value(S &s, const SCol &sc, const CellBits &b, int any_as_string) {
  const DCol &c = sc.c;
  const bool is_text = c.repr == TFGPU_R_STRING || c.repr == TFGPU_R_BYTES;
  uint32_t vn; const uint8_t *vp = cell_text(c, b, vn);
  const uint32_t fl = sc.ch_flags;
  if ((fl & TFGPU_CH_DECIMAL) && is_text) { put_bytes(s, vp, vn); return true; }
  switch (c.dtype) {
    case TFGPU_T_INT8: case TFGPU_T_INT16: case TFGPU_T_INT32: case TFGPU_T_INT64: case TFGPU_T_UINT8: case TFGPU_T_UINT16:
    case TFGPU_T_UINT32: case TFGPU_T_UINT64: case TFGPU_T_FLOAT32: case TFGPU_T_FLOAT64: case TFGPU_T_INTERVAL: {
      const bool q = fl & TFGPU_CH_STRING;
      if (c.repr >= TFGPU_R_INT8 && c.repr <= TFGPU_R_UINT64) { if (q) s.put('"'); emit_int(s, c, b); if (q) s.put('"'); return true; }
      if (c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) { if (q) s.put('"'); emit_float_f(s, c, b); if (q) s.put('"'); return true; }
      if (c.repr == TFGPU_R_JSONNUM) {
        if (q) s.put('"');
        const int o = jsonnum_overflow(vp, vn);
        if (o > 0) put_lit(s, "inf"); else if (o < 0) put_lit(s, "-inf"); else put_bytes(s, vp, vn);
        if (q) s.put('"');
        return true;
      }
      break;
    }
    case TFGPU_T_BYTES: case TFGPU_T_UTF8:
      if (c.repr == TFGPU_R_STRING) { emit_ch_quoted(s, vp, vn); return true; }
      if (c.repr == TFGPU_R_BYTES) {
        if (fl & TFGPU_CH_ARRAY) { s.put('['); for (uint32_t i = 0; i < vn; i++) { if (i) s.put(','); emit_u64(s, vp[i]); } s.put(']'); }
        else emit_ch_quoted(s, vp, vn);
        return true;
      }
      break;
    case TFGPU_T_BOOLEAN:
      if (c.repr == TFGPU_R_BOOL) { put_lit(s, (uint8_t)b.v ? "true" : "false"); return true; }
      break;
    case TFGPU_T_DATE: case TFGPU_T_DATETIME: case TFGPU_T_TIMESTAMP:
      if (c.repr == TFGPU_R_TIME) {  // marshalTime :65-80
        const int64_t sec = (int64_t)b.v; const int32_t ns = b.ns;
        if (fl & TFGPU_CH_STRING) { s.put('"'); emit_time_string(s, sec, ns); s.put('"'); }
        else if (fl & TFGPU_CH_DATETIME64) {
          int64_t full = sec * 1000000000LL + ns;
          if (sc.prec > 0 && sc.prec < 9) full = full / pow10_i64(9 - (int)sc.prec);
          emit_i64(s, full);
        } else if (fl & TFGPU_CH_DATE) { s.put('"'); emit_date(s, sec); s.put('"'); }
        else emit_i64(s, sec);
        return true;
      }
      break;
    default: break;
  }
  // marshalGeneric :318-359
  if (c.repr == TFGPU_R_STRING) { emit_ch_quoted(s, vp, vn); return true; }
  if (c.repr == TFGPU_R_BYTES) {
    if (fl & TFGPU_CH_ARRAY) { s.put('['); for (uint32_t i = 0; i < vn; i++) { if (i) s.put(','); emit_u64(s, vp[i]); } s.put(']'); }
    else emit_ch_quoted(s, vp, vn);
    return true;
  }
  // json.Marshal(v) of the remaining Go types, double-marshalled when the target is a string
  const bool dbl = c.dtype != TFGPU_T_ANY || any_as_string || (fl & TFGPU_CH_STRING);
  if (c.repr >= TFGPU_R_INT8 && c.repr <= TFGPU_R_UINT64) { if (dbl) s.put('"'); emit_int(s, c, b); if (dbl) s.put('"'); return true; }
  if (c.repr == TFGPU_R_BOOL) { if (dbl) s.put('"'); put_lit(s, (uint8_t)b.v ? "true" : "false"); if (dbl) s.put('"'); return true; }
  if (c.repr == TFGPU_R_DURATION) { if (dbl) s.put('"'); emit_i64(s, (int64_t)b.v); if (dbl) s.put('"'); return true; }
  if (c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) { if (dbl) s.put('"'); emit_float_json(s, c, b); if (dbl) s.put('"'); return true; }
  if (c.repr == TFGPU_R_JSONNUM || c.repr == TFGPU_R_JSON) {
    if (c.repr == TFGPU_R_JSONNUM && vn == 0) { if (dbl) s.put('"'); s.put('0'); if (dbl) s.put('"'); return true; }  // json.Number("") encodes as 0
    if (vn == 4 && vp[0] == 'n' && vp[1] == 'u' && vp[2] == 'l' && vp[3] == 'l') return false;  // value is null: skip the column
    if (dbl) emit_json_string(s, vp, vn, true); else put_bytes(s, vp, vn);
    return true;
  }
  return true;  // unreachable: the host rejects the remaining (dtype, repr) pairs
}

// ---- one value, encoding/json (jsonSerializer) ----
template <class S> __device__ __forceinline__ void emit_json_value(S &s, const SCol &sc, const CellBits &b, int any_as_string) {
  const DCol &c = sc.c;
  // the batch / stream serializers strictify first (serializer/strictify.go:24-36): a Go string under "string" becomes []byte
  // (castx.ToByteSliceE), a Go float under "double" becomes json.Number(FormatFloat(v, 'f', -1, bits)) (caste.go:36-49, 59-62)
  if (c.dtype == TFGPU_T_BYTES && c.repr == TFGPU_R_STRING && b.valid) {
    uint32_t vn; const uint8_t *vp = cell_text(c, b, vn);
    s.put('"'); emit_base64(s, vp, vn); s.put('"');
    return;
  }
  if (c.dtype == TFGPU_T_FLOAT64 && (c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) && b.valid) { emit_float_f(s, c, b); return; }
  emit_json_cell(s, c, b, any_as_string, false);  // SetEscapeHTML(false), json_format.go
}

// bytes a Marshaler returned, as json.Marshal's compact(escapeHTML) leaves them: <, >, & and U+2028/9 become \uXXXX
template <class S> __device__ __forceinline__ void put_html_compact(S &s, const uint8_t *p, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t c = p[i];
    if (c == '<' || c == '>' || c == '&') { put_lit(s, "\\u00"); s.put(hexc(c >> 4)); s.put(hexc(c & 15)); }
    else if (c == 0xE2 && i + 2 < n && p[i + 1] == 0x80 && (p[i + 2] & 0xFE) == 0xA8) { put_lit(s, "\\u202"); s.put(hexc(p[i + 2] & 0xF)); i += 2; }
    else s.put(c);
  }
}
// ---- one field, encoding/csv over toCsvValue ----
__device__ bool csv_needs_quotes(const uint8_t *p, uint32_t n) {  // Writer.fieldNeedsQuotes, Comma ','
  if (n == 0) return false;
  if (n == 2 && p[0] == '\\' && p[1] == '.') return true;
  for (uint32_t i = 0; i < n; i++) { const uint32_t c = p[i]; if (c == ',' || c == '"' || c == '\r' || c == '\n') return true; }
  const uint32_t c = p[0];
  if (c == ' ' || (c >= 9 && c <= 13)) return true;
  if (c == 0xC2 && n >= 2 && (p[1] == 0x85 || p[1] == 0xA0)) return true;
  if (n >= 3 && (c == 0xE1 || c == 0xE2 || c == 0xE3)) {
    const uint32_t d = p[1], e = p[2];
    if (c == 0xE1 && d == 0x9A && e == 0x80) return true;
    if (c == 0xE3 && d == 0x80 && e == 0x80) return true;
    if (c == 0xE2 && ((d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) || (d == 0x81 && e == 0x9F))) return true;
  }
  return false;
}
template <class S> __device__ __forceinline__ void emit_csv_field(S &s, const SCol &sc, const CellBits &b) {
  const DCol &c = sc.c;
  if (!b.valid) return;  // nil → ""
  uint32_t vn; const uint8_t *vp = cell_text(c, b, vn);
  uint8_t t[64];
  if (c.dtype == TFGPU_T_BYTES) { emit_base64(s, vp, vn); return; }  // []byte, or a Go string strictified to it (host-checked)
  if (c.dtype == TFGPU_T_ANY && c.repr == TFGPU_R_JSON) {  // json.Marshal(value): unlike the JSON serializer's Encoder it escapes HTML
    if (csv_needs_quotes(vp, vn)) { s.put('"'); CsvQuoteSink<S> q{s}; put_html_compact(q, vp, vn); s.put('"'); }
    else put_html_compact(s, vp, vn);
    return;
  }
  if (c.dtype == TFGPU_T_ANY && c.repr == TFGPU_R_STRING) {           // json.Marshal(string): always holds '"'
    s.put('"'); CsvQuoteSink<S> q{s}; emit_json_string(q, vp, vn, true); s.put('"');
    return;
  }
  switch (c.repr) {
    case TFGPU_R_STRING: case TFGPU_R_BYTES: case TFGPU_R_JSONNUM: case TFGPU_R_JSON:
      if (csv_needs_quotes(vp, vn)) { s.put('"'); CsvQuoteSink<S> q{s}; put_bytes(q, vp, vn); s.put('"'); }
      else put_bytes(s, vp, vn);
      return;
    case TFGPU_R_BOOL: put_lit(s, (uint8_t)b.v ? "true" : "false"); return;
    case TFGPU_R_TIME: emit_time_string(s, (int64_t)b.v, b.ns); return;  // fmt.Stringer
    case TFGPU_R_DURATION: { int n = dev::fmt_duration(t, (int64_t)b.v); emit_small(s, t, n); return; }
    case TFGPU_R_FLOAT32: case TFGPU_R_FLOAT64: emit_float_f(s, c, b); return;
    default: emit_int(s, c, b);
  }
}

// ---- queue formats: ChangeItem.MarshalJSON (change_item.go:568-616) as json.Marshal compacts it, and the queue JSON rows ----
__device__ __forceinline__ bool q_applies(const SerParams &p, uint32_t apply, int64_t r, const uint8_t *absent = nullptr) {
  switch (apply) {
    case QA_LISTED: if (absent && ((absent[r >> 3] >> (r & 7)) & 1u)) return false;
      return !p.q.m_form || p.q.m_form[p.q.src_row ? p.q.src_row[r] : r] == 0;
    case QA_ANY_LISTED: if (p.q.first_listed[r] < 0) return false;
      return !p.q.m_form || p.q.m_form[p.q.src_row ? p.q.src_row[r] : r] == 0;
    case QA_NAMES: return !p.q.m_form || p.q.m_form[p.q.src_row ? p.q.src_row[r] : r] == 0;
    case QA_OLD: return p.q.has_old && (!p.q.old_present || ((p.q.old_present[r >> 3] >> (r & 7)) & 1));
    case QA_ROW_EVENT: return !p.q.kind || p.q.kind[r] <= TFGPU_K_DELETE;
    default: return true;
  }
}
template <class S> __device__ __forceinline__ void emit_meta_string(S &s, const uint32_t *off, const uint8_t *data, int64_t k) {
  if (!off) { s.put('"'); s.put('"'); return; }
  emit_json_string(s, data + off[k], off[k + 1] - off[k], true);
}
template <class S> __device__ __forceinline__ void emit_native_header(S &s, const SerParams &p, int64_t r) {
  const QueueParams &q = p.q;
  const int64_t k = q.src_row ? q.src_row[r] : r;
  uint8_t t[24]; int n;
  put_lit(s, "{\"id\":"); emit_u64(s, q.m_id ? q.m_id[k] : 0u);
  put_lit(s, ",\"nextlsn\":"); emit_u64(s, q.m_lsn ? q.m_lsn[k] : 0ull);
  put_lit(s, ",\"commitTime\":"); emit_u64(s, q.m_commit ? q.m_commit[k] : 0ull);
  put_lit(s, ",\"txPosition\":"); emit_i64(s, q.m_counter ? q.m_counter[k] : 0ll);
  const uint32_t kd = q.kind ? q.kind[r] : (uint32_t)TFGPU_K_INSERT;
  put_lit(s, ",\"kind\":\""); put_lit(s, kd == TFGPU_K_INSERT ? "insert" : kd == TFGPU_K_UPDATE ? "update" : "delete"); s.put('"');
  put_bytes(s, p.blob + q.hdr_off, q.hdr_len);
  if (q.gpart) {  // PartID of the row's group, already a JSON string
    int g = 0; while (g + 1 < q.ngroups && r >= q.gstart[g + 1]) g++;
    put_bytes(s, p.blob + q.gpart[g], q.gpart[g + 1] - q.gpart[g]);
  } else { s.put('"'); if (q.part_id) emit_u64(s, q.part_id[r]); s.put('"'); }
  const uint32_t form = q.m_form ? q.m_form[k] : 0u;
  if (form == 1) put_lit(s, ",\"columnnames\":null"); else if (form != 0) put_lit(s, ",\"columnnames\":[]");
}
template <class S> __device__ __forceinline__ void emit_native_trailer(S &s, const SerParams &p, int64_t r) {
  const QueueParams &q = p.q;
  const int64_t k = q.src_row ? q.src_row[r] : r;
  if (q_applies(p, QA_OLD, r)) s.put(']');
  s.put('}');
  put_lit(s, ",\"tx_id\":"); emit_meta_string(s, q.m_tx_off, q.m_tx, k);
  put_lit(s, ",\"query\":"); emit_meta_string(s, q.m_q_off, q.m_q, k);
  s.put('}');
}
template <class S> __device__ __forceinline__ bool emit_queue_cell(S &s, const SerParams &p, const SCol &sc, int64_t r) {
  if (!q_applies(p, sc.apply, r, sc.absent)) return false;
  switch (sc.kind) {
    case QC_HEADER: emit_native_header(s, p, r); return true;
    case QC_TRAILER: emit_native_trailer(s, p, r); return true;
    case QC_CONST: put_bytes(s, p.blob + sc.pre_off, sc.pre_len); return true;
    default: break;
  }
  const uint32_t lead = (sc.apply == QA_LISTED && p.q.first_listed[r] == (int32_t)sc.prec) ? 1u : 0u;  // the row's first listed column: no comma in front
  put_bytes(s, p.blob + sc.pre_off + lead, sc.pre_len - lead);
  if (sc.kind == QC_NAME) return true;
  if (p.q.qformat == TFGPU_QFMT_NATIVE) {  // encoding/json with escapeHTML on; a pre-marshalled `any` passes through compact()
    const DCol &c = sc.c;
    if (c.repr == TFGPU_R_JSON && is_valid(c, r)) put_html_compact(s, c.data + c.offsets[r], c.offsets[r + 1] - c.offsets[r]);
    else emit_json_cell(s, c, r, 0, true);
  } else emit_json_cell(s, sc.c, r, 0, false);  // serializeQueueItemToJSON: AnyAsString off, SetEscapeHTML(false)
  return true;
}

template <class S> __device__ __forceinline__ bool emit_cell(S &s, const SerParams &p, const SCol &sc, int64_t r) {
  if (p.q.qformat) return emit_queue_cell(s, p, sc, r);
  switch (p.format) {
    case TFGPU_FMT_CH_JSON_EACH_ROW: {
      if (!is_valid(sc.c, r)) return false;  // nil values are omitted (marshal.go:100-102)
      // key, value, then one byte for ',' or '}' written by the caller
      CountSink probe;  // a column whose value marshals to null is dropped together with its key
      (void)probe;
      put_bytes(s, p.blob + sc.pre_off, sc.pre_len);
      return emit_ch_value(s, sc, load_cell(sc.c, r), p.any_as_string);
    }
    case TFGPU_FMT_JSON:
      put_bytes(s, p.blob + sc.pre_off, sc.pre_len);
      emit_json_value(s, sc, load_cell(sc.c, r), p.any_as_string);
      return true;
    default:
      put_bytes(s, p.blob + sc.pre_off, sc.pre_len);
      emit_csv_field(s, sc, load_cell(sc.c, r));
      return true;
  }
}

// (1) cell lengths; item = column * nrows + row
// (grid: x = 256-row blocks, y = column — the column is a scalar, so the descriptor loads are scalar loads and the switches on
//  representation / DataType are scalar branches: with item = column * nrows + row every lane loaded its own copy of the descriptor)
__global__ void __launch_bounds__(256) ser_cell_len(SerParams p) {
  const int32_t ci = (int32_t)blockIdx.y; const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  const int64_t it = (int64_t)ci * p.nrows + r;
  if (p.q.qformat && p.cols[ci].kind == QC_CONST) { p.cell[it] = q_applies(p, p.cols[ci].apply, r) ? p.cols[ci].pre_len : 0u; return; }
  CountSink s;
  if (p.ablate == 3 && (p.cols[ci].c.repr == TFGPU_R_STRING || p.cols[ci].c.repr == TFGPU_R_BYTES)) { p.cell[it] = 8; return; }
  const bool present = emit_cell(s, p, p.cols[ci], r);
  uint32_t n = present ? s.n : 0;
  if (p.format == TFGPU_FMT_CH_JSON_EACH_ROW && present) n += 1;  // its ',' (or the closing '}' for the last one)
  p.cell[it] = n;
}
// (2) per row: lengths → offsets inside the row, row length, last emitted column
__global__ void __launch_bounds__(256) ser_row_layout(SerParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  uint32_t off = (p.format == TFGPU_FMT_CSV) ? 0u : 1u;  // '{'
  if (p.q.qformat) off = p.q.qformat == TFGPU_QFMT_NATIVE ? 1u : 0u;  // '[' or ',' in front of every native element
  int32_t last = -1;
  // eight lengths are requested before the first offset is stored: the loop is a lane's walk over its row's cells, and with a
  // load and a store to the same array in every step the compiler has to wait for each load in turn
  constexpr int U = 8;
  int32_t c = 0;
  if (p.q.col_order) {  // the row's own column order (rare: a collapsed TOAST batch): the name and value cells are walked in it
    const uint16_t *ord = p.q.col_order + r * p.q.ord_n;
    for (; c < p.ncols; c++) {
      int32_t ci = c;
      if (c >= p.q.ord_name0 && c < p.q.ord_name0 + p.q.ord_n) ci = p.q.ord_name0 + ord[c - p.q.ord_name0];
      else if (c >= p.q.ord_val0 && c < p.q.ord_val0 + p.q.ord_n) ci = p.q.ord_val0 + ord[c - p.q.ord_val0];
      const uint32_t n = p.cell[(int64_t)ci * p.nrows + r];
      p.cell[(int64_t)ci * p.nrows + r] = off;
      off += n;
    }
    p.row_len[r] = off;
    return;
  }
  for (; c + U <= p.ncols; c += U) {
    uint32_t n[U];
#pragma unroll
    for (int q = 0; q < U; q++) n[q] = p.cell[(int64_t)(c + q) * p.nrows + r];
#pragma unroll
    for (int q = 0; q < U; q++) { p.cell[(int64_t)(c + q) * p.nrows + r] = off; if (n[q]) last = c + q; off += n[q]; }
  }
  for (; c < p.ncols; c++) {
    const uint32_t n = p.cell[(int64_t)c * p.nrows + r];
    p.cell[(int64_t)c * p.nrows + r] = off;
    if (n) last = c;
    off += n;
  }
  uint32_t tail;
  if (p.q.qformat) { p.row_len[r] = off; return; }  // element length (+ lead byte); the message frame is added once the cut plan is known
  if (p.format == TFGPU_FMT_CH_JSON_EACH_ROW) tail = last < 0 ? 2u : 1u;                       // "}\n" or "\n" (the '}' replaced a ',')
  else if (p.format == TFGPU_FMT_JSON) tail = 1u + ((p.closing_newline || r + 1 < p.nrows) ? 1u : 0u);  // '}' + newline / separator
  else tail = 1u;                                                                                // '\n'
  p.row_len[r] = off + tail;
  if (p.last_present) p.last_present[r] = last;
  atomicAdd(p.total64, (unsigned long long)(off + tail));
}
// (4) cells at their final position
__global__ void __launch_bounds__(256) ser_cell_write(SerParams p) {
  const int32_t ci = (int32_t)blockIdx.y; const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  const int64_t it = (int64_t)ci * p.nrows + r;
  const SCol &sc = p.cols[ci];
  if (p.q.qformat && ((sc.kind == QC_CONST && sc.pre_len > QCONST_INLINE) || !q_applies(p, sc.apply, r, sc.absent))) return;
  if (p.format == TFGPU_FMT_CH_JSON_EACH_ROW && !is_valid(sc.c, r)) return;
  uint64_t base = (uint64_t)p.row_len[r] + p.cell[it];
  if (p.ablate == 4) base = (uint64_t)(it % (int64_t)(p.row_len[p.nrows] / 24)) * 24;  // measurement: every cell at a lane-contiguous 24-byte slot
  if (p.ablate && p.ablate != 4) { const bool text = sc.c.repr == TFGPU_R_STRING || sc.c.repr == TFGPU_R_BYTES; if ((p.ablate == 1 || p.ablate == 3) == text) return; }
  if (p.ablate == 4 && (sc.c.repr == TFGPU_R_STRING || sc.c.repr == TFGPU_R_BYTES)) return;
  if (p.format == TFGPU_FMT_CH_JSON_EACH_ROW) {
    // a null-marshalling value must leave no bytes: probe first (rare: only `any` columns can do it)
    if (sc.c.repr == TFGPU_R_JSON || sc.c.repr == TFGPU_R_JSONNUM) { CountSink probe; if (!emit_cell(probe, p, sc, r)) return; }
    WriteSink w{p.out + base};
    emit_cell(w, p, sc, r);
    w.put((p.last_present[r] == ci) ? '}' : ',');
    w.flush();
    return;
  }
  WriteSink w{p.out + base};
  emit_cell(w, p, sc, r);
  w.flush();
}
// row frame: '{' and the tail
__global__ void __launch_bounds__(256) ser_row_frame(SerParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  uint8_t *row = p.out + p.row_len[r];
  const uint32_t len = p.row_len[r + 1] - p.row_len[r];
  if (p.format == TFGPU_FMT_CSV) { row[len - 1] = '\n'; return; }
  row[0] = '{';
  if (p.format == TFGPU_FMT_CH_JSON_EACH_ROW) { if (p.last_present[r] < 0) row[len - 2] = '}'; row[len - 1] = '\n'; return; }
  const bool nl = p.closing_newline || r + 1 < p.nrows;
  row[len - 1 - (nl ? 1 : 0)] = '}';
  if (nl) row[len - 1] = '\n';
}

// ---- the text formats' passes: chunk walks ---------------------------------------------------------------------------
// A row is cut into CHUNKS of consecutive columns (the host picks the cuts so that a chunk is ~130 bytes of text per row).
// One wave = 64 consecutive rows x one chunk, lane = row: the lane WALKS its chunk's cells in column order with one sink —
// the column descriptor is scalar, every lane of the wave runs the same column's code, value loads are coalesced — so
//   * the length pass stores one number per (row, chunk) instead of one per cell (hits: 19 instead of 105 per row),
//   * the write pass needs no per-cell offsets: a cell starts where the lane's previous cell ended, the sink is flushed once
//     per chunk, and the lane's bytes go to ITS segment of the wave's LDS image with plain 8-byte stores (segments start on
//     8-byte boundaries of the image: no sharing, no atomics, no zeroing), which then leaves as aligned 8-byte words.
// The JSONEachRow separator moves in FRONT of the cell ('{' before the row's first present cell, ',' before the others), so
// the prefix is a per-column constant but for its first byte and nobody has to know which column is a row's last.
#ifndef TF_SER_WAVE_IMG
#define TF_SER_WAVE_IMG 9216
#endif
constexpr uint32_t WAVE_IMG_BYTES = TF_SER_WAVE_IMG;   // per wave: 4 waves x 9 KiB = 36 KiB per workgroup, four workgroups per CU
TF_DYNAMIC_LDS(uint32_t, img);               // named at file scope: the sink's stores are ds_write (through a pointer member: flat)

struct LdsSink {
  uint32_t pos;                 // byte offset in img, a multiple of 8
  uint64_t acc = 0; uint32_t n = 0;
  __device__ __forceinline__ void store() { img[pos >> 2] = (uint32_t)acc; img[(pos >> 2) + 1] = (uint32_t)(acc >> 32); pos += 8; }
  __device__ __forceinline__ void put(uint32_t c) {
    acc |= (uint64_t)(c & 0xFFu) << (8 * n);
    if (++n == 8) { store(); acc = 0; n = 0; }
  }
  __device__ __forceinline__ void put_word(uint64_t w, uint32_t k) {
    acc |= w << (8 * n);
    const uint32_t t = n + k;
    if (t >= 8) { store(); acc = (w >> 1) >> (63 - 8 * n); n = t - 8; } else n = t;  // (w >> 1) >> 63 == 0 when nothing was held
  }
  __device__ __forceinline__ void flush() { if (n) store(); }
};

struct ChunkPlan {
  const int32_t *cut;      // [nchunks + 1] first column of every chunk
  int32_t nchunks;
  uint32_t *chunk;         // [nchunks][nrows]: bytes of the chunk's cells, then (ser_chunk_layout) their offset inside the row
  const uint32_t *recs;    // [ncols][REC_WORDS]: the column records (below)
  int32_t direct;          // measurement: 1 = every tile goes straight to HBM
  int32_t order;           // tile order: 0 = a row group's chunks are neighbours, 1 = a chunk's row groups are
  int64_t ngroups;         // row groups of 64
};
// workgroups of four tiles; the XCD-aware order needs a whole number of workgroups per XCD (the tiles past the end do nothing)
static inline unsigned ser_grid(const ChunkPlan &cp) {
  if (cp.order == 3) return (unsigned)(cp.ngroups * ((cp.nchunks + 3) >> 2));
  const int64_t nb = (cp.ngroups * cp.nchunks + 3) >> 2;
  return (unsigned)(cp.order == 2 ? ((nb + 7) >> 3) << 3 : nb);
}
__device__ __forceinline__ void tile_of(const ChunkPlan &cp, int64_t tile, int32_t &k, int64_t &r0) {
  if (cp.order == 3) {
    // a workgroup's four waves take four NEIGHBOURING chunks of one row group — the seams between their segments are written
    // within microseconds by one CU and leave its L2 as whole lines — while consecutive workgroups take consecutive row groups of
    // the same four chunks: the reads stay within those chunks' columns, as in order 1
    const int64_t b = tile >> 2, quad = b / cp.ngroups;
    k = (int32_t)(quad * 4 + (tile & 3)); r0 = (b - quad * cp.ngroups) * 64;
  }
  else if (cp.order == 2) {
    // a row group's chunks are neighbours AND on one XCD: workgroups are dealt to the eight XCDs round-robin (blockIdx & 7), each
    // XCD has its own L2 — so XCD x takes the x-th eighth of the row groups, and the partial cache lines at the seams between a
    // row's chunks (written by different waves within microseconds) meet in ONE L2 and leave it as whole lines
    const int64_t b = tile >> 2, nb = ((cp.ngroups * cp.nchunks + 3) >> 2), per = (nb + 7) >> 3;
    const int64_t vt = (((b & 7) * per + (b >> 3)) << 2) | (tile & 3);
    k = (int32_t)(vt % cp.nchunks); r0 = (vt / cp.nchunks) * 64;
  }
  else if (cp.order) { k = (int32_t)(tile / cp.ngroups); r0 = (tile - (int64_t)k * cp.ngroups) * 64; }
  else { k = (int32_t)(tile % cp.nchunks); r0 = (tile / cp.nchunks) * 64; }
  k = __builtin_amdgcn_readfirstlane(k);
}
// A column's record: its descriptor and the first bytes of its prefix as 32 dwords.  The wave loads a record with ONE coalesced
// load (lane = dword) and the fields come out of that register by v_readlane — scalars, like s_load results, but without a
// scalar-memory round trip per field group in front of every cell (measured: ~50 s_load + s_waitcnt per 5-cell tile, two thirds
// of the wave time waiting).
constexpr int REC_WORDS = 32, REC_PREFIX_WORD = 16, REC_PREFIX_BYTES = 32;
enum { RW_VALUES = 0, RW_OFFSETS = 2, RW_DATA = 4, RW_NANOS = 6, RW_VALIDITY = 8, RW_REPR = 10, RW_DTYPE = 11, RW_PRE_OFF = 12, RW_PRE_LEN = 13, RW_CH_FLAGS = 14, RW_PREC = 15 };
__device__ __forceinline__ uint32_t rec_word(uint32_t rec, int k) { return (uint32_t)__builtin_amdgcn_readlane((int)rec, k); }
__device__ __forceinline__ uint64_t rec_word64(uint32_t rec, int k) { return (uint64_t)rec_word(rec, k) | ((uint64_t)rec_word(rec, k + 1) << 32); }
__device__ __forceinline__ DCol rec_dcol(uint32_t rec) {
  DCol c;
  c.values = reinterpret_cast<const void *>(rec_word64(rec, RW_VALUES));
  c.offsets = reinterpret_cast<const uint32_t *>(rec_word64(rec, RW_OFFSETS));
  c.data = reinterpret_cast<const uint8_t *>(rec_word64(rec, RW_DATA));
  c.nanos = reinterpret_cast<const int32_t *>(rec_word64(rec, RW_NANOS));
  c.validity = reinterpret_cast<const uint8_t *>(rec_word64(rec, RW_VALIDITY));
  c.repr = (int32_t)rec_word(rec, RW_REPR); c.dtype = (int32_t)rec_word(rec, RW_DTYPE);
  return c;
}

// the cells [c0, c1) of row r, in order.  `first`: JSONEachRow only — the row has emitted nothing yet.
// Groups of WALK_AHEAD cells: the group's records, then every load of the group (validity byte, value or offsets, nanoseconds)
// are issued before its first cell is formatted, so a lane waits for memory twice per group instead of three or four times per
// cell.  ALL 64 lanes of the wave must run the walk (the records live across its lanes).
// four waves per SIMD for the length pass: it is latency-bound (three waves: 1.51 ms, four: 1.26 ms on configs[3]; five spill too much)
#ifndef TF_SER_LEN_ATTR
#define TF_SER_LEN_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
// … and for the write pass (9 KiB of image per wave: 16 waves x 9 KiB per CU; measured 3.04 -> 2.8 ms with a lookahead of four in round 3; on round 6's
// kernels and 144-byte chunks a lookahead of 1 / 2 / 3 / 4 / 6 / 8 gives 2.71 / 2.39 / 2.42 / 2.45 / 2.50 / 2.56 ms: two, profiles/r28e_*, r28f_*)
#ifndef TF_SER_WRITE_ATTR
#define TF_SER_WRITE_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
#ifndef TF_SER_LEN_AHEAD
#define TF_SER_LEN_AHEAD 8
#endif
#ifndef TF_SER_WRITE_AHEAD
#define TF_SER_WRITE_AHEAD 2
#endif
constexpr int LEN_AHEAD = TF_SER_LEN_AHEAD, WRITE_AHEAD = TF_SER_WRITE_AHEAD;
template <int FMT, class S> __device__ __forceinline__ void walk_cell(S &s, const SerParams &p, uint32_t rec, const CellBits &b, bool &first) {
  // every v_readlane up here, in uniform control flow
  SCol sc;
  sc.c = rec_dcol(rec);
  sc.pre_off = rec_word(rec, RW_PRE_OFF); sc.pre_len = rec_word(rec, RW_PRE_LEN); sc.ch_flags = rec_word(rec, RW_CH_FLAGS); sc.prec = rec_word(rec, RW_PREC);
  sc.kind = 0; sc.apply = 0;
  uint64_t pw[REC_PREFIX_BYTES / 8];
#pragma unroll
  for (int w = 0; w < REC_PREFIX_BYTES / 8; w++) pw[w] = rec_word64(rec, REC_PREFIX_WORD + 2 * w);
  if (FMT == TFGPU_FMT_CH_JSON_EACH_ROW) {
    if (!b.valid) return;  // nil values are omitted (marshal.go:100-102)
    if (p.ablate && (p.ablate == 1) == (sc.c.offsets != nullptr)) return;  // measurement only: 1 = no text cells, 2 = only text cells
    // a column whose value marshals to null is dropped together with its key (only `any` columns can do it)
    if (sc.c.repr == TFGPU_R_JSON || sc.c.repr == TFGPU_R_JSONNUM) { CountSink probe; if (!emit_ch_value(probe, sc, b, p.any_as_string)) return; }
    pw[0] |= first ? '{' : ',';  // the record's prefix starts with a zero byte for it
    first = false;
  }
  const uint32_t n = sc.pre_len;
#pragma unroll
  for (int w = 0; w < REC_PREFIX_BYTES / 8; w++) if (n > 8u * w) sink_word(s, pw[w], min(n - 8u * w, 8u), 0);
  if (n > (uint32_t)REC_PREFIX_BYTES) put_bytes(s, p.blob + sc.pre_off + REC_PREFIX_BYTES, n - REC_PREFIX_BYTES);
  if (FMT == TFGPU_FMT_CH_JSON_EACH_ROW) emit_ch_value(s, sc, b, p.any_as_string);
  else if (FMT == TFGPU_FMT_JSON) { if (p.q.qformat) emit_json_cell(s, sc.c, b, 0, false); else emit_json_value(s, sc, b, p.any_as_string); }  // (queue JSON: serializeQueueItemToJSON does not strictify)
  else emit_csv_field(s, sc, b);
}
template <int FMT, int WALK_AHEAD, class S> __device__ __forceinline__ void walk_chunk(S &s, const SerParams &p, const ChunkPlan &cp, int32_t c0, int32_t c1, int64_t r, bool &first) {
  if (FMT == TFGPU_FMT_JSON && c0 == 0) s.put('{');
  const int lane = threadIdx.x & 63;
  for (int32_t cg = c0; cg < c1; cg += WALK_AHEAD) {  // c0, c1 are wave-uniform (the callers say so)
    uint32_t rec[WALK_AHEAD];
    CellBits b[WALK_AHEAD];
#pragma unroll
    for (int j = 0; j < WALK_AHEAD; j++) rec[j] = cp.recs[(int64_t)min(cg + j, c1 - 1) * REC_WORDS + (lane & (REC_WORDS - 1))];
#pragma unroll
    for (int j = 0; j < WALK_AHEAD; j++) { b[j].v = 0; b[j].ns = 0; b[j].valid = false; if (cg + j < c1) b[j] = load_cell(rec_dcol(rec[j]), r); }
    // ONE copy of the formatting code: the group is a register queue, shifted down by one after every cell
    const int32_t cn = min(c1 - cg, WALK_AHEAD);
    for (int32_t j = 0; j < cn; j++) {
      walk_cell<FMT>(s, p, rec[0], b[0], first);
#pragma unroll
      for (int q = 0; q + 1 < WALK_AHEAD; q++) { b[q] = b[q + 1]; rec[q] = rec[q + 1]; }
    }

  }
}
// what closes a row, written by the lane of the row's last chunk
template <int FMT, class S> __device__ __forceinline__ void row_tail(S &s, const SerParams &p, int64_t r, bool first) {
  if (FMT == TFGPU_FMT_CSV) { s.put('\n'); return; }
  if (FMT == TFGPU_FMT_CH_JSON_EACH_ROW) { if (first) s.put('{'); s.put('}'); s.put('\n'); return; }
  s.put('}');
  if (p.q.qformat) { if (p.q.msg_flags && !(p.q.msg_flags[r] & 2)) s.put('\n'); }  // queue JSON (BatchJSON): "\n" after every element but the last of its message; none while the cut plan is unknown (the length passes)
  else if (p.closing_newline || r + 1 < p.nrows) s.put('\n');
}

// (1) bytes per (row, chunk)
template <int FMT>
__global__ void __launch_bounds__(256) TF_SER_LEN_ATTR ser_chunk_len(SerParams p, ChunkPlan cp) {
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int32_t k; int64_t r0;
  tile_of(cp, tile, k, r0);
  if (k >= cp.nchunks || r0 >= p.nrows) return;
  const int32_t c0 = __builtin_amdgcn_readfirstlane(cp.cut[k]), c1 = __builtin_amdgcn_readfirstlane(cp.cut[k + 1]);
  const int64_t rl = r0 + (threadIdx.x & 63), r = min(rl, p.nrows - 1);  // lanes past the end repeat the last row: the walk needs the whole wave
  CountSink s;
  bool first = false;
  walk_chunk<FMT, LEN_AHEAD>(s, p, cp, c0, c1, r, first);
  if (rl < p.nrows) cp.chunk[(int64_t)k * p.nrows + r] = s.n;
}
// (2) per row: chunk lengths → offsets inside the row, row length
template <int FMT>
__global__ void __launch_bounds__(256) ser_chunk_layout(SerParams p, ChunkPlan cp) {
  const int64_t rl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = rl < p.nrows;                    // (lanes past the end stay for the wave's sum: they redo the last row and add nothing)
  const int64_t r = live ? rl : p.nrows - 1;
  if (p.nrows <= 0) return;
  uint32_t off = 0;
  constexpr int U = 8;  // the lengths are requested before the first offset is stored (a load and a store to one array per step)
  int32_t k = 0;
  for (; k + U <= cp.nchunks; k += U) {
    uint32_t n[U];
#pragma unroll
    for (int q = 0; q < U; q++) n[q] = cp.chunk[(int64_t)(k + q) * p.nrows + r];
#pragma unroll
    for (int q = 0; q < U; q++) { if (live) cp.chunk[(int64_t)(k + q) * p.nrows + r] = off; off += n[q]; }
  }
  for (; k < cp.nchunks; k++) { const uint32_t n = cp.chunk[(int64_t)k * p.nrows + r]; if (live) cp.chunk[(int64_t)k * p.nrows + r] = off; off += n; }
  CountSink t;
  row_tail<FMT>(t, p, r, off == 0);
  if (live) p.row_len[r] = off + t.n;
  // the 64-bit total (the offsets are 32-bit: the host refuses an output of 4 GiB and more): one atomic per wave, not per row —
  // a million atomics on one address were most of this kernel's 0.21 ms
  unsigned long long sum = live ? (unsigned long long)(off + t.n) : 0ull;
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t lo = __shfl_down((uint32_t)sum, d, 64), hi = __shfl_down((uint32_t)(sum >> 32), d, 64);
    if ((int)(threadIdx.x & 63) + d < 64) sum += ((unsigned long long)hi << 32) | lo;
  }
  // … and one per workgroup: the waves' sums meet in LDS first (16 384 atomics on one word were still ~0.15 ms of a 2^20-row batch)
  __shared__ unsigned long long wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (unsigned w = 0; w < (blockDim.x + 63) / 64; w++) t += wsum[w];
    if (t) atomicAdd(p.total64, t);
  }
}
// (4) the text
template <int FMT>
__global__ void __launch_bounds__(256) TF_SER_WRITE_ATTR ser_chunk_write(SerParams p, ChunkPlan cp) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  int32_t k; int64_t r0;
  tile_of(cp, tile, k, r0);
  if (k >= cp.nchunks || r0 >= p.nrows) return;
  const bool live = r0 + lane < p.nrows;
  const int64_t r = min(r0 + lane, p.nrows - 1);  // lanes past the end repeat the last row, byte for byte and at the same place: the walk needs the whole wave
  const bool last_chunk = k + 1 == cp.nchunks;
  const uint64_t row_at = p.row_len[r];
  const uint32_t s0 = cp.chunk[(int64_t)k * p.nrows + r];
  const uint32_t len = (last_chunk ? p.row_len[r + 1] - (uint32_t)row_at : cp.chunk[(int64_t)(k + 1) * p.nrows + r]) - s0;
  uint8_t *dst = p.out + row_at + s0;
  const int32_t c0 = __builtin_amdgcn_readfirstlane(cp.cut[k]), c1 = __builtin_amdgcn_readfirstlane(cp.cut[k + 1]);
  bool first = s0 == 0;
  const uint32_t cap = live ? (len + 7u) & ~7u : 0u;
  const uint32_t incl = wave_scan_add(cap);
  const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
  if (cp.direct || total > WAVE_IMG_BYTES) {  // the tile outgrows the image (a long text cell): the same walk, straight to HBM
    WriteSink w{dst};
    walk_chunk<FMT, WRITE_AHEAD>(w, p, cp, c0, c1, r, first);
    if (last_chunk) row_tail<FMT>(w, p, r, first);
    w.flush();
    return;
  }
  const uint32_t base = wave * WAVE_IMG_BYTES;
  const uint32_t last_at = (uint32_t)__builtin_amdgcn_readlane(incl - cap, (int)min(p.nrows - 1 - r0, (int64_t)63));  // (the whole wave reads it)
  const uint32_t at = base + (live ? incl - cap : last_at);
  {
    LdsSink w{at};
    walk_chunk<FMT, WRITE_AHEAD>(w, p, cp, c0, c1, r, first);
    if (last_chunk) row_tail<FMT>(w, p, r, first);
    w.flush();
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the other lanes' image bytes
  __builtin_amdgcn_wave_barrier();
  // segments → HBM, four at a time (16 lanes each): bytes up to the destination's first 8-byte boundary, aligned words, the tail
  const int sub = lane & 15, grp = lane >> 4;
  const uint32_t dlo = (uint32_t)reinterpret_cast<uintptr_t>(dst), dhi = (uint32_t)(reinterpret_cast<uintptr_t>(dst) >> 32);
  auto img_byte = [&](uint32_t o) { return (uint8_t)(img[o >> 2] >> ((o & 3u) * 8)); };
  for (int it = 0; it < 16; it++) {
    const int src = it * 4 + grp;
    const uint32_t a = __shfl(at, src), n = __shfl(live ? len : 0u, src);
    uint8_t *d = reinterpret_cast<uint8_t *>(((uint64_t)__shfl(dhi, src) << 32) | __shfl(dlo, src));
    if (n == 0) continue;
    const uint32_t head = min((uint32_t)((8 - (reinterpret_cast<uintptr_t>(d) & 7)) & 7), n);
    if ((uint32_t)sub < head) d[sub] = img_byte(a + sub);
    const uint32_t nw = (n - head) >> 3, sh = head * 8;   // a is a multiple of 8: every word of the segment sits at the same shift
    for (uint32_t i = sub; i < nw; i += 16) {
      const uint32_t w = (a >> 2) + i * 2;
      const uint64_t lo = img[w] | ((uint64_t)img[w + 1] << 32);
      uint64_t v = lo;
      if (sh) { const uint64_t hi = img[w + 2] | ((uint64_t)img[w + 3] << 32); v = (lo >> sh) | (hi << (64 - sh)); }
      *reinterpret_cast<uint64_t *>(d + head + i * 8) = v;
    }
    const uint32_t done = head + nw * 8;
    if (sub < 8 && done + sub < n) d[done + sub] = img_byte(a + done + sub);
  }
}

// ---- the JSONEachRow length pass, lean form (round 6) -------------------------------------------------------------------------------
// ser_chunk_len runs the write pass's own walk over a counting sink: the column records by v_readlane, the whole emitter family inlined (128
// VGPRs: four waves a SIMD), every text cell's bytes read to count its escapes — 660 vector + 830 scalar instructions and three to five dependent
// memory round trips a (64 rows x 5 cells) tile, 58 % of a wave's life waiting (profiles/r20b_pmc_ser_chunk.txt).  What a LENGTH needs is less:
//   * integers: the digit count (the emitters' own CountSink path: the same code decides what the write pass will write);
//   * text: 2 quotes + the cell's bytes — plus its escapes, and whether a COLUMN holds any byte that escapes at all ('"', '\', < 0x20) is one
//     streaming pass over the column's bytes (ser_text_flags: all text columns in one launch).  A column without such a byte — nearly every column
//     of nearly every batch — costs two offsets a cell and no byte; one with them is counted cell by cell as before;
//   * dates / DateTime64 / epoch seconds: a constant or a digit count.
// Column descriptors are 32-byte records read through the scalar unit at a wave-uniform index, the chunk's cells are loaded eight columns at a
// time before the first is looked at, and the kernel needs a quarter of the registers: eight waves a SIMD hide what is left of the latency.
// Same tiles, same (row, chunk) length table as ser_chunk_len: ser_chunk_layout and ser_chunk_write do not know the difference.  Batches with a
// column outside the list (floats, `any`, []byte as arrays, times as strings) take ser_chunk_len as before.
enum : uint32_t { LK_INT = 1, LK_TEXT = 2, LK_TEXT_RAW = 3, LK_BOOL = 4, LK_TIME_DATE = 5, LK_TIME_SEC = 6, LK_TIME_DT64 = 7 };
struct LenCol {
  const void *values;        // fixed-width values, or the text column's offsets
  const uint8_t *validity;   // or null
  const uint8_t *data;       // text bytes (only read when the column's flag says it holds bytes that escape)
  uint32_t kind;             // LK_*
  uint32_t add;              // prefix bytes (with the separator's) + what frames the value (quotes)
  int32_t repr;              // LK_INT: the Go integer type; LK_TIME_DT64: the precision
  uint32_t flag;             // LK_TEXT: index into the text flags
};
__global__ void __launch_bounds__(256) ser_text_flags(const LenCol *cols, const int32_t *text_cols, int64_t nrows, uint32_t *flags) {
  const int32_t j = (int32_t)blockIdx.y;
  const LenCol c = cols[text_cols[j]];
  // the bytes the column's cells cover, from its own offsets (a sliced column starts above zero; the host's byte count only sizes the grid)
  const uint32_t *off = (const uint32_t *)c.values;
  const uint64_t n = off[nrows];
  bool hit = false;
  for (uint64_t i = off[0] + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (uint64_t)gridDim.x * blockDim.x * 16) {
    if (i + 16 <= n) {
      const uint64_t a = load8(c.data + i), b = load8(c.data + i + 8);
      hit |= (swar_has_less(a, 0x20) | swar_has(a, '"') | swar_has(a, '\\') | swar_has_less(b, 0x20) | swar_has(b, '"') | swar_has(b, '\\')) != 0;
    } else for (uint64_t k = i; k < n; k++) { const uint32_t ch = c.data[k]; hit |= ch < 0x20 || ch == '"' || ch == '\\'; }
  }
  if (__any(hit) && (threadIdx.x & 63) == 0) flags[j] = 1u;
}
__device__ __forceinline__ uint32_t len_of_value(const LenCol &c, uint64_t v, int32_t ns, uint32_t t0, uint32_t t1, const uint32_t *flags) {
  switch (c.kind) {
    case LK_INT: { CountSink s; DCol d; d.repr = c.repr; CellBits b; b.v = v; b.ns = 0; b.valid = true; emit_int(s, d, b); return s.n; }
    case LK_TEXT: {
      const uint32_t n = t1 - t0;
      if (!flags[c.flag]) return n;
      CountSink s; emit_ch_quoted(s, c.data + t0, n); return s.n - 2u;   // (the quotes are in `add`)
    }
    case LK_TEXT_RAW: return t1 - t0;
    case LK_BOOL: return (uint8_t)v ? 4u : 5u;
    case LK_TIME_DATE: {
      const int64_t sec = (int64_t)v;
      if (sec >= -62167219200ll && sec <= 253402300799ll) return 10u;   // years 0000 .. 9999: "2006-01-02"
      CountSink s; emit_date(s, sec); return s.n;
    }
    case LK_TIME_SEC: { CountSink s; emit_i64(s, (int64_t)v); return s.n; }
    default: {  // LK_TIME_DT64
      int64_t full = (int64_t)v * 1000000000LL + ns;
      if (c.repr > 0 && c.repr < 9) full = full / pow10_i64(9 - c.repr);
      CountSink s; emit_i64(s, full); return s.n;
    }
  }
}
#ifndef TF_SER_LENFAST_ATTR
#define TF_SER_LENFAST_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
#ifndef TF_SER_LENFAST_G
#define TF_SER_LENFAST_G 8
#endif
__global__ void __launch_bounds__(256) TF_SER_LENFAST_ATTR ser_chunk_len_fast(SerParams p, ChunkPlan cp, const LenCol *cols, const uint32_t *flags) {
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int32_t k; int64_t r0;
  tile_of(cp, tile, k, r0);
  if (k >= cp.nchunks || r0 >= p.nrows) return;
  const int32_t c0 = __builtin_amdgcn_readfirstlane(cp.cut[k]), c1 = __builtin_amdgcn_readfirstlane(cp.cut[k + 1]);
  const int64_t rl = r0 + (threadIdx.x & 63);
  if (rl >= p.nrows) return;
  const int64_t r = rl;
  uint32_t total = 0;
  constexpr int G = TF_SER_LENFAST_G;
  for (int32_t cg = c0; cg < c1; cg += G) {
    uint64_t v[G]; int32_t ns[G]; uint32_t ok = 0;   // (a text cell's two offsets ride in v: 26 registers of loads in flight, not 50)
#pragma unroll
    for (int j = 0; j < G; j++) {
      v[j] = 0; ns[j] = 0;
      if (cg + j < c1) {
        const LenCol &c = cols[cg + j];
        if (!c.validity || ((c.validity[r >> 3] >> (r & 7)) & 1)) ok |= 1u << j;
        if (c.kind == LK_TEXT || c.kind == LK_TEXT_RAW) { const uint32_t *o = (const uint32_t *)c.values; v[j] = o[r] | ((uint64_t)o[r + 1] << 32); }
        else if (c.kind == LK_BOOL) v[j] = ((const uint8_t *)c.values)[r];
        else if (c.kind == LK_INT) {
          switch (c.repr) {
            case TFGPU_R_INT8: case TFGPU_R_UINT8: v[j] = ((const uint8_t *)c.values)[r]; break;
            case TFGPU_R_INT16: case TFGPU_R_UINT16: v[j] = ((const uint16_t *)c.values)[r]; break;
            case TFGPU_R_INT32: case TFGPU_R_UINT32: v[j] = ((const uint32_t *)c.values)[r]; break;
            default: v[j] = ((const uint64_t *)c.values)[r];
          }
        } else { v[j] = ((const uint64_t *)c.values)[r]; if (c.kind == LK_TIME_DT64 && c.data) ns[j] = ((const int32_t *)c.data)[r]; }   // (a time column's nanoseconds ride in `data`)
      }
    }
#pragma unroll
    for (int j = 0; j < G; j++) if (cg + j < c1 && ((ok >> j) & 1u)) { const LenCol &c = cols[cg + j]; total += c.add + len_of_value(c, v[j], ns[j], (uint32_t)v[j], (uint32_t)(v[j] >> 32), flags); }
  }
  cp.chunk[(int64_t)k * p.nrows + r] = total;
}

#include "tf_serslab.inc"

// queue formats: the message frame once the cut plan is known.  Native: "[" before the first element of a message, ","
// before the others, "]" after the last; JSON: "\n" after every element but the last of its message.
__global__ void __launch_bounds__(256) ser_queue_tail_len(SerParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  const bool last = p.q.msg_flags[r] & 2;
  p.row_len[r] += (p.q.qformat == TFGPU_QFMT_NATIVE) ? (last ? 1u : 0u) : (last ? 0u : 1u);
}
__global__ void __launch_bounds__(256) ser_queue_frame(SerParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.nrows) return;
  uint8_t *row = p.out + p.row_len[r];
  const uint32_t len = p.row_len[r + 1] - p.row_len[r];
  const uint32_t f = p.q.msg_flags[r];
  if (p.q.qformat == TFGPU_QFMT_NATIVE) { row[0] = (f & 1) ? '[' : ','; if (f & 2) row[len - 1] = ']'; }
  else if (!(f & 2)) row[len - 1] = '\n';
}
// batches whose rows list different columns: the first column every row lists (its name and value carry no leading comma), -1 = none
__global__ void __launch_bounds__(256) queue_first_listed(const uint8_t *const *absent, int ncols, int64_t n, const uint16_t *order, int32_t *first) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int32_t f = -1;
  if (order) {  // the row's own order lists its columns first
    const int j = ncols ? order[r * ncols] : -1;
    if (j >= 0 && (!absent[j] || !((absent[j][r >> 3] >> (r & 7)) & 1u))) f = j;
  } else for (int j = 0; j < ncols; j++) if (!absent[j] || !((absent[j][r >> 3] >> (r & 7)) & 1u)) { f = j; break; }
  first[r] = f;
}
// A long constant cell (column names, table schema) of every row: one WAVE per (row, 2 KiB piece) — a workgroup per (row, 4 KiB chunk)
// left most of its 256 lanes without a byte to move for the few hundred bytes to a few KiB such a cell holds, and the launch was bound by
// its 10^5..10^6 workgroups.  The destination's alignment decides the split: bytes up to its next 16-byte boundary, aligned 16-byte
// stores, a byte tail; the source is a few cache-resident KiB of the blob wherever it sits.
constexpr uint32_t QFILL_PIECE = 2048;
__global__ void __launch_bounds__(256) ser_fill_const(SerParams p, int32_t ci, uint32_t pieces) {
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  const int64_t r = item / pieces; const uint32_t piece = (uint32_t)(item % pieces);
  if (r >= p.nrows) return;
  const SCol &sc = p.cols[ci];
  if (!q_applies(p, sc.apply, r)) return;
  const uint8_t *src = p.blob + sc.pre_off;
  uint8_t *dst = p.out + (uint64_t)p.row_len[r] + p.cell[(int64_t)ci * p.nrows + r];
  const uint32_t a = piece * QFILL_PIECE, b = min(a + QFILL_PIECE, sc.pre_len);
  const uint32_t head = min((uint32_t)((16 - ((uintptr_t)(dst + a) & 15)) & 15), b - a);
  if (lane < head) dst[a + lane] = src[a + lane];
  const uint32_t a16 = a + head, quads = (b - a16) / 16;
  struct __attribute__((packed, aligned(1))) U128 { uint32_t x, y, z, w; };
  for (uint32_t q = lane; q < quads; q += 64) { const U128 v = reinterpret_cast<const U128 *>(src + a16)[q]; reinterpret_cast<uint4 *>(dst + a16)[q] = make_uint4(v.x, v.y, v.z, v.w); }
  const uint32_t t0 = a16 + quads * 16;
  if (lane < b - t0) dst[t0 + lane] = src[t0 + lane];
}
// which row kinds occur (bit k = kind k), and whether every src_row indexes the row meta.  A few hundred waves stride over the rows and
// each adds ONE atomic: a wave per 64 rows was ~5 000 atomics on one word for 3 x 10^5 rows — they all read the word as zero before the
// first lands, and the L2 takes them one after another: 47-56 us of a kernel that reads 1.5 MB (profiles/r31b, r35b timelines)
constexpr unsigned QCHECK_BLOCKS = 128;
__global__ void __launch_bounds__(256) queue_check_kernel(const uint8_t *kind, const int32_t *src_row, int64_t n, int64_t meta_n, uint32_t *flags) {
  uint32_t f = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
    f |= 1u << (kind ? (kind[r] & 3u) : 0u);
    if (meta_n >= 0) { const int64_t k = src_row ? src_row[r] : r; if (k < 0 || k >= meta_n) f |= 16u; }
  }
  uint32_t w = 0;
#pragma unroll
  for (int b = 0; b < 5; b++) if (__ballot((f >> b) & 1u)) w |= 1u << b;
  if ((threadIdx.x & 63) == 0 && w) atomicOr(flags, w);
}

static inline unsigned blocks(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + 255) / 256); }

// host-side escaping of a column name the way encoding/json writes a map key (escapeHTML=false)
static void json_key(std::string &out, const std::string &k, bool html = false) {
  static const char *hex = "0123456789abcdef";
  out += '"';
  for (size_t i = 0; i < k.size();) {
    unsigned char c = (unsigned char)k[i];
    if (c < 0x80) {
      if (c >= 0x20 && c != '"' && c != '\\' && !(html && (c == '<' || c == '>' || c == '&'))) out += (char)c;
      else switch (c) {
        case '"': out += "\\\""; break; case '\\': out += "\\\\"; break; case '\b': out += "\\b"; break; case '\f': out += "\\f"; break;
        case '\n': out += "\\n"; break; case '\r': out += "\\r"; break; case '\t': out += "\\t"; break;
        default: out += "\\u00"; out += hex[c >> 4]; out += hex[c & 15];
      }
      i++; continue;
    }
    size_t need = 0; unsigned cp = 0, lo = 0x80, hi = 0xBF;
    if (c >= 0xC2 && c <= 0xDF) { need = 1; cp = c & 0x1F; }
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; cp = c & 0x0F; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; cp = c & 0x07; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
    bool ok = need > 0 && i + need < k.size();
    if (ok) for (size_t q = 1; q <= need; q++) {
      unsigned d = (unsigned char)k[i + q], l = q == 1 ? lo : 0x80u, h = q == 1 ? hi : 0xBFu;
      if (d < l || d > h) { ok = false; break; }
      cp = (cp << 6) | (d & 0x3F);
    }
    if (!ok) { out += "\\ufffd"; i++; continue; }
    if (cp == 0x2028 || cp == 0x2029) { out += cp == 0x2028 ? "\\u2028" : "\\u2029"; i += need + 1; continue; }
    out.append(k, i, need + 1); i += need + 1;
  }
  out += '"';
}

// Which (DataType, Go type) pairs each format renders on device; the rest is the host's (stock Go) business.
static void require_supported(int format, const DColumn &c, int any_as_string) {
  auto bad = [&](const char *why) {
    throw Error(TFGPU_ERR_UNSUPPORTED, "tfgpu_serialize: column " + c.name + ": " + why);
  };
  if (c.repr == TFGPU_R_FLOAT32 || c.repr == TFGPU_R_FLOAT64) {
    // strconv.FormatFloat(f, 'f', -1, bits): marshalNumericValue (marshal.go:257-301), castx.ToStringE (caste.go:57-106);
    // json.Marshal(float) = encoding/json's floatEncoder elsewhere (NaN / Inf there fail the whole call: checked on device)
    if (format == TFGPU_FMT_CSV && (c.dtype == TFGPU_T_ANY || c.dtype == TFGPU_T_BYTES)) bad("Go float under `any` / \"string\" in the CSV serializer");
  }
  if (format == TFGPU_FMT_CH_JSON_EACH_ROW) {
    const bool temporal = c.dtype == TFGPU_T_DATE || c.dtype == TFGPU_T_DATETIME || c.dtype == TFGPU_T_TIMESTAMP;
    if (c.repr == TFGPU_R_TIME && !temporal) bad("time.Time under a non-temporal DataType (generic JSON fallback)");
  } else if (format == TFGPU_FMT_CSV) {
    if (c.dtype == TFGPU_T_BYTES && c.repr != TFGPU_R_BYTES && c.repr != TFGPU_R_STRING) bad("\"string\" (bytes) column holding neither []byte nor a Go string");
    if (c.dtype == TFGPU_T_ANY && (c.repr == TFGPU_R_TIME || c.repr == TFGPU_R_DURATION || c.repr == TFGPU_R_BYTES)) bad("`any` column holding a non-JSON-native Go value");
    if (c.dtype != TFGPU_T_ANY && c.repr == TFGPU_R_JSON) bad("map/slice value under a scalar DataType");
  }
  (void)any_as_string;
}

// json.Marshal fails the whole call on NaN / ±Inf (UnsupportedValueError): find them before any text is produced
__global__ void float_nonfinite_kernel(const void *values, int64_t n, int is32, const uint8_t *validity, uint32_t *flag) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || (validity && !((validity[r >> 3] >> (r & 7)) & 1))) return;
  const double v = is32 ? (double)((const float *)values)[r] : ((const double *)values)[r];
  if (v != v || v == INFINITY || v == -INFINITY) *flag = 1;
}

}  // namespace tf

using namespace tf;

// rawSerializer (pkg/serializer/raw.go:24-63) under the batch serializer: every row's `data` bytes, '\n' behind each (AddClosingNewLine)
// or between them
namespace tf {
struct SegRawParams { const uint32_t *in_off; const uint8_t *in; const uint32_t *out_off; uint8_t *out; int64_t n; };
// lane = row: its bytes as (unaligned) 8-byte words
__global__ void __launch_bounds__(256) ser_raw_copy(SegRawParams p) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.n) return;
  const uint32_t a = p.in_off[r], len = p.in_off[r + 1] - a;
  const uint8_t *src = p.in + a; uint8_t *dst = p.out + p.out_off[r];
  struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
  uint32_t i = 0;
  for (; i + 8 <= len; i += 8) reinterpret_cast<U64 *>(dst + i)->v = reinterpret_cast<const U64 *>(src + i)->v;
  for (; i < len; i++) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) ser_raw_offsets(const uint32_t *in_off, int64_t n, uint32_t *out_off) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r <= n) out_off[r] = in_off[r] + (uint32_t)r;  // one '\n' per row in front of row r
}
__global__ void __launch_bounds__(256) ser_raw_newlines(const uint32_t *out_off, int64_t n, uint64_t total, uint8_t *out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && (uint64_t)out_off[r + 1] - 1 < total) out[out_off[r + 1] - 1] = '\n';
}
}  // namespace tf
// tfgpu_serialize_batch asks the serializers for where each row's text starts (a device array of nrows + 1 offsets)
static thread_local tf::Buf *g_row_offsets_out = nullptr;
static int serialize_raw(const tfgpu_dbatch *b, const tfgpu_serialize_options *opts, tfgpu_dbuf **out) {
  static const char *const MIRROR[] = {"topic", "partition", "seq_no", "write_time", "data", "meta", "sequence_key"};  // changeitem.RawDataColumns (mirror.go:10-33)
  bool mirror = b->cols.size() == 7;
  for (size_t i = 0; mirror && i < 7; i++) mirror = b->cols[i].name == MIRROR[i];
  if (!mirror) return tf::fail(TFGPU_ERR_INVALID, "raw serializer: unexpected input, expect no converted raw data");  // IsMirror (change_item.go:385-395)
  const DColumn &d = b->cols[4];
  if (d.repr != TFGPU_R_BYTES && d.repr != TFGPU_R_STRING) return tf::fail(TFGPU_ERR_INVALID, "raw serializer: unable to construct raw message data: unexpected data type, expected string or []byte");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  hipStream_t st = cx.stream;
  materialize(*b);
  const int64_t n = b->nrows;
  if (d.validity && n) {  // a nil value is no []byte either (GetRawMessageData, mirror.go:78-87)
    std::vector<uint8_t> bm((size_t)(n + 7) / 8);
    d2h(bm.data(), d.validity->p, bm.size());
    tf::sync();
    for (int64_t r = 0; r < n; r++) if (!((bm[(size_t)r >> 3] >> (r & 7)) & 1)) return tf::fail(TFGPU_ERR_INVALID, "raw serializer: unable to construct raw message data: unexpected data type: <nil>");
  }
  const bool closing = opts && opts->add_closing_newline;
  auto res = std::make_unique<tfgpu_dbuf>();
  const uint64_t total = n ? d.data_len + (uint64_t)n - (closing ? 0 : 1) : 0;
  if (total >> 32) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_serialize: output exceeds 4 GiB; split the batch by rows");
  res->size = total;
  res->mem = dalloc(total + 64);
  if (n) {
    Buf ooff = dalloc((size_t)(n + 2) * 4);
    ser_raw_offsets<<<(unsigned)((n + 256) / 256), 256, 0, st>>>(ptr<uint32_t>(d.offsets), n, ptr<uint32_t>(ooff));
    SegRawParams rp{ptr<uint32_t>(d.offsets), ptr<uint8_t>(d.payload()), ptr<uint32_t>(ooff), ptr<uint8_t>(res->mem), n};
    { KernelTimer t("ser_raw"); ser_raw_copy<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(rp); }
    ser_raw_newlines<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(ptr<uint32_t>(ooff), n, total, ptr<uint8_t>(res->mem));
    if (g_row_offsets_out) *g_row_offsets_out = ooff;
  }
  tf::sync();
  *out = res.release();
  return TFGPU_OK;
}

// The chunk plan of the text formats' walks (ser_chunk_len / layout / write): where a row's columns are cut, the per-column records
// the waves read, the (row, chunk) length table.  `order`: the columns in output order; sc / blob: their descriptors and key prefixes.
struct ChunkSetup { ChunkPlan cp{}; Buf bcut, brecs, cell; };
static ChunkSetup make_chunk_setup(const tfgpu_dbatch *b, int format, const std::vector<int> &order, const std::vector<SCol> &sc, const std::string &blob, int64_t n) {
  ChunkSetup cs;
  ChunkPlan &cp = cs.cp;
  Buf &bcut = cs.bcut, &brecs = cs.brecs, &cell = cs.cell;
  const int ncols = (int)order.size();
  // the cuts: consecutive columns until the ESTIMATED text of a row's chunk passes the target (64 rows of a chunk share a 9 KiB
  // image, i.e. 144 bytes per row; an estimate that is off only sends more tiles down the direct path).  The target IS the image's
  // 144 bytes since round 6 (112 until then): swept again on the round's kernels — 112 / 128 / 136 / 144 / 152 / 160 / 192 give a write
  // pass of 2.55 / 2.43 / 2.39 / 2.36 / 2.47 / 2.60 / 3.03 ms on configs[3] and 0.61 / 0.51 / - / 0.49 on configs[2] (profiles/r28c_*,
  // r28d_*): fewer, fuller tiles win until the tiles that outgrow the image (and write straight to HBM) take over
  static const uint32_t target = [] { const char *e = std::getenv("TFGPU_SER_CHUNK_BYTES"); return e ? (uint32_t)std::atoi(e) : 144u; }();
  std::vector<int32_t> cut{0};
  double acc = 0;
  for (int j = 0; j < ncols; j++) {
    const DColumn &c = b->cols[(size_t)order[(size_t)j]];
    double est = sc[(size_t)j].pre_len + 1.0;
    switch (c.repr) {
      case TFGPU_R_INT8: case TFGPU_R_UINT8: est += 2; break;
      case TFGPU_R_INT16: case TFGPU_R_UINT16: est += 4; break;
      case TFGPU_R_INT32: case TFGPU_R_UINT32: est += 8; break;
      case TFGPU_R_INT64: case TFGPU_R_UINT64: case TFGPU_R_DURATION: est += 16; break;
      case TFGPU_R_BOOL: est += 5; break;
      case TFGPU_R_FLOAT32: case TFGPU_R_FLOAT64: est += 14; break;
      case TFGPU_R_TIME: est += 24; break;
      default: est += 2.0 + (double)c.data_len / (double)n * (format == TFGPU_FMT_JSON && c.repr == TFGPU_R_BYTES ? 1.34 : 1.0);
    }
    if (j > cut.back() && acc + est > target) { cut.push_back(j); acc = 0; }
    acc += est;
  }
  cut.push_back(ncols);
  if (ncols == 0) cut = {0, 0};
  cp.nchunks = (int32_t)cut.size() - 1;
  bcut = upload_const(cut.data(), cut.size() * sizeof(int32_t));
  cp.cut = ptr<int32_t>(bcut);
  std::vector<uint32_t> recs((size_t)std::max(ncols, 1) * REC_WORDS, 0u);
  for (int j = 0; j < ncols; j++) {
    const SCol &d = sc[(size_t)j];
    uint32_t *w = recs.data() + (size_t)j * REC_WORDS;
    auto put64 = [&](int k, const void *q) { const uint64_t v = reinterpret_cast<uintptr_t>(q); w[k] = (uint32_t)v; w[k + 1] = (uint32_t)(v >> 32); };
    put64(RW_VALUES, d.c.values); put64(RW_OFFSETS, d.c.offsets); put64(RW_DATA, d.c.data); put64(RW_NANOS, d.c.nanos); put64(RW_VALIDITY, d.c.validity);
    w[RW_REPR] = (uint32_t)d.c.repr; w[RW_DTYPE] = (uint32_t)d.c.dtype; w[RW_PRE_OFF] = d.pre_off; w[RW_PRE_LEN] = d.pre_len; w[RW_CH_FLAGS] = d.ch_flags; w[RW_PREC] = d.prec;
    std::memcpy(w + REC_PREFIX_WORD, blob.data() + d.pre_off, std::min<size_t>(d.pre_len, REC_PREFIX_BYTES));
  }
  brecs = upload_const(recs.data(), recs.size() * 4);
  cp.recs = ptr<uint32_t>(brecs);
  cell = dalloc((size_t)cp.nchunks * (size_t)n * 4);
  cp.chunk = ptr<uint32_t>(cell);
  { const char *e = std::getenv("TFGPU_SER_DIRECT"); cp.direct = e && *e == '1'; }
  { const char *e = std::getenv("TFGPU_SER_ORDER"); cp.order = e ? std::atoi(e) : 1; }  // (2, the XCD-aware form of 0, measured like 0: 3.18 ms against 2.62 for order 1 on configs[3] — the reads of 64 consecutive rows per column decide, not the seams of the writes; profiles/r06o_*)
  cp.ngroups = (n + 63) / 64;
  return cs;
}

// The slab plan of the text formats (tf_serslab.inc): column records + 8-byte aligned key prefixes in ONE allocation (a workgroup stages
// both with one loop), rows per workgroup from the estimated row text so that a group's rows fit the LDS image.
struct SlabSetup { SlabPlan sp{}; Buf brecs; unsigned threads = 256; size_t lds_len = 0, lds_write = 0; };
static int env_int(const char *name, int dflt) { const char *e = std::getenv(name); return e && *e ? std::atoi(e) : dflt; }
static SlabSetup make_slab_setup(const tfgpu_dbatch *b, int format, const std::vector<int> &order, const std::vector<SCol> &sc, const std::string &blob, int64_t n) {
  SlabSetup ss;
  const int ncols = (int)order.size();
  std::vector<uint32_t> recs((size_t)ncols * SLAB_REC_WORDS, 0u);
  std::vector<uint8_t> pblob;
  double est_row = 4;
  for (int j = 0; j < ncols; j++) {
    const SCol &d = sc[(size_t)j];
    const DColumn &c = b->cols[(size_t)order[(size_t)j]];
    uint32_t *w = recs.data() + (size_t)j * SLAB_REC_WORDS;
    auto put64 = [&](int k, const void *q) { const uint64_t v = reinterpret_cast<uintptr_t>(q); w[k] = (uint32_t)v; w[k + 1] = (uint32_t)(v >> 32); };
    put64(SW_VALUES, d.c.values); put64(SW_OFFSETS, d.c.offsets); put64(SW_DATA, d.c.data); put64(SW_NANOS, d.c.nanos); put64(SW_VALIDITY, d.c.validity);
    w[SW_REPR] = (uint32_t)d.c.repr; w[SW_DTYPE] = (uint32_t)d.c.dtype; w[SW_CH_FLAGS] = d.ch_flags; w[SW_PREC] = d.prec;
    // the prefix without the chunk walks' leading placeholder byte (JSONEachRow: the slab cells put '{' or ',' themselves)
    const uint32_t skip = format == TFGPU_FMT_CH_JSON_EACH_ROW ? 1u : 0u;
    w[SW_PRE_OFF] = (uint32_t)pblob.size(); w[SW_PRE_LEN] = d.pre_len - skip;
    pblob.insert(pblob.end(), blob.begin() + d.pre_off + skip, blob.begin() + d.pre_off + d.pre_len);
    pblob.resize((pblob.size() + 7) & ~(size_t)7, 0);
    double est = d.pre_len + 1.0;
    switch (c.repr) {
      case TFGPU_R_INT8: case TFGPU_R_UINT8: est += 3; break;
      case TFGPU_R_INT16: case TFGPU_R_UINT16: est += 5; break;
      case TFGPU_R_INT32: case TFGPU_R_UINT32: est += 10; break;
      case TFGPU_R_INT64: case TFGPU_R_UINT64: case TFGPU_R_DURATION: est += 20; break;
      case TFGPU_R_BOOL: est += 5; break;
      case TFGPU_R_FLOAT32: case TFGPU_R_FLOAT64: est += 18; break;
      case TFGPU_R_TIME: est += 32; break;
      default: est += 2.0 + 1.1 * (double)c.data_len / (double)std::max<int64_t>(n, 1) * (format == TFGPU_FMT_JSON && c.repr == TFGPU_R_BYTES ? 1.34 : 1.0);
    }
    est_row += est;
  }
  pblob.resize((pblob.size() + 15) & ~(size_t)15, 0);
  std::vector<uint8_t> both(recs.size() * 4 + pblob.size());
  std::memcpy(both.data(), recs.data(), recs.size() * 4);
  std::memcpy(both.data() + recs.size() * 4, pblob.data(), pblob.size());
  ss.brecs = upload_const(both.data(), both.size());
  SlabPlan &sp = ss.sp;
  sp.recs = ptr<uint32_t>(ss.brecs);
  sp.pblob = ptr<uint8_t>(ss.brecs) + recs.size() * 4;
  sp.pblob_bytes = (uint32_t)pblob.size();
  // rows per workgroup: the largest power of two whose estimated text fits the image budget
  // (measurement and test knobs, read per call: the rows-per-workgroup budget, R itself, the workgroup size, a clamp on the image that forces the
  //  sub-run and the straight-to-HBM paths on ordinary rows)
  const int cap_kb = env_int("TFGPU_SLAB_CAP_KB", 24), force_r = env_int("TFGPU_SLAB_RSHIFT", -1), threads = env_int("TFGPU_SLAB_THREADS", 256), image_max = env_int("TFGPU_SLAB_IMAGE_BYTES", 0);
  int rshift = 6;
  while (rshift > 2 && (double)((size_t)1 << rshift) * est_row > (double)cap_kb * 1024.0) rshift--;
  while (rshift > 2 && ((int64_t)1 << rshift) > std::max<int64_t>(n, 4)) rshift--;
  if (force_r >= 2 && force_r <= 6) rshift = force_r;
  ss.threads = (unsigned)std::min(512, std::max(64 << 0, threads));
  while ((ss.threads >> rshift) == 0) rshift--;             // at least one thread a row
  sp.rshift = rshift;
  const size_t fixed = slab_lds_bytes(ncols, sp.pblob_bytes, rshift, 0);
  size_t cap = (size_t)((double)((size_t)1 << rshift) * est_row * 1.15) + 256;
  cap = std::max<size_t>(cap, (size_t)ss.threads * 4 + 64);
  cap = std::min<size_t>(cap, 64 * 1024 - 64 - fixed);
  if (image_max > 0) cap = std::max<size_t>((size_t)ss.threads * 4 + 64, std::min<size_t>(cap, (size_t)image_max));
  sp.cap = (uint32_t)(cap & ~(size_t)15);
  sp.ngroups = (n + ((int64_t)1 << rshift) - 1) >> rshift;
  sp.xcd_order = env_int("TFGPU_SLAB_XCD", 1);
  ss.lds_write = slab_lds_bytes(ncols, sp.pblob_bytes, rshift, sp.cap);
  ss.lds_len = slab_lds_bytes(ncols, sp.pblob_bytes, rshift, ss.threads * 4 + 64);
  return ss;
}
static inline unsigned slab_grid(const SlabPlan &sp) { return (unsigned)(sp.xcd_order ? ((sp.ngroups + 7) >> 3) << 3 : sp.ngroups); }
// can the slab form take this batch?  (the fixed part — records, prefixes, one length per cell — must leave room for an image)
static bool slab_fits(int ncols, size_t blob_bytes) { return ncols > 0 && (size_t)ncols * (SLAB_REC_WORDS * 4 + 16 + 4 * 4) + blob_bytes + 8192 < 60 * 1024; }

extern "C" int tfgpu_serialize_ex(int format, const tfgpu_dbatch *b, const tfgpu_serialize_options *opts, tfgpu_dbuf **out) {
  try {
    // ClickHouse JSONEachRow walks the row's OWN ColumnNames by name and skips nils (MarshalCItoJSON, marshal.go:82-125): a cell the row does not
    // list prints exactly like a nil one — nothing — and an ABSENT cell's validity bit is clear, so the walk needs no change.  The json / csv
    // serializers type a row's i-th VALUE by the schema's i-th column (buildJsonKV / buildCsvCells: columns[i]) — a row that leaves a column out is
    // typed by the wrong columns from there on: that quirk stays with the stock path, as do rows with their own name ORDER (col_order).
    tf::dense(b, b && format == TFGPU_FMT_CH_JSON_EACH_ROW && !b->col_order);  // its rows may still be a selection (tfgpu_dbatch::pending)
    if (!b || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_serialize: null argument");
    if (format == TFGPU_FMT_RAW) return serialize_raw(b, opts, out);
    if (format != TFGPU_FMT_CH_JSON_EACH_ROW && format != TFGPU_FMT_JSON && format != TFGPU_FMT_CSV)
      return tf::fail(TFGPU_ERR_INVALID, "tfgpu_serialize: unknown format");
    Context &cx = ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    hipStream_t st = cx.stream;
    const int64_t n = b->nrows;
    const int ncols = (int)b->cols.size();
    const int any_as_string = opts ? opts->any_as_string : 0;
    materialize(*b);
    for (auto &c : b->cols) require_supported(format, c, any_as_string);
    {
      Buf flag;
      for (auto &c : b->cols) {
        if (c.repr != TFGPU_R_FLOAT32 && c.repr != TFGPU_R_FLOAT64) continue;
        const bool numeric = (c.dtype >= TFGPU_T_INT8 && c.dtype <= TFGPU_T_FLOAT64) || c.dtype == TFGPU_T_INTERVAL;
        if (format == TFGPU_FMT_CSV || (format == TFGPU_FMT_CH_JSON_EACH_ROW && numeric)) continue;  // FormatFloat prints NaN / +Inf
        if (!flag) flag = dalloc_zero(4);
        if (n) float_nonfinite_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(c.values->p, n, c.repr == TFGPU_R_FLOAT32, ptr<uint8_t>(c.validity), ptr<uint32_t>(flag));
      }
      if (flag) {
        const uint32_t *h = d2h_u32(flag->p);
        tf::sync();
        if (*h) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_serialize: json: unsupported value: NaN or Inf (json.Marshal fails the batch in the reference)");
      }
    }
    if (opts && opts->ncols && opts->ncols != ncols) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_serialize: options describe another column count");

    // column order: CH and CSV keep ColumnNames order; encoding/json sorts map keys
    std::vector<int> order((size_t)ncols);
    for (int i = 0; i < ncols; i++) order[(size_t)i] = i;
    if (format == TFGPU_FMT_JSON) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return b->cols[(size_t)x].name < b->cols[(size_t)y].name; });
    if (format == TFGPU_FMT_JSON)  // a map holds each key once: the last value of a duplicated name wins
      for (int i = 0; i + 1 < ncols; i++) if (b->cols[(size_t)order[(size_t)i]].name == b->cols[(size_t)order[(size_t)i + 1]].name)
        return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_serialize: duplicate column names");

    static const bool scatter = [] { const char *e = std::getenv("TFGPU_SER_SCATTER"); return e && *e == '1'; }();  // the per-cell passes (the queue formats' own), for A/B runs
    std::string blob;
    std::vector<SCol> sc((size_t)ncols);
    for (int j = 0; j < ncols; j++) {
      const DColumn &c = b->cols[(size_t)order[(size_t)j]];
      SCol &s = sc[(size_t)j];
      s.c = dcol_of(c);
      s.pre_off = (uint32_t)blob.size();
      if (format == TFGPU_FMT_CH_JSON_EACH_ROW) { if (!scatter) blob += '\0'; blob += '"'; blob += c.name; blob += "\":"; }  // writeColName: raw name (the chunk walks put '{' or ',' into the leading byte)
      else if (format == TFGPU_FMT_JSON) { if (j) blob += ','; json_key(blob, c.name); blob += ':'; }
      else if (j) blob += ',';
      s.pre_len = (uint32_t)blob.size() - s.pre_off;
      s.ch_flags = 0; s.prec = 0;
      if (format == TFGPU_FMT_CH_JSON_EACH_ROW) {
        if (opts && opts->ncols && opts->ch_flags) { s.ch_flags = opts->ch_flags[order[(size_t)j]]; s.prec = opts->ch_precision ? opts->ch_precision[order[(size_t)j]] : 0; }
        else switch (c.dtype) {  // pkg/providers/clickhouse/typesystem.md
          case TFGPU_T_BYTES: case TFGPU_T_UTF8: case TFGPU_T_ANY: s.ch_flags = TFGPU_CH_STRING; break;
          case TFGPU_T_DATE: s.ch_flags = TFGPU_CH_DATE; break;
          case TFGPU_T_TIMESTAMP: s.ch_flags = TFGPU_CH_DATETIME64; s.prec = 9; break;
          default: break;
        }
      }
    }
    auto res = std::make_unique<tfgpu_dbuf>();
    if (n == 0 || (ncols == 0 && format == TFGPU_FMT_CSV && n == 0)) { res->mem = dalloc(64); res->size = 0; *out = res.release(); return TFGPU_OK; }

    Buf bsc = upload_const(sc.data(), sc.size() * sizeof(SCol)), bblob = upload_const(blob.data(), blob.size());
    Buf row_len = dalloc((size_t)(n + 1) * 4 + 16), last = dalloc((size_t)n * 4 + 16);
    SerParams p{};
    p.cols = ptr<SCol>(bsc); p.ncols = ncols; p.nrows = n; p.format = format; p.any_as_string = any_as_string;
    { const char *ab = std::getenv("TFGPU_SER_ABLATE"); p.ablate = ab ? std::atoi(ab) : 0; }
    p.closing_newline = opts ? opts->add_closing_newline : 0;
    p.blob = ptr<uint8_t>(bblob); p.row_len = ptr<uint32_t>(row_len); p.last_present = ptr<int32_t>(last);
    Buf tot64 = dalloc_zero(8);
    p.total64 = reinterpret_cast<unsigned long long *>(tot64->p);
    const int64_t ncell = (int64_t)ncols * n;
    Buf cell, bcut, brecs;
    ChunkPlan cp{};
    // the slab form (rows in LDS, one contiguous run a workgroup: tf_serslab.inc) is OPT-IN (TFGPU_SER_SLAB=1): it writes exactly the text's bytes, but on
    // wide tables a workgroup's rows x ALL columns means 8-16 rows of 300 arrays per workgroup (32-byte reads, a TLB entry each) and 4-8 columns a wave —
    // measured 3.3 + 10.2 ms against the chunk walks' 1.28 + 2.62 ms on configs[3] (profiles/r20d_tune_configs3.txt)
    const bool slab_on = env_int("TFGPU_SER_SLAB", 0) != 0;
    const bool slab = !scatter && slab_on && slab_fits(ncols, blob.size());
    SlabSetup slabs;
    if (scatter) {
      cell = dalloc((size_t)std::max(ncols, 1) * (size_t)n * 4);
      p.cell = ptr<uint32_t>(cell);
      { KernelTimer t("ser_cell_len"); if (ncell) ser_cell_len<<<dim3(blocks(n), (unsigned)ncols), 256, 0, st>>>(p); }
      { KernelTimer t("ser_row_layout"); ser_row_layout<<<blocks(n), 256, 0, st>>>(p); }
    } else if (slab) {
      slabs = make_slab_setup(b, format, order, sc, blob, n);
      SlabPlan lp = slabs.sp;
      lp.cap = slabs.threads * 4 + 64;
      KernelTimer t("ser_slab_len");
      if (format == TFGPU_FMT_CH_JSON_EACH_ROW) ser_slab_len<TFGPU_FMT_CH_JSON_EACH_ROW><<<slab_grid(lp), slabs.threads, slabs.lds_len, st>>>(p, lp);
      else if (format == TFGPU_FMT_JSON) ser_slab_len<TFGPU_FMT_JSON><<<slab_grid(lp), slabs.threads, slabs.lds_len, st>>>(p, lp);
      else ser_slab_len<TFGPU_FMT_CSV><<<slab_grid(lp), slabs.threads, slabs.lds_len, st>>>(p, lp);
    } else {
      ChunkSetup csu = make_chunk_setup(b, format, order, sc, blob, n);
      cp = csu.cp; bcut = csu.bcut; brecs = csu.brecs; cell = csu.cell;
      const unsigned grid = ser_grid(cp);
      // JSONEachRow over integers, text, booleans, dates / DateTime64 / epoch seconds: the lean length pass (ser_chunk_len_fast); anything else in the
      // batch: the walk over a counting sink.  TFGPU_SER_LEN_FAST=0: always the walk (A/B).
      std::vector<LenCol> lc;
      std::vector<int32_t> tcols; std::vector<uint64_t> tbytes;
      bool fast = format == TFGPU_FMT_CH_JSON_EACH_ROW && env_int("TFGPU_SER_LEN_FAST", 1) != 0 && p.ablate == 0;
      for (int j = 0; j < ncols && fast; j++) {
        const DColumn &c = b->cols[(size_t)order[(size_t)j]];
        const SCol &q = sc[(size_t)j];
        LenCol l{};
        l.validity = q.c.validity; l.add = q.pre_len; l.repr = c.repr;
        const uint32_t fl = q.ch_flags;
        const bool numeric = (c.dtype >= TFGPU_T_INT8 && c.dtype <= TFGPU_T_FLOAT64) || c.dtype == TFGPU_T_INTERVAL;
        const bool generic_quotes = c.dtype != TFGPU_T_ANY || any_as_string || (fl & TFGPU_CH_STRING);   // marshalGeneric's double marshal
        if (c.repr >= TFGPU_R_INT8 && c.repr <= TFGPU_R_UINT64) {
          l.kind = LK_INT; l.values = q.c.values;
          if (numeric ? (fl & TFGPU_CH_STRING) != 0 : generic_quotes) l.add += 2;
        } else if (c.repr == TFGPU_R_STRING || (c.repr == TFGPU_R_BYTES && !(fl & TFGPU_CH_ARRAY))) {
          l.values = q.c.offsets; l.data = q.c.data;
          if (fl & TFGPU_CH_DECIMAL) l.kind = LK_TEXT_RAW;
          else { l.kind = LK_TEXT; l.add += 2; l.flag = (uint32_t)tcols.size(); tcols.push_back(j); tbytes.push_back(c.data_len); }
        } else if (c.repr == TFGPU_R_BOOL) {
          l.kind = LK_BOOL; l.values = q.c.values;
          if (c.dtype != TFGPU_T_BOOLEAN && generic_quotes) l.add += 2;
        } else if (c.repr == TFGPU_R_TIME && (c.dtype == TFGPU_T_DATE || c.dtype == TFGPU_T_DATETIME || c.dtype == TFGPU_T_TIMESTAMP) && !(fl & TFGPU_CH_STRING)) {
          l.values = q.c.values;
          if (fl & TFGPU_CH_DATETIME64) { l.kind = LK_TIME_DT64; l.repr = (int32_t)q.prec; l.data = reinterpret_cast<const uint8_t *>(q.c.nanos); }
          else if (fl & TFGPU_CH_DATE) { l.kind = LK_TIME_DATE; l.add += 2; }
          else l.kind = LK_TIME_SEC;
        } else fast = false;
        lc.push_back(l);
      }
      if (fast) {
        Buf blc = upload_const(lc.data(), lc.size() * sizeof(LenCol));
        Buf flags = dalloc_zero((tcols.size() + 1) * 4);
        if (!tcols.empty()) {
          Buf btc = upload_const(tcols.data(), tcols.size() * 4);
          uint64_t most = 0;
          for (uint64_t x : tbytes) most = std::max(most, x);
          const unsigned gx = (unsigned)std::min<uint64_t>(std::max<uint64_t>((most + 4095) / 4096, 1), 2048);
          KernelTimer t("ser_text_flags");
          ser_text_flags<<<dim3(gx, (unsigned)tcols.size()), 256, 0, st>>>(ptr<LenCol>(blc), ptr<int32_t>(btc), n, ptr<uint32_t>(flags));
        }
        KernelTimer t("ser_chunk_len");
        ser_chunk_len_fast<<<grid, 256, 0, st>>>(p, cp, ptr<LenCol>(blc), ptr<uint32_t>(flags));
      } else {
        KernelTimer t("ser_chunk_len");
        if (format == TFGPU_FMT_CH_JSON_EACH_ROW) ser_chunk_len<TFGPU_FMT_CH_JSON_EACH_ROW><<<grid, 256, 0, st>>>(p, cp);
        else if (format == TFGPU_FMT_JSON) ser_chunk_len<TFGPU_FMT_JSON><<<grid, 256, 0, st>>>(p, cp);
        else ser_chunk_len<TFGPU_FMT_CSV><<<grid, 256, 0, st>>>(p, cp);
      }
      {
        KernelTimer t("ser_chunk_layout");
        if (format == TFGPU_FMT_CH_JSON_EACH_ROW) ser_chunk_layout<TFGPU_FMT_CH_JSON_EACH_ROW><<<blocks(n), 256, 0, st>>>(p, cp);
        else if (format == TFGPU_FMT_JSON) ser_chunk_layout<TFGPU_FMT_JSON><<<blocks(n), 256, 0, st>>>(p, cp);
        else ser_chunk_layout<TFGPU_FMT_CSV><<<blocks(n), 256, 0, st>>>(p, cp);
      }
    }
        exclusive_scan_u32(p.row_len, p.row_len, n, true);
    if (g_row_offsets_out) *g_row_offsets_out = row_len; const uint32_t *htot = d2h_u32(p.row_len + n);
    const uint32_t *h64 = d2h_u32(tot64->p, 2);
    tf::sync();
    // a uint32 offset space: batches whose text exceeds 4 GiB must be split by rows (batch.go chunks at 25 000 rows)
    if (h64[1] != 0) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_serialize: output exceeds 4 GiB; split the batch by rows (the reference serialises 25 000-row chunks, batch.go:18)");
    res->size = *htot;
    res->mem = dalloc(res->size + 64);
    p.out = ptr<uint8_t>(res->mem);
    if (scatter) {
      { KernelTimer t("ser_cell_write"); if (ncell) ser_cell_write<<<dim3(blocks(n), (unsigned)ncols), 256, 0, st>>>(p); }
      { KernelTimer t("ser_row_frame"); ser_row_frame<<<blocks(n), 256, 0, st>>>(p); }
    } else if (slab) {
      KernelTimer t("ser_slab_write");
      const SlabPlan &sp = slabs.sp;
      if (format == TFGPU_FMT_CH_JSON_EACH_ROW) ser_slab_write<TFGPU_FMT_CH_JSON_EACH_ROW><<<slab_grid(sp), slabs.threads, slabs.lds_write, st>>>(p, sp);
      else if (format == TFGPU_FMT_JSON) ser_slab_write<TFGPU_FMT_JSON><<<slab_grid(sp), slabs.threads, slabs.lds_write, st>>>(p, sp);
      else ser_slab_write<TFGPU_FMT_CSV><<<slab_grid(sp), slabs.threads, slabs.lds_write, st>>>(p, sp);
    } else {
      const unsigned grid = ser_grid(cp);
      KernelTimer t("ser_chunk_write");
      if (format == TFGPU_FMT_CH_JSON_EACH_ROW) ser_chunk_write<TFGPU_FMT_CH_JSON_EACH_ROW><<<grid, 256, 4 * WAVE_IMG_BYTES + 64, st>>>(p, cp);
      else if (format == TFGPU_FMT_JSON) ser_chunk_write<TFGPU_FMT_JSON><<<grid, 256, 4 * WAVE_IMG_BYTES + 64, st>>>(p, cp);
      else ser_chunk_write<TFGPU_FMT_CSV><<<grid, 256, 4 * WAVE_IMG_BYTES + 64, st>>>(p, cp);
    }
    *out = res.release();
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}

extern "C" int tfgpu_serialize(int format, const tfgpu_dbatch *b, tfgpu_dbuf **out) { return tfgpu_serialize_ex(format, b, nullptr, out); }

// batchSerializer (pkg/serializer/batch.go): Serialize (:73-117) cuts batches above the threshold into parts of `threshold`
// items, serializes them side by side and joins them with the separator — the text of the whole batch — then trims ONE
// trailing separator (bytes.TrimSuffix, :112-114), which the undivided path (:75-82) does not; SerializeAndWrite (:119-209)
// writes the parts in order, each but the last followed by the separator, and trims nothing.  The device serializes the
// batch at once; what this call adds is exactly those two differences: the trim, and where the reference's Write calls end.
extern "C" int tfgpu_serialize_batch(int format, const tfgpu_dbatch *b, const tfgpu_serialize_options *opts, const tfgpu_batch_serializer_config *cfg,
                                     int for_writer, tfgpu_dbuf **out, uint64_t *part_ends, int64_t part_cap, int64_t *nparts) {
  try {
  tf::dense(b);  // its rows may still be a selection (tfgpu_dbatch::pending)
    if (!b || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_serialize_batch: null argument");
    if (format != TFGPU_FMT_JSON && format != TFGPU_FMT_CSV && format != TFGPU_FMT_RAW)
      return tf::fail(TFGPU_ERR_INVALID, "tfgpu_serialize_batch: the batch serializers are raw, json and csv (batch_factory.go:31-63)");
    // newBatchSerializer (:37-71)
    int64_t concurrency = 1, threshold = 0;
    if (!(cfg && cfg->disable_concurrency)) {
      concurrency = cfg && cfg->concurrency ? cfg->concurrency : (cfg && cfg->gomaxprocs > 0 ? cfg->gomaxprocs : 1);  // runtime.GOMAXPROCS(0) is the caller's
      threshold = cfg && cfg->threshold ? cfg->threshold : 25000;                                                        // DefaultBatchSerializerThreshold
    }
    const int64_t n = b->nrows;
    const bool parts = concurrency >= 2 && n > threshold;
    const bool sep = format != TFGPU_FMT_CSV && !(opts && opts->add_closing_newline);  // "\n" (batch_factory.go:36-39); the csv serializer has none (:54-58)
    tf::Buf row_off;
    g_row_offsets_out = (parts && for_writer) ? &row_off : nullptr;
    tfgpu_dbuf *text = nullptr;
    const int rc = tfgpu_serialize_ex(format, b, opts, &text);
    g_row_offsets_out = nullptr;
    if (rc != TFGPU_OK) return rc;
    std::unique_ptr<tfgpu_dbuf> res(text);
    int64_t np = 1;
    if (parts && !for_writer && sep && res->size) {  // bytes.TrimSuffix(joined, separator)
      tf::Context &cx = tf::ctx();
      std::lock_guard<std::mutex> lk(cx.mu);
      uint8_t last = 0;
      tf::d2h(&last, (const uint8_t *)res->mem->p + res->size - 1, 1);
      tf::sync();
      if (last == '\n') res->size--;
    }
    if (for_writer) {
      np = parts ? (n + threshold - 1) / threshold : 1;  // the Write calls: one per part, or one for the undivided batch
      if (part_ends) {
        if (np > part_cap) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_serialize_batch: part_ends is too small");
        if (!parts) part_ends[0] = res->size;
        else {
          tf::Context &cx = tf::ctx();
          std::lock_guard<std::mutex> lk(cx.mu);
          std::vector<uint32_t> at((size_t)np);
          // part i ends where row (i + 1) * threshold starts: its text, then the separator in front of the next part
          for (int64_t i = 0; i + 1 < np; i++) tf::d2h(&at[(size_t)i], (const uint32_t *)row_off->p + (i + 1) * threshold, 4);
          tf::sync();
          for (int64_t i = 0; i + 1 < np; i++) part_ends[i] = at[(size_t)i];
          part_ends[np - 1] = res->size;
        }
      }
    }
    if (nparts) *nparts = np;
    *out = res.release();
    return TFGPU_OK;
  } catch (const tf::Error &e) { g_row_offsets_out = nullptr; return tf::fail(e.code, e.what()); }
  catch (const std::bad_alloc &) { g_row_offsets_out = nullptr; return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
  catch (const std::exception &e) { g_row_offsets_out = nullptr; return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}

// ======================================================================================================
// The queue serializers whose messages ARE column bytes, and the Kafka writer's partitioner: RawColumnSerializer
// (raw_column_serializer.go:21-73), MirrorSerializer (mirror_serializer.go:15-52; changeitem/mirror.go:23-87), splitByTablePartID
// (split.go:5-12), kafka-go's Hash balancer (vendor_patched/github.com/segmentio/kafka-go/balancer.go:153-181).  No kernels of their
// own: a message value is the named column's cell, so the result is that column's (offsets, bytes) — shared when every row has a
// value, gathered over the rows that do otherwise — and the rules around it are the reference's, row for row.
// ======================================================================================================
namespace tf {
std::unique_ptr<tfgpu_dbatch> gather_rows(const tfgpu_dbatch &in, const Buf &sel, int64_t m);  // tf_transform.hip
static const char *const RAW_DATA_COLUMNS[7] = {"topic", "partition", "seq_no", "write_time", "data", "meta", "sequence_key"};
static bool is_mirror_batch(const tfgpu_dbatch &b) {  // ChangeItem.IsMirror (change_item.go:385-395): ColumnNames are exactly RawDataColumns, in order
  if (b.cols.size() != 7) return false;
  for (size_t i = 0; i < 7; i++) if (b.cols[i].name != RAW_DATA_COLUMNS[i]) return false;
  return true;
}
// GetSequenceKey compares the TableSchema POINTER with RawDataSchema (mirror.go:71-73); value identity is the closest a columnar
// batch has: names, types, key / required flags and original types, position by position
static bool is_raw_data_schema(const tfgpu_schema &s) {
  static const struct { const char *name; int dtype; uint32_t flags; const char *ot; } W[7] = {
      {"topic", TFGPU_T_UTF8, TFGPU_COL_KEY | TFGPU_COL_REQUIRED, ""}, {"partition", TFGPU_T_UINT32, TFGPU_COL_KEY | TFGPU_COL_REQUIRED, ""},
      {"seq_no", TFGPU_T_UINT64, TFGPU_COL_KEY | TFGPU_COL_REQUIRED, ""}, {"write_time", TFGPU_T_DATETIME, TFGPU_COL_KEY | TFGPU_COL_REQUIRED, ""},
      {"data", TFGPU_T_UTF8, 0, "mirror:binary"}, {"meta", TFGPU_T_ANY, 0, ""}, {"sequence_key", TFGPU_T_BYTES, 0, ""}};
  if (s.ncols != 7) return false;
  for (int i = 0; i < 7; i++) {
    const tfgpu_colschema &c = s.cols[i];
    if (std::string(c.name ? c.name : "") != W[i].name || c.dtype != W[i].dtype || (c.flags & (TFGPU_COL_KEY | TFGPU_COL_REQUIRED)) != W[i].flags || std::string(c.original_type ? c.original_type : "") != W[i].ot) return false;
  }
  return true;
}
// the cells of one text column as messages: every row (all_rows) or the rows that hold a value; offsets to the host
static void column_messages(const tfgpu_dbatch &b, const DColumn &c, bool all_rows, tfgpu_dbuf **values, uint32_t *msg_start, uint8_t *nil, int64_t cap, int64_t *nmsg) {
  const int64_t n = b.nrows;
  std::vector<uint8_t> bm;
  int64_t valid = n;
  if (c.validity && n) {
    bm.resize((size_t)(n + 7) / 8);
    d2h(bm.data(), c.validity->p, bm.size());
    sync();
    valid = 0;
    for (int64_t r = 0; r < n; r++) valid += (bm[(size_t)r >> 3] >> (r & 7)) & 1;
  }
  const int64_t m = all_rows ? n : valid;
  if (m > cap) throw Error(TFGPU_ERR_INVALID, "queue: more messages than msg_start has room for");
  auto res = std::make_unique<tfgpu_dbuf>();
  if (all_rows || valid == n) {  // the column's own buffers (a nil cell has no bytes)
    res->mem = c.payload(); res->size = c.data_len;
    if (m) d2h(msg_start, c.offsets->p, (size_t)(m + 1) * 4); else msg_start[0] = 0;
    if (nil) for (int64_t r = 0; r < n; r++) nil[r] = bm.empty() ? 0 : (uint8_t)(1 ^ ((bm[(size_t)r >> 3] >> (r & 7)) & 1));
    sync();
  } else {
    std::vector<int32_t> sel;
    for (int64_t r = 0; r < n; r++) if ((bm[(size_t)r >> 3] >> (r & 7)) & 1) sel.push_back((int32_t)r);
    tfgpu_dbatch one;
    one.nrows = n; one.cols.push_back(c); one.cols[0].validity = nullptr;
    Buf bsel = upload_small(sel.data(), std::max<size_t>(sel.size(), 1) * 4);
    std::unique_ptr<tfgpu_dbatch> g = gather_rows(one, bsel, m);
    const DColumn &gc = g->cols[0];
    res->mem = gc.payload(); res->size = gc.data_len;
    if (m) d2h(msg_start, gc.offsets->p, (size_t)(m + 1) * 4); else msg_start[0] = 0;
    sync();
  }
  *nmsg = m;
  *values = res.release();
}
}  // namespace tf

extern "C" int tfgpu_queue_raw_column(const tfgpu_dbatch *b, const char *column, const tfgpu_schema *schema, tfgpu_dbuf **values, uint32_t *msg_start, int64_t cap, int64_t *nmsg) {
  try {
  tf::dense(b);  // its rows may still be a selection (tfgpu_dbatch::pending)
    if (!b || !column || !values || !msg_start || !nmsg || cap < 0) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_raw_column: null argument");
    tf::Context &cx = tf::ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    *nmsg = 0; *values = nullptr; msg_start[0] = 0;
    auto none = [&]() { auto r = std::make_unique<tfgpu_dbuf>(); r->mem = tf::dalloc(64); r->size = 0; *values = r.release(); return TFGPU_OK; };  // every row skipped with a warning
    const tf::DColumn *c = nullptr;
    for (auto &x : b->cols) if (x.name == column) { c = &x; break; }
    if (!c) return none();  // "column is not found in the change item" (:37-41)
    int dtype = c->dtype;
    if (schema) {
      const tfgpu_colschema *sc = nullptr;
      for (int i = 0; i < schema->ncols; i++) if (schema->cols[i].name && std::string(schema->cols[i].name) == column) { sc = &schema->cols[i]; break; }
      if (!sc) return none();  // "table schema does not contain column" (:43-49)
      dtype = sc->dtype;
    } else if (!b->schema.empty()) {
      bool found = false;
      for (auto &pr : b->schema) if (pr.first == column) { dtype = pr.second; found = true; break; }
      if (!found) return none();
    }
    if (dtype != TFGPU_T_UTF8 && dtype != TFGPU_T_BYTES) return none();                      // "unexpected column type" (:51-55)
    if (c->repr != TFGPU_R_STRING && c->repr != TFGPU_R_BYTES) return none();                // "unexpected column value type" (:57-70): nil included, per row below
    tf::materialize(*b);
    tf::column_messages(*b, *c, false, values, msg_start, nullptr, cap, nmsg);
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}

extern "C" int tfgpu_queue_mirror(const tfgpu_dbatch *b, const tfgpu_schema *schema, tfgpu_dbuf **values, uint32_t *msg_start, tfgpu_dbuf **keys, uint32_t *key_start,
                                  uint8_t *key_nil, int64_t cap, int64_t *nmsg) {
  try {
  tf::dense(b);  // its rows may still be a selection (tfgpu_dbatch::pending)
    if (!b || !values || !msg_start || !keys || !key_start || !key_nil || !nmsg || cap < 0) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_mirror: null argument");
    tf::Context &cx = tf::ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    if (!tf::is_mirror_batch(*b)) return tf::fail(TFGPU_ERR_INVALID, "MirrorSerializer should be used only with 'Mirror' changeItems");  // mirror_serializer.go:16-18
    if (schema && !tf::is_raw_data_schema(*schema)) return tf::fail(TFGPU_ERR_INVALID, "unable to get sequence key: changeItem should be 'mirror'");  // mirror.go:71-73
    const tf::DColumn &data = b->cols[4], &key = b->cols[6];
    if (data.repr != TFGPU_R_STRING && data.repr != TFGPU_R_BYTES) return tf::fail(TFGPU_ERR_INVALID, "unable to get message: unexpected data type, expected string or []byte");  // mirror.go:78-87
    tf::materialize(*b);
    int64_t nk = 0, nv = 0;
    std::vector<uint8_t> dnil((size_t)std::max<int64_t>(b->nrows, 1));
    tfgpu_dbuf *v = nullptr, *k = nullptr;
    tf::column_messages(*b, data, true, &v, msg_start, dnil.data(), cap, &nv);
    std::unique_ptr<tfgpu_dbuf> hv(v);
    for (int64_t r = 0; r < b->nrows; r++) if (dnil[(size_t)r]) return tf::fail(TFGPU_ERR_INVALID, "unable to get message: unexpected data type: <nil>, expected string or []byte");
    if (key.repr == TFGPU_R_BYTES) tf::column_messages(*b, key, true, &k, key_start, key_nil, cap, &nk);
    else {  // every producer passes a []byte; a null is MakeRawMessage's typed nil — anything else fails the `.([]byte)` assertion (mirror.go:74)
      std::vector<uint8_t> bm((size_t)(b->nrows + 7) / 8, 0);
      if (key.validity && b->nrows) { tf::d2h(bm.data(), key.validity->p, bm.size()); tf::sync(); }
      for (int64_t r = 0; r < b->nrows; r++) if (!key.validity || ((bm[(size_t)r >> 3] >> (r & 7)) & 1)) return tf::fail(TFGPU_ERR_INVALID, "interface conversion: sequence_key is not []byte");
      auto e = std::make_unique<tfgpu_dbuf>(); e->mem = tf::dalloc(64); e->size = 0; k = e.release();
      for (int64_t r = 0; r <= b->nrows; r++) key_start[r] = 0;
      for (int64_t r = 0; r < b->nrows; r++) key_nil[r] = 1;
    }
    *values = hv.release(); *keys = k; *nmsg = nv;
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}

// splitByTablePartID for one table (split.go:5-12): rows grouped by ChangeItem.PartID; Go maps iterate in no defined order — the groups
// come out by first appearance.  order[nrows]: the rows, group after group, input order inside a group; group_rows / group_part_id per group.
extern "C" int tfgpu_queue_part_groups(const tfgpu_dbatch *b, int32_t *order, int64_t *group_rows, uint32_t *group_part_id, int64_t cap, int64_t *ngroups) {
  try {
  tf::dense(b);  // its rows may still be a selection (tfgpu_dbatch::pending)
    if (!b || !order || !group_rows || !group_part_id || !ngroups || cap < 1) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_part_groups: bad argument");
    tf::Context &cx = tf::ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    const int64_t n = b->nrows;
    if (!b->part_id) { for (int64_t r = 0; r < n; r++) order[r] = (int32_t)r; group_rows[0] = n; group_part_id[0] = 0; *ngroups = 1; return TFGPU_OK; }
    std::vector<uint32_t> pid((size_t)n);
    if (n) { tf::d2h(pid.data(), b->part_id->p, (size_t)n * 4); tf::sync(); }
    // two passes over the ids: the groups by first appearance with their sizes (the last row's group is tried first: a partitioned batch holds
    // its parts in runs; a thousand parts looked up by a linear walk per row was 10^8 compares for 3 x 10^5 rows), then every row to its slot
    std::vector<uint32_t> ids; std::vector<int64_t> count;
    std::unordered_map<uint32_t, uint32_t> slot;
    std::vector<uint32_t> grp((size_t)n);
    uint32_t last_id = 0, last_g = 0; bool have = false;
    for (int64_t r = 0; r < n; r++) {
      const uint32_t id = pid[(size_t)r];
      uint32_t g;
      if (have && id == last_id) g = last_g;
      else {
        auto it = slot.find(id);
        if (it == slot.end()) {
          if ((int64_t)ids.size() >= cap) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_part_groups: more groups than the arrays hold");
          g = (uint32_t)ids.size(); slot.emplace(id, g); ids.push_back(id); count.push_back(0);
        } else g = it->second;
        last_id = id; last_g = g; have = true;
      }
      grp[(size_t)r] = g; count[g]++;
    }
    std::vector<int64_t> at(ids.size() + 1, 0);
    for (size_t g = 0; g < ids.size(); g++) { at[g + 1] = at[g] + count[g]; group_rows[g] = count[g]; group_part_id[g] = ids[g]; }
    for (int64_t r = 0; r < n; r++) order[at[grp[(size_t)r]]++] = (int32_t)r;
    *ngroups = (int64_t)ids.size();
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}

// kafka-go's Hash balancer with its default hasher (balancer.go:153-181; Sarama's hashPartitioner): FNV-1a(32) of the key, taken as an
// int32, Go's truncated remainder, a negative result negated.  -1: a nil key (round robin there).
extern "C" int32_t tfgpu_kafka_hash_partition(const void *key, int64_t len, int32_t npartitions) {
  if (!key || len < 0 || npartitions <= 0) return -1;
  uint32_t h = 0x811C9DC5u;
  for (int64_t i = 0; i < len; i++) h = (h ^ ((const uint8_t *)key)[i]) * 0x01000193u;
  int32_t p = (int32_t)h % npartitions;
  return p < 0 ? -p : p;
}
namespace tf {
__global__ void __launch_bounds__(256) kafka_partition_kernel(const uint32_t *off, const uint8_t *data, const uint8_t *validity, int64_t n, int32_t np, int32_t *out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  if (validity && !((validity[r >> 3] >> (r & 7)) & 1)) { out[r] = -1; return; }
  uint32_t h = 0x811C9DC5u;
  for (uint32_t i = off[r]; i < off[r + 1]; i++) h = (h ^ data[i]) * 0x01000193u;
  const int32_t p = (int32_t)h % np;
  out[r] = p < 0 ? -p : p;
}
}  // namespace tf
// … for a column of keys at once (the Mirror serializer's `sequence_key`, a RawColumn key column): partitions[r], -1 for a nil key
extern "C" int tfgpu_kafka_partitions(const tfgpu_dbatch *b, const char *key_column, int32_t npartitions, int32_t *partitions) {
  try {
  tf::dense(b);  // its rows may still be a selection (tfgpu_dbatch::pending)
    if (!b || !key_column || !partitions || npartitions <= 0) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_kafka_partitions: bad argument");
    tf::Context &cx = tf::ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    const tf::DColumn *c = nullptr;
    for (auto &x : b->cols) if (x.name == key_column) { c = &x; break; }
    if (!c) return tf::fail(TFGPU_ERR_INVALID, std::string("tfgpu_kafka_partitions: no column ") + key_column);
    if (c->repr != TFGPU_R_STRING && c->repr != TFGPU_R_BYTES) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_kafka_partitions: the key column holds no bytes");
    tf::materialize(*b);
    const int64_t n = b->nrows;
    if (!n) return TFGPU_OK;
    tf::Buf out = tf::dalloc((size_t)n * 4);
    tf::kafka_partition_kernel<<<(unsigned)((n + 255) / 256), 256, 0, cx.stream>>>(tf::ptr<uint32_t>(c->offsets), tf::ptr<uint8_t>(c->payload()), tf::ptr<uint8_t>(c->validity), n, npartitions, tf::ptr<int32_t>(out));
    tf::d2h(partitions, out->p, (size_t)n * 4);
    tf::sync();
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}

// ======================================================================================================
// tfgpu_queue_serialize — pkg/serializer/queue: NativeSerializer / JSONSerializer for one table's rows.
// Same four passes as tfgpu_serialize_ex over a longer cell list (header, constants, values, old-key values, trailer);
// between the length pass and the scan the element lengths go to the host once, where BatchNative / BatchJSON's greedy
// cut (native_batcher.go:10-63, json_batcher.go:11-66) runs over them — it is the reference's own sequential loop over
// integers — and the cut comes back as one flag byte per row.
// ======================================================================================================
namespace tf {

static const char *const QDTYPE_NAMES[] = {"", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float", "double",
                                           "boolean", "string", "utf8", "date", "datetime", "timestamp", "interval", "any"};

static void json_string_array(std::string &out, const std::vector<std::string> &v) {
  out += '[';
  for (size_t i = 0; i < v.size(); i++) { if (i) out += ','; json_key(out, v[i], true); }
  out += ']';
}
// json.Marshal([]ColSchema): col_schema.go:14-29 (Properties omitempty; the fields the ABI schema does not carry are zero)
static void table_schema_json(std::string &out, const tfgpu_queue_options *o, const tfgpu_dbatch *b) {
  struct Col { std::string name, path, ot; int dtype; bool key, req; std::string ts, tn, ex, props; bool fake; };
  std::vector<Col> cols;
  if (o->table_schema) {
    for (int i = 0; i < o->table_schema->ncols; i++) {
      const tfgpu_colschema &c = o->table_schema->cols[i];
      cols.push_back({c.name ? c.name : "", c.path ? c.path : "", c.original_type ? c.original_type : "", c.dtype, (c.flags & TFGPU_COL_KEY) != 0, (c.flags & TFGPU_COL_REQUIRED) != 0,
                      c.table_schema ? c.table_schema : "", c.table_name ? c.table_name : "", c.expression ? c.expression : "", c.properties_json ? c.properties_json : "",
                      (c.flags & TFGPU_COL_FAKE_KEY) != 0});
    }
  } else {
    auto is_key = [&](const std::string &n) { for (auto &k : b->key_names) if (k == n) return true; return false; };
    if (!b->schema.empty()) for (auto &pr : b->schema) cols.push_back({pr.first, "", "", pr.second, is_key(pr.first), false, "", "", "", "", false});
    else for (auto &c : b->cols) cols.push_back({c.name, "", "", c.dtype, is_key(c.name), false, "", "", "", "", false});
  }
  if (cols.empty()) return;  // len(c.TableSchema.columns) == 0: the key is omitted
  out += ",\"table_schema\":[";
  for (size_t i = 0; i < cols.size(); i++) {
    const Col &c = cols[i];
    if (i) out += ',';
    // json.Marshal(ColSchema): every field by its tag (col_schema.go:14-29), `properties` omitempty
    out += "{\"table_schema\":"; json_key(out, c.ts, true);
    out += ",\"table_name\":"; json_key(out, c.tn, true);
    out += ",\"path\":"; json_key(out, c.path, true);
    out += ",\"name\":"; json_key(out, c.name, true);
    out += ",\"type\":"; json_key(out, c.dtype > 0 && c.dtype < TFGPU_T__COUNT ? QDTYPE_NAMES[c.dtype] : "", true);
    out += ",\"key\":"; out += c.key ? "true" : "false";
    out += ",\"fake_key\":"; out += c.fake ? "true" : "false";
    out += ",\"required\":"; out += c.req ? "true" : "false";
    out += ",\"expression\":"; json_key(out, c.ex, true);
    out += ",\"original_type\":"; json_key(out, c.ot, true);
    if (!c.props.empty()) { out += ",\"properties\":"; out += c.props; }
    out += '}';
  }
  out += ']';
}

// BatchNative / BatchJSON over one group: flags[i] |= 1 first, 2 last of its message
static void queue_cut_plan(const tfgpu_queue_options *o, const uint32_t *elem_len, int64_t n, uint32_t lead, uint8_t *flags) {
  if (n == 0) return;
  if (!o->batching_enabled) { for (int64_t i = 0; i < n; i++) flags[i] = 3; return; }
  const uint64_t wrap = o->format == TFGPU_QFMT_NATIVE ? 2 : 0;
  auto emit = [&](int64_t a, int64_t z) { flags[a] |= 1; flags[z - 1] |= 2; };
  int64_t start = 0; uint64_t sum = 0;
  for (int64_t i = 0; i < n; i++) {
    const uint64_t len = elem_len[i] - lead;
    const int64_t num_new = i - start + 1;
    bool violates = false;
    if (o->max_message_size != 0 && sum + (uint64_t)(num_new - 1) + len + wrap > (uint64_t)o->max_message_size) violates = true;
    if (o->max_change_items != 0 && num_new > o->max_change_items) violates = true;
    if (violates) {
      if (i - start == 0) { emit(start, start + 1); start = i + 1; sum = 0; }
      else { emit(start, i); start = i; sum = len; }
    } else sum += len;
  }
  if (start != n) emit(start, n);
}

// The cut plan AND the messages' places in one sweep per MESSAGE instead of two loops per ELEMENT (the two loops were 0.25-0.3 ms of
// host time per 3 x 10^5 rows: a fifth of a configs[4] pass, profiles/r31b_timeline_configs4.txt).  Q[i] = sum over j < i of (len_j + 1)
// turns the batchers' running test  sum + (num_new - 1) + len + wrap > max  (native_batcher.go:10-63, json_batcher.go:11-66; queue_cut_plan
// above is its literal restatement) into  Q[i + 1] - Q[start] + wrap - 1 > max, monotone in i: the first violating element of a message
// is found by a galloping search from its start.  An element that opens a message is never tested alone by the batchers — but an
// element too large for a message of its own violates again with the next one and is cut off as a message of one either way, so
// restarting the search AT it emits the same ranges (len >= 0).  A message's place: the element lengths in front of it plus one byte per
// message closed (native: ']' joins, ',' between elements is counted in the lengths) or per element not closing one (JSON: '\n').
static int64_t queue_cut_and_place(const tfgpu_queue_options *o, const uint32_t *elen, int64_t n, const std::vector<int64_t> &gstart, uint32_t lead, bool native,
                                   uint8_t *flags, uint64_t *msg_start, int64_t *msg_row, int64_t cap, uint64_t *total_out) {
  static thread_local std::vector<uint64_t> qstore;
  if (qstore.size() < (size_t)n + 1) qstore.resize((size_t)n + 1);
  uint64_t *Q = qstore.data();
  {
    uint64_t q = 0;
    Q[0] = 0;
    for (int64_t i = 0; i < n; i++) { q += (uint64_t)(uint32_t)(elen[i] - lead) + 1u; Q[i + 1] = q; }
  }
  auto pe = [&](int64_t i) { return Q[i] + (uint64_t)i * lead - (uint64_t)i; };  // sum of elen[0 .. i)
  int64_t nm = 0;
  auto emit = [&](int64_t a, int64_t z) {
    flags[a] |= 1; flags[z - 1] |= 2;
    if (nm < cap) { msg_start[nm] = pe(a) + (native ? (uint64_t)nm : (uint64_t)(a - nm)); msg_row[nm] = a; }
    nm++;
  };
  const uint64_t wrap = o->format == TFGPU_QFMT_NATIVE ? 2 : 0, maxsz = (uint64_t)o->max_message_size;
  for (size_t g = 0; g + 1 < gstart.size(); g++) {
    const int64_t g0 = gstart[g], g1 = gstart[g + 1];
    if (g0 == g1) continue;
    if (!o->batching_enabled) { for (int64_t i = g0; i < g1; i++) emit(i, i + 1); continue; }
    int64_t start = g0;
    while (start < g1) {
      const uint64_t q0 = Q[start];
      auto too_big = [&](int64_t i) { return Q[i + 1] - q0 + wrap - 1 > maxsz; };  // element i joins the message that starts at `start`
      int64_t lim = g1;  // the first element the item count refuses
      if (o->max_change_items != 0) lim = o->max_change_items > 0 ? std::min<int64_t>(g1, start + o->max_change_items) : start;
      int64_t iv = lim;
      if (maxsz != 0 && lim > start) {
        if (too_big(start)) iv = start;
        else {  // gallop: element lo fits, the first that does not lies in (lo, hi]
          int64_t lo = start, step = 1, hi = -1;
          while (lo + step < lim) { if (too_big(lo + step)) { hi = lo + step; break; } lo += step; step <<= 1; }
          if (hi < 0) { if (lo + 1 < lim && too_big(lim - 1)) hi = lim - 1; else hi = lim; }
          if (hi < lim) { while (hi - lo > 1) { const int64_t mid = lo + ((hi - lo) >> 1); if (too_big(mid)) hi = mid; else lo = mid; } }
          iv = hi;
        }
      }
      if (iv >= g1) { emit(start, g1); break; }
      if (iv == start) { emit(start, start + 1); start = iv + 1; }
      else { emit(start, iv); start = iv; }
    }
  }
  *total_out = pe(n) + (native ? (uint64_t)nm : (uint64_t)(n - nm));
  // TFGPU_CUT_PLAN_CHECK=1 (tests): the literal restatement of the batchers' loop must give the same flags
  const char *chk = std::getenv("TFGPU_CUT_PLAN_CHECK");  // (read every call: a test switches it on in a process that has serialized before)
  if (chk && chk[0] == '1') {
    std::vector<uint8_t> f2((size_t)n, 0);
    for (size_t g = 0; g + 1 < gstart.size(); g++) queue_cut_plan(o, elen + gstart[g], gstart[g + 1] - gstart[g], lead, f2.data() + gstart[g]);
    if (n && std::memcmp(f2.data(), flags, (size_t)n) != 0) throw Error(TFGPU_ERR_DEVICE, "tfgpu_queue_serialize: internal: the galloping cut plan and the batchers' loop disagree");
  }
  return nm;
}

template <class T> static const T *meta_array(const tfgpu_row_meta *m, const T *p, size_t count, std::vector<Buf> &keep) {
  if (!p) return nullptr;
  if (m->mem == TFGPU_MEM_DEVICE) return p;
  Buf d = dalloc(count * sizeof(T) + 16);
  h2d(d->p, p, count * sizeof(T));
  keep.push_back(d);
  return reinterpret_cast<const T *>(d->p);
}

}  // namespace tf

// The element lengths for the host's cut plan, and the plan's flags back: through the lane's page-locked ring when they fit (a pageable
// vector is a staged copy behind a fresh allocation: 0.2 ms of a 2^18-row batch), the flag bytes kept across calls.
struct CutBuffers {
  const uint32_t *elen = nullptr;
  std::vector<uint32_t> elen_own;
  uint8_t *flags = nullptr;
  static std::vector<uint8_t> &flag_store() { static thread_local std::vector<uint8_t> v; return v; }
  CutBuffers(const uint32_t *dev_len, int64_t n) {
    using namespace tf;
    if ((size_t)n * 4 <= (2u << 20)) elen = d2h_u32(dev_len, (size_t)n);
    else { elen_own.resize((size_t)n); d2h(elen_own.data(), dev_len, (size_t)n * 4); elen = elen_own.data(); }
    tf::sync();
    std::vector<uint8_t> &f = flag_store();
    if (f.size() < (size_t)n) f.resize((size_t)n);
    std::memset(f.data(), 0, (size_t)n);
    flags = f.data();
  }
  void upload_flags(void *dev, int64_t n) const { using namespace tf; if ((size_t)n <= (1u << 20)) h2d_small(dev, flags, (size_t)n); else { h2d(dev, flags, (size_t)n); tf::sync(); } }
};

// queue JSON (BatchJSON over pkg/serializer/json.go rows) by chunk walks; the caller holds the lane's mutex and has run the checks
static int queue_json_chunks(const tfgpu_queue_options *o, const tfgpu_dbatch *b, const std::vector<int64_t> &gstart, tfgpu_dbuf **values, uint64_t *msg_start, int64_t *msg_row, int64_t cap, int64_t *nmsg) {
  using namespace tf;
  hipStream_t st = ctx().stream;
  const int64_t n = b->nrows;
  const int ncols = (int)b->cols.size();
  std::vector<int> order((size_t)ncols);
  for (int i = 0; i < ncols; i++) order[(size_t)i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return b->cols[(size_t)x].name < b->cols[(size_t)y].name; });
  for (int i = 0; i + 1 < ncols; i++) if (b->cols[(size_t)order[(size_t)i]].name == b->cols[(size_t)order[(size_t)i + 1]].name)
    return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_queue_serialize: duplicate column names");
  std::string blob;
  std::vector<SCol> sc((size_t)ncols);
  for (int j = 0; j < ncols; j++) {
    const DColumn &c = b->cols[(size_t)order[(size_t)j]];
    SCol &x = sc[(size_t)j];
    x.c = dcol_of(c);
    x.pre_off = (uint32_t)blob.size();
    if (j) blob += ',';
    json_key(blob, c.name); blob += ':';
    x.pre_len = (uint32_t)blob.size() - x.pre_off;
    x.ch_flags = 0; x.prec = 0;
  }
  Buf bsc = upload_const(sc.data(), sc.size() * sizeof(SCol)), bblob = upload_const(blob.data(), blob.size());
  Buf row_len = dalloc((size_t)(n + 1) * 4 + 16), mflags = dalloc((size_t)n + 16), tot64 = dalloc_zero(8);
  ChunkSetup cs = make_chunk_setup(b, TFGPU_FMT_JSON, order, sc, blob, n);
  SerParams p{};
  p.cols = ptr<SCol>(bsc); p.ncols = ncols; p.nrows = n; p.format = TFGPU_FMT_JSON; p.any_as_string = 0; p.closing_newline = 0;
  p.blob = ptr<uint8_t>(bblob); p.row_len = ptr<uint32_t>(row_len); p.last_present = nullptr;
  p.total64 = reinterpret_cast<unsigned long long *>(tot64->p);
  p.q.qformat = TFGPU_QFMT_JSON; p.q.msg_flags = nullptr;  // (no frame yet: the length passes give the ELEMENT lengths the cut plan is made of)
  const unsigned grid = ser_grid(cs.cp);
  { KernelTimer t("ser_chunk_len"); ser_chunk_len<TFGPU_FMT_JSON><<<grid, 256, 0, st>>>(p, cs.cp); }
  { KernelTimer t("ser_chunk_layout"); ser_chunk_layout<TFGPU_FMT_JSON><<<blocks(n), 256, 0, st>>>(p, cs.cp); }
  // ---- the cut plan (host: the batchers' sequential greedy loop over the element lengths) ----
  CutBuffers cb(p.row_len, n);
  const uint32_t *elen = cb.elen; uint8_t *hflags = cb.flags;
  uint64_t total = 0;
  const int64_t nm = queue_cut_and_place(o, elen, n, gstart, 0u, false, hflags, msg_start, msg_row, cap, &total);
  if (nm > cap) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_serialize: more messages than msg_start / msg_row hold");
  if (total >> 32) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_queue_serialize: output exceeds 4 GiB; split the batch by rows");
  msg_start[nm] = total; msg_row[nm] = n;
  cb.upload_flags(mflags->p, n);
  p.q.msg_flags = ptr<uint8_t>(mflags);
  ser_queue_tail_len<<<blocks(n), 256, 0, st>>>(p);
  exclusive_scan_u32(p.row_len, p.row_len, n, true);
  const uint32_t *htot = d2h_u32(p.row_len + n);
  tf::sync();  // (hflags may die after this)
  if (*htot != (uint32_t)total) return tf::fail(TFGPU_ERR_DEVICE, "tfgpu_queue_serialize: internal: device and host disagree on the output size");
  auto res = std::make_unique<tfgpu_dbuf>();
  res->size = total;
  res->mem = dalloc(res->size + 64);
  p.out = ptr<uint8_t>(res->mem);
  { KernelTimer t("ser_chunk_write"); ser_chunk_write<TFGPU_FMT_JSON><<<grid, 256, 4 * WAVE_IMG_BYTES + 64, st>>>(p, cs.cp); }
  tf::sync();
  *nmsg = nm;
  *values = res.release();
  return TFGPU_OK;
}

extern "C" int tfgpu_queue_serialize(const tfgpu_queue_options *o, const tfgpu_dbatch *b, const tfgpu_row_meta *meta, tfgpu_dbuf **values,
                                     uint64_t *msg_start, int64_t *msg_row, int64_t cap, int64_t *nmsg) {
  try {
  tf::dense(b, o && o->format == TFGPU_QFMT_NATIVE);  // its rows may still be a selection (tfgpu_dbatch::pending); the native format writes every row's own ColumnNames (ABSENT cells)
    if (!o || !b || !values || !nmsg || (cap > 0 && (!msg_start || !msg_row))) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_serialize: null argument");
    if (o->format != TFGPU_QFMT_NATIVE && o->format != TFGPU_QFMT_JSON) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_serialize: unknown format");
    const bool native = o->format == TFGPU_QFMT_NATIVE;
    Context &cx = ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    hipStream_t st = cx.stream;
    const int64_t n = b->nrows;
    const int ncols = (int)b->cols.size(), nold = native ? (int)b->old_keys.size() : 0;
    // groups: contiguous row runs, each serialised on its own (splitByTablePartID)
    std::vector<int64_t> gstart{0};
    if (o->group_rows) { for (int g = 0; g < o->ngroups; g++) { if (o->group_rows[g] < 0) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_serialize: negative group size"); gstart.push_back(gstart.back() + o->group_rows[g]); } }
    else gstart.push_back(n);
    if (gstart.back() != n) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_serialize: group_rows do not add up to the batch's rows");
    if (meta && meta->n < 0) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_serialize: row meta with a negative length");
    auto res = std::make_unique<tfgpu_dbuf>();
    if (n == 0) { res->mem = dalloc(64); res->size = 0; *values = res.release(); *nmsg = 0; if (cap >= 0 && msg_start) { msg_start[0] = 0; msg_row[0] = 0; } return TFGPU_OK; }
    materialize(*b);

    // ---- what must fail the call before any text exists: kinds, meta range, NaN / Inf ----
    Buf flags = dalloc_zero(8);
    queue_check_kernel<<<std::min(blocks(n), QCHECK_BLOCKS), 256, 0, st>>>(ptr<uint8_t>(b->kind), ptr<int32_t>(b->src_row), n, (native && meta) ? meta->n : -1, ptr<uint32_t>(flags));
    auto nonfinite = [&](const DColumn &c) {
      if (c.repr != TFGPU_R_FLOAT32 && c.repr != TFGPU_R_FLOAT64) return;
      float_nonfinite_kernel<<<blocks(n), 256, 0, st>>>(c.values->p, n, c.repr == TFGPU_R_FLOAT32, ptr<uint8_t>(c.validity), ptr<uint32_t>(flags) + 1);
    };
    for (auto &c : b->cols) nonfinite(c);
    for (int k = 0; k < nold; k++) nonfinite(b->old_keys[(size_t)k]);
    bool has_nonrow = false;
    {
      const uint32_t *h = d2h_u32(flags->p, 2);
      tf::sync();
      has_nonrow = (h[0] & 8u) != 0;
      if (h[0] & 16u) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_serialize: src_row outside the row meta");
      if (native && (h[0] & 8u)) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_queue_serialize: non-row kinds are serialised by the stock NativeSerializer");
      if (!native && (h[0] & 6u)) return tf::fail(TFGPU_ERR_UNSUPPORTED, "JsonSerializer: unsupported kind: update / delete (json.go:54-56)");
      if (h[1]) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_queue_serialize: json: unsupported value: NaN or Inf");
    }

    // The JSON format's element is the text serializer's row (pkg/serializer/json.go: one map per item, keys sorted) — it goes through
    // the same chunk walks (ser_chunk_len / layout / write<JSON>, twice as fast per byte as the per-cell passes below), with the
    // BatchJSON frame as the row tail: "\n" after every element but the last of its message, known once the cut plan is.  Non-row
    // items (an empty element each) keep the per-cell passes.
    static const bool cell_passes = [] { const char *e = std::getenv("TFGPU_SER_SCATTER"); return e && *e == '1'; }();
    if (!native && !has_nonrow && !cell_passes) return queue_json_chunks(o, b, gstart, values, msg_start, msg_row, cap, nmsg);

    // ---- the cell list ----
    const bool ragged = native && has_absent(*b);
    int ord_name0 = 0, ord_val0 = 0;
    std::string blob;
    std::vector<SCol> sc;
    auto add = [&](uint32_t kind, uint32_t apply, const std::string &pre, const DColumn *col) {
      SCol s{};
      if (col) s.c = dcol_of(*col);
      s.pre_off = (uint32_t)blob.size(); blob += pre; s.pre_len = (uint32_t)pre.size();
      s.kind = kind; s.apply = apply;
      sc.push_back(s);
    };
    QueueParams q{};
    q.qformat = o->format;
    std::vector<int> order((size_t)ncols);
    for (int i = 0; i < ncols; i++) order[(size_t)i] = i;
    if (native) {
      add(QC_HEADER, QA_ALWAYS, "", nullptr);
      { std::string t = ",\"schema\":"; json_key(t, b->ns, true); t += ",\"table\":"; json_key(t, b->table, true); t += ",\"part\":"; q.hdr_off = (uint32_t)blob.size(); q.hdr_len = (uint32_t)t.size(); blob += t; }
      if (!ragged) {
        { std::string t = ",\"columnnames\":"; std::vector<std::string> nm; for (auto &c : b->cols) nm.push_back(c.name); json_string_array(t, nm); if (ncols) t += ",\"columnvalues\":["; add(QC_CONST, QA_NAMES, t, nullptr); }
        for (int j = 0; j < ncols; j++) add(QC_VALUE, QA_NAMES, j ? "," : "", &b->cols[(size_t)j]);
        if (ncols) add(QC_CONST, QA_NAMES, "]", nullptr);
      } else {  // rows list different columns (tfgpu_column.absent): every row writes its own columnnames, and columnvalues only when it lists any
        auto listed = [&](uint32_t kind, const std::string &pre, int j) {
          add(kind, QA_LISTED, pre, kind == QC_VALUE ? &b->cols[(size_t)j] : nullptr);
          sc.back().absent = ptr<uint8_t>(b->cols[(size_t)j].absent); sc.back().prec = (uint32_t)j;
        };
        add(QC_CONST, QA_NAMES, ",\"columnnames\":[", nullptr);
        ord_name0 = (int)sc.size();
        for (int j = 0; j < ncols; j++) { std::string t = ","; json_key(t, b->cols[(size_t)j].name, true); listed(QC_NAME, t, j); }
        add(QC_CONST, QA_NAMES, "]", nullptr);
        add(QC_CONST, QA_ANY_LISTED, ",\"columnvalues\":[", nullptr);
        ord_val0 = (int)sc.size();
        for (int j = 0; j < ncols; j++) listed(QC_VALUE, ",", j);
        add(QC_CONST, QA_ANY_LISTED, "]", nullptr);
      }
      { std::string t; if (!o->omit_table_schema) { if (o->table_schema_json) { t += ",\"table_schema\":"; t += o->table_schema_json; } else table_schema_json(t, o, b); } t += ",\"oldkeys\":{"; add(QC_CONST, QA_ALWAYS, t, nullptr); }
      for (int k = 0; k < nold; k++) {
        std::string t;
        if (k == 0) {
          std::vector<std::string> nm, ty; for (auto &c : b->old_keys) nm.push_back(c.name);
          t = "\"keynames\":"; json_string_array(t, nm);
          if (o->old_key_types) { for (int i = 0; i < nold; i++) ty.push_back(o->old_key_types[i] ? o->old_key_types[i] : ""); t += ",\"keytypes\":"; json_string_array(t, ty); }
          t += ",\"keyvalues\":[";
        } else t = ",";
        add(QC_VALUE, QA_OLD, t, &b->old_keys[(size_t)k]);
      }
      add(QC_TRAILER, QA_ALWAYS, "", nullptr);
    } else {  // one pkg/serializer/json.go row per element: a map, keys sorted; non-row items are empty elements
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return b->cols[(size_t)x].name < b->cols[(size_t)y].name; });
      for (int i = 0; i + 1 < ncols; i++) if (b->cols[(size_t)order[(size_t)i]].name == b->cols[(size_t)order[(size_t)i + 1]].name)
        return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_queue_serialize: duplicate column names");
      add(QC_CONST, QA_ROW_EVENT, "{", nullptr);
      for (int j = 0; j < ncols; j++) { const DColumn &c = b->cols[(size_t)order[(size_t)j]]; std::string t; if (j) t += ','; json_key(t, c.name); t += ':'; add(QC_VALUE, QA_ROW_EVENT, t, &c); }
      add(QC_CONST, QA_ROW_EVENT, "}", nullptr);
    }
    const int ncell = (int)sc.size();

    // ---- row meta and group part ids ----
    std::vector<Buf> keep;
    q.kind = ptr<uint8_t>(b->kind); q.src_row = ptr<int32_t>(b->src_row); q.part_id = ptr<uint32_t>(b->part_id);
    q.old_present = ptr<uint8_t>(b->old_present); q.has_old = nold > 0;
    if (native && meta) {
      const size_t mn = (size_t)meta->n;
      q.m_id = meta_array(meta, meta->id, mn, keep); q.m_lsn = meta_array(meta, meta->lsn, mn, keep); q.m_commit = meta_array(meta, meta->commit_time, mn, keep);
      q.m_counter = meta_array(meta, meta->counter, mn, keep); q.m_form = meta_array(meta, meta->names_form, mn, keep);
      auto var = [&](const uint32_t *off, const uint8_t *data, const uint32_t *&doff, const uint8_t *&ddata) {
        if (!off) return;
        if (meta->mem == TFGPU_MEM_DEVICE) { doff = off; ddata = data; return; }
        doff = meta_array(meta, off, mn + 1, keep);
        ddata = (data && off[mn]) ? meta_array(meta, data, (size_t)off[mn], keep) : reinterpret_cast<const uint8_t *>(doff);  // all strings empty: never read
      };
      var(meta->tx_id_offsets, meta->tx_id_data, q.m_tx_off, q.m_tx);
      var(meta->query_offsets, meta->query_data, q.m_q_off, q.m_q);
    }
    std::vector<uint32_t> gpart;
    if (native && o->group_part_ids) {
      for (size_t g = 0; g + 1 < gstart.size(); g++) { gpart.push_back((uint32_t)blob.size()); json_key(blob, o->group_part_ids[g] ? o->group_part_ids[g] : "", true); }
      gpart.push_back((uint32_t)blob.size());
      q.ngroups = (int32_t)gstart.size() - 1;
      Buf a = upload_small(gstart.data(), gstart.size() * sizeof(int64_t)), c = upload_small(gpart.data(), gpart.size() * sizeof(uint32_t));
      keep.push_back(a); keep.push_back(c);
      q.gstart = ptr<int64_t>(a); q.gpart = ptr<uint32_t>(c);
    }

    Buf first_listed;
    if (ragged) {
      std::vector<const uint8_t *> abs((size_t)ncols);
      for (int j = 0; j < ncols; j++) abs[(size_t)j] = ptr<uint8_t>(b->cols[(size_t)j].absent);
      Buf babs = upload_small(abs.data(), abs.size() * sizeof(uint8_t *));
      keep.push_back(babs);
      first_listed = dalloc((size_t)n * 4);
      queue_first_listed<<<blocks(n), 256, 0, st>>>(ptr<const uint8_t *>(babs), ncols, n, ptr<uint16_t>(b->col_order), ptr<int32_t>(first_listed));
      q.first_listed = ptr<int32_t>(first_listed);
      if (b->col_order) { q.col_order = ptr<uint16_t>(b->col_order); q.ord_n = ncols; q.ord_name0 = ord_name0; q.ord_val0 = ord_val0; }
    }
    Buf bsc = upload_const(sc.data(), sc.size() * sizeof(SCol)), bblob = upload_const(blob.data(), blob.size());
    Buf cell = dalloc((size_t)ncell * (size_t)n * 4), row_len = dalloc((size_t)(n + 1) * 4 + 16), mflags = dalloc((size_t)n + 16);
    SerParams p{};
    p.cols = ptr<SCol>(bsc); p.ncols = ncell; p.nrows = n; p.format = TFGPU_FMT_JSON; p.any_as_string = 0; p.closing_newline = 0;
    p.blob = ptr<uint8_t>(bblob); p.cell = ptr<uint32_t>(cell); p.row_len = ptr<uint32_t>(row_len); p.last_present = nullptr;
    Buf tot64 = dalloc_zero(8);
    p.total64 = reinterpret_cast<unsigned long long *>(tot64->p);
    q.msg_flags = ptr<uint8_t>(mflags);
    p.q = q;
    { KernelTimer t("ser_cell_len"); if (ncell && n) ser_cell_len<<<dim3(blocks(n), (unsigned)ncell), 256, 0, st>>>(p); }
    { KernelTimer t("ser_row_layout"); ser_row_layout<<<blocks(n), 256, 0, st>>>(p); }

    // ---- the cut plan (host: the batchers' sequential greedy loop over the element lengths) ----
    CutBuffers cb(p.row_len, n);
    const uint32_t *elen = cb.elen; uint8_t *hflags = cb.flags;
    const uint32_t lead = native ? 1u : 0u;
    uint64_t total = 0;
    const int64_t nm = queue_cut_and_place(o, elen, n, gstart, lead, native, hflags, msg_start, msg_row, cap, &total);
    if (nm > cap) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_queue_serialize: more messages than msg_start / msg_row hold");
    if (total >> 32) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_queue_serialize: output exceeds 4 GiB; split the batch by rows");
    msg_start[nm] = total; msg_row[nm] = n;
    cb.upload_flags(mflags->p, n);
    ser_queue_tail_len<<<blocks(n), 256, 0, st>>>(p);
    exclusive_scan_u32(p.row_len, p.row_len, n, true);
    const uint32_t *htot = d2h_u32(p.row_len + n);
    tf::sync();  // also: hflags / meta staging vectors may die after this
    if (*htot != (uint32_t)total) return tf::fail(TFGPU_ERR_DEVICE, "tfgpu_queue_serialize: internal: device and host disagree on the output size");
    res->size = total;
    res->mem = dalloc(res->size + 64);
    p.out = ptr<uint8_t>(res->mem);
    { KernelTimer t("ser_cell_write"); if (ncell && n) ser_cell_write<<<dim3(blocks(n), (unsigned)ncell), 256, 0, st>>>(p); }
    for (int ci = 0; ci < ncell; ci++) if (sc[(size_t)ci].kind == QC_CONST && sc[(size_t)ci].pre_len > QCONST_INLINE) {
      const uint32_t pieces = (sc[(size_t)ci].pre_len + QFILL_PIECE - 1) / QFILL_PIECE;
      const uint64_t items = (uint64_t)n * pieces;
      if (items > 0x1FFFFFFFFull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_queue_serialize: too many rows for one call");
      KernelTimer t("ser_fill_const");
      ser_fill_const<<<(unsigned)((items + 3) / 4), 256, 0, st>>>(p, ci, pieces);
    }
    { KernelTimer t("ser_queue_frame"); ser_queue_frame<<<blocks(n), 256, 0, st>>>(p); }
    tf::sync();  // `keep` (row meta staging) is released on return
    *nmsg = nm;
    *values = res.release();
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}
