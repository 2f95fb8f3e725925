// tf_parquet.hip — Parquet column chunks → device columns (the source format of BASELINE.json configs[3]).
//
// The reference reads Parquet through github.com/parquet-go/parquet-go (not under /root/reference) one ROW at a time into a
// map[string]any and boxes every value (pkg/providers/s3/reader/registry/parquet/reader_parquet.go:137-283: pr.Read(&row),
// constructCI, parseParquetField: DATE → time.Unix(0, 0).Add(24h * days), everything else through abstract.Restore).  The file
// format itself is columnar, so here a column chunk goes to its device column without ever becoming rows:
//
//   host   the footer and the page headers (Thrift compact protocol, a few hundred small structs), and the RUN HEADERS of the
//          RLE / bit-packed hybrid streams (definition levels, dictionary indices): one table of segments per column, each
//          (first ordinal, count, kind, where the packed bits are | the run's value), plus the (offset, length) of every entry
//          of a byte-array dictionary.  The host never touches a value.
//   device the whole file is uploaded once; per column: definition levels → validity (a row finds its run by binary search),
//          a scan gives every present row the ordinal of its value, and the value comes from its segment — PLAIN (the bytes
//          themselves), a dictionary index (RLE run or bit-packed, any width) into the chunk's dictionary, PLAIN booleans;
//          PLAIN byte arrays are length-prefixed, so one lane per page walks the prefixes once; text is then packed with the
//          destination-centric copy the CSV ingest uses (tf_segcopy.hpp), straight out of the file image.
//
// Scope: top-level leaves (required / optional) of any schema — a requested GROUP (nested / repeated: the `any` tree parquet-go builds)
// is refused by name, the flat columns beside it are read; data pages v1 and v2; PLAIN, PLAIN_DICTIONARY / RLE_DICTIONARY, RLE
// (booleans), DELTA_BINARY_PACKED and DELTA_LENGTH_BYTE_ARRAY (block headers on the host, the deltas unpacked and prefix-summed by one
// wave per page: pq_delta), DELTA_BYTE_ARRAY (a prefix chain: expanded by the host while it walks the page); BOOLEAN, INT32, INT64,
// INT96 (→ the decimal text of its 96 bits), FLOAT, DOUBLE, BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY; the DATE / TIMESTAMP / INT / DECIMAL
// annotations as abstract.Restore treats their values; UNCOMPRESSED, SNAPPY, GZIP, ZSTD and LZ4_RAW pages (a compressed object's pages
// are inflated on the host while it walks them — it reads the run headers there anyway — and the device image is then the inflated
// payloads instead of the file).  BYTE_STREAM_SPLIT and the other codecs are refused by name (TFGPU_ERR_UNSUPPORTED).  Every length,
// offset and dictionary index the object states is checked against the bytes that are there — on the host where it walks them, on
// the device where the values are (an error word read back at the existing sync): a corrupt object is TFGPU_ERR_INVALID, never an
// out-of-bounds read.
// PARITY: the schema resolver (parquet_schema_resolver.go:81-158) and the value mapping (parseParquetField, abstract.Restore) are pinned
// to the reference's own reader canon — 30 files under tests/canon/s3/parquet/canondata, extracted into tests/golden/parquet_reader.json
// and reproduced on inputs re-created from the canon's values (tests/test_parquet_canon.py): the 20 flat files value for value, the
// 10 with nested columns refused by name.  The page decoder stands against pyarrow's reading of the same bytes (tests/test_parquet.py):
// parquet-go itself is not under /root/reference.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <atomic>
#include <thread>
#include <unordered_map>
#include <cstdio>

#include "tf_common.hpp"
#include "tf_segcopy.hpp"
#define PQD(...) do { if (std::getenv("TFGPU_PQ_DEBUG")) { std::fprintf(stderr, __VA_ARGS__); std::fflush(stderr); } } while (0)  /* the page walk, line by line */
#include <chrono>
// TFGPU_PQ_TIMES=1 (measurement only): host milliseconds since the call began, at the ends of its phases, on stderr
struct PqClock {
  const bool on = std::getenv("TFGPU_PQ_TIMES") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void at(const char *what) const { if (on) { std::fprintf(stderr, "pq %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); std::fflush(stderr); } }
  ~PqClock() { at("left (locals destroyed)"); }
};

namespace tf {
namespace pq {

// ---- Thrift compact protocol (the subset Parquet's metadata uses) -------------------------------------------------------
struct TReader {
  const uint8_t *p, *e;
  bool ok = true;
  uint64_t varint() {
    uint64_t v = 0; int sh = 0;
    while (p < e) { const uint8_t b = *p++; v |= (uint64_t)(b & 0x7F) << sh; if (!(b & 0x80)) return v; sh += 7; if (sh > 63) break; }
    ok = false; return 0;
  }
  int64_t zz() { const uint64_t v = varint(); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }
  // field header: returns the compact type (0 = stop), sets id
  int field(int &id) {
    if (p >= e) { ok = false; return 0; }
    const uint8_t b = *p++;
    if (b == 0) return 0;
    const int delta = b >> 4, ty = b & 15;
    if (delta) id += delta; else id = (int)zz();
    return ty;
  }
  std::string binary() { const uint64_t n = varint(); if (!ok || n > (uint64_t)(e - p)) { ok = false; return ""; } std::string s((const char *)p, (size_t)n); p += n; return s; }
  void list(int &ety, uint32_t &n) {
    if (p >= e) { ok = false; n = 0; ety = 0; return; }
    const uint8_t b = *p++;
    n = b >> 4; ety = b & 15;
    if (n == 15) n = (uint32_t)varint();
  }
  int depth = 0;  // (a footer is a few structs deep; a hostile one must not take the host's stack: nested skips stop at 64)
  void skip(int ty) {
    if (!ok) return;
    struct Depth { int &d; explicit Depth(int &x) : d(x) { d++; } ~Depth() { d--; } } guard(depth);
    if (depth > 64) { ok = false; return; }
    switch (ty) {
      case 1: case 2: break;                       // bool in the header
      case 3: p++; break;
      case 4: case 5: case 6: varint(); break;
      case 7: p += 8; break;
      case 8: { const uint64_t n = varint(); if (n > (uint64_t)(e - p)) ok = false; else p += n; break; }
      case 9: case 10: { int et; uint32_t n; list(et, n); for (uint32_t i = 0; i < n && ok; i++) { if (et == 1 || et == 2) p++; else skip(et); } break; }
      case 11: { const uint64_t n = varint(); if (n) { if (p >= e) { ok = false; break; } const uint8_t kv = *p++; for (uint64_t i = 0; i < n && ok; i++) { skip(kv >> 4); skip(kv & 15); } } break; }
      case 12: { int id = 0; for (;;) { const int t = field(id); if (!t || !ok) break; skip(t); } break; }
      default: ok = false;
    }
    if (p > e) ok = false;
  }
};

enum { T_BOOLEAN = 0, T_INT32 = 1, T_INT64 = 2, T_INT96 = 3, T_FLOAT = 4, T_DOUBLE = 5, T_BYTE_ARRAY = 6, T_FLBA = 7 };
enum { E_PLAIN = 0, E_PLAIN_DICT = 2, E_RLE = 3, E_DELTA_BINARY_PACKED = 5, E_DELTA_LENGTH_BYTE_ARRAY = 6, E_DELTA_BYTE_ARRAY = 7, E_RLE_DICT = 8, E_BYTE_STREAM_SPLIT = 9 };
// LogicalType's union members (parquet.thrift)
enum { L_NONE = 0, L_STRING = 1, L_MAP = 2, L_LIST = 3, L_ENUM = 4, L_DECIMAL = 5, L_DATE = 6, L_TIME = 7, L_TIMESTAMP = 8, L_INTEGER = 10, L_UNKNOWN = 11, L_JSON = 12, L_BSON = 13, L_UUID = 14, L_FLOAT16 = 15 };

struct SchemaEl {
  int type = -1, type_len = 0, rep = 0, conv = -1, nchildren = 0, logical = L_NONE;
  int scale = 0, precision = 0;          // DECIMAL (the logical type's, else the element's own fields 7 / 8)
  int int_bits = 0; bool int_signed = true;  // INTEGER
  bool utc = true; int unit = 0;         // TIME / TIMESTAMP: 1 MILLIS, 2 MICROS, 3 NANOS
  std::string name;
};
struct ColChunk { int type = -1, codec = 0; int64_t num_values = 0, data_off = -1, dict_off = -1, total_comp = 0, total_uncomp = 0; uint32_t enc_mask = 0 /* bit e: the chunk lists encoding e */; std::vector<std::string> path; };
struct RowGroup { int64_t num_rows = 0; std::vector<ColChunk> cols; };
struct FileMeta { std::vector<SchemaEl> schema; std::vector<RowGroup> groups; int64_t num_rows = 0; };

static void parse_logical(TReader &r, SchemaEl &s) {  // the union's set member, with the fields the type strings and the resolver read
  int lid = 0;
  for (;;) {
    const int lt = r.field(lid);
    if (!lt || !r.ok) break;
    s.logical = lid;
    if (lt != 12) { r.skip(lt); continue; }
    int fid = 0;
    for (;;) {
      const int ft = r.field(fid);
      if (!ft || !r.ok) break;
      if (lid == L_DECIMAL && fid == 1) s.scale = (int)r.zz();
      else if (lid == L_DECIMAL && fid == 2) s.precision = (int)r.zz();
      else if (lid == L_INTEGER && fid == 1) { if (r.p < r.e) s.int_bits = (int8_t)*r.p++; else r.ok = false; }
      else if (lid == L_INTEGER && fid == 2) s.int_signed = ft == 1;
      else if ((lid == L_TIME || lid == L_TIMESTAMP) && fid == 1) s.utc = ft == 1;
      else if ((lid == L_TIME || lid == L_TIMESTAMP) && fid == 2 && ft == 12) { int uid = 0; for (;;) { const int ut = r.field(uid); if (!ut || !r.ok) break; s.unit = uid; r.skip(ut); } }
      else r.skip(ft);
    }
  }
}
static void parse_schema_el(TReader &r, SchemaEl &s) {
  int id = 0, el_scale = 0, el_precision = 0;
  for (;;) {
    const int t = r.field(id);
    if (!t || !r.ok) break;
    switch (id) {
      case 1: s.type = (int)r.zz(); break;
      case 2: s.type_len = (int)r.zz(); break;
      case 3: s.rep = (int)r.zz(); break;
      case 4: s.name = r.binary(); break;
      case 5: s.nchildren = (int)r.zz(); break;
      case 6: s.conv = (int)r.zz(); break;
      case 7: el_scale = (int)r.zz(); break;
      case 8: el_precision = (int)r.zz(); break;
      case 10: parse_logical(r, s); break;
      default: r.skip(t);
    }
  }
  // A leaf that carries only the legacy converted type is typed by it (parquet-go builds the node's Type from whichever annotation
  // is there — the canon's fixed_length_decimal_legacy prints DECIMAL(13,2)); a GROUP is not (the canon's datapage_v2 `e` stays "group").
  if (s.nchildren == 0 && (s.logical == L_NONE || s.logical == L_UNKNOWN) && s.conv >= 0) {
    switch (s.conv) {
      case 0: s.logical = L_STRING; break;
      case 4: s.logical = L_ENUM; break;
      case 5: s.logical = L_DECIMAL; s.scale = el_scale; s.precision = el_precision; break;
      case 6: s.logical = L_DATE; break;
      case 7: case 8: s.logical = L_TIME; s.utc = true; s.unit = s.conv == 7 ? 1 : 2; break;
      case 9: case 10: s.logical = L_TIMESTAMP; s.utc = true; s.unit = s.conv == 9 ? 1 : 2; break;
      case 11: case 12: case 13: case 14: s.logical = L_INTEGER; s.int_signed = false; s.int_bits = 8 << (s.conv - 11); break;
      case 15: case 16: case 17: case 18: s.logical = L_INTEGER; s.int_signed = true; s.int_bits = 8 << (s.conv - 15); break;
      case 19: s.logical = L_JSON; break;
      case 20: s.logical = L_BSON; break;
      default: break;
    }
  }
}
static void parse_col_meta(TReader &r, ColChunk &c) {
  int id = 0;
  for (;;) {
    const int t = r.field(id);
    if (!t || !r.ok) break;
    switch (id) {
      case 1: c.type = (int)r.zz(); break;
      case 2: { int et; uint32_t n; r.list(et, n); for (uint32_t i = 0; i < n && r.ok; i++) { const int64_t e = r.zz(); if (e >= 0 && e < 32) c.enc_mask |= 1u << e; } break; }
      case 3: { int et; uint32_t n; r.list(et, n); for (uint32_t i = 0; i < n && r.ok; i++) c.path.push_back(r.binary()); break; }
      case 4: c.codec = (int)r.zz(); break;
      case 5: c.num_values = r.zz(); break;
      case 6: c.total_uncomp = r.zz(); break;
      case 7: c.total_comp = r.zz(); break;
      case 9: c.data_off = r.zz(); break;
      case 11: c.dict_off = r.zz(); break;
      default: r.skip(t);
    }
  }
}

// ---- the file's top-level fields, as the reference's resolver and row reader see them ---------------------------------------
// parquet_schema_resolver.go:92-153 walks meta.Schema().Fields(): a leaf is typed by its physical type, then its logical type,
// then its converted type; a group is TypeAny.  OriginalType = "parquet:" + el.Type().String().  The strings of DECIMAL(p,s),
// STRING, INT32 … FIXED_LEN_BYTE_ARRAY(n), group and LIST are pinned by the reference's reader canon (tests/golden/parquet_reader.json);
// INT(bits,signed), TIME(…) and TIMESTAMP(…) restate parquet-go's format package and are not.
struct TopField {
  SchemaEl el;
  bool group = false;
  int leaf = -1;        // index of the leaf's column chunk in a row group (leaves in depth-first order)
  int dtype = TFGPU_T_ANY;
  std::string type_string;
};
static bool top_fields(const FileMeta &m, std::vector<TopField> &out, int &nleaves, std::string &why) {
  if (m.schema.empty()) { why = "empty schema"; return false; }
  size_t at = 1;
  nleaves = 0;
  // leaves under schema[at] (inclusive), advancing `at`; depth-guarded (a hostile footer must not take the host's stack)
  struct Walk { const FileMeta &m; size_t &at; int &nleaves; bool ok = true;
    void sub(int depth) {
      if (at >= m.schema.size() || depth > 64) { ok = false; return; }
      const SchemaEl &e = m.schema[at++];
      if (e.nchildren <= 0) { nleaves++; return; }
      for (int i = 0; i < e.nchildren && ok; i++) sub(depth + 1);
    } } w{m, at, nleaves};
  for (int i = 0; i < m.schema[0].nchildren; i++) {
    if (at >= m.schema.size()) { why = "schema tree runs past its elements"; return false; }
    TopField f;
    f.el = m.schema[at];
    if (f.el.nchildren > 0 || f.el.rep == 2) {  // a group — or a repeated leaf, which is a list of its values
      f.group = true;
      f.type_string = f.el.nchildren > 0 ? (f.el.logical == L_LIST ? "LIST" : f.el.logical == L_MAP ? "MAP" : "group") : "";
      const int before = nleaves;
      w.sub(0);
      if (!w.ok) { why = "schema tree runs past its elements"; return false; }
      f.leaf = before;
    } else {
      f.leaf = nleaves++;
      at++;
    }
    out.push_back(f);
  }
  if (at != m.schema.size()) { why = "schema elements outside the tree"; return false; }
  static const char *phys[] = {"BOOLEAN", "INT32", "INT64", "INT96", "FLOAT", "DOUBLE", "BYTE_ARRAY", "FIXED_LEN_BYTE_ARRAY"};
  static const int phys_t[] = {TFGPU_T_BOOLEAN, TFGPU_T_INT32, TFGPU_T_INT64, TFGPU_T_UTF8, TFGPU_T_FLOAT32, TFGPU_T_FLOAT64, TFGPU_T_BYTES, TFGPU_T_BYTES};
  static const char *units[] = {"", "MILLIS", "MICROS", "NANOS"};
  for (auto &f : out) {
    const SchemaEl &e = f.el;
    if (f.group) { f.dtype = TFGPU_T_ANY; if (f.type_string.empty()) f.type_string = (e.type >= 0 && e.type <= 7) ? phys[e.type] : "group"; continue; }
    if (e.type < 0 || e.type > 7) { why = "column " + e.name + ": unknown physical type"; return false; }
    f.dtype = phys_t[e.type];
    f.type_string = e.type == T_FLBA ? "FIXED_LEN_BYTE_ARRAY(" + std::to_string(e.type_len) + ")" : phys[e.type];
    switch (e.logical) {
      case L_DATE: f.dtype = TFGPU_T_DATE; f.type_string = "DATE"; break;
      case L_STRING: f.dtype = TFGPU_T_UTF8; f.type_string = "STRING"; break;
      case L_INTEGER: f.dtype = e.int_signed ? TFGPU_T_INT64 : TFGPU_T_UINT64; f.type_string = "INT(" + std::to_string(e.int_bits) + "," + (e.int_signed ? "true" : "false") + ")"; break;
      case L_DECIMAL: f.dtype = TFGPU_T_FLOAT64; f.type_string = "DECIMAL(" + std::to_string(e.precision) + "," + std::to_string(e.scale) + ")"; break;  // (> 8 digits: TypeString, then the converted type DECIMAL every decimal node also has makes it Float64)
      case L_TIMESTAMP: f.dtype = TFGPU_T_TIMESTAMP; f.type_string = std::string("TIMESTAMP(isAdjustedToUTC=") + (e.utc ? "true" : "false") + ",unit=" + units[e.unit & 3] + ")"; break;
      case L_TIME: f.type_string = std::string("TIME(isAdjustedToUTC=") + (e.utc ? "true" : "false") + ",unit=" + units[e.unit & 3] + ")"; break;
      case L_UUID: f.dtype = TFGPU_T_UTF8; f.type_string = "UUID"; break;
      case L_ENUM: f.dtype = TFGPU_T_UTF8; f.type_string = "ENUM"; break;
      case L_JSON: f.type_string = "JSON"; break;
      case L_BSON: f.type_string = "BSON"; break;
      case L_FLOAT16: f.type_string = "FLOAT16"; break;
      default: break;
    }
  }
  return true;
}
static bool parse_footer(const uint8_t *f, uint64_t len, FileMeta &m, std::string &why) {
  if (len < 12 || std::memcmp(f, "PAR1", 4) || std::memcmp(f + len - 4, "PAR1", 4)) { why = "not a Parquet file (PAR1 magic)"; return false; }
  uint32_t flen; std::memcpy(&flen, f + len - 8, 4);
  if ((uint64_t)flen + 12 > len) { why = "footer length out of range"; return false; }
  TReader r{f + len - 8 - flen, f + len - 8};
  int id = 0;
  for (;;) {
    const int t = r.field(id);
    if (!t || !r.ok) break;
    if (id == 2) { int et; uint32_t n; r.list(et, n); for (uint32_t i = 0; i < n && r.ok; i++) { SchemaEl s; parse_schema_el(r, s); m.schema.push_back(s); } }
    else if (id == 3) m.num_rows = r.zz();
    else if (id == 4) {
      int et; uint32_t n; r.list(et, n);
      for (uint32_t i = 0; i < n && r.ok; i++) {
        RowGroup g; int gid = 0;
        for (;;) {
          const int gt = r.field(gid);
          if (!gt || !r.ok) break;
          if (gid == 1) {
            int cet; uint32_t cn; r.list(cet, cn);
            for (uint32_t k = 0; k < cn && r.ok; k++) {
              ColChunk c; int cid = 0;
              for (;;) { const int ct = r.field(cid); if (!ct || !r.ok) break; if (cid == 3) parse_col_meta(r, c); else r.skip(ct); }
              g.cols.push_back(c);
            }
          } else if (gid == 3) g.num_rows = r.zz();
          else r.skip(gt);
        }
        m.groups.push_back(g);
      }
    } else r.skip(t);
  }
  if (!r.ok) { why = "malformed footer (Thrift)"; return false; }
  return true;
}
struct PageHeader { int type = -1, usize = 0, csize = 0, nvalues = 0, enc = 0, def_enc = E_RLE, def_len = 0, rep_len = 0, v2 = 0; bool compressed_v2 = true; };
static bool parse_page_header(TReader &r, PageHeader &h) {
  int id = 0;
  for (;;) {
    const int t = r.field(id);
    if (!t || !r.ok) break;
    if (id == 1) h.type = (int)r.zz();
    else if (id == 2) h.usize = (int)r.zz();
    else if (id == 3) h.csize = (int)r.zz();
    else if (id == 5 || id == 7 || id == 8) {
      int sid = 0;
      if (id == 8) h.v2 = 1;
      for (;;) {
        const int st = r.field(sid);
        if (!st || !r.ok) break;
        if (sid == 1) h.nvalues = (int)r.zz();
        else if (id == 5 && sid == 2) h.enc = (int)r.zz();
        else if (id == 5 && sid == 3) h.def_enc = (int)r.zz();
        else if (id == 7 && sid == 2) h.enc = (int)r.zz();
        else if (id == 8 && sid == 4) h.enc = (int)r.zz();
        else if (id == 8 && sid == 5) h.def_len = (int)r.zz();
        else if (id == 8 && sid == 6) h.rep_len = (int)r.zz();
        else if (id == 8 && sid == 7) h.compressed_v2 = st == 1;
        else r.skip(st);
      }
    } else r.skip(t);
  }
  return r.ok;
}

// ---- page codecs (host) --------------------------------------------------------------------------------------------------
// Snappy's raw format (format_description.txt): a varint length, then literal / copy elements.  false: malformed or not `usize` bytes.
static bool snappy_inflate(const uint8_t *p, const uint8_t *e, uint8_t *out, uint64_t usize) {
  TReader r{p, e};
  const uint64_t n = r.varint();
  if (!r.ok || n != usize) return false;
  p = r.p;
  uint64_t o = 0;
  while (p < e) {
    const uint32_t tag = *p++;
    uint64_t len, off;
    switch (tag & 3) {
      case 0: {
        len = (tag >> 2) + 1;
        if (len > 60) { const uint32_t nb = (uint32_t)len - 60; if ((uint64_t)(e - p) < nb) return false; len = 0; for (uint32_t i = 0; i < nb; i++) len |= (uint64_t)p[i] << (8 * i); len += 1; p += nb; }
        if (len > (uint64_t)(e - p) || len > usize - o) return false;
        std::memcpy(out + o, p, (size_t)len); p += len; o += len;
        continue;
      }
      case 1: if (p >= e) return false; len = ((tag >> 2) & 7) + 4; off = ((uint64_t)(tag >> 5) << 8) | *p++; break;
      case 2: if (e - p < 2) return false; len = (tag >> 2) + 1; off = (uint64_t)p[0] | (uint64_t)p[1] << 8; p += 2; break;
      default: if (e - p < 4) return false; len = (tag >> 2) + 1; off = (uint64_t)p[0] | (uint64_t)p[1] << 8 | (uint64_t)p[2] << 16 | (uint64_t)p[3] << 24; p += 4;
    }
    if (off == 0 || off > o || len > usize - o) return false;
    for (uint64_t i = 0; i < len; i++) out[o + i] = out[o - off + i];  // (may overlap: byte by byte, as the format says)
    o += len;
  }
  return o == usize;
}
// GZIP (zlib's inflate with the gzip wrapper) and ZSTD through the system's libraries, bound at first use
static bool gzip_inflate(const uint8_t *p, uint64_t n, uint8_t *out, uint64_t usize, std::string &why) {
  struct ZS { const uint8_t *next_in; unsigned avail_in; unsigned long total_in; uint8_t *next_out; unsigned avail_out; unsigned long total_out; const char *msg; void *state; void *zalloc, *zfree, *opaque; int data_type; unsigned long adler, reserved; };
  using Init2 = int (*)(ZS *, int, const char *, int); using Inflate = int (*)(ZS *, int); using End = int (*)(ZS *);
  static void *h = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
  static Init2 init2 = h ? (Init2)dlsym(h, "inflateInit2_") : nullptr; static Inflate inf = h ? (Inflate)dlsym(h, "inflate") : nullptr; static End end = h ? (End)dlsym(h, "inflateEnd") : nullptr;
  static const char *(*ver)() = h ? (const char *(*)())dlsym(h, "zlibVersion") : nullptr;
  if (!init2 || !inf || !end || !ver) { why = "GZIP pages need libz.so.1"; return false; }
  ZS z; std::memset(&z, 0, sizeof z);
  if (init2(&z, 15 + 32, ver(), (int)sizeof z) != 0) { why = "inflateInit2 failed"; return false; }
  z.next_in = p; z.avail_in = (unsigned)n; z.next_out = out; z.avail_out = (unsigned)usize;
  const int rc = inf(&z, 4 /* Z_FINISH */);
  const bool ok = rc == 1 /* Z_STREAM_END */ && z.total_out == usize;
  end(&z);
  if (!ok) why = "malformed GZIP page";
  return ok;
}
static bool zstd_inflate(const uint8_t *p, uint64_t n, uint8_t *out, uint64_t usize, std::string &why) {
  using Dec = size_t (*)(void *, size_t, const void *, size_t); using IsErr = unsigned (*)(size_t);
  static void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
  static Dec dec = h ? (Dec)dlsym(h, "ZSTD_decompress") : nullptr; static IsErr iserr = h ? (IsErr)dlsym(h, "ZSTD_isError") : nullptr;
  if (!dec || !iserr) { why = "ZSTD pages need libzstd.so.1"; return false; }
  const size_t got = dec(out, (size_t)usize, p, (size_t)n);
  if (iserr(got) || got != usize) { why = "malformed ZSTD page"; return false; }
  return true;
}
// LZ4's block format (lz4_Block_format.md) — the LZ4_RAW codec: sequences of (token, literals, 2-byte offset, match).  false: malformed
// or not `usize` bytes.
static bool lz4_raw_inflate(const uint8_t *p, const uint8_t *e, uint8_t *out, uint64_t usize) {
  uint64_t o = 0;
  while (p < e) {
    const uint32_t tok = *p++;
    uint64_t lit = tok >> 4;
    if (lit == 15) { uint8_t b; do { if (p >= e) return false; b = *p++; lit += b; } while (b == 255); }
    if (lit > (uint64_t)(e - p) || lit > usize - o) return false;
    std::memcpy(out + o, p, (size_t)lit); p += lit; o += lit;
    if (p == e) break;  // the last sequence is literals only
    if (e - p < 2) return false;
    const uint64_t off = (uint64_t)p[0] | (uint64_t)p[1] << 8; p += 2;
    uint64_t ml = tok & 15;
    if (ml == 15) { uint8_t b; do { if (p >= e) return false; b = *p++; ml += b; } while (b == 255); }
    ml += 4;
    if (off == 0 || off > o || ml > usize - o) return false;
    for (uint64_t i = 0; i < ml; i++) out[o + i] = out[o - off + i];  // (may overlap)
    o += ml;
  }
  return o == usize;
}
// The first `want` bytes of what [p, e) inflates to (the device inflates the page; the host's walk reads a page's definition levels and the
// width byte of its dictionary indices, which sit at its front).  `cap` = the page's stated size (room behind `out`).  false: malformed so far.
static bool snappy_inflate_prefix(const uint8_t *p, const uint8_t *e, uint8_t *out, uint64_t cap, uint64_t want) {
  TReader r{p, e};
  const uint64_t n = r.varint();
  if (!r.ok || n != cap) return false;
  p = r.p;
  uint64_t o = 0;
  want = std::min(want, cap);
  while (p < e && o < want) {
    const uint32_t tag = *p++;
    uint64_t len, off;
    switch (tag & 3) {
      case 0: {
        len = (tag >> 2) + 1;
        if (len > 60) { const uint32_t nb = (uint32_t)len - 60; if ((uint64_t)(e - p) < nb) return false; len = 0; for (uint32_t i = 0; i < nb; i++) len |= (uint64_t)p[i] << (8 * i); len += 1; p += nb; }
        if (len > (uint64_t)(e - p) || len > cap - o) return false;
        const uint64_t take = std::min(len, want - o);   // (the rest of a long literal is the device's)
        std::memcpy(out + o, p, (size_t)take); p += len; o += len;
        continue;
      }
      case 1: if (p >= e) return false; len = ((tag >> 2) & 7) + 4; off = ((uint64_t)(tag >> 5) << 8) | *p++; break;
      case 2: if (e - p < 2) return false; len = (tag >> 2) + 1; off = (uint64_t)p[0] | (uint64_t)p[1] << 8; p += 2; break;
      default: if (e - p < 4) return false; len = (tag >> 2) + 1; off = (uint64_t)p[0] | (uint64_t)p[1] << 8 | (uint64_t)p[2] << 16 | (uint64_t)p[3] << 24; p += 4;
    }
    if (off == 0 || off > o || len > cap - o) return false;
    for (uint64_t i = 0; i < len; i++) out[o + i] = out[o - off + i];
    o += len;
  }
  return o >= want;
}
static bool lz4_raw_inflate_prefix(const uint8_t *p, const uint8_t *e, uint8_t *out, uint64_t cap, uint64_t want) {
  uint64_t o = 0;
  want = std::min(want, cap);
  while (p < e && o < want) {
    const uint32_t tok = *p++;
    uint64_t lit = tok >> 4;
    if (lit == 15) { uint8_t b; do { if (p >= e) return false; b = *p++; lit += b; } while (b == 255); }
    if (lit > (uint64_t)(e - p) || lit > cap - o) return false;
    std::memcpy(out + o, p, (size_t)std::min(lit, want - o)); p += lit; o += lit;
    if (p == e || o >= want) break;
    if (e - p < 2) return false;
    const uint64_t off = (uint64_t)p[0] | (uint64_t)p[1] << 8; p += 2;
    uint64_t ml = tok & 15;
    if (ml == 15) { uint8_t b; do { if (p >= e) return false; b = *p++; ml += b; } while (b == 255); }
    ml += 4;
    if (off == 0 || off > o || ml > cap - o) return false;
    for (uint64_t i = 0; i < ml; i++) out[o + i] = out[o - off + i];
    o += ml;
  }
  return o >= want;
}
// The arena: bytes without a constructor run over them (a 556 MB object is not zero-filled page by page before it is overwritten), in PINNED host
// memory (its upload is one DMA at the link's rate, not a staged copy out of pageable memory), and KEPT by the calling thread between calls — a
// fresh half-gigabyte allocation is a hundred thousand page faults, more host time than the inflating itself once that runs on every core.
struct RawVec {
  uint8_t *p = nullptr; size_t n = 0, cap = 0;
  RawVec() = default;
  RawVec(const RawVec &) = delete;
  RawVec &operator=(const RawVec &) = delete;
  ~RawVec() { if (p) (void)hipHostFree(p); }
  uint8_t *data() { return p; }
  const uint8_t *data() const { return p; }
  size_t size() const { return n; }
  void reserve(size_t c) {
    if (c <= cap) return;
    void *q = nullptr;
    if (hipHostMalloc(&q, c, hipHostMallocPortable) != hipSuccess || !q) throw std::bad_alloc();
    if (n) std::memcpy(q, p, n);
    if (p) (void)hipHostFree(p);
    p = (uint8_t *)q; cap = c;
  }
  void resize(size_t m) { if (m > cap) reserve(std::max(m, cap + cap / 2 + 4096)); n = m; }
  void append(const uint8_t *b, const uint8_t *e) { const size_t at = n; resize(n + (size_t)(e - b)); if (e > b) std::memcpy(p + at, b, (size_t)(e - b)); }
};
static RawVec &thread_arena() { static thread_local RawVec a; a.n = 0; return a; }
enum { C_UNCOMPRESSED = 0, C_SNAPPY = 1, C_GZIP = 2, C_ZSTD = 6, C_LZ4_RAW = 7 };
static bool page_inflate(int codec, const uint8_t *p, uint64_t n, uint8_t *out, uint64_t usize, std::string &why) {
  switch (codec) {
    case C_UNCOMPRESSED: if (n != usize) { why = "page sizes disagree"; return false; } std::memcpy(out, p, (size_t)n); return true;
    case C_SNAPPY: if (!snappy_inflate(p, p + n, out, usize)) { why = "malformed SNAPPY page"; return false; } return true;
    case C_GZIP: return gzip_inflate(p, n, out, usize, why);
    case C_ZSTD: return zstd_inflate(p, n, out, usize, why);
    case C_LZ4_RAW: if (!lz4_raw_inflate(p, p + n, out, usize)) { why = "malformed LZ4_RAW page"; return false; } return true;
    default: why = "codec " + std::to_string(codec); return false;
  }
}
// What `csize` compressed bytes can at most inflate to: the page header's uncompressed_page_size is the object's own claim, and the
// arena is sized from it BEFORE a byte has been inflated — a few KB of headers each claiming 2 GiB must not make the host zero-fill
// gigabytes.  Ratios of the formats: snappy copies 64 bytes per 3-byte element, LZ4 255 bytes per extra length byte, deflate 1032 : 1,
// zstd an RLE block of 128 KiB from 4 bytes.
// host threads for the pages the host inflates: the cores this process may use (the cgroup's CPU quota: a container on a 256-thread host is typically
// given a handful), at most 32; TFGPU_PQ_INFLATE_THREADS overrides (1 = the walk's own thread, as before round 6)
static size_t inflate_threads() {
  if (const char *e = std::getenv("TFGPU_PQ_INFLATE_THREADS")) { const int v = std::atoi(e); if (v >= 1) return (size_t)std::min(v, 64); }
  size_t n = std::max(1u, std::thread::hardware_concurrency());
  if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char a[32] = {0}; long period = 0;
    if (std::fscanf(fq, "%31s %ld", a, &period) == 2 && std::strcmp(a, "max") != 0 && period > 0) n = std::min<size_t>(n, (size_t)std::max(1L, std::atol(a) / period));
    std::fclose(fq);
  }
  return std::min<size_t>(n, 32);
}
static uint64_t max_inflated(int codec, uint64_t csize) {
  switch (codec) {
    case C_UNCOMPRESSED: return csize;
    case C_SNAPPY: return 64 + csize * 22;
    case C_LZ4_RAW: return 64 + csize * 255;
    case C_GZIP: return 4096 + csize * 1032;
    default: return 131072 + csize * 32768;
  }
}

// DELTA_BINARY_PACKED (Encodings.md): <block size> <miniblocks per block> <total count> <first value: zigzag> then blocks of
// <min delta: zigzag> <bit width per miniblock> <miniblocks>.  The host walks the HEADERS only: where every miniblock's bits are, its
// width and its block's min delta — the deltas themselves are unpacked and prefix-summed on the device (pq_delta), one wave per page.
struct DMini { uint64_t at; int64_t min_delta; uint32_t count, bw; };        // `count` live values of this miniblock (the last one is padded)
struct DStream { int64_t first; uint64_t total; size_t mini0, mini1; const uint8_t *end; };  // minis [mini0, mini1) of `minis`; end: one past the stream
static bool delta_headers(const uint8_t *fb, const uint8_t *p, const uint8_t *e, std::vector<DMini> &minis, DStream &ds) {
  TReader r{p, e};
  const uint64_t block = r.varint(), mpb = r.varint(), total = r.varint();
  const int64_t first = r.zz();
  if (!r.ok || mpb == 0 || mpb > 4096 || block == 0 || block % 128 || block % mpb || (block / mpb) % 32 || total > 0x7FFFFFF0ull) return false;
  const uint64_t vpm = block / mpb;
  ds.first = first; ds.total = total; ds.mini0 = minis.size();
  uint64_t left = total ? total - 1 : 0;
  while (left) {
    const int64_t md = r.zz();
    if (!r.ok || (uint64_t)(r.e - r.p) < mpb) return false;
    const uint8_t *bws = r.p;
    r.p += mpb;
    for (uint64_t j = 0; j < mpb && left; j++) {
      const uint32_t bw = bws[j];
      if (bw > 64) return false;
      const uint64_t nbytes = vpm * bw / 8;
      if (nbytes > (uint64_t)(r.e - r.p)) return false;
      const uint64_t n = std::min<uint64_t>(vpm, left);
      minis.push_back(DMini{(uint64_t)(r.p - fb), md, (uint32_t)n, bw});
      r.p += nbytes; left -= n;
    }
  }
  ds.mini1 = minis.size(); ds.end = r.p;
  return true;
}
// the same stream decoded on the HOST: only for DELTA_BYTE_ARRAY pages, whose values are prefixes of their predecessors — a chain the
// host expands into plain length-prefixed values while it walks the page (the rare encoding of sorted string columns)
static bool delta_decode_host(const uint8_t *fb, const std::vector<DMini> &minis, const DStream &ds, std::vector<int64_t> &out) {
  out.clear();
  if (!ds.total) return true;
  uint64_t v = (uint64_t)ds.first;
  out.push_back((int64_t)v);
  for (size_t m = ds.mini0; m < ds.mini1; m++) {
    const DMini &mi = minis[m];
    for (uint32_t i = 0; i < mi.count; i++) {
      uint64_t d = 0;
      const uint64_t bit = (uint64_t)i * mi.bw;
      for (uint32_t b = 0; b < mi.bw; b++) { const uint64_t q = bit + b; d |= (uint64_t)((fb[mi.at + (q >> 3)] >> (q & 7)) & 1) << b; }
      v += d + (uint64_t)mi.min_delta;
      out.push_back((int64_t)v);
    }
  }
  return out.size() == ds.total;
}

// ---- the segment table a column's rows / values look themselves up in ----------------------------------------------------
enum : uint32_t { SG_RLE = 0, SG_PACKED = 1, SG_PLAIN = 2, SG_PLAIN_BOOL = 3, SG_PLAIN_TEXT = 4, SG_FIXED_TEXT = 5 /* FIXED_LEN_BYTE_ARRAY: value k at `at` + k * bw */,
                      SG_INDEX = 6 /* a page's dictionary indices (or RLE booleans) expanded by pq_hybrid: value k = the bw-BYTE word at tail + at + k * bw */,
                      SG_BSS = 7 /* BYTE_STREAM_SPLIT (Encodings.md: K = the type's width byte streams of `count` bytes each, back to back): byte j of value k at `at` + j * count + k */ };
struct Seg {
  uint32_t start, count;   // first ordinal (levels: row of the column; values: ordinal among the present values) and length
  uint32_t kind, bw;       // SG_*; bit width of packed values
  uint64_t at;             // file offset of the packed bits / plain bytes; SG_RLE: the run's value
  uint32_t dict_base;      // dictionary indices: first entry of this chunk's dictionary in the column's concatenated dictionary
  uint32_t in_tail;        // 1: `at` counts from the image's TAIL (values a kernel decoded there: DELTA_BINARY_PACKED pages)
};
// host: RLE / bit-packed hybrid run headers of [p, e) holding `total` values of width bw; appends segments with ordinals from `ord`
static bool hybrid_runs(const uint8_t *file, const uint8_t *p, const uint8_t *e, uint32_t bw, uint64_t total, uint64_t ord, uint32_t dict_base, std::vector<Seg> &out, uint64_t *ones) {
  uint64_t got = 0;
  const uint32_t vbytes = (bw + 7) / 8;
  while (got < total) {
    TReader r{p, e};
    const uint64_t h = r.varint();
    if (!r.ok) return false;
    p = r.p;
    if (h & 1) {
      const uint64_t groups = h >> 1, nbytes = groups * bw;
      if (nbytes > (uint64_t)(e - p)) return false;
      const uint64_t n = std::min<uint64_t>(groups * 8, total - got);
      out.push_back(Seg{(uint32_t)(ord + got), (uint32_t)n, SG_PACKED, bw, (uint64_t)(p - file), dict_base, 0});
      if (ones) {  // (levels of width 1: the present rows)
        const uint64_t full = n >> 3;
        uint64_t i = 0;
        for (; i + 8 <= full; i += 8) { uint64_t w; std::memcpy(&w, p + i, 8); *ones += (uint64_t)__builtin_popcountll(w); }  // (eight level bytes a step: megabytes of them per object)
        for (; i < full; i++) *ones += (uint64_t)__builtin_popcount(p[i]);
        for (uint64_t j = full * 8; j < n; j++) *ones += (p[j >> 3] >> (j & 7)) & 1;
      }
      p += nbytes; got += n;
    } else {
      const uint64_t n = std::min<uint64_t>(h >> 1, total - got);
      if (vbytes > (uint64_t)(e - p)) return false;
      uint64_t v = 0;
      std::memcpy(&v, p, vbytes);
      p += vbytes;
      if ((h >> 1) == 0) return false;
      out.push_back(Seg{(uint32_t)(ord + got), (uint32_t)n, SG_RLE, bw, v, dict_base, 0});
      if (ones && v) *ones += n;
      got += n;
    }
  }
  return true;
}

__device__ __forceinline__ const Seg &find_seg(const Seg *segs, int32_t n, uint32_t ord) {
  int lo = 0, hi = n - 1;  // the last segment whose start <= ord
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segs[mid].start <= ord) lo = mid; else hi = mid - 1; }
  return segs[lo];
}
// the same, for a workgroup of consecutive rows: ordinals grow with the rows, so ONE lane searches for the block's first ordinal and
// every row walks forward from there — usually not at all (a page holds thousands of rows), instead of log2(pages) dependent loads a row
__device__ __forceinline__ const Seg &find_seg_block(const Seg *segs, int32_t n, uint32_t ord, uint32_t first_ord, int *hint /* shared */) {
  if (threadIdx.x == 0) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segs[mid].start <= first_ord) lo = mid; else hi = mid - 1; }
    *hint = lo;
  }
  __syncthreads();
  int i = *hint;
  while (i + 1 < n && segs[i + 1].start <= ord) i++;
  return segs[i];
}
__device__ __forceinline__ uint64_t load_unaligned(const uint8_t *file, uint64_t at, uint32_t nbytes) {  // nbytes <= 8
  uint64_t v = seg_read8(file, at);
  return nbytes >= 8 ? v : v & ((1ull << (8 * nbytes)) - 1);
}
__device__ __forceinline__ uint32_t seg_value(const uint8_t *file, const Seg &s, uint32_t ord, uint64_t tail_base = 0) {  // an RLE / packed integer (level, dictionary index, boolean)
  if (s.kind == SG_RLE) return (uint32_t)s.at;
  if (s.kind == SG_INDEX) {
    const uint8_t *q = file + tail_base + s.at + (uint64_t)(ord - s.start) * s.bw;
    return s.bw == 1 ? (uint32_t)*q : s.bw == 2 ? (uint32_t)*reinterpret_cast<const uint16_t *>(q) : *reinterpret_cast<const uint32_t *>(q);
  }
  const uint64_t bit = (uint64_t)(ord - s.start) * s.bw;
  const uint64_t w = seg_read8(file, s.at + (bit >> 3)) >> (bit & 7);
  return (uint32_t)(w & ((s.bw >= 32) ? 0xFFFFFFFFull : ((1ull << s.bw) - 1)));
}
// What abstract.Restore (pkg/abstract/restore.go:20-260) makes of the physical value under the column's DataType — the cases a Parquet
// value can meet: float32 under "double" → float64; an int32 under "int64" → cast.ToInt64; an int32 / int64 under "uint64" →
// cast.ToUint64 (a negative value does not cast: 0); an int64 under "timestamp" → ytschema.Timestamp(v).Time(): MICROseconds
// whatever unit the file states; DATE (parseLogicalDate, reader_parquet.go:285-297) → time.Unix(0, 0).Add(24h * days).
enum : int32_t { CV_SAME = 0, CV_DATE = 1, CV_TS_MICROS = 2, CV_I32_I64 = 3, CV_I32_U64 = 4, CV_I64_U64 = 5, CV_F32_F64 = 6 };
struct FixedOut { void *values; int32_t *nanos; int32_t in_width, out_width, conv; };
// what the object's own numbers may not exceed, checked where the values are read (the error word is read back at the sync)
enum : uint32_t { PQE_DICT_INDEX = 1, PQE_TEXT_LENGTH = 2, PQE_HYBRID = 3, PQE_INFLATE = 4 };
// DELTA_BINARY_PACKED / DELTA_LENGTH_BYTE_ARRAY pages: ONE WAVE per page walks its miniblocks in order; a miniblock's deltas are
// unpacked one per lane (any width up to 64 bits), prefix-summed across the wave (wrapping 64-bit adds, as the format says) and carried
// into the next one.  mode 0: value i → `width` bytes at tail[out_at + i * width] (the column's PLAIN values of this page);
// mode 1: the values are LENGTHS of a DELTA_LENGTH_BYTE_ARRAY page → val_len[ord + i], val_off[ord + i] = data_at + their running sum
// (a length that is negative or runs past `data_end` empties the rest of the page and fails the call).
struct DPage { int64_t first; uint64_t out_at, data_at, data_end; uint32_t total, mini0, mini1, mode, width, ord, pad0, pad1; };
__global__ void __launch_bounds__(256) pq_delta(const uint8_t *file, uint8_t *tail, const DMini *minis, const DPage *pages, int32_t npages, uint32_t *val_off, uint32_t *val_len, uint32_t *err) {
  const int lane = threadIdx.x & 63;
  const int32_t pi = (int32_t)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (pi >= npages) return;
  const DPage pg = pages[pi];
  if (!pg.total) return;
  uint64_t run = (uint64_t)pg.first;   // the last decoded value
  uint64_t pos = pg.data_at;           // mode 1: where the next value's bytes start
  bool bad = false;                    // mode 1: a length ran past the page — this value and the rest read as empty
  auto put = [&](uint32_t i, uint64_t v, uint64_t at, bool empty) {  // value i of the page
    if (pg.mode == 0) {
      if (pg.width == 4) *reinterpret_cast<uint32_t *>(tail + pg.out_at + (uint64_t)i * 4) = (uint32_t)v;
      else *reinterpret_cast<uint64_t *>(tail + pg.out_at + (uint64_t)i * 8) = v;
    } else { val_len[pg.ord + i] = empty ? 0u : (uint32_t)v; val_off[pg.ord + i] = empty ? (uint32_t)pg.data_at : (uint32_t)at; }
  };
  if (pg.mode == 1 && ((int64_t)run < 0 || run > pg.data_end - pos)) bad = true;
  if (lane == 0) put(0, run, pos, bad);
  if (pg.mode == 1 && !bad) pos += run;
  uint32_t done = 1;
  for (uint32_t m = pg.mini0; m < pg.mini1; m++) {
    const DMini mi = minis[m];
    for (uint32_t b = 0; b < mi.count; b += 64) {
      const uint32_t i = b + lane;
      const bool live = i < mi.count;
      uint64_t d = 0;
      if (live) {
        if (mi.bw) {
          const uint64_t bit = (uint64_t)i * mi.bw;
          d = seg_read8(file, mi.at + (bit >> 3)) >> (bit & 7);
          if ((bit & 7) + mi.bw > 64) d |= (uint64_t)file[mi.at + (bit >> 3) + 8] << (64 - (bit & 7));
          if (mi.bw < 64) d &= (1ull << mi.bw) - 1;
        }
        d += (uint64_t)mi.min_delta;
      }
      uint64_t inc = d;  // inclusive scan of the deltas across the wave (dead lanes add 0)
#pragma unroll
      for (int sh = 1; sh < 64; sh <<= 1) { const uint64_t t = __shfl_up(inc, sh, 64); if (lane >= sh) inc += t; }
      const uint64_t v = run + inc;
      const uint32_t n = mi.count - b < 64 ? mi.count - b : 64;
      if (pg.mode == 0) { if (live) put(done + lane, v, 0, false); }
      else {
        // the values are lengths: their own running sum places the bytes
        const bool mine_bad = live && ((int64_t)v < 0 || v > 0xFFFFFFFFull);
        const uint64_t len = live && !mine_bad ? v : 0;
        uint64_t lsum = len;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) { const uint64_t t = __shfl_up(lsum, sh, 64); if (lane >= sh) lsum += t; }
        const uint64_t at = pos + lsum - len;
        const bool over = live && (mine_bad || at > pg.data_end || len > pg.data_end - at);
        const unsigned long long anybad = __ballot(over);
        const int first_bad = bad ? 0 : (anybad ? __ffsll(anybad) - 1 : 64);
        if (live) put(done + lane, v, at, lane >= first_bad);
        if (anybad) bad = true;
        pos += __shfl(lsum, (int)n - 1, 64);
      }
      run = __shfl(v, (int)n - 1, 64);
      done += n;
    }
  }
  if (bad && lane == 0) *err = PQE_TEXT_LENGTH;
}
// ---- page codecs on the device: SNAPPY and LZ4_RAW (round 6) --------------------------------------------------------------------
// parquet-go inflates every page in its reader (reader_parquet.go:137-283); until round 6 so did this library — single-threaded, on the
// host, inside the page walk: ~1 GB/s for an object whose uncompressed twin is decoded at 25 GB/s.  Pages are independent and both
// formats are byte-oriented LZ77s (snappy format_description.txt: a varint length, then literal / copy elements with 1-, 2- or 4-byte
// offsets; lz4_Block_format.md: token, literals, 2-byte offset, match), so ONE WAVE takes one page:
//   * the compressed bytes ride in two 256-byte register windows (lane l holds dword l of each); an element's tag and operands are read
//     with v_readlane at a wave-uniform index — the parse is a scalar chain without a memory round trip per element; the next window's
//     load is in flight while the current one is parsed;
//   * the last 64 KiB of OUTPUT live in an LDS ring: a copy reads its source bytes there, lane i byte i (an overlapping copy — offset <
//     length — repeats its period: out[o + i] = out[o - off + i mod off]); snappy's compressor works on 64 KiB fragments and LZ4's
//     offsets are 16 bits, so a source outside the ring (snappy's 4-byte offsets) is rare and is read back from the image;
//   * the ring leaves for the image in 16 KiB pieces of aligned 16-byte stores (ring positions and image addresses are kept congruent
//     mod 16).
// Every length is checked against the page's own numbers (input and output ends, offsets that reach in front of the page): a page
// that lies fails the call (PQE_INFLATE) — the kernels behind it bound-check whatever bytes they find anyway.
struct InfPage { uint64_t src, src_end, dst; uint32_t usize, lead, codec, pad; };  // [src + lead, src_end) of the compressed object inflates to usize bytes at image[dst + lead]; the `lead` bytes (a v2 page's levels) are copied as they are
constexpr uint32_t INF_RING_MAX = 64u * 1024u;   // the ring a workgroup asks for is a launch parameter (a power of two, 16 .. 64 KiB): the kernels are templates over it
TF_DYNAMIC_LDS(uint4, inf_ring4);
struct InfWindow {
  const uint8_t *base;   // 4-byte aligned address of window byte 0
  uint32_t w0, w1;       // lane l: the dwords at base + 4 l and base + 256 + 4 l
  __device__ __forceinline__ void load(const uint8_t *b, int lane) {
    base = b;
    w0 = *reinterpret_cast<const uint32_t *>(b + 4 * lane);
    w1 = *reinterpret_cast<const uint32_t *>(b + 256 + 4 * lane);
  }
  __device__ __forceinline__ void shift(int lane) {  // the second window becomes the first; the next 256 bytes are requested
    base += 256; w0 = w1;
    w1 = *reinterpret_cast<const uint32_t *>(base + 256 + 4 * lane);
  }
  __device__ __forceinline__ uint32_t word(uint32_t d) const {  // dword d (0 .. 127) of the window, d wave-uniform
    const int k = __builtin_amdgcn_readfirstlane((int)(d & 63u));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)w0, k), b = (uint32_t)__builtin_amdgcn_readlane((int)w1, k);
    return d < 64u ? a : b;
  }
  __device__ __forceinline__ uint64_t bytes8(uint32_t p) const {  // the eight bytes at window position p (p + 8 <= 508), p wave-uniform
    const uint32_t d = p >> 2, sh = (p & 3u) * 8u;
    const uint32_t x0 = word(d), x1 = word(d + 1), x2 = word(d + 2);
    const uint64_t lo = (uint64_t)x0 | ((uint64_t)x1 << 32);
    return sh ? (lo >> sh) | ((uint64_t)x2 << (64 - sh)) : lo;
  }
  __device__ __forceinline__ uint64_t bytes5(uint32_t p) const {  // at least five bytes at window position p: two dwords hold them ((p & 3) + 5 <= 8)
    const uint32_t d = p >> 2, sh = (p & 3u) * 8u;
    return ((uint64_t)word(d) | ((uint64_t)word(d + 1) << 32)) >> sh;
  }
  __device__ __forceinline__ uint32_t lane_byte(uint32_t p) const {  // the byte at window position p (< 512), p per lane
    const uint32_t d = p >> 2;
    const uint32_t a = (uint32_t)__shfl((int)w0, (int)(d & 63u), 64), b = (uint32_t)__shfl((int)w1, (int)(d & 63u), 64);
    return ((d < 64u ? a : b) >> ((p & 3u) * 8u)) & 0xFFu;
  }
};
template <int CODEC, uint32_t INF_RING>
__device__ __forceinline__ bool inflate_page(const uint8_t *cfile, uint8_t *image, const InfPage &pg, int lane) {
  constexpr uint32_t INF_MASK = INF_RING - 1u, INF_PIECE = INF_RING / 4u;
  uint8_t *const ring = reinterpret_cast<uint8_t *>(inf_ring4);
  const uint8_t *const in0 = cfile + pg.src + pg.lead;
  const uint64_t in_len = pg.src_end - pg.src - pg.lead;
  uint8_t *const out = image + pg.dst + pg.lead;
  const uint32_t usize = pg.usize;
  const uint32_t phase = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);   // ring position of output byte o = (o + phase) & INF_MASK
  for (uint32_t i = (uint32_t)lane; i < pg.lead; i += 64) image[pg.dst + i] = cfile[pg.src + i];
  InfWindow W;
  const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in0) & 3u);
  W.load(in0 - mis, lane);
  uint64_t ip = 0;          // input bytes consumed
  uint32_t wp = mis;        // window position of input byte ip
  uint32_t o = 0, flushed = 0;
  // window position wp may have run past the first 256 bytes (or the whole window, after a long literal)
  auto settle = [&]() {
    if (wp >= 512u) { const uint32_t skip = wp & ~255u; W.load(W.base + skip, lane); wp -= skip; }
    else if (wp >= 256u) { W.shift(lane); wp -= 256u; }
  };
  auto flush_to = [&](uint32_t upto, bool last) {  // output bytes [flushed, upto) → the image; whole 16-byte lines but for the page's two ends
    if (upto <= flushed) return;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t a = flushed + phase, b = upto + phase;          // phased positions
    const uint32_t a16 = (a + 15u) & ~15u, b16 = last ? (b & ~15u) : b;   // (not last: upto + phase is a multiple of 16 by construction)
    for (uint32_t q = a + (uint32_t)lane; q < min(a16, b); q += 64) out[q - phase] = ring[q & INF_MASK];
    if (b16 > a16) for (uint32_t q = a16 + 16u * (uint32_t)lane; q < b16; q += 16u * 64u)
      *reinterpret_cast<uint4 *>(out + (q - phase)) = *reinterpret_cast<const uint4 *>(ring + (q & INF_MASK));
    if (last) for (uint32_t q = max(b16, a16 > b ? b : a16) + (uint32_t)lane; q < b; q += 64) out[q - phase] = ring[q & INF_MASK];
    flushed = upto;
  };
  auto maybe_flush = [&]() {  // keep less than a ring minus one piece unflushed
    const uint32_t done = (o + phase) & ~(INF_PIECE - 1u);   // (a phased position: a multiple of 16)
    if (done > phase + flushed) flush_to(done - phase, false);
  };
  auto literal = [&](uint32_t len) -> bool {
    if ((uint64_t)len > in_len - ip || len > usize - o) return false;
    uint32_t done = 0;
    while (done < len) {  // pieces that never outrun the ring's unflushed room
      const uint32_t n = min(len - done, INF_PIECE);
      if (wp + n <= 512u) {
        for (uint32_t i = (uint32_t)lane; i < ((n + 63u) & ~63u); i += 64) { const uint32_t v = W.lane_byte(min(wp + i, 511u)); if (i < n) ring[(o + i + phase) & INF_MASK] = (uint8_t)v; }
      } else {
        // a long literal (incompressible bytes come as literals of up to 64 KiB): 16 bytes a lane, four loads in flight — a byte a lane
        // leaves 64 bytes a wave in flight and two waves a CU cannot cover the memory latency with that (measured: 6 GB/s a chip).
        // Ring-ALIGNED 16-byte chunks (the source is read unaligned), the bytes in front of the first and behind the last one by one.
        const uint8_t *src = in0 + ip;
        const uint32_t rp = o + phase;                               // ring position of the literal's first byte (before masking)
        const uint32_t head = min((16u - (rp & 15u)) & 15u, n);
        if ((uint32_t)lane < head) ring[(rp + (uint32_t)lane) & INF_MASK] = src[lane];
        const uint32_t chunks = (n - head) >> 4;
        struct __attribute__((packed, aligned(1))) U64u { uint64_t v; };
        for (uint32_t c0 = 0; c0 < chunks; c0 += 256) {
          uint64_t lo[4], hi[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t c = c0 + 64u * (uint32_t)u + (uint32_t)lane;
            lo[u] = 0; hi[u] = 0;
            if (c < chunks) { const uint8_t *q = src + head + 16u * c; lo[u] = reinterpret_cast<const U64u *>(q)->v; hi[u] = reinterpret_cast<const U64u *>(q + 8)->v; }
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t c = c0 + 64u * (uint32_t)u + (uint32_t)lane;
            if (c < chunks) *reinterpret_cast<uint4 *>(ring + ((rp + head + 16u * c) & INF_MASK)) = make_uint4((uint32_t)lo[u], (uint32_t)(lo[u] >> 32), (uint32_t)hi[u], (uint32_t)(hi[u] >> 32));
          }
        }
        const uint32_t done_b = head + 16u * chunks;
        if (done_b + (uint32_t)lane < n) ring[(rp + done_b + (uint32_t)lane) & INF_MASK] = src[done_b + lane];
      }
      o += n; ip += n; wp += n; done += n;
      maybe_flush();
    }
    settle();
    return true;
  };
  auto match = [&](uint32_t len, uint32_t off) -> bool {
    if (off == 0 || off > o || len > usize - o) return false;
    uint32_t done = 0;
    while (done < len) {
      const uint32_t n = min(len - done, 64u);
      if (off <= INF_RING - 128u) {
        // (LDS is in order: the bytes earlier elements wrote are there)
        const uint32_t i = (uint32_t)lane;
        uint32_t k = i;
        if (off < 64u) {                                // (uniform) the period of an overlapping copy: i mod off through a reciprocal, exact for i, off < 64
          const uint32_t q = (uint32_t)(((uint32_t)i * (65536u / off + 1u)) >> 16);
          k = i - q * off;
        }
        const uint8_t v = ring[(o - off + k + phase) & INF_MASK];
        __builtin_amdgcn_wave_barrier();
        if (i < n) ring[(o + i + phase) & INF_MASK] = v;
      } else {
        // a source the ring no longer holds: everything that far back has left for the image (the ring is flushed a quarter at a time, so what is
        // more than three quarters of it behind `o` is in HBM; only a copy whose last bytes reach into the unflushed part — period-long copies
        // over a far source do not exist, this is defensive — flushes first)
        if (o - off + n > flushed) flush_to(((o + phase) & ~15u) > phase ? ((o + phase) & ~15u) - phase : 0u, false);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        const uint32_t i = (uint32_t)lane;
        uint8_t v = 0;
        if (i < n) { const uint32_t s = o - off + i; v = s < flushed ? out[s] : ring[(s + phase) & INF_MASK]; }
        __builtin_amdgcn_wave_barrier();
        if (i < n) ring[(o + i + phase) & INF_MASK] = v;
      }
      o += n; done += n;
      __builtin_amdgcn_wave_barrier();   // (the next piece of a long overlapping copy reads what this one wrote)
      maybe_flush();
    }
    return true;
  };
  if (CODEC == 1) {  // SNAPPY: the uncompressed length first
    uint64_t n = 0; uint32_t shb = 0;
    for (;;) {
      if (ip >= in_len || shb > 28) return false;
      const uint32_t b = (uint32_t)(W.bytes8(wp) & 0xFFu);
      ip++; wp++; settle();
      n |= (uint64_t)(b & 0x7Fu) << shb; shb += 7;
      if (!(b & 0x80u)) break;
    }
    if (n != usize) return false;
    while (ip < in_len) {
      const uint64_t w = W.bytes5(wp);
      const uint32_t tag = (uint32_t)w & 0xFFu;
      if ((tag & 3u) == 0u) {
        uint32_t len = (tag >> 2) + 1u, hdr = 1u;
        if (len > 60u) { const uint32_t nb = len - 60u; if (in_len - ip < 1u + nb) return false; len = (uint32_t)((w >> 8) & (nb == 4u ? 0xFFFFFFFFull : ((1ull << (8 * nb)) - 1ull))) + 1u; hdr += nb; if (len == 0u) return false; }
        ip += hdr; wp += hdr; settle();
        if (!literal(len)) return false;
      } else {
        uint32_t len, off, hdr;
        if ((tag & 3u) == 1u) { len = ((tag >> 2) & 7u) + 4u; off = ((tag >> 5) << 8) | ((uint32_t)(w >> 8) & 0xFFu); hdr = 2u; }
        else if ((tag & 3u) == 2u) { len = (tag >> 2) + 1u; off = (uint32_t)(w >> 8) & 0xFFFFu; hdr = 3u; }
        else { len = (tag >> 2) + 1u; off = (uint32_t)(w >> 8); hdr = 5u; }
        if (in_len - ip < hdr) return false;
        ip += hdr; wp += hdr; settle();
        if (!match(len, off)) return false;
      }
    }
  } else {  // LZ4_RAW
    while (ip < in_len) {
      uint64_t w = W.bytes8(wp);
      const uint32_t tok = (uint32_t)w & 0xFFu;
      ip++; wp++; settle();
      uint32_t lit = tok >> 4;
      if (lit == 15u) for (;;) { if (ip >= in_len) return false; const uint32_t b = (uint32_t)(W.bytes8(wp) & 0xFFu); ip++; wp++; settle(); lit += b; if (lit > usize) return false; if (b != 255u) break; }
      if (lit && !literal(lit)) return false;
      if (ip == in_len) break;  // the last sequence is literals only
      if (in_len - ip < 2u) return false;
      w = W.bytes8(wp);
      const uint32_t off = (uint32_t)w & 0xFFFFu;
      ip += 2; wp += 2; settle();
      uint32_t ml = tok & 15u;
      if (ml == 15u) for (;;) { if (ip >= in_len) return false; const uint32_t b = (uint32_t)(W.bytes8(wp) & 0xFFu); ip++; wp++; settle(); ml += b; if (ml > usize) return false; if (b != 255u) break; }
      if (!match(ml + 4u, off)) return false;
    }
  }
  if (o != usize) return false;
  flush_to(usize, true);
  return true;
}
template <uint32_t RING>
__global__ void __launch_bounds__(64) pq_inflate(const uint8_t *cfile, uint8_t *image, const InfPage *pages, int32_t npages, uint32_t *err) {
  const int lane = threadIdx.x & 63;
  const int32_t pi = (int32_t)blockIdx.x;
  if (pi >= npages) return;
  const InfPage pg = pages[pi];
  const bool ok = pg.codec == 1u ? inflate_page<1, RING>(cfile, image, pg, lane) : inflate_page<7, RING>(cfile, image, pg, lane);
  if (!ok && lane == 0) *err = PQE_INFLATE;
}

// The RLE / bit-packed hybrid stream of a page's dictionary indices (or RLE booleans) → one bw-byte word per value in the tail.  A
// writer that alternates short literal groups and short repeats leaves a run header every few dozen values — a million and a half
// of them in a 2^20-row object of a hundred dictionary-coded columns: walked on the host (a 32-byte segment each) they cost more than
// the upload of the object.  Here ONE WAVE per page walks the headers (every lane reads the same bytes) and its 64 lanes expand each
// run together.  A stream that ends early, a run of no values or a width above 32 fails the call (PQE_HYBRID).
struct HPage { uint64_t at, end, out_at; uint32_t bw, total, out_w, pad; };
__global__ void __launch_bounds__(256) pq_hybrid(const uint8_t *file, uint8_t *tail, const HPage *pages, int32_t npages, uint32_t *err) {
  const int lane = threadIdx.x & 63;
  const int32_t pi = (int32_t)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (pi >= npages) return;
  const HPage pg = pages[pi];
  uint8_t *out = tail + pg.out_at;
  auto put = [&](uint32_t i, uint32_t v) {
    if (pg.out_w == 1) out[i] = (uint8_t)v; else if (pg.out_w == 2) reinterpret_cast<uint16_t *>(out)[i] = (uint16_t)v; else reinterpret_cast<uint32_t *>(out)[i] = v;
  };
  const uint32_t vbytes = (pg.bw + 7) / 8;
  const uint32_t mask = pg.bw >= 32 ? 0xFFFFFFFFu : ((1u << pg.bw) - 1u);
  uint64_t pos = pg.at;
  uint32_t done = 0;
  bool bad = false;
  while (done < pg.total && !bad) {
    // the run header: a varint of at most five bytes for counts below 2^32 (uniform across the wave)
    if (pos >= pg.end) { bad = true; break; }
    const uint64_t w = seg_read8(file, pos);
    uint64_t h = 0; uint32_t nb = 0;
    for (; nb < 8; nb++) { const uint32_t b = (uint32_t)(w >> (8 * nb)) & 0xFFu; h |= (uint64_t)(b & 0x7Fu) << (7 * nb); if (!(b & 0x80u)) { nb++; break; } }
    if (nb == 8 && ((w >> 56) & 0x80u)) { bad = true; break; }
    if (pos + nb > pg.end) { bad = true; break; }
    pos += nb;
    if (h & 1) {
      const uint64_t groups = h >> 1, nbytes = groups * pg.bw;
      if (groups == 0 || nbytes > pg.end - pos) { bad = true; break; }  // (a run of no groups makes no progress)
      const uint32_t n = (uint32_t)(groups * 8 < (uint64_t)(pg.total - done) ? groups * 8 : (uint64_t)(pg.total - done));
      for (uint32_t i = (uint32_t)lane; i < n; i += 64) {
        const uint64_t bit = (uint64_t)i * pg.bw;
        put(done + i, (uint32_t)(seg_read8(file, pos + (bit >> 3)) >> (bit & 7)) & mask);
      }
      pos += nbytes; done += n;
    } else {
      const uint64_t cnt = h >> 1;
      if (cnt == 0 || vbytes > pg.end - pos) { bad = true; break; }
      const uint32_t v = (uint32_t)load_unaligned(file, pos, vbytes ? vbytes : 1) & (vbytes ? 0xFFFFFFFFu : 0u);
      const uint32_t n = (uint32_t)(cnt < (uint64_t)(pg.total - done) ? cnt : (uint64_t)(pg.total - done));
      for (uint32_t i = (uint32_t)lane; i < n; i += 64) put(done + i, v);
      pos += vbytes; done += n;
    }
  }
  if (bad) {  // the rest of the page reads as index 0 (its dictionary's first entry, or past an empty one: caught there); the call fails
    for (uint32_t i = done + (uint32_t)lane; i < pg.total; i += 64) put(i, 0u);
    if (lane == 0) *err = PQE_HYBRID;
  }
}
// INT96 → the text of deprecated.Int96.String() (parseParquetField, reader_parquet.go:303-305): the 96-bit two's-complement integer
// (nanoseconds of the day in the low 64 bits, the Julian day above them) in decimal.  One row per lane: the 12 bytes by ordinal (PLAIN
// or through the chunk's dictionary), at most 29 digits and a sign into the row's 32-byte slot of the tail; the packing copy reads it there.
__global__ void __launch_bounds__(256) pq_int96_text(const uint8_t *file, uint8_t *tail, uint64_t tail_base, uint64_t slot0, const Seg *segs, int32_t nsegs, const uint32_t *rank, int64_t nrows,
                                                     const uint64_t *dict_at, const uint32_t *dict_n, uint32_t *src_off, uint32_t *lens, uint32_t *err) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const uint32_t k = rank ? rank[r] : (uint32_t)r;
  if (rank && rank[r + 1] == k) { src_off[r] = SEG_NONE; lens[r] = 0; return; }
  const Seg &s = find_seg(segs, nsegs, k);
  uint64_t at = 0; bool ok = true;
  if (s.kind == SG_PLAIN) at = s.at + (uint64_t)(k - s.start) * 12;
  else { const uint32_t idx = seg_value(file, s, k, tail_base); if (idx >= dict_n[s.dict_base]) { *err = PQE_DICT_INDEX; ok = false; } else at = dict_at[s.dict_base] + (uint64_t)idx * 12; }
  uint32_t w0 = 0, w1 = 0, w2 = 0;
  if (ok) { const uint64_t lo = seg_read8(file, at); w0 = (uint32_t)lo; w1 = (uint32_t)(lo >> 32); w2 = (uint32_t)load_unaligned(file, at + 8, 4); }
  const bool neg = (w2 >> 31) != 0;
  if (neg) {  // magnitude of the two's-complement value
    w0 = ~w0; w1 = ~w1; w2 = ~w2;
    if (++w0 == 0) if (++w1 == 0) ++w2;
  }
  char dig[40]; int nd = 0;
  while (w0 | w1 | w2) {  // long division by 10^9, nine digits a round
    uint64_t rem = 0, cur;
    cur = (rem << 32) | w2; w2 = (uint32_t)(cur / 1000000000u); rem = cur % 1000000000u;
    cur = (rem << 32) | w1; w1 = (uint32_t)(cur / 1000000000u); rem = cur % 1000000000u;
    cur = (rem << 32) | w0; w0 = (uint32_t)(cur / 1000000000u); rem = cur % 1000000000u;
    uint32_t part = (uint32_t)rem;
    const bool more = (w0 | w1 | w2) != 0;
    for (int j = 0; j < 9 && (more || part || j == 0); j++) { dig[nd++] = (char)('0' + part % 10); part /= 10; }
  }
  if (!nd) dig[nd++] = '0';
  uint8_t *dst = tail + slot0 + (uint64_t)r * 32;
  int n = 0;
  if (neg) dst[n++] = '-';
  for (int j = nd - 1; j >= 0; j--) dst[n++] = (uint8_t)dig[j];
  src_off[r] = (uint32_t)(tail_base + slot0 + (uint64_t)r * 32);
  lens[r] = (uint32_t)n;
}
// PLAIN byte arrays are length-prefixed: the position of value i + 1 is known only after the length of value i has been read, a
// chain no amount of lanes shortens.  What CAN be shortened is each link.  One workgroup per page stages the page 16 KiB at a time
// (256 threads, the window starting ON a boundary, so a length field sits at a known byte of two staged words), ONE lane walks the
// prefixes there — a link is one two-word LDS read, a funnel shift, a bound check and a two-word LDS write: the (offset, length)
// pairs collect in LDS — and all threads then write the window's pairs out coalesced.  (Round 4's walk read the prefix byte by byte and
// stored each pair to HBM from the walking lane: ~270 ns a value.  Pointer jumping over all 8192 byte positions of a window — every
// position computes where a value starting there would end, log2 rounds of mark / jump-doubling find the chain — was built and measured in
// round 5: 4.6 ms against the old walk's 3.5 on the hits object; with one workgroup per page the rounds are LDS-latency-bound too.)
struct TextPage { uint64_t at, end; uint32_t ord, count; };  // [at, end): the page's values
constexpr uint32_t PQ_STAGE = 16384, PQ_OUT = PQ_STAGE / 4;
constexpr uint32_t PQ_SLICE = PQ_STAGE / 64;
__global__ void __launch_bounds__(256) pq_walk_text(const uint8_t *file, uint64_t limit, const TextPage *pages, uint32_t *val_off, uint32_t *val_len, uint32_t *err, int spec_on, uint32_t *dbg) {
  // (one word of padding per 64: lane i of the wave walk reads words 64 i + j — without it all 64 lanes would meet in one LDS bank)
  __shared__ uint32_t stage_[PQ_STAGE / 4 + 4 + PQ_STAGE / 256 + 1];
  auto S = [&](uint32_t w) -> uint32_t & { return stage_[w + (w >> 6)]; };
  __shared__ uint32_t o_off[PQ_OUT], o_len[PQ_OUT];
  __shared__ uint64_t s_at;
  __shared__ uint32_t s_done, s_cnt, s_bad, s_spec;
  const TextPage pg = pages[blockIdx.x];
  const int tid = threadIdx.x;
  if (tid == 0) { s_at = pg.at; s_done = 0; s_bad = 0; }
  __syncthreads();
  for (;;) {
    // the window is staged from the 16-byte line its first value starts in: `at` is that line, `sh` where the value starts in it — every offset
    // below counts from `at`.  Four (one thread: five) aligned 16-byte loads a thread, all in flight together; a word at a time through
    // unaligned 8-byte pairs they were most of this kernel's time
    const uint64_t at0 = s_at, at = at0 & ~15ull;
    const uint32_t sh = (uint32_t)(at0 - at);
    const uint32_t done = s_done;
    if (done >= pg.count || s_bad) break;  // (uniform: written before the round's last barrier)
    {
      uint4 v[5];
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const uint32_t c = (uint32_t)tid + 256u * (uint32_t)j;
        const uint64_t a = at + 16ull * c;
        v[j] = (c <= PQ_STAGE / 16 && a + 16 <= limit) ? *reinterpret_cast<const uint4 *>(file + a) : make_uint4(0, 0, 0, 0);  // (a read past the page is harmless, one past the image is not)
      }
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const uint32_t c = (uint32_t)tid + 256u * (uint32_t)j;
        if (c <= PQ_STAGE / 16) { S(4 * c) = v[j].x; S(4 * c + 1) = v[j].y; S(4 * c + 2) = v[j].z; S(4 * c + 3) = v[j].w; }
      }
    }
    __syncthreads();
    // ---- the window by all 64 lanes of wave 0 (round 5): lane i owns the values that START in bytes [256 i, 256 i + 256) of the window.  Where its
    //      first value starts is what the lane in front hands over — a chain — so every lane first GUESSES (the first offset of its slice from
    //      which three prefixes in a row stay inside the page), walks its slice, and the guesses are then checked against the neighbours' exits
    //      and re-walked where they differ, a few rounds; the counts are scanned and the pairs written.  A window that does not settle, holds a
    //      prefix that points past its page, or more values than the page has left takes the one-lane walk below, which carries those rules. ----
    bool spec_ok = false;
    if (spec_on == 2) {  // TFGPU_PQ_WALK_SPEC=2, profiling only (the values are NOT valid): no walk at all — what staging, barriers and the flush cost
      if (tid == 0) { const uint32_t want = pg.count - done; s_cnt = want < 800u ? want : 800u; s_at = at0 + PQ_STAGE - 16; s_spec = 1; }
    } else
    if (tid < 64 && spec_on) {
      const uint32_t want = pg.count - done;
      const uint32_t s0 = (uint32_t)tid * PQ_SLICE, s1 = s0 + PQ_SLICE;
      auto rd = [&](uint32_t o) { return __builtin_amdgcn_alignbyte(S((o >> 2) + 1), S(o >> 2), o & 3u); };
      auto fits = [&](uint32_t o, uint32_t n) { const uint64_t abs = at + o; return abs + 4 <= pg.end && (uint64_t)n <= pg.end - (abs + 4); };
      // walk the values that start in [e, s1): where the chain leaves the slice, how many, whether a prefix pointed past the page
      auto walk = [&](uint32_t e, uint32_t &cnt, bool &bad, bool write, uint32_t base) {
        uint32_t o = e; cnt = 0; bad = false;
        while (o < s1 && o + 4 <= PQ_STAGE && at + o < pg.end) {   // (the page's last value ends at pg.end: a clean stop, not a prefix)
          // four empty values at once where sixteen zero bytes lie ahead inside the slice and the page (their reads do not depend on each other)
          if (o + 16 <= s1 && at + o + 16 <= pg.end) {
            if ((rd(o) | rd(o + 4) | rd(o + 8) | rd(o + 12)) == 0u) {
              if (write) {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) { o_off[base + cnt + k] = (uint32_t)(at + o + 4 * k + 4); o_len[base + cnt + k] = 0; }
              }
              cnt += 4; o += 16;
              continue;
            }
          }
          const uint32_t n = rd(o);
          if (!fits(o, n)) { bad = true; break; }
          if (write) { o_off[base + cnt] = (uint32_t)(at + o + 4); o_len[base + cnt] = n; }
          cnt++;
          o += 4u + n;   // (n <= the page's bytes < 2^32 - 2^24: no wrap worth guarding beyond `fits`)
        }
        return o;
      };
      // the last non-zero byte in front of the slice (all slices before it): where a run of empty values (zero bytes) that reaches into the slice began
      int lnz = -1;
      for (uint32_t w = s0 / 4; w < s1 / 4; w++) { const uint32_t v = S(w); if (v) lnz = (int)(4 * w + (3u - (uint32_t)__builtin_clz(v) / 8u)); }
      if (tid == 0 && lnz < (int)sh) lnz = (int)sh - 1;   // (the window's first value starts at sh: what lies in front of it is not this window's)
      int left = lnz;
      for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(left, d, 64); if (tid >= d) left = max(left, t); }
      left = __shfl_up(left, 1, 64);
      if (tid == 0) left = -1;
      uint32_t entry = tid == 0 ? sh : s1;   // s1: nothing starts in the slice as far as the lane can tell — the lanes in front will say
      bool pass = false;                     // … and the lane hands its neighbour's exit on
      const bool dead = tid != 0 && at + s0 >= pg.end;   // the slice lies behind the page's last value: whatever arrives passes through
      if (dead) pass = true;
      if (tid != 0 && !dead) {
        if ((S(s0 / 4) | S(s0 / 4 + 1)) == 0u) {   // inside a run of zeros: empty values, aligned with the byte behind the last non-zero one
          const uint32_t first = (uint32_t)(left + 1);
          entry = s0 + ((first + 4u - (s0 & 3u)) & 3u);
        } else {
          // a prefix of a value shorter than 16 MiB ends in a zero byte, text holds none, and what follows a prefix is text: the candidates are the
          // offsets three bytes in front of the LAST zero of a run of zeros.  The first four are collected first (64 lanes, 65 words, no
          // branches that depend on the data), then proven in lockstep — a proof is a chain of dependent reads, and 64 lanes proving candidates
          // at 64 different moments would run them one after another
          uint32_t cq[4] = {s1, s1, s1, s1};
          uint32_t nc = 0;
          for (uint32_t k = s0 / 4; k <= (s1 + 2) / 4; k++) {
            const uint32_t v = S(k), nx = (v >> 8) | (S(k + 1) << 24);   // nx: the byte behind each byte of v
#pragma unroll
            for (uint32_t b = 0; b < 4; b++) {
              const bool hit = ((v >> (8 * b)) & 0xFFu) == 0u && ((nx >> (8 * b)) & 0xFFu) != 0u;
              const uint32_t z = 4 * k + b;
              if (hit && z >= s0 + 3 && z - 3 < s1 && nc < 4) { cq[nc < 3 ? nc : 3] = z - 3; nc++; }
            }
          }
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const uint32_t q = cq[t];
            // three non-empty values in a row (at most twelve links) that stay inside the page, all of them READ: a link that leaves the window
            // proves nothing (two bytes in front of a true prefix read as a length of 65536 and more), the page's end reached is proof enough
            uint32_t x = q, hops = 0, real = 0; bool ok = entry == s1 && q < s1;
            while (ok && real < 3 && hops < 12) {
              if (at + x == pg.end) break;
              if (x + 4 > PQ_STAGE) { ok = false; break; }
              const uint32_t n = rd(x);
              if (!fits(x, n)) { ok = false; break; }
              hops++; real += n ? 1u : 0u;
              x += 4u + n;
            }
            if (ok) entry = q;
          }
          pass = entry == s1;
        }
      }
      uint32_t cnt = 0; bool bad = false;
      uint32_t ex = walk(entry, cnt, bad, false, 0u);
      // settle: the exit of the nearest lane in front that holds values must be this lane's entry.  Only the FIRST lane that disagrees takes its
      // neighbour's word and walks again (the lanes in front of it all agree: their word is final) — a wrong guess costs one round, and cannot
      // push a wrong exit through lanes that had guessed right
      int rounds = 0;
      for (;; rounds++) {
        int src = pass ? -1 : tid;   // the nearest lane in front that holds values (lane 0 always does)
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(src, d, 64); if (tid >= d) src = max(src, t); }
        src = __shfl_up(src, 1, 64);
        const uint32_t front = (uint32_t)__shfl((int)ex, tid ? src : 0, 64);   // (lane 0 never looks at it)
        bool need = false;
        if (tid != 0) {
          if (pass) { if (front >= s1 || dead) { entry = front; ex = front; cnt = 0; } else need = true; }
          else need = front != entry;
        }
        const uint64_t m = __ballot(need);
        if (!m) { spec_ok = true; break; }
        if (rounds >= 10 || (rounds == 0 && __popcll(m) > 12)) break;
        if (tid == __ffsll((long long)m) - 1) { pass = false; entry = front; ex = walk(entry, cnt, bad, false, 0u); }
      }
      uint32_t incl = cnt;
      for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64); if (tid >= d) incl += t; }
      const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
      const uint32_t last_exit = (uint32_t)__shfl((int)ex, 63, 64);
      spec_ok = spec_ok && !__ballot(bad) && total <= want && total <= PQ_OUT && total > 0;
      if (spec_ok) {
        walk(entry, cnt, bad, true, incl - cnt);
        if (tid == 0) { s_cnt = total; s_at = at + last_exit; }
      }
      if (tid == 0) s_spec = spec_ok ? 1u : 0u;
      if (dbg && tid == 0) {  // TFGPU_PQ_DEBUG=1: how the windows went
        atomicAdd(dbg, 1u); atomicAdd(dbg + 5, (uint32_t)rounds);
        if (spec_ok) atomicAdd(dbg + 1, 1u); else if (rounds >= 10 || rounds == 0) atomicAdd(dbg + 2, 1u); else if (__ballot(bad)) atomicAdd(dbg + 3, 1u); else atomicAdd(dbg + 4, 1u);
      }
    }
    if (tid == 0 && !spec_on) s_spec = 0;
    if (spec_on == 2) for (uint32_t i = (uint32_t)tid; i < 800u; i += 256) { o_off[i] = (uint32_t)at; o_len[i] = 0; }
    __syncthreads();
    if (tid == 0 && !s_spec) {
      uint64_t o = sh;  // byte offset of the next prefix in the window
      uint32_t cnt = 0;
      const uint32_t want = pg.count - done;
      while (cnt < want && o + 4 <= PQ_STAGE) {
        const uint32_t oo = (uint32_t)o;
        const uint64_t abs = at + o;
        const uint32_t n = __builtin_amdgcn_alignbyte(S((oo >> 2) + 1), S(oo >> 2), oo & 3u);
        if (abs + 4 > pg.end || (uint64_t)n > pg.end - (abs + 4)) { s_bad = 1; break; }  // a prefix that points past its page: this value and the rest of the page read as empty, the call fails
        o_off[cnt] = (uint32_t)(abs + 4); o_len[cnt] = n;
        cnt++;
        o += 4ull + n;
      }
      s_cnt = cnt;
      s_at = at + o;
    }
    __syncthreads();
    const uint32_t cnt = s_cnt;
    for (uint32_t i = (uint32_t)tid; i < cnt; i += 256) { val_off[pg.ord + done + i] = o_off[i]; val_len[pg.ord + done + i] = o_len[i]; }
    __syncthreads();
    if (tid == 0) s_done = done + cnt;
    __syncthreads();
  }
  if (dbg && tid == 0) { atomicMax(dbg + 6, (uint32_t)((pg.end - pg.at + PQ_STAGE - 1) / PQ_STAGE)); atomicAdd(dbg + 7, 1u); }
  if (s_bad) {
    for (uint32_t o = s_done + (uint32_t)tid; o < pg.count; o += 256) { val_off[pg.ord + o] = (uint32_t)pg.at; val_len[pg.ord + o] = 0; }
    if (tid == 0) *err = PQE_TEXT_LENGTH;
  }
}
// … every text column in one launch (grid.y = column), and the words the host wants from many buffers in one read-back
struct PackDesc { const uint32_t *dst_off; uint8_t *dst; const uint32_t *src_off; };
template <int RPT>
__global__ void __launch_bounds__(256) pq_pack_text_all(const PackDesc *descs, int64_t nrows, const uint8_t *file) {
  __shared__ uint32_t doff[256 * RPT + 1], soff[256 * RPT];
  const PackDesc d = descs[blockIdx.y];
  const int64_t k0 = (int64_t)blockIdx.x * (256 * RPT);
  if (k0 >= nrows) return;
  segcopy_run<RPT>(d.dst_off, nrows, k0, file, d.dst, [&](int64_t r) { return d.src_off[r]; }, doff, soff);
}
__global__ void pq_gather_words(const uint32_t *const *at, int n, uint32_t *out) {
  for (int k = threadIdx.x; k < n; k += blockDim.x) out[k] = *at[k];
}

// ---- pass 2, one launch per step over ALL columns of a kind (grid.y = column): a hundred columns launched one by one cost a
//      hundred times a HIP call's ~15 us for every step, more than the kernels run ----
struct LevDesc { const Seg *segs; int32_t nsegs, pad; uint32_t *present; uint8_t *bits; };   // present: u32[stride] (row flags → ranks after the scan); bits: the validity bitmap or null
struct FixedDesc { const Seg *segs; int32_t nsegs, pad; const uint32_t *rank; const uint64_t *dict_at; const uint32_t *dict_n; FixedOut o; };
struct TextDesc { const Seg *segs; int32_t nsegs, pad; const uint32_t *rank, *val_off, *val_len, *dict_off, *dict_len, *dict_n; uint32_t *src_off, *lens; unsigned long long *total; };
__global__ void __launch_bounds__(256) pq_levels_all(const uint8_t *file, const LevDesc *descs, int64_t nrows) {
  const LevDesc &d = descs[blockIdx.y];
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  d.present[r] = d.nsegs ? (seg_value(file, find_seg(d.segs, d.nsegs, (uint32_t)r), (uint32_t)r) != 0 ? 1u : 0u) : 1u;
}
__global__ void __launch_bounds__(256) pq_pack_validity_all(const LevDesc *descs, int64_t nrows) {
  const LevDesc &d = descs[blockIdx.y];
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (!d.bits || b * 8 >= nrows) return;
  const uint32_t *rank = d.present;
  uint32_t v = 0;
  for (int j = 0; j < 8; j++) { const int64_t r = b * 8 + j; if (r < nrows && rank[r + 1] != rank[r]) v |= 1u << j; }
  d.bits[b] = (uint8_t)v;
}
__device__ __forceinline__ void pq_fixed_row(const uint8_t *file, uint64_t tail_base, const Seg *segs, int32_t nsegs, const uint32_t *rank, int64_t r, const uint64_t *dict_at, const uint32_t *dict_n, const FixedOut &o, uint32_t *err, const Seg &s) {
  const uint32_t k = rank ? rank[r] : (uint32_t)r;   // (no rank: every row holds a value, its ordinal is the row)
  uint64_t v = 0;
  int32_t ns = 0;
  if (!rank || rank[r + 1] != k) {
    const uint32_t w = (uint32_t)o.in_width;
    if (s.kind == SG_PLAIN) v = load_unaligned(file, s.at + (s.in_tail ? tail_base : 0) + (uint64_t)(k - s.start) * w, w);
    else if (s.kind == SG_PLAIN_BOOL) v = (file[s.at + ((k - s.start) >> 3)] >> ((k - s.start) & 7)) & 1;
    else if (s.kind == SG_BSS) { for (uint32_t j = 0; j < w; j++) v |= (uint64_t)file[s.at + (uint64_t)j * s.count + (k - s.start)] << (8 * j); }   // (a wave reads w runs of 64 consecutive bytes)
    else {
      const uint32_t idx = seg_value(file, s, k, tail_base);
      if (o.in_width == 1 && !dict_at) v = idx;  // (booleans under RLE are their own values)
      else if (idx >= dict_n[s.dict_base]) *err = PQE_DICT_INDEX;  // an index past the chunk's dictionary: the value stays 0, the call fails
      else v = load_unaligned(file, dict_at[s.dict_base] + (uint64_t)idx * w, w);
    }
    switch (o.conv) {
      case CV_DATE: v = (uint64_t)((int64_t)(int32_t)v * 86400); break;
      case CV_TS_MICROS: { const int64_t us = (int64_t)v; int64_t sec = us / 1000000, rem = us % 1000000; if (rem < 0) { rem += 1000000; sec--; } v = (uint64_t)sec; ns = (int32_t)(rem * 1000); break; }
      case CV_I32_I64: v = (uint64_t)(int64_t)(int32_t)v; break;
      case CV_I32_U64: v = (int32_t)v < 0 ? 0 : (uint64_t)(int32_t)v; break;
      case CV_I64_U64: v = (int64_t)v < 0 ? 0 : v; break;
      case CV_F32_F64: v = (uint64_t)__double_as_longlong((double)__uint_as_float((uint32_t)v)); break;
      default: break;
    }
  }
  switch (o.out_width) {
    case 1: ((uint8_t *)o.values)[r] = (uint8_t)v; break;
    case 4: ((uint32_t *)o.values)[r] = (uint32_t)v; break;
    default: ((uint64_t *)o.values)[r] = v;
  }
  if (o.nanos) o.nanos[r] = ns;
}
__global__ void __launch_bounds__(256) pq_values_all(const uint8_t *file, uint64_t tail_base, const FixedDesc *descs, int64_t nrows, uint32_t *err) {
  __shared__ int hint;
  const FixedDesc &d = descs[blockIdx.y];
  const int64_t r0 = (int64_t)blockIdx.x * blockDim.x, rl = r0 + threadIdx.x, r = rl < nrows ? rl : nrows - 1;  // (every thread reaches the barrier of the block search)
  const Seg &s = find_seg_block(d.segs, d.nsegs, d.rank ? d.rank[r] : (uint32_t)r, d.rank ? d.rank[r0] : (uint32_t)r0, &hint);
  if (rl >= nrows) return;
  pq_fixed_row(file, tail_base, d.segs, d.nsegs, d.rank, r, d.dict_at, d.dict_n, d.o, err, s);
}
// four rows a thread (a workgroup = 1024 consecutive rows of one column: one block search of the segment table per 1024 rows); the
// column's bytes in 64 bits are summed from the lengths by sum_u32_segments_u64 afterwards (an atomic per wave on one word per column was
// most of this kernel's time)
constexpr int PQ_TEXT_RPT = 4;
__global__ void __launch_bounds__(256) pq_text_cells_all(const uint8_t *file, uint64_t tail_base, const TextDesc *descs, int64_t nrows, uint32_t *err) {
  __shared__ int hint;
  const TextDesc &d = descs[blockIdx.y];
  const int64_t r0 = (int64_t)blockIdx.x * (256 * PQ_TEXT_RPT);
  if (r0 >= nrows) return;  // (uniform)
  if (threadIdx.x == 0) {
    const uint32_t first_ord = d.rank ? d.rank[r0] : (uint32_t)r0;
    int lo = 0, hi = d.nsegs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (d.segs[mid].start <= first_ord) lo = mid; else hi = mid - 1; }
    hint = lo;
  }
  __syncthreads();
  int si = hint;
#pragma unroll
  for (int j = 0; j < PQ_TEXT_RPT; j++) {
    const int64_t r = r0 + (int64_t)j * 256 + threadIdx.x;
    if (r >= nrows) break;
    const uint32_t k = d.rank ? d.rank[r] : (uint32_t)r;
    while (si + 1 < d.nsegs && d.segs[si + 1].start <= k) si++;   // ordinals grow with the rows: forward only
    const Seg &s = d.segs[si];
    uint32_t so = SEG_NONE, n = 0;
    if (!d.rank || d.rank[r + 1] != k) {
      if (s.kind == SG_PLAIN_TEXT) { so = d.val_off[k]; n = d.val_len[k]; }
      else if (s.kind == SG_FIXED_TEXT) { so = (uint32_t)(s.at + (uint64_t)(k - s.start) * s.bw); n = s.bw; }
      else {
        const uint32_t i = seg_value(file, s, k, tail_base);
        if (i >= d.dict_n[s.dict_base]) *err = PQE_DICT_INDEX;  // dict_n[first entry of a chunk's dictionary] = its entries
        else { so = d.dict_off[s.dict_base + i]; n = d.dict_len[s.dict_base + i]; }
      }
    }
    d.src_off[r] = n ? so : SEG_NONE; d.lens[r] = n;
  }
}
__global__ void __launch_bounds__(256) pq_row_index(uint64_t *v, int64_t nrows) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nrows) v[r] = (uint64_t)r + 1;
}
__global__ void __launch_bounds__(256) pq_file_name(const uint8_t *name, uint32_t n, int64_t nrows, uint32_t *off, uint8_t *data) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > nrows) return;
  off[r] = (uint32_t)((uint64_t)r * n);
  if (r < nrows) for (uint32_t i = 0; i < n; i++) data[(uint64_t)r * n + i] = name[i];
}

}  // namespace pq
}  // namespace tf

using namespace tf;
using namespace tf::pq;

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

struct NeedArena {};  // thrown by the walk when a page must be rewritten on the host (DELTA_BYTE_ARRAY) and the object was to be uploaded as it is

// An uncompressed object goes up in pieces on the lane's copy stream, an event behind each: the page kernels of a piece are queued behind
// ITS event, so the walks of the pages already in HBM run beside the upload of the rest instead of after all of it.
struct UploadPieces {
  std::vector<uint64_t> end;      // piece i = [end[i - 1], end[i]) of the object
  std::vector<hipEvent_t> ev;
  hipStream_t main = nullptr;
  ~UploadPieces() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); if (first_) (void)hipEventDestroy(first_); }
  int piece_of(uint64_t last_byte_end) const { int i = 0; while (i + 1 < (int)end.size() && end[(size_t)i] < last_byte_end) i++; return i; }
  void wait(int i) const { if (i >= 0 && i < (int)ev.size()) TF_HIP(hipStreamWaitEvent(main, ev[(size_t)i], 0)); }
  void start(void *dst, const uint8_t *src, uint64_t len) {
    Context &cx = ctx();
    main = cx.stream;
    if (!cx.copy_stream) TF_HIP(hipStreamCreateWithFlags(&cx.copy_stream, hipStreamNonBlocking));
    static const int forced = [] { const char *e = std::getenv("TFGPU_PQ_PIECES"); return e ? std::atoi(e) : 0; }();  // A/B runs (1 = one copy, as before)
    // ONE piece unless asked: measured on the MI355X (gpurun r11n, the 556 MB hits object) eight pieces on the copy stream make the read 41.8 ms
    // against 22.8 with one copy on the lane's stream — the copy stream's transfers and the kernels waiting on their events do not run side by side here
    const int k = forced > 0 ? forced : 1;
    TF_HIP(hipEventCreateWithFlags(&first_, hipEventDisableTiming));
    TF_HIP(hipEventRecord(first_, main));
    TF_HIP(hipStreamWaitEvent(cx.copy_stream, first_, 0));   // whoever used this block before on the lane's stream is done first
    uint64_t at = 0;
    for (int i = 0; i < k; i++) {
      const uint64_t gran = forced > 0 ? 0xFFFull : 0xFFFFFull;   // (forced: small test objects are cut too)
      const uint64_t z = i + 1 == k ? len : std::min<uint64_t>(len, ((len / (uint64_t)k) * (uint64_t)(i + 1) + gran) & ~gran);
      if (z > at) TF_HIP(hipMemcpyAsync((char *)dst + at, src + at, (size_t)(z - at), hipMemcpyHostToDevice, cx.copy_stream));
      hipEvent_t e;
      TF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      TF_HIP(hipEventRecord(e, cx.copy_stream));
      ev.push_back(e); end.push_back(z);
      at = z;
    }
  }
  hipEvent_t first_ = nullptr;
};
// pages of one kind cut into runs by the piece their last byte arrives with: run i is launched behind piece i's event
template <class T, class EndOf> static std::vector<std::pair<size_t, size_t>> by_piece(std::vector<T> &pages, const UploadPieces &up, EndOf end_of) {
  std::vector<std::pair<size_t, size_t>> runs;
  if (up.end.empty()) { runs.push_back({0, pages.size()}); return runs; }
  std::stable_sort(pages.begin(), pages.end(), [&](const T &a, const T &b) { return end_of(a) < end_of(b); });
  size_t at = 0;
  for (size_t i = 0; i < up.end.size(); i++) {
    size_t z = at;
    while (z < pages.size() && (end_of(pages[z]) <= up.end[i] || i + 1 == up.end.size())) z++;
    runs.push_back({at, z});
    at = z;
  }
  return runs;
}

// What the tail (the bytes decoded beside the object: expanded dictionary indices, DELTA_BINARY_PACKED values, INT96 texts) can need at
// most, from the footer alone: a chunk lists its encodings.
static uint64_t tail_bound(const FileMeta &m, const std::vector<TopField> &fields, int64_t nrows) {
  uint64_t bound = 64;
  for (size_t k = 0; k < fields.size(); k++) {
    if (fields[k].group) continue;
    const SchemaEl &e = fields[k].el;
    if (e.type == T_INT96) bound += (uint64_t)std::max<int64_t>(nrows, 1) * 32 + 16;
    for (auto &g : m.groups) {
      const ColChunk &c = g.cols[(size_t)fields[k].leaf];
      if (c.enc_mask & ((1u << E_PLAIN_DICT) | (1u << E_RLE_DICT) | (e.type == T_BOOLEAN ? (1u << E_RLE) : 0u)))   // expanded indices: at most 4 bytes a value
        bound += (uint64_t)std::max<int64_t>(c.num_values, 0) * 4 + 16 * (uint64_t)(std::max<int64_t>(c.num_values, 0) / 64 + 2);
      if ((c.enc_mask >> E_DELTA_BINARY_PACKED) & 1u) bound += (uint64_t)std::max<int64_t>(c.num_values, 0) * 8 + 16 * (uint64_t)(std::max<int64_t>(c.num_values, 0) / 64 + 2);  // (every page's share is rounded up to 16 bytes)
    }
  }
  return bound;
}

static int parquet_read_impl(const uint8_t *f, uint64_t len, const tfgpu_schema *schema, const char *table_ns, const char *table_name, const char *file_name, bool force_arena, tfgpu_dbatch **out, Buf staged = nullptr) {
  PqClock clk;
  FileMeta m; std::string why;
  if (!parse_footer(f, len, m, why)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: " + why);
  clk.at("footer parsed");
  PQD("footer: %zu schema els, %zu groups, %lld rows\n", m.schema.size(), m.groups.size(), (long long)m.num_rows);
  std::vector<TopField> fields; int nleaves = 0;
  if (!top_fields(m, fields, nleaves, why)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: " + why);
  int64_t nrows = 0;
  for (auto &g : m.groups) { nrows += g.num_rows; if ((int)g.cols.size() != nleaves) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: a row group does not hold every leaf column"); }
  if (nrows > 0x7FFFFFF0ll) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: more than 2^31 rows in one object");
  Context &cx = ctx();
  std::lock_guard<std::mutex> lk(cx.mu);
  hipStream_t st = cx.stream;
  // An object without compressed chunks is uploaded as it is and every offset below is a file offset.  With compressed chunks (or a
  // page the host has to rewrite) the host inflates each page it walks into `arena`, offsets are arena offsets, and the arena is
  // what the device gets.  Either way the device image gets a TAIL behind it: room for what kernels decode for other kernels
  // (DELTA_BINARY_PACKED values, INT96 texts), sized by the walk.
  bool inflate = force_arena;
  for (auto &g : m.groups) for (auto &c : g.cols) if (c.codec != C_UNCOMPRESSED) inflate = true;
  RawVec &arena = thread_arena();   // (empty again; its pinned block is the calling thread's from the last call)
  // Pages of the byte-oriented codecs (SNAPPY, LZ4_RAW) whose content the host's walk does not need — PLAIN values, dictionary indices,
  // RLE booleans — are inflated ON THE DEVICE (pq_inflate, one wave a page): the compressed object goes up as it is, beside this walk,
  // and the first `dev_region` bytes of the image are theirs (the arena keeps that room free and is uploaded from there on).  What the
  // walk reads at a page's front (definition levels, the index width) the host inflates as a prefix.  Dictionary pages, DELTA_* pages
  // and the other codecs (GZIP, ZSTD: entropy coders, a chain per bit) are the host's, as before.  TFGPU_PQ_DEVICE_INFLATE=0: all on the host (A/B).
  // Which pages: measured (profiles/r20j_pq_inflate_shapes.txt, one page a wave) the kernel copies stored LITERALS at 8-20 GB/s a page but parses a
  // stream of short elements at ~6 M elements/s — 50 MB/s on text, and hardly more on values that "barely compress" (a 24-byte value with a repeated
  // 4-byte length prefix is two elements) — and a wave's pages cannot be cut: the launch lasts as long as its LARGEST page (91 ms for the hits
  // object's multi-megabyte text pages), while sixteen host cores inflate the same object's pages side by side in 46 ms.  So by default
  // (TFGPU_PQ_DEVICE_INFLATE unset or "auto") the device takes the pages it is at least as good at — small ones (<= 128 KiB inflated: a few
  // milliseconds each, hundreds in flight: objects of many small row groups or pages) and stored ones (compressed >= inflated: one long literal) —
  // and the host's cores the rest; "1": every eligible page on the device; "0": none.  (Read per call: bench.py and the tests A/B it in one process.)
  const int dev_inflate_mode = [] { const char *e = std::getenv("TFGPU_PQ_DEVICE_INFLATE"); return !e || !*e || *e == 'a' ? 2 : (*e == '0' ? 0 : 1); }();
  const bool dev_inflate_on = dev_inflate_mode != 0;
  auto dev_page = [dev_inflate_mode](const ColChunk &c, const PageHeader &h, int leaf_type) {
    if (c.codec != C_SNAPPY && c.codec != C_LZ4_RAW) return false;
    if (!(h.type == 0 || h.type == 3) || (h.type == 3 && !h.compressed_v2)) return false;
    const bool enc_ok = h.enc == E_PLAIN || h.enc == E_BYTE_STREAM_SPLIT || h.enc == E_RLE_DICT || h.enc == E_PLAIN_DICT || (h.enc == E_RLE && leaf_type == T_BOOLEAN);
    const int64_t lead = h.type == 3 ? (int64_t)std::max(h.rep_len, 0) + (int64_t)std::max(h.def_len, 0) : 0;
    if (!(enc_ok && (int64_t)h.usize > lead && (int64_t)h.csize > lead)) return false;
    if (dev_inflate_mode == 2 && (int64_t)h.usize - lead > 128 * 1024 && (int64_t)h.csize < (int64_t)h.usize) return false;   // a large page of many elements: the host's cores
    return true;
  };
  uint64_t dev_region = 0, dev_cursor = 0;
  std::vector<InfPage> infpages;
  Buf cfile;
  if (inflate && dev_inflate_on) {
    // an upper bound of the region from the page headers alone (the walk below places the pages; a header that does not parse ends the
    // count — the walk will say why)
    uint64_t d = 0; bool any = false;
    for (size_t k = 0; k < fields.size() && d < 0xF0000000ull; k++) {
      if (fields[k].group) continue;
      for (auto &g : m.groups) {
        const ColChunk &c = g.cols[(size_t)fields[k].leaf];
        if ((c.codec != C_SNAPPY && c.codec != C_LZ4_RAW) || c.data_off < 0 || c.total_comp < 0) continue;
        uint64_t pos = (uint64_t)((c.dict_off > 0 && c.dict_off < c.data_off) ? c.dict_off : c.data_off);
        const uint64_t end = pos + (uint64_t)c.total_comp;
        if (end > len || end < pos) continue;
        int64_t seen = 0;
        while (pos < end && seen < c.num_values) {
          TReader r{f + pos, f + end};
          PageHeader h;
          if (!parse_page_header(r, h) || h.csize < 0 || h.usize < 0 || h.nvalues < 0 || (uint64_t)h.csize > (uint64_t)(f + end - r.p)) break;
          if (dev_page(c, h, fields[k].el.type)) { d = ((d + 15) & ~15ull) + (uint64_t)h.usize; any = true; }
          if (h.type == 0 || h.type == 3) seen += h.nvalues;
          pos = (uint64_t)(r.p - f) + (uint64_t)h.csize;
        }
      }
    }
    if (any && d < 0xF0000000ull) {
      dev_region = (d + 15) & ~15ull;
      cfile = dalloc((size_t)len + 1024);   // (the kernel's input windows read up to 512 bytes past a page's last byte)
    }
  }
  UploadPieces up;
  if (cfile) up.start(cfile->p, f, len);
  if (inflate) { arena.reserve((size_t)dev_region + (size_t)len + (size_t)len / 2 + 4096); arena.resize((size_t)dev_region); }
  // An uncompressed object starts uploading NOW, beside the host's page walk: what the tail can need at most is in the footer (a
  // chunk lists its encodings: DELTA_BINARY_PACKED values decode into the tail, an INT96 column's texts live there, a
  // DELTA_BYTE_ARRAY chunk sends the whole object through the arena instead).
  Buf file;
  uint64_t tail_base = 0, tail_cap = 0;
  if (!inflate) {
    tail_base = (len + 64 + 15) & ~15ull;
    tail_cap = tail_bound(m, fields, nrows);
    if (tail_base + tail_cap + 64 >= 0xFFFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: the object and what is decoded beside it exceed 4 GiB: read it row group by row group");
    if (staged) {  // the object is in HBM already (tfgpu_parquet_read_staged: somebody else's copy engine brought it), with room for the tail behind it
      if (staged->bytes < (size_t)(tail_base + tail_cap) + 64) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read_staged: the staging buffer is smaller than tfgpu_parquet_staging_size asks for");
      file = staged;
    } else {
      file = dalloc((size_t)(tail_base + tail_cap) + 64);
      up.start(file->p, f, len);
    }
    TF_HIP(hipMemsetAsync((char *)file->p + len, 0, (size_t)(tail_base - len), st));
    clk.at(staged ? "staged object taken" : "upload issued");
  }

  auto db = std::make_unique<tfgpu_dbatch>();
  db->nrows = nrows; db->ns = table_ns ? table_ns : ""; db->table = table_name ? table_name : "";
  enum { W_NIL = -1, W_FILE_NAME = -2, W_ROW_INDEX = -3 };
  std::vector<int> want;  // field index per output column, or W_*: the file does not have it → nil (reader_parquet.go:256-259); a system column
  std::vector<std::pair<std::string, int>> outcols;
  if (schema) {
    for (int i = 0; i < schema->ncols; i++) {
      const char *nm = schema->cols[i].name ? schema->cols[i].name : "";
      int at = W_NIL;
      for (size_t k = 0; k < fields.size(); k++) if (fields[k].el.name == nm) at = (int)k;
      // constructCI (reader_parquet.go:243-255): the system columns are the reader's own, whatever the file holds under those names
      if (file_name && !std::strcmp(nm, "__file_name")) at = W_FILE_NAME;
      if (file_name && !std::strcmp(nm, "__row_index")) at = W_ROW_INDEX;
      want.push_back(at); outcols.push_back({nm, schema->cols[i].dtype});
      db->schema.push_back({nm, schema->cols[i].dtype});
      if (schema->cols[i].flags & TFGPU_COL_KEY) db->key_names.push_back(nm);
    }
  } else for (size_t k = 0; k < fields.size(); k++) { want.push_back((int)k); outcols.push_back({fields[k].el.name, -1}); }

  std::vector<Buf> keep;  // tables the kernels read until the final sync
  // ---- the pages the HOST inflates (GZIP / ZSTD everywhere; dictionary and DELTA_* pages of every codec; everything with the device codec
  //      switched off) are inflated AHEAD of the walk, side by side on the host's cores: pages are independent, the reference's reader inflates them
  //      one by one in its read loop (reader_parquet.go:137-283) and so did the walk below until round 6 (one thread, ~1 GB/s).  Each gets its
  //      place in the arena here; the walk finds it by the page's file offset and reads it as if it had just inflated it. ----
  std::unordered_map<uint64_t, size_t> preinflated;   // file offset of a page's payload → its offset in the arena
  if (inflate) {
    struct HostPage { const uint8_t *raw; uint64_t csize, usize, lead; int codec; size_t at0; };
    std::vector<HostPage> hp;
    size_t cur = arena.size();
    bool fits = true;
    for (size_t oc = 0; oc < want.size() && fits; oc++) {
      if (want[oc] < 0) continue;
      const TopField &fld = fields[(size_t)want[oc]];
      if (fld.group || fld.el.logical == L_DECIMAL) continue;
      for (auto &g : m.groups) {
        const ColChunk &c = g.cols[(size_t)fld.leaf];
        if (c.codec == C_UNCOMPRESSED || c.data_off < 0 || c.total_comp < 0) continue;
        if (c.codec != C_SNAPPY && c.codec != C_GZIP && c.codec != C_ZSTD && c.codec != C_LZ4_RAW) continue;
        uint64_t pos = (uint64_t)((c.dict_off > 0 && c.dict_off < c.data_off) ? c.dict_off : c.data_off);
        const uint64_t end = pos + (uint64_t)c.total_comp;
        if (end > len || end < pos) continue;
        int64_t seen = 0;
        while (pos < end && seen < c.num_values) {
          TReader r{f + pos, f + end};
          PageHeader h;
          if (!parse_page_header(r, h) || h.csize < 0 || h.usize < 0 || h.nvalues < 0 || (uint64_t)h.csize > (uint64_t)(f + end - r.p)) break;
          const uint8_t *raw = r.p;
          if (h.type == 0 || h.type == 3) seen += h.nvalues;
          pos = (uint64_t)(raw - f) + (uint64_t)h.csize;
          if (!(h.type == 0 || h.type == 2 || h.type == 3)) continue;
          const uint64_t lead = h.type == 3 ? (uint64_t)std::max(h.rep_len, 0) + (uint64_t)std::max(h.def_len, 0) : 0;
          if (lead > (uint64_t)h.csize || lead > (uint64_t)h.usize) continue;   // (the walk reports it)
          const int codec = (h.type == 3 && !h.compressed_v2) ? (int)C_UNCOMPRESSED : c.codec;
          if (dev_region && dev_page(c, h, fld.el.type)) continue;               // the device's
          if ((uint64_t)h.usize - lead > max_inflated(codec, (uint64_t)h.csize - lead)) continue;
          const size_t at0 = (cur + 15) & ~(size_t)15;
          if ((uint64_t)at0 + (uint64_t)h.usize + 64 >= 0xFFFFFFF0ull) { fits = false; break; }
          hp.push_back(HostPage{raw, (uint64_t)h.csize, (uint64_t)h.usize, lead, codec, at0});
          cur = at0 + (size_t)h.usize;
        }
        if (!fits) break;
      }
    }
    if (fits && !hp.empty()) {
      arena.resize(cur);
      uint8_t *const base = arena.data();
      std::atomic<size_t> next{0};
      std::atomic<int> failed{-1};
      std::vector<std::string> whys(hp.size());
      auto work = [&]() {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= hp.size()) return;
          const HostPage &q = hp[i];
          std::memcpy(base + q.at0, q.raw, (size_t)q.lead);
          if (!page_inflate(q.codec, q.raw + q.lead, q.csize - q.lead, base + q.at0 + q.lead, q.usize - q.lead, whys[i])) { int none = -1; failed.compare_exchange_strong(none, (int)i); }
        }
      };
      const unsigned nthreads = (unsigned)std::min<size_t>(hp.size(), inflate_threads());
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nthreads; t++) pool.emplace_back(work);
      work();
      for (auto &t : pool) t.join();
      if (failed.load() >= 0) {   // (the walk meets the page again and reports it with its column's name: it inflates it inline)
        arena.resize((size_t)dev_region);
      } else for (auto &q : hp) preinflated[(uint64_t)(q.raw - f)] = q.at0;
      clk.at("host pages inflated");
    }
  }
  // ---- pass 1 (host): every column's segment tables ----
  struct ColPlan {
    DColumn d; bool nil = false, optional = false, is_text = false, is_int96 = false; uint32_t width = 0; int32_t conv = CV_SAME, out_width = 0;
    std::vector<Seg> lev, val; std::vector<TextPage> tpages; std::vector<uint64_t> dict_at; std::vector<uint32_t> dict_off, dict_len, dict_n;  // dict_n: entries per chunk dictionary (fixed: by chunk; text: at the chunk's first entry)
    uint64_t ord = 0; size_t arena_at = 0; uint64_t slot0 = 0;
  };
  std::vector<ColPlan> plans(want.size());
  std::vector<DMini> minis; std::vector<DPage> dpages; std::vector<HPage> hpages;
  uint64_t tail_need = 0;
  auto tail_take = [&](uint64_t bytes) { const uint64_t at = tail_need; tail_need += (bytes + 15) & ~15ull; return at; };
  for (size_t oc = 0; oc < want.size(); oc++) {
    ColPlan &P = plans[oc];
    DColumn &d = P.d;
    d.name = outcols[oc].first;
    auto all_nil = [&](int dtype) {
      P.nil = true;
      d.repr = TFGPU_R_STRING; d.dtype = dtype;
      d.offsets = dalloc_zero((size_t)(nrows + 1) * 4); d.data = dalloc(8); d.data_len = 0; d.validity = dalloc_zero((size_t)(nrows + 7) / 8 + 8);
    };
    if (want[oc] == W_FILE_NAME || want[oc] == W_ROW_INDEX) { P.nil = true; continue; }  // filled in pass 2
    if (want[oc] < 0) { all_nil(outcols[oc].second >= 0 ? outcols[oc].second : TFGPU_T_UTF8); continue; }
    const TopField &fld = fields[(size_t)want[oc]];
    const SchemaEl &leaf = fld.el;
    if (fld.group) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: column " + leaf.name + ": nested / repeated columns (the `any` tree parquet-go builds) are read by the stock reader");
    PQD("column %s type %d rep %d logical %d conv %d\n", leaf.name.c_str(), leaf.type, leaf.rep, leaf.logical, leaf.conv);
    const int dtype = outcols[oc].second >= 0 ? outcols[oc].second : fld.dtype;
    // A DECIMAL is typed `double` by the resolver and its int32 / int64 / []byte value falls to Restore's default: nil in every row
    // (pinned by the canon's five decimal files).  The pages are not read.
    if (leaf.logical == L_DECIMAL) {
      if (dtype != TFGPU_T_FLOAT64) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: column " + leaf.name + ": a DECIMAL under another DataType than the resolver's double is read by the stock reader");
      all_nil(dtype);
      continue;
    }
    const bool optional = P.optional = leaf.rep == 1;
    const bool is_date = leaf.logical == L_DATE;
    const bool is_int96 = P.is_int96 = leaf.type == T_INT96;
    const bool is_flba = leaf.type == T_FLBA;
    const bool is_text = P.is_text = leaf.type == T_BYTE_ARRAY || is_flba;
    if (is_flba && (leaf.type_len <= 0 || leaf.type_len > (1 << 20))) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: column " + leaf.name + ": FIXED_LEN_BYTE_ARRAY length");
    P.width = leaf.type == T_BOOLEAN ? 1 : (leaf.type == T_INT32 || leaf.type == T_FLOAT) ? 4 : is_int96 ? 12 : is_flba ? (uint32_t)leaf.type_len : 8;
    P.out_width = (int32_t)P.width;
    switch (leaf.type) {
      case T_BOOLEAN: d.repr = TFGPU_R_BOOL; break;
      case T_INT32:
        if (is_date) { d.repr = TFGPU_R_TIME; P.conv = CV_DATE; P.out_width = 8; }
        else if (dtype == TFGPU_T_INT64) { d.repr = TFGPU_R_INT64; P.conv = CV_I32_I64; P.out_width = 8; }
        else if (dtype == TFGPU_T_UINT64) { d.repr = TFGPU_R_UINT64; P.conv = CV_I32_U64; P.out_width = 8; }
        else d.repr = TFGPU_R_INT32;
        break;
      case T_INT64:
        if (dtype == TFGPU_T_TIMESTAMP) { d.repr = TFGPU_R_TIME; P.conv = CV_TS_MICROS; }
        else if (dtype == TFGPU_T_UINT64) { d.repr = TFGPU_R_UINT64; P.conv = CV_I64_U64; }
        else d.repr = TFGPU_R_INT64;
        break;
      case T_FLOAT: if (dtype == TFGPU_T_FLOAT64) { d.repr = TFGPU_R_FLOAT64; P.conv = CV_F32_F64; P.out_width = 8; } else d.repr = TFGPU_R_FLOAT32; break;
      case T_DOUBLE: d.repr = TFGPU_R_FLOAT64; break;
      // []byte under a "string" / "utf8" column is string(v) (restore.go:222-229) — both BYTE_ARRAY flavours and FIXED_LEN_BYTE_ARRAY;
      // INT96 is its decimal text
      default: d.repr = (dtype == TFGPU_T_BYTES || dtype == TFGPU_T_UTF8) ? TFGPU_R_STRING : TFGPU_R_BYTES; break;
    }
    if (is_int96) d.repr = TFGPU_R_STRING;
    d.dtype = dtype;
    if (is_int96) P.slot0 = tail_take((uint64_t)std::max<int64_t>(nrows, 1) * 32);
    auto &lev = P.lev; auto &val = P.val; auto &tpages = P.tpages;
    auto &dict_at = P.dict_at;                               // fixed-width dictionaries: file offset of each chunk's entries
    auto &dict_off = P.dict_off; auto &dict_len = P.dict_len;  // byte-array dictionaries: every entry of every chunk
    auto &dict_n = P.dict_n;
    uint64_t row0 = 0; uint64_t &ord = P.ord;
    for (auto &g : m.groups) {
      const ColChunk &c = g.cols[(size_t)fld.leaf];
      if (c.codec != C_UNCOMPRESSED && c.codec != C_SNAPPY && c.codec != C_GZIP && c.codec != C_ZSTD && c.codec != C_LZ4_RAW)
        return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: column " + leaf.name + ": pages of codec " + std::to_string(c.codec) + " (BROTLI / LZ4 / LZO) are read by the stock reader");
      if (c.data_off < 0 || c.total_comp < 0) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: column chunk out of range");
      uint64_t pos = (uint64_t)((c.dict_off > 0 && c.dict_off < c.data_off) ? c.dict_off : c.data_off);
      const uint64_t end = pos + (uint64_t)c.total_comp;
      if (end > len || end < pos) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: column chunk out of range");
      uint32_t dict_base = 0;
      bool have_dict = false;
      int64_t seen = 0;
      while (pos < end && seen < c.num_values) {
        TReader r{f + pos, f + end};
        PageHeader h;
        if (!parse_page_header(r, h) || h.csize < 0 || h.usize < 0 || h.nvalues < 0) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: malformed page header in column " + leaf.name);
        if ((uint64_t)h.csize > (uint64_t)(f + end - r.p)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: page out of range in column " + leaf.name);
        const uint8_t *raw = r.p;                 // the page as the file holds it
        const uint64_t next = (uint64_t)(raw - f) + (uint64_t)h.csize;
        // fb: what offsets are counted from; [pl, pe): the page's payload as the device will see it
        const uint8_t *fb = f, *pl = raw, *pe = raw + h.csize;
        PQD("  page at %llu type %d csize %d usize %d nvalues %d enc %d v2 %d\n", (unsigned long long)pos, h.type, h.csize, h.usize, h.nvalues, h.enc, h.v2);
        size_t page_at0 = 0;
        if (inflate && (h.type == 0 || h.type == 2 || h.type == 3)) {
          // v2 keeps the levels in front of the (possibly compressed) values; v1 and dictionary pages are compressed whole
          const uint64_t lead = h.type == 3 ? (uint64_t)std::max(h.rep_len, 0) + (uint64_t)std::max(h.def_len, 0) : 0;
          if (lead > (uint64_t)h.csize || lead > (uint64_t)h.usize) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: level lengths of a v2 page in column " + leaf.name);
          const int codec = (h.type == 3 && !h.compressed_v2) ? (int)C_UNCOMPRESSED : c.codec;
          if ((uint64_t)h.usize - lead > max_inflated(codec, (uint64_t)h.csize - lead))
            return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: column " + leaf.name + ": a page states an uncompressed size its " + std::to_string(h.csize) + " bytes cannot inflate to");
          const uint64_t dev_at = (dev_cursor + 15) & ~15ull;
          if (dev_region && dev_page(c, h, leaf.type) && dev_at + (uint64_t)h.usize <= dev_region) {
            // the device inflates this page into its place in front of the arena; the host reads only what the walk needs of its front
            const size_t at0 = page_at0 = (size_t)dev_at;
            dev_cursor = dev_at + (uint64_t)h.usize;
            std::memcpy(arena.data() + at0, raw, (size_t)lead);
            const uint8_t *cp = raw + lead, *ce = raw + h.csize;
            uint8_t *dst = arena.data() + at0 + lead;
            const uint64_t body = (uint64_t)h.usize - lead;
            auto prefix = [&](uint64_t want) { return codec == C_SNAPPY ? snappy_inflate_prefix(cp, ce, dst, body, want) : lz4_raw_inflate_prefix(cp, ce, dst, body, want); };
            const bool bad_page_msg = false; (void)bad_page_msg;
            uint64_t want = 0;
            if (leaf.rep == 1 && !h.v2) {  // v1: a 4-byte length, then the RLE levels
              if (!prefix(4)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: column " + leaf.name + ": malformed " + (codec == C_SNAPPY ? "SNAPPY" : "LZ4_RAW") + " page");
              uint32_t L = 0; if (body >= 4) std::memcpy(&L, dst, 4);
              want = 4 + (uint64_t)L;
            }
            want += (h.enc == E_RLE && leaf.type == T_BOOLEAN) ? 4 : ((h.enc == E_PLAIN || h.enc == E_BYTE_STREAM_SPLIT) ? 0 : 1);
            if (want && !prefix(want)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: column " + leaf.name + ": malformed " + (codec == C_SNAPPY ? "SNAPPY" : "LZ4_RAW") + " page");
            infpages.push_back(InfPage{(uint64_t)(raw - f), (uint64_t)(raw - f) + (uint64_t)h.csize, (uint64_t)at0, (uint32_t)body, (uint32_t)lead, (uint32_t)codec, 0u});
            fb = arena.data(); pl = fb + at0; pe = pl + h.usize;
          } else if (auto pre = preinflated.find((uint64_t)(raw - f)); pre != preinflated.end()) {
            page_at0 = pre->second;   // inflated ahead of the walk, beside the other host pages
            fb = arena.data(); pl = fb + page_at0; pe = pl + h.usize;
          } else {
          const size_t at0 = page_at0 = (arena.size() + 15) & ~(size_t)15;
          if ((uint64_t)at0 + (uint64_t)h.usize + 64 >= 0xFFFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: the inflated pages of one object exceed 4 GiB: read it row group by row group");
          arena.resize(at0 + (size_t)h.usize);
          std::memcpy(arena.data() + at0, raw, (size_t)lead);
          std::string cw;
          if (!page_inflate(codec, raw + lead, (uint64_t)h.csize - lead, arena.data() + at0 + lead, (uint64_t)h.usize - lead, cw))
            return tf::fail(cw.find("need") != std::string::npos || cw.find("codec") != std::string::npos ? TFGPU_ERR_UNSUPPORTED : TFGPU_ERR_INVALID, "tfgpu_parquet_read: column " + leaf.name + ": " + cw);
          fb = arena.data(); pl = fb + at0; pe = pl + h.usize;
          }
        } else if (c.codec != C_UNCOMPRESSED && (h.type == 0 || h.type == 2 || h.type == 3)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: internal: compressed page outside the inflating walk");
        if (h.type == 2) {  // dictionary page: PLAIN entries
          if (h.enc != E_PLAIN && h.enc != E_PLAIN_DICT) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: dictionary page encoding " + std::to_string(h.enc));
          have_dict = true;
          if (is_text) {
            dict_base = (uint32_t)dict_off.size();
            const uint8_t *q = pl;
            for (int i = 0; i < h.nvalues; i++) {
              uint32_t n;
              if (is_flba) { n = P.width; if ((uint64_t)n > (uint64_t)(pe - q)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: truncated dictionary in column " + leaf.name); dict_off.push_back((uint32_t)(q - fb)); q += n; }
              else {
                if (q + 4 > pe) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: truncated dictionary in column " + leaf.name);
                std::memcpy(&n, q, 4);
                if ((uint64_t)n > (uint64_t)(pe - (q + 4))) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: a dictionary entry of column " + leaf.name + " runs past its page");
                dict_off.push_back((uint32_t)(q + 4 - fb));
                q += 4 + (uint64_t)n;
              }
              dict_len.push_back(n); dict_n.push_back(0);
            }
            if (h.nvalues) dict_n[dict_base] = (uint32_t)h.nvalues;
            if (!h.nvalues) { dict_off.push_back(0); dict_len.push_back(0); dict_n.push_back(0); }  // an empty dictionary still has a slot that says so
          } else {
            if ((uint64_t)h.nvalues * P.width > (uint64_t)(pe - pl)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: the dictionary of column " + leaf.name + " runs past its page");
            dict_base = (uint32_t)dict_at.size(); dict_at.push_back((uint64_t)(pl - fb)); dict_n.push_back((uint32_t)h.nvalues);
          }
        } else if (h.type == 0 || h.type == 3) {
          if (h.v2 && h.rep_len) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: repetition levels");
          const uint8_t *q = pl;
          uint64_t present = (uint64_t)h.nvalues;
          if (optional) {
            const uint8_t *ls, *le;
            if (h.v2) { if (h.def_len < 0 || (uint64_t)h.def_len > (uint64_t)(pe - q)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: truncated page"); ls = q; le = q + h.def_len; q = le; }
            else { if (q + 4 > pe) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: truncated page"); uint32_t L; std::memcpy(&L, q, 4); if ((uint64_t)L > (uint64_t)(pe - (q + 4))) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: truncated page"); ls = q + 4; le = ls + L; q = le; }
            if (!h.v2 && h.def_enc != E_RLE) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: definition levels of column " + leaf.name + " are not RLE");
            present = 0;
            if (!hybrid_runs(fb, ls, le, 1, (uint64_t)h.nvalues, row0 + (uint64_t)seen, 0, lev, &present)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: malformed definition levels in column " + leaf.name);
          } else if (h.v2 && h.def_len) { if ((uint64_t)h.def_len > (uint64_t)(pe - q)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: truncated page"); q += h.def_len; }
          if (h.enc == E_PLAIN) {
            if (is_flba) {
              if (present * P.width > (uint64_t)(pe - q)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: the values of a PLAIN page of column " + leaf.name + " run past it");
              val.push_back(Seg{(uint32_t)ord, (uint32_t)present, SG_FIXED_TEXT, P.width, (uint64_t)(q - fb), 0, 0});
            } else if (is_text) { val.push_back(Seg{(uint32_t)ord, (uint32_t)present, SG_PLAIN_TEXT, 0, 0, 0, 0}); tpages.push_back(TextPage{(uint64_t)(q - fb), (uint64_t)(pe - fb), (uint32_t)ord, (uint32_t)present}); }
            else {
              const uint64_t need = leaf.type == T_BOOLEAN ? (present + 7) / 8 : present * P.width;
              if (need > (uint64_t)(pe - q)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: the values of a PLAIN page of column " + leaf.name + " run past it");
              val.push_back(Seg{(uint32_t)ord, (uint32_t)present, leaf.type == T_BOOLEAN ? SG_PLAIN_BOOL : SG_PLAIN, 0, (uint64_t)(q - fb), 0, 0});
            }
          } else if (h.enc == E_BYTE_STREAM_SPLIT && (leaf.type == T_FLOAT || leaf.type == T_DOUBLE || leaf.type == T_INT32 || leaf.type == T_INT64)) {
            // the values' bytes transposed: stream j holds byte j of every value (floats compress better that way); read back per value on the device
            if (present * P.width > (uint64_t)(pe - q)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: the values of a BYTE_STREAM_SPLIT page of column " + leaf.name + " run past it");
            val.push_back(Seg{(uint32_t)ord, (uint32_t)present, SG_BSS, 0, (uint64_t)(q - fb), 0, 0});
          } else if (h.enc == E_RLE_DICT || h.enc == E_PLAIN_DICT) {
            if (q >= pe && present) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: truncated page");
            if (present && !have_dict) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: dictionary indices without a dictionary page in column " + leaf.name);
            const uint32_t bw = present ? *q : 0;
            if (bw > 32) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: dictionary index width");
            if (present) {  // the run headers stay unread here: pq_hybrid expands the stream into the tail, one wave per page
              const uint32_t ow = bw <= 8 ? 1u : bw <= 16 ? 2u : 4u;
              const uint64_t out_at = tail_take(present * ow);
              hpages.push_back(HPage{(uint64_t)(q + 1 - fb), (uint64_t)(pe - fb), out_at, bw, (uint32_t)present, ow, 0});
              val.push_back(Seg{(uint32_t)ord, (uint32_t)present, SG_INDEX, ow, out_at, dict_base, 1});
            }
          } else if (h.enc == E_RLE && leaf.type == T_BOOLEAN) {  // RLE booleans: a length-prefixed hybrid of width 1
            if (q + 4 > pe) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: truncated page");
            uint32_t L; std::memcpy(&L, q, 4);
            if (present) {
              const uint8_t *he = (uint64_t)L < (uint64_t)(pe - (q + 4)) ? q + 4 + L : pe;
              const uint64_t out_at = tail_take(present);
              hpages.push_back(HPage{(uint64_t)(q + 4 - fb), (uint64_t)(he - fb), out_at, 1u, (uint32_t)present, 1u, 0});
              val.push_back(Seg{(uint32_t)ord, (uint32_t)present, SG_INDEX, 1u, out_at, 0, 1});
            }
          } else if (h.enc == E_DELTA_BINARY_PACKED && (leaf.type == T_INT32 || leaf.type == T_INT64)) {
            // the block headers say where every miniblock is; pq_delta decodes the page into the image's tail as PLAIN values
            DStream ds;
            if (!delta_headers(fb, q, pe, minis, ds) || ds.total != present) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: malformed DELTA_BINARY_PACKED page in column " + leaf.name);
            const uint64_t out_at = tail_take(present * P.width);
            dpages.push_back(DPage{ds.first, out_at, 0, 0, (uint32_t)present, (uint32_t)ds.mini0, (uint32_t)ds.mini1, 0, P.width, 0, 0, 0});
            val.push_back(Seg{(uint32_t)ord, (uint32_t)present, SG_PLAIN, 0, out_at, 0, 1});
          } else if (h.enc == E_DELTA_LENGTH_BYTE_ARRAY && leaf.type == T_BYTE_ARRAY) {
            // the lengths are a DELTA_BINARY_PACKED stream, the bytes follow it back to back: pq_delta turns the lengths into (offset, length)
            DStream ds;
            if (!delta_headers(fb, q, pe, minis, ds) || ds.total != present) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: malformed DELTA_LENGTH_BYTE_ARRAY page in column " + leaf.name);
            dpages.push_back(DPage{ds.first, 0, (uint64_t)(ds.end - fb), (uint64_t)(pe - fb), (uint32_t)present, (uint32_t)ds.mini0, (uint32_t)ds.mini1, 1, 4, (uint32_t)ord, (uint32_t)oc, 0});
            val.push_back(Seg{(uint32_t)ord, (uint32_t)present, SG_PLAIN_TEXT, 0, 0, 0, 0});
          } else if (h.enc == E_DELTA_BYTE_ARRAY && (leaf.type == T_BYTE_ARRAY || is_flba)) {
            // every value is a prefix of its predecessor plus a suffix: a chain.  The host expands the page into plain values in the
            // arena (FIXED_LEN_BYTE_ARRAY: back to back; BYTE_ARRAY: length-prefixed) and the page is then a PLAIN one.
            if (!inflate) throw NeedArena{};
            std::vector<DMini> hm; DStream pre, suf;
            std::vector<int64_t> pl_, sl_;
            if (!delta_headers(fb, q, pe, hm, pre) || pre.total != present || !delta_decode_host(fb, hm, pre, pl_)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: malformed DELTA_BYTE_ARRAY prefixes in column " + leaf.name);
            hm.clear();
            if (!delta_headers(fb, pre.end, pe, hm, suf) || suf.total != present || !delta_decode_host(fb, hm, suf, sl_)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: malformed DELTA_BYTE_ARRAY suffixes in column " + leaf.name);
            const size_t src_off = (size_t)(suf.end - fb), src_end = (size_t)(pe - fb);
            std::vector<uint8_t> exp;
            std::string prev, cur;
            size_t sp = src_off;
            for (uint64_t i = 0; i < present; i++) {
              const int64_t a = pl_[(size_t)i], b = sl_[(size_t)i];
              if (a < 0 || b < 0 || (uint64_t)a > prev.size() || (uint64_t)b > src_end - sp) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: a DELTA_BYTE_ARRAY value of column " + leaf.name + " runs past its page");
              cur.assign(prev, 0, (size_t)a);
              cur.append((const char *)fb + sp, (size_t)b); sp += (size_t)b;
              if (is_flba) { if (cur.size() != P.width) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: a DELTA_BYTE_ARRAY value of column " + leaf.name + " has another length than the type"); }
              else { const uint32_t n = (uint32_t)cur.size(); exp.insert(exp.end(), (const uint8_t *)&n, (const uint8_t *)&n + 4); }
              exp.insert(exp.end(), cur.begin(), cur.end());
              if (exp.size() > 0x7FFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: the expanded pages of one object exceed 4 GiB: read it row group by row group");
              prev.swap(cur);
            }
            const size_t at0 = (arena.size() + 15) & ~(size_t)15;
            if ((uint64_t)at0 + exp.size() + 64 >= 0xFFFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: the inflated pages of one object exceed 4 GiB: read it row group by row group");
            arena.resize(at0);  // (fb may move: nothing below reads through it)
            arena.append(exp.data(), exp.data() + exp.size());
            if (is_flba) val.push_back(Seg{(uint32_t)ord, (uint32_t)present, SG_FIXED_TEXT, P.width, (uint64_t)at0, 0, 0});
            else { val.push_back(Seg{(uint32_t)ord, (uint32_t)present, SG_PLAIN_TEXT, 0, 0, 0, 0}); tpages.push_back(TextPage{(uint64_t)at0, (uint64_t)(at0 + exp.size()), (uint32_t)ord, (uint32_t)present}); }
          } else return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: column " + leaf.name + ": value encoding " + std::to_string(h.enc) + " (BYTE_STREAM_SPLIT over FIXED_LEN_BYTE_ARRAY, or a DELTA_* encoding over this type) is read by the stock reader");
          ord += present; seen += h.nvalues;
          if (ord > 0x7FFFFFF0ull) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: more values than rows in column " + leaf.name);
        }
        (void)page_at0;
        pos = next;
      }
      if (seen != c.num_values || c.num_values != g.num_rows) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: column " + leaf.name + ": pages do not add up to the row group's rows");
      row0 += (uint64_t)g.num_rows;
    }
    PQD("  %zu level segs, %zu value segs, %zu text pages, ord %llu\n", lev.size(), val.size(), tpages.size(), (unsigned long long)ord);
  }
  clk.at("page walk done (host)");
  // ---- the device image: the object (or the inflated pages), 64 bytes of slack, then the tail ----
  const uint64_t img_len = inflate ? arena.size() : len;
  if (inflate) {
    tail_base = (img_len + 64 + 15) & ~15ull;
    if (tail_base + tail_need + 64 >= 0xFFFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: the object and what is decoded beside it exceed 4 GiB: read it row group by row group");
    file = dalloc((size_t)(tail_base + tail_need) + 64);
    // (the first dev_region bytes are the device-inflated pages': not uploaded)
    if (img_len > dev_region) h2d((char *)file->p + dev_region, arena.data() + dev_region, (size_t)(img_len - dev_region));
    TF_HIP(hipMemsetAsync((char *)file->p + img_len, 0, (size_t)(tail_base - img_len), st));
  } else if (tail_need > tail_cap) {
    // a chunk whose pages use an encoding its metadata does not list: the footer's bound does not hold — through the arena, sized by the walk
    throw NeedArena{};
  }
  const uint8_t *dfile = ptr<uint8_t>(file);
  uint8_t *dtail = ptr<uint8_t>(file) + tail_base;
  Buf derr = dalloc_zero(16);  // [0]: PQE_* raised by a kernel
  if (!infpages.empty()) {
    clk.at("device pages: walk done");
    Buf bip = upload_small(infpages.data(), infpages.size() * sizeof(InfPage));
    keep.push_back(bip); keep.push_back(cfile);
    up.wait(0);   // the compressed object is in HBM
    // the ring: 64 KiB holds every source a snappy / LZ4 copy can name with a 16-bit offset (two waves a CU); a smaller ring lets more pages run side
    // by side and sends the copies that reach further back to the image (TFGPU_PQ_RING_KB = 16 | 32 | 64, measurement: DESIGN 10)
    const int ring_kb = [] { const char *e = std::getenv("TFGPU_PQ_RING_KB"); const int v = e ? std::atoi(e) : 64; return v == 16 || v == 32 ? v : 64; }();
    KernelTimer t("pq_inflate");
    const InfPage *dp = reinterpret_cast<const InfPage *>(bip->p);
    if (ring_kb == 16) pq_inflate<16384u><<<(unsigned)infpages.size(), 64, 16384, st>>>(ptr<uint8_t>(cfile), ptr<uint8_t>(file), dp, (int32_t)infpages.size(), ptr<uint32_t>(derr));
    else if (ring_kb == 32) pq_inflate<32768u><<<(unsigned)infpages.size(), 64, 32768, st>>>(ptr<uint8_t>(cfile), ptr<uint8_t>(file), dp, (int32_t)infpages.size(), ptr<uint32_t>(derr));
    else pq_inflate<65536u><<<(unsigned)infpages.size(), 64, 65536, st>>>(ptr<uint8_t>(cfile), ptr<uint8_t>(file), dp, (int32_t)infpages.size(), ptr<uint32_t>(derr));
  }
  // ---- (offset, length) of every PLAIN / DELTA_LENGTH byte-array value: the length-prefix walks of the PLAIN pages in one launch (the
  //      pages are independent; a page is a serial chain), the DELTA pages by prefix sums ----
  Buf arena_off, arena_len;
  {
    size_t total = 0; std::vector<TextPage> all;
    for (size_t oc = 0; oc < plans.size(); oc++) {
      ColPlan &P = plans[oc];
      if (!P.is_text || P.nil) continue;
      P.arena_at = total;
      for (auto tp : P.tpages) { tp.ord += (uint32_t)total; all.push_back(tp); }
      for (auto &dp : dpages) if (dp.mode == 1 && dp.pad0 == (uint32_t)oc) dp.ord += (uint32_t)total;
      total += (size_t)P.ord;
    }
    arena_off = dalloc(std::max<size_t>(total, 1) * 4); arena_len = dalloc(std::max<size_t>(total, 1) * 4);
    // (an object that is going up in pieces: the pages of each kind in runs by the piece their last byte arrives with, a run queued
    //  behind its piece's event; otherwise one run each)
    const auto truns = by_piece(all, up, [](const TextPage &q) { return q.end; });
    const auto hruns = by_piece(hpages, up, [](const HPage &q) { return q.end; });
    const auto druns = by_piece(dpages, up, [](const DPage &q) { return q.mode == 1 ? q.data_end : ~0ull; });   // (a page of values reads its miniblocks wherever they lie: with the last piece)
    Buf btp, bh, bm, bp;
    if (!all.empty() && nrows) { btp = upload_small(all.data(), all.size() * sizeof(TextPage)); keep.push_back(btp); }
    if (!hpages.empty() && nrows) { bh = upload_small(hpages.data(), hpages.size() * sizeof(HPage)); keep.push_back(bh); }
    if (!dpages.empty() && nrows) {
      DMini none{};
      bm = upload_small(minis.empty() ? &none : minis.data(), std::max<size_t>(minis.size(), 1) * sizeof(DMini)); bp = upload_small(dpages.data(), dpages.size() * sizeof(DPage));
      keep.push_back(bm); keep.push_back(bp);
    }
    // TFGPU_PQ_WALK_SPEC=1: the windows by all 64 lanes of wave 0 (guessed slice entries, settled against the neighbours' exits).  OFF by default:
    // measured on the MI355X (gpurun r11p-r11t, the 556 MB hits object, 721 pages of <= 69 windows) 88 % of the windows settle in 0.6 rounds
    // and the kernel takes 3.77 ms against the one-lane walk's 3.73-4.02 — a lane's slice is walked three times (guess proofs, count, write)
    // in lockstep with the slowest of 64 slices, every link two LDS reads behind an index computation; without any walk (=2, profiling
    // only) the kernel is 0.075 ms: staging, barriers and the flush are not what it waits for
    static const int walk_spec = [] { const char *e = std::getenv("TFGPU_PQ_WALK_SPEC"); return e ? std::atoi(e) : 0; }();   // A/B runs: 0 = the one-lane walk for every window
    static const bool walk_debug = [] { const char *e = std::getenv("TFGPU_PQ_DEBUG"); return e && e[0] == '1'; }();
    Buf wdbg = walk_debug ? dalloc_zero(32) : nullptr;
    const size_t nruns = std::max<size_t>(up.end.size(), 1);
    for (size_t g = 0; g < nruns; g++) {
      up.wait((int)g);
      if (btp && truns[g].second > truns[g].first) {
        KernelTimer t("pq_walk_text");
        pq_walk_text<<<(unsigned)(truns[g].second - truns[g].first), 256, 0, st>>>(dfile, (img_len + 64) & ~15ull, reinterpret_cast<const TextPage *>(btp->p) + truns[g].first, ptr<uint32_t>(arena_off), ptr<uint32_t>(arena_len), ptr<uint32_t>(derr), walk_spec, wdbg ? ptr<uint32_t>(wdbg) : nullptr);
      }
      if (bh && hruns[g].second > hruns[g].first) {
        const size_t n = hruns[g].second - hruns[g].first;
        KernelTimer t("pq_hybrid");
        pq_hybrid<<<(unsigned)((n * 64 + 255) / 256), 256, 0, st>>>(dfile, dtail, reinterpret_cast<const HPage *>(bh->p) + hruns[g].first, (int32_t)n, ptr<uint32_t>(derr));
      }
      if (bp && druns[g].second > druns[g].first) {
        const size_t n = druns[g].second - druns[g].first;
        KernelTimer t("pq_delta");
        pq_delta<<<(unsigned)((n * 64 + 255) / 256), 256, 0, st>>>(dfile, dtail, reinterpret_cast<const DMini *>(bm->p), reinterpret_cast<const DPage *>(bp->p) + druns[g].first, (int32_t)n, ptr<uint32_t>(arena_off), ptr<uint32_t>(arena_len), ptr<uint32_t>(derr));
      }
    }
    if (wdbg) {
      uint32_t h[8]; d2h(h, wdbg->p, 32); tf::sync();
      std::fprintf(stderr, "tfgpu pq walk: %u windows, %u by the wave, %u did not settle, %u held a bad prefix, %u other; %.2f rounds a window; %u pages, the longest %u windows\n", h[0], h[1], h[2], h[3], h[4], h[0] ? (double)h[5] / h[0] : 0.0, h[7], h[6]);
    }
  }
  clk.at("text walks enqueued");
  // ---- pass 2 (device): levels → validity → ordinals, then the values — one launch per step over all the columns of a kind ----
  std::vector<size_t> totals;  // the text columns, in order: their scanned totals are read back together
  std::vector<std::pair<size_t, Buf>> soffs;
  Buf tot64_all;               // the text columns' byte totals in 64 bits
  // every small table the kernels read (segment tables, dictionary tables, the descriptors themselves) travels in ONE copy
  std::vector<uint8_t> blob;
  auto blob_put = [&](const void *src, size_t n) -> size_t { const size_t at = (blob.size() + 15) & ~(size_t)15; blob.resize(at + n); if (n) std::memcpy(blob.data() + at, src, n); return at; };
  struct Pending2 { size_t oc; size_t val_at, lev_at, dict_at_at, dict_n_at, dict_off_at, dict_len_at; };
  std::vector<Pending2> fixed_cols, text_cols, lev_cols, int96_cols;
  for (size_t oc = 0; oc < plans.size(); oc++) {
    ColPlan &P = plans[oc];
    DColumn &d = P.d;
    if (want[oc] == W_ROW_INDEX) {  // constructCI: vals[i] = idx, the 1-based row of the object (uint64)
      d.repr = TFGPU_R_UINT64; d.dtype = outcols[oc].second >= 0 ? outcols[oc].second : TFGPU_T_UINT64;
      d.values = dalloc((size_t)std::max<int64_t>(nrows, 1) * 8);
      if (nrows) pq_row_index<<<(unsigned)((nrows + 255) / 256), 256, 0, st>>>(ptr<uint64_t>(d.values), nrows);
      continue;
    }
    if (want[oc] == W_FILE_NAME) {  // vals[i] = fileName in every row
      const size_t fl = std::strlen(file_name);
      if ((uint64_t)fl * (uint64_t)nrows >= 0xFFFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: __file_name holds 4 GiB of text or more: read the object row group by row group");
      d.repr = TFGPU_R_STRING; d.dtype = outcols[oc].second >= 0 ? outcols[oc].second : TFGPU_T_UTF8;
      d.offsets = dalloc((size_t)(nrows + 1) * 4 + 16); d.data_len = (uint64_t)fl * (uint64_t)nrows; d.data = dalloc((size_t)d.data_len + 16);
      Buf nm = upload_small(file_name, fl + 1);
      keep.push_back(nm);
      pq_file_name<<<(unsigned)((nrows + 1 + 255) / 256), 256, 0, st>>>(ptr<uint8_t>(nm), (uint32_t)fl, nrows, ptr<uint32_t>(d.offsets), ptr<uint8_t>(d.data));
      continue;
    }
    if (P.nil) continue;
    if (nrows == 0) {
      if (P.is_text || P.is_int96) { d.offsets = dalloc_zero(8); d.data = dalloc(8); } else { d.values = dalloc(8); if (d.repr == TFGPU_R_TIME) d.nanos = dalloc(8); }
      continue;
    }
    Pending2 q{oc, 0, 0, 0, 0, 0, 0};
    q.val_at = blob_put(P.val.data(), P.val.size() * sizeof(Seg));
    // The host counted the present values while it read the level runs' headers: a column whose every row holds a value (required, or
    // optional without a nil — the usual case) needs no levels, no ordinals and no validity: the ordinal of a row's value is the row.
    if ((int64_t)P.ord != nrows) { q.lev_at = blob_put(P.lev.data(), P.lev.size() * sizeof(Seg)); lev_cols.push_back(q); }
    if (P.is_text) { q.dict_off_at = blob_put(P.dict_off.data(), P.dict_off.size() * 4); q.dict_len_at = blob_put(P.dict_len.data(), P.dict_len.size() * 4); }
    else q.dict_at_at = blob_put(P.dict_at.data(), P.dict_at.size() * 8);
    q.dict_n_at = blob_put(P.dict_n.data(), P.dict_n.size() * 4);
    (P.is_text ? text_cols : P.is_int96 ? int96_cols : fixed_cols).push_back(q);
  }
  if (nrows && !(fixed_cols.empty() && text_cols.empty() && int96_cols.empty())) {
    // ranks of the columns that hold nils: flags → one segmented scan; a column's rank array is a slice of rank_all
    const int64_t rstride = ((nrows + 2 + 3) / 4) * 4;
    Buf rank_all = lev_cols.empty() ? nullptr : dalloc((size_t)lev_cols.size() * (size_t)rstride * 4 + 16);
    std::map<size_t, uint32_t *> rank_of;
    for (size_t k = 0; k < lev_cols.size(); k++) rank_of[lev_cols[k].oc] = ptr<uint32_t>(rank_all) + (int64_t)k * rstride;
    const size_t nt = text_cols.size() + int96_cols.size();
    const int64_t lstride = ((nrows + 1 + 3) / 4) * 4;
    Buf lens_all = nt ? dalloc(nt * (size_t)lstride * 4 + 16) : nullptr;
    tot64_all = dalloc_zero((nt + 1) * 8);
    // descriptors (device pointers into the blob, whose address is known before it is filled)
    std::vector<LevDesc> ld; std::vector<FixedDesc> fd; std::vector<TextDesc> td;
    const size_t blob_cap = blob.size() + 64 + (lev_cols.size() * sizeof(LevDesc) + fixed_cols.size() * sizeof(FixedDesc) + text_cols.size() * sizeof(TextDesc)) + 64;
    Buf bblob = dalloc(blob_cap);
    const uint8_t *B = ptr<uint8_t>(bblob);
    keep.push_back(bblob); keep.push_back(rank_all);
    for (auto &q : lev_cols) {
      ColPlan &P = plans[q.oc];
      if (P.optional) P.d.validity = dalloc((size_t)(nrows + 7) / 8 + 8);
      ld.push_back(LevDesc{reinterpret_cast<const Seg *>(B + q.lev_at), (int32_t)P.lev.size(), 0, rank_of[q.oc], ptr<uint8_t>(P.d.validity)});
    }
    for (auto &q : fixed_cols) {
      ColPlan &P = plans[q.oc];
      DColumn &d = P.d;
      d.values = dalloc((size_t)nrows * (size_t)P.out_width);
      if (d.repr == TFGPU_R_TIME) d.nanos = P.conv == CV_TS_MICROS ? dalloc((size_t)nrows * 4) : dalloc_zero((size_t)nrows * 4);
      fd.push_back(FixedDesc{reinterpret_cast<const Seg *>(B + q.val_at), (int32_t)P.val.size(), 0, rank_of.count(q.oc) ? rank_of[q.oc] : nullptr,
                             P.dict_at.empty() ? nullptr : reinterpret_cast<const uint64_t *>(B + q.dict_at_at), P.dict_n.empty() ? nullptr : reinterpret_cast<const uint32_t *>(B + q.dict_n_at),
                             FixedOut{d.values->p, P.conv == CV_TS_MICROS ? ptr<int32_t>(d.nanos) : nullptr, (int32_t)P.width, P.out_width, P.conv}});
    }
    size_t ti = 0;
    for (auto &q : text_cols) {
      ColPlan &P = plans[q.oc];
      DColumn &d = P.d;
      Buf soff = dalloc((size_t)nrows * 4);
      d.offsets = subbuf(lens_all, ti * (size_t)lstride * 4, (size_t)(nrows + 1) * 4);
      td.push_back(TextDesc{reinterpret_cast<const Seg *>(B + q.val_at), (int32_t)P.val.size(), 0, rank_of.count(q.oc) ? rank_of[q.oc] : nullptr,
                            ptr<uint32_t>(arena_off) + P.arena_at, ptr<uint32_t>(arena_len) + P.arena_at,
                            P.dict_off.empty() ? nullptr : reinterpret_cast<const uint32_t *>(B + q.dict_off_at), P.dict_len.empty() ? nullptr : reinterpret_cast<const uint32_t *>(B + q.dict_len_at),
                            P.dict_n.empty() ? nullptr : reinterpret_cast<const uint32_t *>(B + q.dict_n_at), ptr<uint32_t>(soff), ptr<uint32_t>(d.offsets),
                            reinterpret_cast<unsigned long long *>(tot64_all->p) + ti});
      totals.push_back(q.oc); soffs.push_back({q.oc, soff}); keep.push_back(soff);
      ti++;
    }
    const size_t ld_at = blob_put(ld.data(), ld.size() * sizeof(LevDesc)), fd_at = blob_put(fd.data(), fd.size() * sizeof(FixedDesc)), td_at = blob_put(td.data(), td.size() * sizeof(TextDesc));
    if (blob.size() > blob_cap) return tf::fail(TFGPU_ERR_DEVICE, "tfgpu_parquet_read: internal: table blob outgrew its buffer");
    h2d_small(bblob->p, blob.data(), blob.size());
    const unsigned gx = (unsigned)((nrows + 255) / 256);
    if (!ld.empty()) {
      pq_levels_all<<<dim3(gx, (unsigned)ld.size()), 256, 0, st>>>(dfile, reinterpret_cast<const LevDesc *>(B + ld_at), nrows);
      exclusive_scan_u32_segments(ptr<uint32_t>(rank_all), nrows, (int)ld.size(), rstride);
      pq_pack_validity_all<<<dim3((unsigned)(((nrows + 7) / 8 + 255) / 256), (unsigned)ld.size()), 256, 0, st>>>(reinterpret_cast<const LevDesc *>(B + ld_at), nrows);
    }
    if (!fd.empty()) { KernelTimer t("pq_values"); pq_values_all<<<dim3(gx, (unsigned)fd.size()), 256, 0, st>>>(dfile, tail_base, reinterpret_cast<const FixedDesc *>(B + fd_at), nrows, ptr<uint32_t>(derr)); }
    if (!td.empty()) {
      KernelTimer t("pq_text");
      pq_text_cells_all<<<dim3((unsigned)((nrows + 256 * PQ_TEXT_RPT - 1) / (256 * PQ_TEXT_RPT)), (unsigned)td.size()), 256, 0, st>>>(dfile, tail_base, reinterpret_cast<const TextDesc *>(B + td_at), nrows, ptr<uint32_t>(derr));
      sum_u32_segments_u64(ptr<uint32_t>(lens_all), nrows, (int)td.size(), lstride, reinterpret_cast<unsigned long long *>(tot64_all->p));   // (the text columns are the first segments)
    }
    for (auto &q : int96_cols) {  // (rare: one launch per column)
      ColPlan &P = plans[q.oc];
      DColumn &d = P.d;
      Buf soff = dalloc((size_t)nrows * 4);
      d.offsets = subbuf(lens_all, ti * (size_t)lstride * 4, (size_t)(nrows + 1) * 4);
      KernelTimer t("pq_text");
      pq_int96_text<<<gx, 256, 0, st>>>(dfile, dtail, tail_base, P.slot0, reinterpret_cast<const Seg *>(B + q.val_at), (int32_t)P.val.size(), rank_of.count(q.oc) ? rank_of[q.oc] : nullptr, nrows,
                                        P.dict_at.empty() ? nullptr : reinterpret_cast<const uint64_t *>(B + q.dict_at_at), P.dict_n.empty() ? nullptr : reinterpret_cast<const uint32_t *>(B + q.dict_n_at),
                                        ptr<uint32_t>(soff), ptr<uint32_t>(d.offsets), ptr<uint32_t>(derr));
      // (29 digits and a sign per row at most: the column cannot reach 4 GiB under the 2^31-row bound, its 64-bit total stays 0)
      totals.push_back(q.oc); soffs.push_back({q.oc, soff}); keep.push_back(soff);
      ti++;
    }
    if (nt) exclusive_scan_u32_segments(ptr<uint32_t>(lens_all), nrows, (int)nt, lstride);
  }
  // the text columns' sizes, their 64-bit sums and the error word: one gather launch, three read-backs into page-locked memory (a four-byte
  // copy per column into a pageable vector was 22 us each)
  const uint32_t *htot = nullptr, *htot64w = nullptr;
  if (!totals.empty()) {
    std::vector<const uint32_t *> at(totals.size());
    for (size_t k = 0; k < totals.size(); k++) at[k] = ptr<uint32_t>(plans[totals[k]].d.offsets) + nrows;
    Buf bat = upload_small(at.data(), at.size() * sizeof(at[0])), gathered = dalloc(totals.size() * 4);
    pq_gather_words<<<1, 256, 0, st>>>(reinterpret_cast<const uint32_t *const *>(bat->p), (int)totals.size(), ptr<uint32_t>(gathered));
    keep.push_back(bat); keep.push_back(gathered);
    htot = d2h_u32(gathered->p, totals.size());
    htot64w = d2h_u32(tot64_all->p, totals.size() * 2);
  }
  const uint32_t *herr = d2h_u32(derr->p, 4);
  clk.at("columns enqueued");
  tf::sync();  // ONE wait for the text columns' sizes (a dictionary-coded column can be far longer than its chunk), then the copies
  clk.at("first sync");
  if (herr[0]) return tf::fail(TFGPU_ERR_INVALID, herr[0] == PQE_DICT_INDEX ? "tfgpu_parquet_read: a dictionary index past its dictionary" : herr[0] == PQE_INFLATE ? "tfgpu_parquet_read: malformed SNAPPY / LZ4_RAW page (an element that runs past its page, or a copy from in front of it)" : herr[0] == PQE_HYBRID ? "tfgpu_parquet_read: malformed dictionary indices (an RLE / bit-packed run that ends outside its page)" : "tfgpu_parquet_read: a byte-array length that runs past its page");
  for (size_t k = 0; k < totals.size(); k++) if (((uint64_t)htot64w[2 * k] | (uint64_t)htot64w[2 * k + 1] << 32) >= 0xFFFFFFF0ull) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: column " + plans[totals[k]].d.name + " holds 4 GiB of text or more: read the object row group by row group");
  if (!totals.empty()) {
    std::vector<PackDesc> pd(totals.size());
    for (size_t k = 0; k < totals.size(); k++) {
      DColumn &d = plans[totals[k]].d;
      d.data_len = htot[k];
      d.data = dalloc((size_t)d.data_len + 16);
      pd[k] = PackDesc{ptr<uint32_t>(d.offsets), ptr<uint8_t>(d.data), ptr<uint32_t>(soffs[k].second)};
    }
    Buf bpd = upload_small(pd.data(), pd.size() * sizeof(PackDesc));
    constexpr int RPT = 4;
    KernelTimer t("pq_pack_text");
    pq_pack_text_all<RPT><<<dim3((unsigned)((nrows + 256 * RPT - 1) / (256 * RPT)), (unsigned)totals.size()), 256, 0, st>>>(reinterpret_cast<const PackDesc *>(bpd->p), nrows, dfile);
    keep.push_back(bpd);
  }
  herr = d2h_u32(derr->p, 4);
  tf::sync();
  if (herr[0]) return tf::fail(TFGPU_ERR_INVALID, herr[0] == PQE_DICT_INDEX ? "tfgpu_parquet_read: a dictionary index past its dictionary" : herr[0] == PQE_INFLATE ? "tfgpu_parquet_read: malformed SNAPPY / LZ4_RAW page (an element that runs past its page, or a copy from in front of it)" : herr[0] == PQE_HYBRID ? "tfgpu_parquet_read: malformed dictionary indices (an RLE / bit-packed run that ends outside its page)" : "tfgpu_parquet_read: a byte-array length that runs past its page");
  clk.at("second sync");
  for (auto &P : plans) db->cols.push_back(std::move(P.d));
  *out = db.release();
  return TFGPU_OK;
}

extern "C" int tfgpu_parquet_read_object(const void *bytes, uint64_t len, int mem, const tfgpu_schema *schema, const char *table_ns, const char *table_name, const char *file_name, tfgpu_dbatch **out) {
  TF_API_BEGIN
  if (!bytes || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read: null argument");
  if (mem != TFGPU_MEM_HOST) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: the footer and the page headers are walked on the host: pass the object in host memory (pinned for an asynchronous upload)");
  if (len >> 32) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: objects of 4 GiB and more: read them row group by row group");
  const uint8_t *f = static_cast<const uint8_t *>(bytes);
  try { return parquet_read_impl(f, len, schema, table_ns, table_name, file_name, false, out); }
  catch (const NeedArena &) { return parquet_read_impl(f, len, schema, table_ns, table_name, file_name, true, out); }
  TF_API_END
}
// The object brought into HBM by the caller (its own lane's copy engine, ahead of time): `bytes` stays the host copy the footer and the
// page headers are walked in; `staged` holds the same `len` bytes at offset 0 and is at least tfgpu_parquet_staging_size long (the decoded
// tail is written behind the object).  A compressed object's pages are inflated on the host: `staged` is then not used.
extern "C" int tfgpu_parquet_staging_size(const void *bytes, uint64_t len, uint64_t *need) {
  TF_API_BEGIN
  if (!bytes || !need) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_staging_size: null argument");
  FileMeta m; std::string why;
  if (!parse_footer(static_cast<const uint8_t *>(bytes), len, m, why)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_staging_size: " + why);
  std::vector<TopField> fields; int nleaves = 0;
  if (!top_fields(m, fields, nleaves, why)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_staging_size: " + why);
  for (auto &g : m.groups) if ((int)g.cols.size() != nleaves) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_staging_size: a row group does not hold every column");
  int64_t nrows = 0;
  for (auto &g : m.groups) nrows += g.num_rows;
  *need = ((len + 64 + 15) & ~15ull) + tail_bound(m, fields, nrows) + 64;
  return TFGPU_OK;
  TF_API_END
}
extern "C" int tfgpu_parquet_read_staged(const void *bytes, uint64_t len, const tfgpu_dbuf *staged, const tfgpu_schema *schema, const char *table_ns, const char *table_name, const char *file_name, tfgpu_dbatch **out) {
  TF_API_BEGIN
  if (!bytes || !out || !staged || !staged->mem) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read_staged: null argument");
  if (len >> 32) return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_parquet_read: objects of 4 GiB and more: read them row group by row group");
  if (staged->size < len) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_read_staged: the staging buffer is shorter than the object");
  const uint8_t *f = static_cast<const uint8_t *>(bytes);
  try { return parquet_read_impl(f, len, schema, table_ns, table_name, file_name, false, out, staged->mem); }
  catch (const NeedArena &) { return parquet_read_impl(f, len, schema, table_ns, table_name, file_name, true, out); }
  TF_API_END
}
extern "C" int tfgpu_parquet_read(const void *bytes, uint64_t len, int mem, const tfgpu_schema *schema, const char *table_ns, const char *table_name, tfgpu_dbatch **out) {
  return tfgpu_parquet_read_object(bytes, len, mem, schema, table_ns, table_name, nullptr, out);
}

// parquet_schema_resolver.go:81-158 resolveSchema + s3_reader.AppendSystemColsTableSchema (the inferred schema has no key: the two
// system columns become it) — host only, the footer is all it reads.
extern "C" int tfgpu_parquet_resolve_schema(const void *bytes, uint64_t len, int hide_system_cols, tfgpu_schema **out) {
  TF_API_BEGIN
  if (!bytes || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_resolve_schema: null argument");
  FileMeta m; std::string why;
  if (!parse_footer(static_cast<const uint8_t *>(bytes), len, m, why)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_resolve_schema: " + why);
  std::vector<TopField> fields; int nleaves = 0;
  if (!top_fields(m, fields, nleaves, why)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parquet_resolve_schema: " + why);
  const int nsys = hide_system_cols ? 0 : 2;
  auto *s = (tfgpu_schema *)std::calloc(1, sizeof(tfgpu_schema));
  if (!s) throw std::bad_alloc();
  s->ncols = (int32_t)fields.size() + nsys;
  s->cols = (tfgpu_colschema *)std::calloc((size_t)std::max(s->ncols, 1), sizeof(tfgpu_colschema));
  if (!s->cols) { std::free(s); throw std::bad_alloc(); }
  if (nsys) {
    s->cols[0].name = strdup("__file_name"); s->cols[0].dtype = TFGPU_T_UTF8; s->cols[0].flags = TFGPU_COL_KEY;
    s->cols[1].name = strdup("__row_index"); s->cols[1].dtype = TFGPU_T_UINT64; s->cols[1].flags = TFGPU_COL_KEY;
  }
  for (size_t k = 0; k < fields.size(); k++) {
    tfgpu_colschema &c = s->cols[(size_t)nsys + k];
    c.name = strdup(fields[k].el.name.c_str());
    c.dtype = fields[k].dtype;
    c.original_type = strdup(("parquet:" + fields[k].type_string).c_str());
  }
  *out = s;
  return TFGPU_OK;
  TF_API_END
}
