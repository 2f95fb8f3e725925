// tf_dbzrecv.cpp — the host halves of the two schema-carrying Kafka parsers behind the C ABI.  Confluent-SR JSON schemas: at the
// end of the file (tfgpu_sr_compile_schema).  The Debezium receiver: what the reference does ONCE PER SCHEMA before any
// value is touched, and the loop that hands each group of messages to the device.
//
//   Receiver.receiveSchema / receiveTableSchema    pkg/debezium/receiver.go:60-96, 45-59
//   receiveFieldColSchema                          pkg/debezium/receiver_engine.go:108-146
//   TypeToDefault, Point / VariableScaleDecimal /
//   Decimal matchers                               pkg/debezium/common/field_receiver_default.go:14-31, 258-355
//   Schema (encoding/json struct binding)          pkg/debezium/common/debezium_schema.go:12-29, 84-101
//   DebeziumImpl.DoBatch                           pkg/parsers/registry/debezium/engine/parser.go:120-130
//
// Scope: NewDebeziumImpl(logger, nil, threads) — no schema registry, no original-type table.  The reference caches the compiled
// schema by a hash of its bytes (receiver.go:61-66); so does tfgpu_dbz_receiver, keyed by the device's hash of the same bytes,
// and it keeps the head of the batch's opening message for tfgpu_debezium_unpack_cached.  Per message everything is
// tfgpu_debezium_unpack / tfgpu_debezium_parse (tf_debezium.hip); nothing here touches a value.
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "tf_common.hpp"
#include "tf_plan.hpp"

namespace tf {
namespace dbzrecv {

struct SchemaError { std::string why; };  // receiveSchema fails: every message of the schema becomes an `_unparsed` item
struct HostOnly { std::string why; };     // the stock code must handle this schema

// strings.EqualFold against an ASCII field name: ASCII case folding plus the two runes Unicode folds onto ASCII letters
// (U+017F LATIN SMALL LETTER LONG S → s, U+212A KELVIN SIGN → k)
static bool equal_fold_ascii(const std::string &key, const char *name) {
  size_t i = 0, j = 0;
  const size_t n = key.size(), m = std::strlen(name);
  while (i < n && j < m) {
    unsigned char c = (unsigned char)key[i];
    char folded;
    if (c < 0x80) { folded = (char)std::tolower(c); i++; }
    else if (c == 0xC5 && i + 1 < n && (unsigned char)key[i + 1] == 0xBF) { folded = 's'; i += 2; }
    else if (c == 0xE2 && i + 2 < n && (unsigned char)key[i + 1] == 0x84 && (unsigned char)key[i + 2] == 0xAA) { folded = 'k'; i += 3; }
    else return false;
    if (folded != (char)std::tolower((unsigned char)name[j])) return false;
    j++;
  }
  return i == n && j == m;
}
// encoding/json's struct-field binding: a key names the field exactly or under case folding; the LAST such key wins.  A key
// that matches only by folding is left to the host (the device binds exact keys only).
static const Json *member(const Json &obj, const char *name) {
  const Json *out = nullptr;
  for (auto &kv : obj.obj) {
    if (kv.first == name) out = &kv.second;
    else if (equal_fold_ascii(kv.first, name)) throw HostOnly{"key " + kv.first + " binds " + name + " by case folding"};
  }
  return out;
}
static std::string string_field(const Json &obj, const char *name) {
  const Json *v = member(obj, name);
  if (!v || v->type == Json::Null) return "";
  if (v->type != Json::Str) throw SchemaError{std::string("json: cannot unmarshal into Go struct field Schema.") + name + " of type string"};
  return v->str;
}
static bool go_int64_literal(const std::string &t) {  // what json.Unmarshal takes for an int field: an integer literal in range
  size_t i = 0;
  if (i < t.size() && t[i] == '-') i++;
  if (i >= t.size()) return false;
  for (size_t k = i; k < t.size(); k++) if (t[k] < '0' || t[k] > '9') return false;
  const std::string digits = t.substr(i);
  if (digits.size() > 19) return false;
  if (digits.size() == 19) { const std::string lim = t[0] == '-' ? "9223372036854775808" : "9223372036854775807"; if (digits > lim) return false; }
  return true;
}

// debezium_schema.go:12-22, as json.Unmarshal fills it
struct Schema {
  std::string field, name, type, scale;
  bool optional = false, has_parameters = false, has_dt_info = false;
  std::vector<Schema> fields;
  const Schema *find(const std::string &f) const { for (auto &s : fields) if (s.field == f) return &s; return nullptr; }  // FindSchemaDescr: the first one
};
static Schema unmarshal(const Json *node, int depth) {
  Schema s;
  if (!node || node->type == Json::Null) return s;
  if (depth > 64) throw HostOnly{"schema nested deeper than 64 levels (encoding/json reads up to 10000: the stock code decides)"};
  if (node->type != Json::Obj) throw SchemaError{"json: cannot unmarshal into Go value of type common.Schema"};
  s.field = string_field(*node, "field"); s.name = string_field(*node, "name"); s.type = string_field(*node, "type");
  if (const Json *v = member(*node, "optional"); v && v->type != Json::Null) { if (v->type != Json::Bool) throw SchemaError{"Schema.optional"}; s.optional = v->b; }
  if (const Json *v = member(*node, "version"); v && v->type != Json::Null) { if (v->type != Json::Num || !go_int64_literal(v->str)) throw SchemaError{"Schema.version"}; }
  if (const Json *v = member(*node, "parameters"); v && v->type != Json::Null) {
    if (v->type != Json::Obj) throw SchemaError{"Schema.parameters"};
    s.has_parameters = true;
    for (const char *k : {"length", "connect.decimal.precision", "allowed"}) string_field(*v, k);
    s.scale = string_field(*v, "scale");
  }
  if (const Json *v = member(*node, "items"); v && v->type != Json::Null) { if (v->type != Json::Obj) throw SchemaError{"Schema.items"}; unmarshal(v, depth + 1); }
  if (const Json *v = member(*node, "__dt_original_type_info"); v && v->type != Json::Null) s.has_dt_info = true;
  if (const Json *v = member(*node, "fields"); v && v->type != Json::Null) {
    if (v->type != Json::Arr) throw SchemaError{"Schema.fields"};
    for (auto &x : v->arr) s.fields.push_back(unmarshal(&x, depth + 1));
  }
  return s;
}
struct Field { std::string name; int32_t op, optional, scale; };
// (DBZ_* op, scale) of one field with an empty original type: receiveFieldColSchema + the default matchers
static void receiver_of(const Schema &f, int32_t &op, int32_t &scale) {
  scale = 0;
  if (f.has_dt_info || f.type == "array") { op = TFGPU_DBZ_HOST; return; }
  static const std::map<std::string, int> simple = {{"int8", TFGPU_DBZ_INT8}, {"int16", TFGPU_DBZ_INT16}, {"int32", TFGPU_DBZ_INT32}, {"int64", TFGPU_DBZ_INT64}, {"boolean", TFGPU_DBZ_BOOLEAN},
                                                    {"string", TFGPU_DBZ_STRING}, {"float", TFGPU_DBZ_FLOAT64}, {"double", TFGPU_DBZ_FLOAT64}};
  auto it = simple.find(f.type);
  if (it != simple.end()) { op = it->second; return; }
  if (f.type == "struct") {
    if (f.name == "io.debezium.data.geometry.Point") { op = TFGPU_DBZ_POINT; return; }
    if (f.name == "io.debezium.data.VariableScaleDecimal") { op = TFGPU_DBZ_VSD; return; }
  }
  if (f.type == "bytes") {
    if (f.name == "org.apache.kafka.connect.data.Decimal") {
      op = TFGPU_DBZ_DECIMAL;
      if (f.has_parameters && !f.scale.empty()) {  // strconv.Atoi; a scale that does not parse fails every non-nil value ("unable to parse scale")
        const std::string &t = f.scale;
        size_t i = (t[0] == '+' || t[0] == '-') ? 1 : 0;
        bool ok = i < t.size() && t.size() - i <= 11;
        for (size_t k = i; k < t.size() && ok; k++) ok = t[k] >= '0' && t[k] <= '9';
        long long v = 0;
        if (ok) { v = std::strtoll(t.c_str(), nullptr, 10); ok = v >= INT32_MIN && v <= INT32_MAX; }
        scale = ok ? (int32_t)v : INT32_MIN;
      }
      return;
    }
    op = TFGPU_DBZ_BYTES; return;
  }
  throw SchemaError{"unable to find field receiver - even default, for kafka type: " + f.type};
}
// receiveSchema for one distinct schema: the fields of the `after` struct; code = TFGPU_ROW_OK, or what the whole schema is
// ---- Schema-registry framed events (NewReceiver with a registry client: receiver.go:221-235) ------------------------------------
// The registry's schema text is a ConfluentJSONSchema; convertSchemaFormat (receiver.go:118-139) unmarshals it, turns it into a
// KafkaJSONSchema (ToKafkaJSONSchema, pkg/schemaregistry/format/json_schema_format.go:120-164) and marshals that for
// UnmarshalSchema.  from_confluent does the three steps in one: the ConfluentJSONSchema as json.Unmarshal binds it (:52-68; every
// field's JSON type checked like the struct's), straight into the Schema UnmarshalSchema would read from the marshalled text.
//   oneOf: the first entry whose type is not "null", made optional (:121-129)
//   properties: sorted by *connect.index (:139-141; sort.Slice — a nil index among two or more properties is the reference's nil
//     dereference, equal indexes leave the order to pdqsort: both host)
//   type: confluentTypeToKafka(type, connect.type) (:70-96)
static std::string confluent_type_to_kafka(const std::string &json_type, const std::string &connect_type) {
  if (json_type == "object") return "struct";
  if (json_type == "string") return connect_type == "bytes" ? "bytes" : "string";
  if (json_type == "boolean") return "boolean";
  if (json_type == "integer") return connect_type;
  if (json_type == "number") return connect_type == "float64" ? "double" : connect_type == "float32" ? "float" : "bytes";
  if (json_type == "array") return "array";
  return "";
}
static std::string cstring_field(const Json &obj, const char *name) {
  const Json *v = member(obj, name);
  if (!v || v->type == Json::Null) return "";
  if (v->type != Json::Str) throw SchemaError{std::string("json: cannot unmarshal into Go struct field ConfluentJSONSchema.") + name + " of type string"};
  return v->str;
}
static Schema from_confluent(const Json *node, int depth) {
  Schema s;  // (a null binds nothing: the zero ConfluentJSONSchema, whose Kafka form is {"type":"","optional":false})
  if (!node || node->type == Json::Null) return s;
  if (depth > 64) throw HostOnly{"schema nested deeper than 64 levels (encoding/json reads up to 10000: the stock code decides)"};
  if (node->type != Json::Obj) throw SchemaError{"json: cannot unmarshal into Go value of type format.ConfluentJSONSchema"};
  // json.Unmarshal binds every field before ToKafkaJSONSchema looks at any: type errors anywhere fail the schema
  const std::string type = cstring_field(*node, "type"), connect_type = cstring_field(*node, "connect.type"), title = cstring_field(*node, "title");
  cstring_field(*node, "description");
  if (const Json *v = member(*node, "connect.version"); v && v->type != Json::Null) { if (v->type != Json::Num || !go_int64_literal(v->str)) throw SchemaError{"ConfluentJSONSchema.connect.version"}; }
  if (const Json *v = member(*node, "additionalProperties"); v && v->type != Json::Null) { if (v->type != Json::Bool) throw SchemaError{"ConfluentJSONSchema.additionalProperties"}; }
  bool has_parameters = false; std::string scale;
  if (const Json *v = member(*node, "connect.parameters"); v && v->type != Json::Null) {
    if (v->type != Json::Obj) throw SchemaError{"ConfluentJSONSchema.connect.parameters"};
    has_parameters = true;
    for (const char *k : {"length", "connect.decimal.precision", "allowed"}) cstring_field(*v, k);
    scale = cstring_field(*v, "scale");
  }
  const Json *dt = member(*node, "__dt_original_type_info");
  const bool has_dt = dt && dt->type != Json::Null;
  Schema items; bool has_items = false;
  if (const Json *v = member(*node, "items"); v && v->type != Json::Null) { if (v->type != Json::Obj) throw SchemaError{"ConfluentJSONSchema.items"}; items = from_confluent(v, depth + 1); has_items = true; }
  std::vector<Schema> one_of; std::vector<std::string> one_of_type;
  if (const Json *v = member(*node, "oneOf"); v && v->type != Json::Null) {
    if (v->type != Json::Arr) throw SchemaError{"ConfluentJSONSchema.oneOf"};
    for (auto &x : v->arr) {
      if (x.type != Json::Null && x.type != Json::Obj) throw SchemaError{"ConfluentJSONSchema.oneOf"};
      one_of_type.push_back(x.type == Json::Obj ? cstring_field(x, "type") : std::string());
      one_of.push_back(from_confluent(&x, depth + 1));
    }
  }
  struct Prop { std::string name; const Json *node; bool has_index; long long index; Schema conv; };
  std::vector<Prop> props;
  if (const Json *v = member(*node, "connect.index"); v && v->type != Json::Null) { if (v->type != Json::Num || !go_int64_literal(v->str)) throw SchemaError{"ConfluentJSONSchema.connect.index"}; }
  if (const Json *v = member(*node, "properties"); v && v->type != Json::Null) {
    if (v->type != Json::Obj) throw SchemaError{"ConfluentJSONSchema.properties"};
    for (auto &kv : v->obj) {  // a Go map: the last occurrence of a name is the entry
      if (kv.second.type != Json::Null && kv.second.type != Json::Obj) throw SchemaError{"ConfluentJSONSchema.properties"};
      Prop pr{kv.first, &kv.second, false, 0, from_confluent(&kv.second, depth + 1)};
      if (kv.second.type == Json::Obj) if (const Json *ix = member(kv.second, "connect.index"); ix && ix->type != Json::Null) { pr.has_index = true; pr.index = std::strtoll(ix->str.c_str(), nullptr, 10); }
      bool replaced = false;
      for (auto &q : props) if (q.name == pr.name) { q = pr; replaced = true; break; }
      if (!replaced) props.push_back(std::move(pr));
    }
  }
  (void)has_items;
  // ---- ToKafkaJSONSchema ----
  for (size_t i = 0; i < one_of.size(); i++) {
    if (one_of_type[i] == "null") continue;
    Schema f = one_of[i];
    f.optional = true;
    return f;
  }
  if (props.size() >= 2) {
    for (auto &q : props) if (!q.has_index) throw HostOnly{"a property without connect.index: the reference dereferences a nil *int while sorting"};
    std::stable_sort(props.begin(), props.end(), [](const Prop &a, const Prop &b) { return a.index < b.index; });
    for (size_t i = 1; i < props.size(); i++) if (props[i].index == props[i - 1].index) throw HostOnly{"two properties share a connect.index: sort.Slice leaves their order open"};
  }
  for (auto &q : props) { Schema f = q.conv; f.field = q.name; s.fields.push_back(std::move(f)); }
  s.type = confluent_type_to_kafka(type, connect_type);
  s.name = title;
  s.has_parameters = has_parameters; s.scale = scale;
  s.has_dt_info = has_dt;
  return s;
}

static int compile(const uint8_t *bytes, size_t len, std::vector<Field> &out, std::string &why, bool confluent = false) {
  out.clear();
  try {
    if (!len) throw SchemaError{"unexpected end of JSON input"};
    Json node;
    try { node = Json::parse(std::string((const char *)bytes, len)); }
    catch (const Error &e) {
      if (std::strstr(e.what(), "nesting deeper")) throw HostOnly{e.what()};  // (encoding/json would still read it: the stock code decides)
      throw SchemaError{e.what()};
    }
    if (node.type != Json::Null && node.type != Json::Obj) throw SchemaError{confluent ? "json: cannot unmarshal into Go value of type format.ConfluentJSONSchema" : "json: cannot unmarshal into Go value of type common.Schema"};
    const Schema top = confluent ? from_confluent(&node, 0) : unmarshal(&node, 0);
    const Schema *before = top.find("before"), *after = top.find("after");
    if (!before || !after) throw HostOnly{"receiveTableSchema(nil): the reference dereferences a nil schema"};
    std::vector<Field> b, a;
    for (auto *which : {before, after}) {
      auto &dst = which == before ? b : a;
      for (auto &f : which->fields) { Field x; x.name = f.field; x.optional = f.optional ? 1 : 0; receiver_of(f, x.op, x.scale); dst.push_back(x); }
    }
    bool same = a.size() == b.size();
    for (size_t i = 0; i < a.size() && same; i++) same = a[i].name == b[i].name && a[i].op == b[i].op && a[i].optional == b[i].optional && a[i].scale == b[i].scale;
    if (!same) throw HostOnly{"before and after structs differ: Delete rows would have another TableSchema than the rest"};
    for (size_t i = 0; i < a.size(); i++) for (size_t j = i + 1; j < a.size(); j++) if (a[i].name == a[j].name) throw HostOnly{"a field name repeats"};
    out = a;
    return TFGPU_ROW_OK;
  } catch (const SchemaError &e) { why = e.why; return TFGPU_ROW_DBZ_SCHEMA; }
  catch (const HostOnly &e) { why = e.why; return TFGPU_ROW_HOST_FALLBACK; }
}

}  // namespace dbzrecv
}  // namespace tf

using namespace tf;
using namespace tf::dbzrecv;

namespace tf { std::unique_ptr<tfgpu_dbatch> compact_rows(const tfgpu_dbatch &in, Buf keep); }  // tf_transform.hip
namespace tf { namespace dbz { void dbz_trust_frames(bool on); void dbz_tentative_frames(bool on); bool dbz_last_parse_was_quick(); void dbz_lazy_frames(bool on, uint64_t h0, uint64_t h1); bool dbz_last_unpack_uniform(); } }

struct tfgpu_dbz_schema {
  int code = 0; std::string why;
  std::vector<Field> fields;
  std::vector<tfgpu_dbz_field> cfields;  // views of `fields` for the C side
  void seal() { cfields.clear(); for (auto &f : fields) cfields.push_back(tfgpu_dbz_field{f.name.c_str(), f.op, f.optional, f.scale, 0}); }
};
// a host array the device copies into and out of: page-locked (hipHostMalloc), so the copy is one DMA at link speed instead of the
// runtime's staged pageable copy (5 MB of frames and 4 MB of row meta per 2^17 messages, both ways, every batch)
template <class T> struct PinVec {
  T *p = nullptr; size_t n = 0;
  PinVec() = default;
  PinVec(const PinVec &) = delete; PinVec &operator=(const PinVec &) = delete;
  PinVec(PinVec &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  PinVec &operator=(PinVec &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
  ~PinVec() { release(); }
  void release() { if (p) tfgpu_host_free(p); p = nullptr; n = 0; }
  void resize(size_t k) {  // grows only; contents are not kept (every user fills it before reading)
    if (k <= n) return;
    release();
    void *q = nullptr;
    if (tfgpu_host_alloc(k * sizeof(T), &q) != TFGPU_OK) throw std::bad_alloc();
    p = (T *)q; n = k;
  }
  void swap(PinVec &o) { std::swap(p, o.p); std::swap(n, o.n); }
  T *data() { return p; } const T *data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T &operator[](size_t i) { return p[i]; } const T &operator[](size_t i) const { return p[i]; }
};

struct tfgpu_dbz_receiver {
  std::map<std::pair<uint64_t, uint64_t>, std::shared_ptr<tfgpu_dbz_schema>> cache;
  std::map<uint32_t, std::shared_ptr<tfgpu_dbz_schema>> registry;  // schema id → the compiled registry schema (tfgpu_dbz_receiver_add_registry_schema)
  // the head of an earlier batch's opening message up to its payload value (tfgpu_debezium_unpack_cached)
  bool last_quick = false;  // the last batch's messages went through dbz_parse_quick
  bool have_known = false; std::string known_bytes; uint32_t known_off = 0, known_len = 0; uint64_t known_hash[2] = {0, 0};
  // the last batch's results
  struct Group { std::shared_ptr<tfgpu_dbz_schema> schema; tfgpu_dbatch *batch = nullptr; PinVec<tfgpu_dbz_row> rows; int64_t nrows = 0; };
  std::vector<Group> groups;
  PinVec<tfgpu_dbz_frame> frames;
  std::vector<uint64_t> ev_starts, ev_offs; std::vector<int64_t> ev_wts;   // registry form: one slot per event, kept across batches
  PinVec<tfgpu_sr_frame> events;   // registry form: tfgpu_sr_frames' list (page-locked: one DMA down, and its device copy serves the next call)
  std::vector<tfgpu_row_error> errs;
  std::vector<PinVec<tfgpu_dbz_row>> spare_rows;
  void drop() { for (auto &g : groups) { if (g.batch) tfgpu_dbatch_free(g.batch); spare_rows.push_back(std::move(g.rows)); } groups.clear(); }
};

extern "C" {

int tfgpu_debezium_compile_schema(const void *schema_bytes, uint64_t len, tfgpu_dbz_schema **out) {
  if (!out || (len && !schema_bytes)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_compile_schema: null argument");
  auto s = std::make_unique<tfgpu_dbz_schema>();
  try { s->code = compile((const uint8_t *)schema_bytes, (size_t)len, s->fields, s->why); }
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
  s->seal();
  *out = s.release();
  return TFGPU_OK;
}
int tfgpu_dbz_schema_info(const tfgpu_dbz_schema *s, int32_t *code, const tfgpu_dbz_field **fields, int32_t *nfields, const char **why) {
  if (!s) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbz_schema_info: null schema");
  if (code) *code = s->code;
  if (fields) *fields = s->cfields.data();
  if (nfields) *nfields = (int32_t)s->cfields.size();
  if (why) *why = s->why.c_str();
  return TFGPU_OK;
}
void tfgpu_dbz_schema_free(tfgpu_dbz_schema *s) { delete s; }

int tfgpu_dbz_receiver_create(tfgpu_dbz_receiver **out) {
  if (!out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbz_receiver_create: null argument");
  *out = new tfgpu_dbz_receiver();
  return TFGPU_OK;
}
void tfgpu_dbz_receiver_destroy(tfgpu_dbz_receiver *r) { if (r) { r->drop(); delete r; } }
// the head the receiver keeps for tfgpu_debezium_unpack_cached (0 = none yet); `known->bytes` stays the receiver's
int tfgpu_dbz_receiver_known(const tfgpu_dbz_receiver *r, tfgpu_dbz_prefix *known) {
  if (!r || !known) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbz_receiver_known: null argument");
  std::memset(known, 0, sizeof *known);
  if (!r->have_known) return TFGPU_OK;
  known->bytes = r->known_bytes.data(); known->len = (uint32_t)r->known_bytes.size(); known->schema_off = r->known_off; known->schema_len = r->known_len;
  known->schema_hash[0] = r->known_hash[0]; known->schema_hash[1] = r->known_hash[1];
  return TFGPU_OK;
}

// One message batch.  host_copy (optional): the same bytes in host memory when `bytes` is a device buffer — a new schema's text
// and the opening message's head are taken from it instead of being copied back from the device.
int tfgpu_dbz_receive(tfgpu_dbz_receiver *r, const void *bytes, uint64_t len, int mem, const void *host_copy, const tfgpu_messages *msgs, int32_t *ngroups, int32_t *msg_codes) {
  if (!r || (len && !bytes) || !ngroups) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbz_receive: null argument");
  try {
    r->drop();
    const int64_t nmsg = msgs ? msgs->nmsg : 1;
    if ((int64_t)r->frames.size() < std::max<int64_t>(nmsg, 1)) r->frames.resize((size_t)std::max<int64_t>(nmsg, 1));  // (filled by the unpack call: no need to clear 48 bytes per message per batch)
    int rc;
    bool uniform = false;
    if (r->have_known) {
      tfgpu_dbz_prefix k{};
      k.bytes = r->known_bytes.data(); k.len = (uint32_t)r->known_bytes.size(); k.schema_off = r->known_off; k.schema_len = r->known_len;
      k.schema_hash[0] = r->known_hash[0]; k.schema_hash[1] = r->known_hash[1];
      // payload spans claimed from the messages' ends instead of walked (tf_debezium.hip, Params::tent) once the tile parser has
      // taken a batch of this receiver: it proves them, or the walker walks them
      struct Tent { bool on; explicit Tent(bool o) : on(o) { if (on) tf::dbz::dbz_tentative_frames(true); } ~Tent() { if (on) tf::dbz::dbz_tentative_frames(false); } } tent(r->last_quick);
      // … and the frames themselves only when they are not all alike (tf_debezium.hip, FrameCache::uniform): the usual batch of the
      // usual topic answers "every message OK, this schema, a payload the tile parser takes" with frame 0 and nothing else
      struct Lazy { Lazy() { tf::dbz::dbz_lazy_frames(true, 0, 0); } ~Lazy() { tf::dbz::dbz_lazy_frames(false, 0, 0); } } lazy;
      rc = tfgpu_debezium_unpack_cached(bytes, len, mem, msgs, &k, r->frames.data());
      uniform = rc == TFGPU_OK && nmsg > 0 && tf::dbz::dbz_last_unpack_uniform();
    } else rc = tfgpu_debezium_unpack(bytes, len, mem, msgs, r->frames.data());
    if (rc) return rc;
    if (msg_codes) {
      if (uniform) std::fill(msg_codes, msg_codes + nmsg, 0);
      else for (int64_t m = 0; m < nmsg; m++) msg_codes[m] = r->frames[(size_t)m].code;
    }
    // the first message of every distinct schema, in order of first appearance
    std::vector<std::pair<std::pair<uint64_t, uint64_t>, int64_t>> firsts;
    if (uniform) firsts.push_back({{r->frames[0].schema_hash[0], r->frames[0].schema_hash[1]}, 0});
    else {
      std::pair<uint64_t, uint64_t> last{0, 0}; bool have_last = false;
      std::map<std::pair<uint64_t, uint64_t>, bool> seen;
      for (int64_t m = 0; m < nmsg; m++) {
        const tfgpu_dbz_frame &f = r->frames[(size_t)m];
        if (f.code) continue;
        const std::pair<uint64_t, uint64_t> key{f.schema_hash[0], f.schema_hash[1]};
        if (have_last && key == last) continue;  // the usual topic: one schema
        last = key; have_last = true;
        if (seen.emplace(key, true).second) firsts.push_back({key, m});
      }
    }
    const uint8_t *host = mem == TFGPU_MEM_HOST ? (const uint8_t *)bytes : (const uint8_t *)host_copy;
    std::vector<tfgpu_row_error> &errs = r->errs;
    if ((int64_t)errs.size() < std::max<int64_t>(nmsg, 1)) errs.resize((size_t)std::max<int64_t>(nmsg, 1));
    for (auto &fm : firsts) {
      const int64_t m = fm.second;
      const tfgpu_dbz_frame &f0 = r->frames[(size_t)m];
      auto it = r->cache.find(fm.first);
      if (it == r->cache.end()) {
        std::string raw((size_t)f0.schema_len, '\0');
        if (f0.schema_len) {
          if (host) std::memcpy(&raw[0], host + f0.schema_start, f0.schema_len);
          else { d2h(&raw[0], (const uint8_t *)bytes + f0.schema_start, f0.schema_len); tf::sync(); }  // a device buffer without a host copy: once per schema per process
        }
        if (!r->have_known && m == 0 && f0.payload_len && f0.schema_start < f0.payload_start) {
          // the opening message's head up to its payload value, for the next batches; usable when the payload is the message's
          // last member (message 0 starts at offset 0)
          const uint64_t ps = f0.payload_start;
          r->known_bytes.resize((size_t)ps);
          if (host) std::memcpy(&r->known_bytes[0], host, (size_t)ps);
          else { d2h(&r->known_bytes[0], bytes, (size_t)ps); tf::sync(); }
          r->known_off = (uint32_t)f0.schema_start; r->known_len = f0.schema_len; r->known_hash[0] = fm.first.first; r->known_hash[1] = fm.first.second;
          r->have_known = true;
        }
        auto s = std::make_shared<tfgpu_dbz_schema>();
        s->code = compile((const uint8_t *)raw.data(), raw.size(), s->fields, s->why);
        s->seal();
        it = r->cache.emplace(fm.first, s).first;
      }
      const std::shared_ptr<tfgpu_dbz_schema> &s = it->second;
      tfgpu_dbz_options o{};
      o.schema_hash[0] = fm.first.first; o.schema_hash[1] = fm.first.second;
      o.nfields = s->code ? 0 : (int32_t)s->cfields.size(); o.fields = s->code ? nullptr : s->cfields.data(); o.schema_code = s->code;
      tfgpu_dbz_receiver::Group g;
      g.schema = s;
      if (!r->spare_rows.empty()) { g.rows.swap(r->spare_rows.back()); r->spare_rows.pop_back(); }  // (buffers of earlier batches: no 32 bytes per message to clear)
      if ((int64_t)g.rows.size() < std::max<int64_t>(nmsg, 1)) g.rows.resize((size_t)std::max<int64_t>(nmsg, 1));
      int64_t ne = 0;
      {
        struct Trust { Trust() { tf::dbz::dbz_trust_frames(true); } ~Trust() { tf::dbz::dbz_trust_frames(false); } } trust;  // r->frames is what the unpack call above wrote, untouched: its device copy serves (tf_debezium.hip)
        rc = tfgpu_debezium_parse(&o, bytes, len, mem, msgs, r->frames.data(), &g.batch, g.rows.data(), nmsg, errs.data(), (int64_t)errs.size(), &ne);
      }
      if (rc) return rc;
      r->last_quick = tf::dbz::dbz_last_parse_was_quick();
      if (msg_codes) for (int64_t i = 0; i < std::min<int64_t>(ne, (int64_t)errs.size()); i++) if (errs[(size_t)i].row >= 0 && errs[(size_t)i].row < nmsg) msg_codes[errs[(size_t)i].row] = errs[(size_t)i].code;
      tfgpu_batch v{};
      if (g.batch && tfgpu_dbatch_view(g.batch, &v) == TFGPU_OK) g.nrows = v.nrows;
      if (s->code || g.nrows == 0) { if (g.batch) tfgpu_dbatch_free(g.batch); r->spare_rows.push_back(std::move(g.rows)); continue; }  // a schema that fails as a whole produces no rows
      r->groups.push_back(std::move(g));
    }
    *ngroups = (int32_t)r->groups.size();
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}
// ---- events framed by a schema registry ------------------------------------------------------------------------------------------
int tfgpu_debezium_compile_registry_schema(const void *schema_text, uint64_t len, tfgpu_dbz_schema **out) {
  if (!out || (len && !schema_text)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_debezium_compile_registry_schema: null argument");
  auto s = std::make_unique<tfgpu_dbz_schema>();
  try { s->code = compile((const uint8_t *)schema_text, (size_t)len, s->fields, s->why, true); }
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
  s->seal();
  *out = s.release();
  return TFGPU_OK;
}
int tfgpu_dbz_receiver_add_registry_schema(tfgpu_dbz_receiver *r, uint32_t schema_id, const void *schema_text, uint64_t len) {
  if (!r || (len && !schema_text)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbz_receiver_add_registry_schema: null argument");
  try {
    auto s = std::make_shared<tfgpu_dbz_schema>();
    s->code = compile((const uint8_t *)schema_text, (size_t)len, s->fields, s->why, true);
    s->seal();
    r->registry[schema_id] = s;
    return TFGPU_OK;
  } catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
}
// DoBatch with a registry: frame (tfgpu_sr_frames), payload spans (tfgpu_debezium_registry_frames), one parse per schema id, then
// DoBuf's rule over the events of each Kafka message.
// TFGPU_DBZ_HOSTTIME=1 (profiling only): wall time between marks of the receiver's host code, to stderr
struct RecvClock {
  bool on; std::chrono::steady_clock::time_point t0;
  RecvClock() : on([] { const char *e = std::getenv("TFGPU_DBZ_HOSTTIME"); return e && e[0] == '1'; }()), t0(std::chrono::steady_clock::now()) {}
  void mark(const char *name) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "tfgpu hosttime receiver: %-28s %.3f ms\n", name, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};
int tfgpu_dbz_receive_registry(tfgpu_dbz_receiver *r, const void *bytes, uint64_t len, int mem, const tfgpu_messages *msgs,
                               tfgpu_sr_frame *events, int64_t events_cap, int64_t *nevents, int32_t *event_codes,
                               uint32_t *missing_ids, int32_t missing_cap, int32_t *nmissing, int32_t *ngroups) {
  if (!r || (len && !bytes) || !ngroups || !nevents || !nmissing || (events_cap && (!events || !event_codes)) || events_cap < 0 || missing_cap < 0 || (missing_cap && !missing_ids))
    return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbz_receive_registry: bad argument");
  try {
    RecvClock rc_clock;
    r->drop();
    rc_clock.mark("drop");
    *ngroups = 0; *nmissing = 0;
    // into the receiver's own page-locked list first (the caller's array is pageable: a staged copy of 32 bytes per event), grown to
    // what the batch needs; the caller's copy is made at the end
    if ((int64_t)r->events.size() < std::max<int64_t>(events_cap, 1)) r->events.resize((size_t)std::max<int64_t>(events_cap, 1));
    tfgpu_sr_frame *const caller_events = events;
    events = r->events.data();
    int rc = tfgpu_sr_frames(bytes, len, mem, msgs, events, events_cap, nevents);
    if (rc) return rc;
    rc_clock.mark("tfgpu_sr_frames");
    const int64_t n = *nevents;
    struct CopyOut { tfgpu_sr_frame *dst; const tfgpu_sr_frame *src; const int64_t &n; ~CopyOut() { if (n > 0) std::memcpy(dst, src, (size_t)n * sizeof(tfgpu_sr_frame)); } } copy_out{caller_events, events, *nevents};
    // schema ids in order of first appearance; the ones nobody registered
    std::vector<uint32_t> ids;
    {
      std::map<uint32_t, bool> seen;
      uint32_t last = 0; bool have_last = false;
      for (int64_t e = 0; e < n; e++) {
        if (events[e].code) continue;
        const uint32_t id = events[e].schema_id;
        if (have_last && id == last) continue;  // the usual topic: one schema
        last = id; have_last = true;
        if (seen.emplace(id, true).second) ids.push_back(id);
      }
    }
    for (uint32_t id : ids) if (!r->registry.count(id)) { if (*nmissing < missing_cap) missing_ids[*nmissing] = id; ++*nmissing; }
    if (*nmissing) return TFGPU_OK;
    if (!n) return TFGPU_OK;
    rc_clock.mark("schema ids");
    // one slot per event: [where it starts, where the next one starts)
    std::vector<uint64_t> &starts = r->ev_starts, &offs = r->ev_offs;   // (kept across batches: 24 bytes per event are not allocated and paged in per call)
    std::vector<int64_t> &wts = r->ev_wts;
    if ((int64_t)starts.size() < n + 1) starts.resize((size_t)n + 1);
    if ((int64_t)offs.size() < n) offs.resize((size_t)n);
    if ((int64_t)wts.size() < n) wts.resize((size_t)n);
    for (int64_t e = 0; e < n; e++) {
      const tfgpu_sr_frame &f = events[e];
      starts[(size_t)e] = f.code ? f.start : f.start - 5;
      if (msgs && msgs->offset) offs[(size_t)e] = msgs->offset[f.msg];
      if (msgs && msgs->write_time_ns) wts[(size_t)e] = msgs->write_time_ns[f.msg];
    }
    starts[(size_t)n] = len;
    for (int64_t e = n - 1; e >= 0; e--) if (starts[(size_t)e] > starts[(size_t)e + 1]) starts[(size_t)e] = starts[(size_t)e + 1];  // (cannot happen with tfgpu_sr_frames' list; the device call checks the spans)
    tfgpu_messages em{};
    em.nmsg = n; em.start = starts.data(); em.offset = msgs && msgs->offset ? offs.data() : nullptr; em.write_time_ns = msgs && msgs->write_time_ns ? wts.data() : nullptr;
    if ((int64_t)r->frames.size() < n) r->frames.resize((size_t)n);
    rc_clock.mark("event slots");
    bool uniform = false;
    {
      // payload spans claimed from the events' ends instead of walked once the tile parser has taken a batch of this receiver: it
      // proves them, or the walker walks them (tf_debezium.hip, Params::tent)
      struct Tent { bool on; explicit Tent(bool o) : on(o) { if (on) tf::dbz::dbz_tentative_frames(true); } ~Tent() { if (on) tf::dbz::dbz_tentative_frames(false); } } tent(r->last_quick);
      struct Trust { Trust() { tf::dbz::dbz_trust_frames(true); } ~Trust() { tf::dbz::dbz_trust_frames(false); } } trust;  // `events` is what tfgpu_sr_frames wrote, untouched
      // … and the frames themselves only when they are not all alike (every event OK, the batch's one schema id, a payload the tile parser takes)
      struct Lazy { bool on; Lazy(bool o, uint32_t id) : on(o) { if (on) tf::dbz::dbz_lazy_frames(true, id, TFGPU_DBZ_REGISTRY_HASH); } ~Lazy() { if (on) tf::dbz::dbz_lazy_frames(false, 0, 0); } } lazy(ids.size() == 1, ids.empty() ? 0u : ids[0]);
      rc = tfgpu_debezium_registry_frames(bytes, len, mem, &em, events, r->frames.data());
      uniform = rc == TFGPU_OK && ids.size() == 1 && tf::dbz::dbz_last_unpack_uniform();
    }
    if (rc) return rc;
    rc_clock.mark("registry_frames");
    if (uniform) std::fill(event_codes, event_codes + n, 0);
    else for (int64_t e = 0; e < n; e++) event_codes[e] = r->frames[(size_t)e].code;
    rc_clock.mark("event codes");
    bool codes_clean = uniform;
    std::vector<tfgpu_row_error> &errs = r->errs;
    if ((int64_t)errs.size() < n) errs.resize((size_t)n);
    for (uint32_t id : ids) {
      const std::shared_ptr<tfgpu_dbz_schema> &s = r->registry[id];
      tfgpu_dbz_options o{};
      o.schema_hash[0] = id; o.schema_hash[1] = TFGPU_DBZ_REGISTRY_HASH;
      o.nfields = s->code ? 0 : (int32_t)s->cfields.size(); o.fields = s->code ? nullptr : s->cfields.data(); o.schema_code = s->code;
      tfgpu_dbz_receiver::Group g;
      g.schema = s;
      if (!r->spare_rows.empty()) { g.rows.swap(r->spare_rows.back()); r->spare_rows.pop_back(); }
      if ((int64_t)g.rows.size() < n) g.rows.resize((size_t)n);
      int64_t ne = 0;
      {
        struct Trust { Trust() { tf::dbz::dbz_trust_frames(true); } ~Trust() { tf::dbz::dbz_trust_frames(false); } } trust;  // r->frames is what the call above wrote, untouched
        rc = tfgpu_debezium_parse(&o, bytes, len, mem, &em, r->frames.data(), &g.batch, g.rows.data(), n, errs.data(), (int64_t)errs.size(), &ne);
      }
      if (rc) return rc;
      r->last_quick = tf::dbz::dbz_last_parse_was_quick();
      for (int64_t i = 0; i < std::min<int64_t>(ne, (int64_t)errs.size()); i++) if (errs[(size_t)i].row >= 0 && errs[(size_t)i].row < n) event_codes[errs[(size_t)i].row] = errs[(size_t)i].code;
      if (ne) codes_clean = false;
      tfgpu_batch v{};
      if (g.batch && tfgpu_dbatch_view(g.batch, &v) == TFGPU_OK) g.nrows = v.nrows;
      if (s->code || g.nrows == 0) { if (g.batch) tfgpu_dbatch_free(g.batch); r->spare_rows.push_back(std::move(g.rows)); continue; }
      r->groups.push_back(std::move(g));
    }
    rc_clock.mark("parse per schema");
    // DoBuf (parser.go:59-71): the first event that fails ends its Kafka message — DoOne hands back a nil rest; an event the stock
    // code must redo takes its whole message there
    bool any_lost = false;
    for (int64_t a = 0; a < n && !codes_clean;) {   // (every code zero: nothing ends, nothing is redone)
      int64_t z = a;
      while (z < n && events[z].msg == events[a].msg) z++;
      bool host = false;
      for (int64_t e = a; e < z && !host; e++) { if (event_codes[e] == TFGPU_ROW_HOST_FALLBACK) host = true; else if (event_codes[e]) break; }
      bool dead = false;
      for (int64_t e = a; e < z; e++) {
        if (host) { any_lost |= event_codes[e] == TFGPU_ROW_OK; event_codes[e] = TFGPU_ROW_HOST_FALLBACK; }
        else if (dead) { any_lost |= event_codes[e] == TFGPU_ROW_OK; event_codes[e] = TFGPU_ROW_DROPPED; }
        else if (event_codes[e]) dead = true;
      }
      a = z;
    }
    rc_clock.mark("DoBuf rule");
    if (any_lost) {  // rows of events that are no items after all: cut out of their group's batch (a keep mask, compact_rows) and of its row meta
      std::vector<tfgpu_dbz_receiver::Group> kept;
      for (auto &G : r->groups) {
        std::vector<uint32_t> keep((size_t)G.nrows + 1, 0u);
        int64_t nk = 0;
        for (int64_t i = 0; i < G.nrows; i++) {
          if (event_codes[G.rows[(size_t)i].msg]) continue;
          keep[(size_t)i] = 1u;
          G.rows[(size_t)nk++] = G.rows[(size_t)i];
        }
        if (nk == G.nrows) { kept.push_back(std::move(G)); continue; }
        if (nk == 0) { tfgpu_dbatch_free(G.batch); G.batch = nullptr; r->spare_rows.push_back(std::move(G.rows)); continue; }
        std::unique_ptr<tfgpu_dbatch> cut;
        {
          Context &cx = ctx();
          std::lock_guard<std::mutex> lk(cx.mu);
          Buf dk = dalloc(keep.size() * 4 + 16);
          h2d(dk->p, keep.data(), keep.size() * 4);
          cut = tf::compact_rows(*G.batch, dk);  // syncs
        }
        tfgpu_dbatch_free(G.batch);
        G.batch = cut.release(); G.nrows = nk;
        kept.push_back(std::move(G));
      }
      r->groups = std::move(kept);
    }
    *ngroups = (int32_t)r->groups.size();
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }
}

// One table's rows of the last batch.  *batch passes to the caller (tfgpu_dbatch_free); rows / fields stay the receiver's until
// the next tfgpu_dbz_receive.
int tfgpu_dbz_receive_group(tfgpu_dbz_receiver *r, int32_t g, tfgpu_dbatch **batch, const tfgpu_dbz_row **rows, int64_t *nrows, const tfgpu_dbz_field **fields, int32_t *nfields) {
  if (!r || g < 0 || g >= (int32_t)r->groups.size() || !batch) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbz_receive_group: bad argument");
  auto &G = r->groups[(size_t)g];
  if (!G.batch) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbz_receive_group: the group's batch was taken already");
  *batch = G.batch; G.batch = nullptr;
  if (rows) *rows = G.rows.data();
  if (nrows) *nrows = G.nrows;
  if (fields) *fields = G.schema->cfields.data();
  if (nfields) *nfields = (int32_t)G.schema->cfields.size();
  return TFGPU_OK;
}

// The ChangeItem members that are not columns, laid out by MESSAGE index (= src_row of the group's rows) — the arrays a
// tfgpu_row_meta of the whole message batch takes (ids u32, lsns u64, commit_times u64, names_form u8; nmsg entries each, zeroed
// first): what the shim would otherwise scatter row by row.  Call it before the next tfgpu_dbz_receive.
int tfgpu_dbz_receive_group_meta(tfgpu_dbz_receiver *r, int32_t g, int64_t nmsg, uint32_t *ids, uint64_t *lsns, uint64_t *commit_times, uint8_t *names_form) {
  if (!r || g < 0 || g >= (int32_t)r->groups.size() || nmsg < 0) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_dbz_receive_group_meta: bad argument");
  const auto &G = r->groups[(size_t)g];
  if (ids) std::memset(ids, 0, (size_t)nmsg * 4);
  if (lsns) std::memset(lsns, 0, (size_t)nmsg * 8);
  if (commit_times) std::memset(commit_times, 0, (size_t)nmsg * 8);
  if (names_form) std::memset(names_form, 0, (size_t)nmsg);
  for (int64_t i = 0; i < G.nrows; i++) {
    const tfgpu_dbz_row &w = G.rows[(size_t)i];
    if (w.msg < 0 || w.msg >= nmsg) continue;
    if (ids) ids[w.msg] = w.id;
    if (lsns) lsns[w.msg] = w.lsn;
    if (commit_times) commit_times[w.msg] = w.commit_time;
    if (names_form) names_form[w.msg] = w.names_form;
  }
  return TFGPU_OK;
}

}  // extern "C"

// ======================================================================================================================
// Confluent Schema Registry, JSON schemas: the per-schema set-up of the reference — unmarshal the JSON schema, resolve every
// property to a column (type through `oneOf`, the required set), derive the table id from the title.
//   JSONProperties, jsonPropertyToJSONSchemaRow   pkg/parsers/registry/confluentschemaregistry/engine/utils_json.go:15-21, 71-95
//   jsonSchemaTypes                               engine/types_json.go:25-32
//   BuildJSONTableID                              table_name_policy/table_name_policy.go:73-92
// Runs once per schema id (the shim caches it next to its registry client); per message: tfgpu_sr_frames / tfgpu_sr_json_parse.
// ======================================================================================================================
namespace tf {
namespace srschema {
// encoding/json binds every key of the document to the struct field it names — exactly or, failing that, ignoring case — in
// document order, so the LAST key that names the field is the one whose value stays (decode.go object())
static const Json *last_field(const Json &obj, const char *name) {
  const Json *out = nullptr;
  if (obj.type != Json::Obj) return nullptr;
  for (auto &kv : obj.obj) if (kv.first == name || dbzrecv::equal_fold_ascii(kv.first, name)) out = &kv.second;
  return out;
}
static int json_type_of(const Json *t) {  // types_json.go:25-32 — 0: no column type
  if (!t || t->type != Json::Str) return 0;
  const std::string &s = t->str;
  if (s == "array" || s == "object") return TFGPU_SRT_ANY;
  if (s == "boolean") return TFGPU_SRT_BOOLEAN;
  if (s == "integer") return TFGPU_SRT_INTEGER;
  if (s == "number") return TFGPU_SRT_NUMBER;
  if (s == "string") return TFGPU_SRT_STRING;
  return 0;
}
}  // namespace srschema
}  // namespace tf

struct tfgpu_sr_schema {
  std::string title, ns, table;
  std::vector<std::string> names;
  std::vector<tfgpu_sr_property> props;
};

extern "C" {

int tfgpu_sr_compile_schema(const char *schema_text, uint64_t len, const char *policy, const char *manual_table_name, tfgpu_sr_schema **out) {
  using namespace tf::srschema;
  if (!out || (len && !schema_text)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_sr_compile_schema: null argument");
  try {
    Json js = Json::parse(std::string(schema_text ? schema_text : "", (size_t)len));
    const Json *ty = last_field(js, "type");
    if (js.type != Json::Obj || !ty || ty->type != Json::Str || ty->str != "object") return tf::fail(TFGPU_ERR_CONFIG, "json schema type must be 'object'");  // utils_json.go:35-37
    auto s = std::make_unique<tfgpu_sr_schema>();
    std::vector<std::string> required;
    if (const Json *r = last_field(js, "required"); r && r->type == Json::Arr) for (auto &x : r->arr) if (x.type == Json::Str) required.push_back(x.str);
    std::map<std::string, const Json *> props;  // std::string compares bytewise, as Go's sort.Strings does; a repeated key: the last one
    if (const Json *p = last_field(js, "properties"); p && p->type == Json::Obj) for (auto &kv : p->obj) props[kv.first] = &kv.second;
    for (auto &kv : props) {
      const Json &p = *kv.second;
      int t = json_type_of(last_field(p, "type"));
      bool req = std::find(required.begin(), required.end(), kv.first) != required.end();
      if (const Json *one = last_field(p, "oneOf"); one && one->type == Json::Arr)
        for (auto &q : one->arr) {
          const Json *qt = last_field(q, "type");
          if (qt && qt->type == Json::Str && qt->str == "null") req = false;
          else t = json_type_of(qt);
        }
      if (!t) return tf::fail(TFGPU_ERR_CONFIG, "property " + kv.first + ": JSON-schema type without a column type (DataType \"\" in the reference)");
      s->names.push_back(kv.first);
      s->props.push_back(tfgpu_sr_property{nullptr, t, req ? 1 : 0});
    }
    for (size_t i = 0; i < s->props.size(); i++) s->props[i].name = s->names[i].c_str();
    if (const Json *t = last_field(js, "title"); t && t->type == Json::Str) s->title = t->str;
    // BuildJSONTableID
    const std::string pol = policy && policy[0] ? policy : "debezium_style", manual = manual_table_name ? manual_table_name : "";
    if (!manual.empty()) s->table = manual;
    else if (pol == "debezium_style") {
      const size_t dot = s->title.find('.');
      if (dot == std::string::npos) return tf::fail(TFGPU_ERR_CONFIG, "Can't split title '" + s->title + "' from json into schema and table names");
      s->ns = s->title.substr(0, dot); s->table = s->title.substr(dot + 1);
    } else if (pol == "title") s->table = s->title;
    else return tf::fail(TFGPU_ERR_CONFIG, "invalid JSONTableNamePolicy");
    *out = s.release();
    return TFGPU_OK;
  } catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
}
int tfgpu_sr_schema_info(const tfgpu_sr_schema *s, const tfgpu_sr_property **props, int32_t *nprops, const char **table_ns, const char **table_name, const char **title) {
  if (!s) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_sr_schema_info: null schema");
  if (props) *props = s->props.data();
  if (nprops) *nprops = (int32_t)s->props.size();
  if (table_ns) *table_ns = s->ns.c_str();
  if (table_name) *table_name = s->table.c_str();
  if (title) *title = s->title.c_str();
  return TFGPU_OK;
}
void tfgpu_sr_schema_free(tfgpu_sr_schema *s) { delete s; }

}  // extern "C"
