// tf_protoschema.cpp — the per-schema half of the Confluent-SR parser's PROTOBUF branch: compile the registry's .proto text into the
// column list the device decodes by (tf_protobuf.hip).  What the reference does once per (schema id, message name):
//
//   mdBuilder.toMD, getRecordName                 pkg/parsers/registry/confluentschemaregistry/engine/md_builder.go:26-70, utils_protobuf.go:27-32
//   handleField, protoSchemaTypes                 engine/utils_protobuf.go:58-85, types_protobuf.go:16-35
//   BuildProtobufTableID                          table_name_policy/table_name_policy.go:51-71
//
// The compiler itself (jhump/protoreflect protoparse) is a dependency of the reference, not part of it.  Restated here: the proto3
// language subset whose messages the device takes — singular and repeated scalar / enum fields, and singular message fields whose own
// fields are singular scalars / enums (the shape of both PROTOBUF schemas in the reference's parser test) — with protobuf's scoping rule for type
// names; since round 6 also REPEATED one-level message fields and map<string, V> fields (V scalar / enum).  Everything else is named and handed
// to the stock code (TFGPU_ROW_HOST_FALLBACK for every message of the schema): maps keyed by anything but a string or holding messages, a oneof inside a nested message, proto2 (required / default / groups / extensions), services, imports other than confluent/meta.proto and
// confluent/type/decimal.proto (whose Decimal message is built in).  A text that does not parse is TFGPU_ROW_SR_PROTO ("unable to
// build MessageDescriptor": every message of the schema becomes `_unparsed`).
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "tf_common.hpp"

namespace tf {
namespace pb {

struct SyntaxError { std::string why; };
struct HostOnly { std::string why; };

struct Tok { int kind; std::string s; };  // 0 eof, 1 string, 2 name, 3 number, 4 symbol
static bool name_start(unsigned char c) { return std::isalpha(c) || c == '_'; }
static bool name_char(unsigned char c) { return std::isalnum(c) || c == '_'; }
static std::vector<Tok> tokens(const std::string &t) {
  std::vector<Tok> out;
  size_t i = 0;
  const size_t n = t.size();
  while (i < n) {
    const unsigned char c = (unsigned char)t[i];
    if (std::isspace(c)) { i++; continue; }
    if (c == '/' && i + 1 < n && t[i + 1] == '/') { while (i < n && t[i] != '\n') i++; continue; }
    if (c == '/' && i + 1 < n && t[i + 1] == '*') { const size_t e = t.find("*/", i + 2); if (e == std::string::npos) throw SyntaxError{"unterminated comment"}; i = e + 2; continue; }
    if (c == '"' || c == '\'') {
      size_t j = i + 1;
      while (j < n && t[j] != (char)c) { if (t[j] == '\\') j++; j++; }
      if (j >= n) throw SyntaxError{"unterminated string"};
      out.push_back({1, t.substr(i + 1, j - i - 1)});
      i = j + 1;
      continue;
    }
    if (name_start(c) || (c == '.' && i + 1 < n && name_start((unsigned char)t[i + 1]))) {
      size_t j = i + (c == '.' ? 1 : 0);
      for (;;) {
        while (j < n && name_char((unsigned char)t[j])) j++;
        if (j + 1 < n && t[j] == '.' && name_start((unsigned char)t[j + 1])) { j++; continue; }
        break;
      }
      out.push_back({2, t.substr(i, j - i)});
      i = j;
      continue;
    }
    if (std::isdigit(c) || ((c == '-' || c == '+') && i + 1 < n && std::isdigit((unsigned char)t[i + 1]))) {
      size_t j = i + 1;
      while (j < n && (std::isalnum((unsigned char)t[j]) || t[j] == '.' || t[j] == '_' || ((t[j] == '+' || t[j] == '-') && (t[j - 1] == 'e' || t[j - 1] == 'E')))) j++;
      out.push_back({3, t.substr(i, j - i)});
      i = j;
      continue;
    }
    if (std::strchr("{}[]()<>=;,:", c)) { out.push_back({4, std::string(1, (char)c)}); i++; continue; }
    throw SyntaxError{"unexpected character in the schema text"};
  }
  return out;
}

struct FieldDecl { std::string name, type, label, map_key; long long number = 0; bool has_default = false, is_map = false; int oneof = 0; /* k > 0: a member of the message's k-th oneof */ };
struct Msg { std::string name, full; std::vector<FieldDecl> fields; std::vector<std::unique_ptr<Msg>> messages; std::vector<std::string> enums; int noneof = 0; };

struct Parser {
  std::vector<Tok> t; size_t i = 0;
  std::string package, syntax = "proto2";
  std::vector<std::string> imports, enums;
  std::vector<std::unique_ptr<Msg>> messages;
  const Tok &peek() const { static const Tok eof{0, ""}; return i < t.size() ? t[i] : eof; }
  Tok next() { Tok k = peek(); i++; return k; }
  Tok peek2() const { return i + 1 < t.size() ? t[i + 1] : Tok{0, ""}; }
  bool is_sym(const Tok &k, const char *s) const { return k.kind == 4 && k.s == s; }
  void expect(const char *s) { const Tok k = next(); if (k.s != s) throw SyntaxError{std::string("expected ") + s + ", got " + k.s}; }
  std::string ident() { const Tok k = next(); if (k.kind != 2) throw SyntaxError{"expected a name, got " + k.s}; return k.s; }
  void skip_statement() {
    int depth = 0;
    for (;;) {
      const Tok k = next();
      if (k.kind == 0) throw SyntaxError{"unexpected end"};
      if (k.kind != 4) continue;
      if (std::strchr("{[(<", k.s[0])) depth++;
      else if (std::strchr("}])>", k.s[0])) depth--;
      else if (k.s == ";" && depth == 0) return;
    }
  }
  void skip_block() {
    expect("{");
    int depth = 1;
    while (depth) {
      const Tok k = next();
      if (k.kind == 0) throw SyntaxError{"unexpected end"};
      if (is_sym(k, "{")) depth++; else if (is_sym(k, "}")) depth--;
    }
  }
  void file() {
    while (peek().kind) {
      const Tok k = peek();
      if (k.kind == 2 && k.s == "syntax") { next(); expect("="); syntax = next().s; expect(";"); }
      else if (k.kind == 2 && k.s == "package") { next(); package = ident(); expect(";"); }
      else if (k.kind == 2 && k.s == "import") {
        next();
        if (peek().kind == 2 && (peek().s == "public" || peek().s == "weak")) next();
        const Tok f = next();
        if (f.kind != 1) throw SyntaxError{"import"};
        imports.push_back(f.s);
        expect(";");
      } else if (k.kind == 2 && k.s == "option") skip_statement();
      else if (k.kind == 2 && k.s == "message") messages.push_back(message(package));
      else if (k.kind == 2 && k.s == "enum") { next(); const std::string n = ident(); skip_block(); enums.push_back((package.empty() ? "" : package + ".") + n); }
      else if (k.kind == 2 && (k.s == "service" || k.s == "extend")) throw HostOnly{k.s};
      else if (is_sym(k, ";")) next();
      else throw SyntaxError{"unexpected " + k.s};
    }
  }
  int depth = 0;
  std::unique_ptr<Msg> message(const std::string &scope) {
    struct Depth { int &d; explicit Depth(int &x) : d(x) { if (++d > 64) throw HostOnly{"messages nested deeper than 64 levels"}; } ~Depth() { d--; } } guard(depth);
    expect("message");
    auto m = std::make_unique<Msg>();
    m->name = ident();
    m->full = (scope.empty() ? "" : scope + ".") + m->name;
    expect("{");
    for (;;) {
      const Tok k = peek();
      if (is_sym(k, "}")) { next(); return m; }
      if (k.kind == 0) throw SyntaxError{"unexpected end"};
      if (k.kind == 2 && k.s == "message") m->messages.push_back(message(m->full));
      else if (k.kind == 2 && k.s == "enum") { next(); const std::string n = ident(); skip_block(); m->enums.push_back(m->full + "." + n); }
      else if (k.kind == 2 && (k.s == "option" || k.s == "reserved")) skip_statement();
      else if (k.kind == 2 && (k.s == "extensions" || k.s == "extend" || k.s == "group")) throw HostOnly{k.s};
      else if (k.kind == 2 && k.s == "oneof") {
        // oneof name { type member = N; … }: its members are fields of the message like any other, in declaration order (the descriptor's field list; GetKnownFields,
        // types_protobuf.go:103) — a member takes no label and is no map (language guide); what a oneof adds is on the wire: setting a member clears the others
        next(); ident(); expect("{");
        const int g = ++m->noneof;
        for (;;) {
          const Tok q = peek();
          if (is_sym(q, "}")) { next(); break; }
          if (q.kind == 0) throw SyntaxError{"unexpected end"};
          if (q.kind == 2 && q.s == "option") { skip_statement(); continue; }
          if (is_sym(q, ";")) { next(); continue; }
          if (q.kind == 2 && q.s == "group") throw HostOnly{"group"};
          FieldDecl f = field();
          if (!f.label.empty() || f.is_map) throw SyntaxError{"a oneof member takes no label and is no map"};
          f.oneof = g;
          m->fields.push_back(std::move(f));
        }
      }
      else if (is_sym(k, ";")) next();
      else m->fields.push_back(field());
    }
  }
  FieldDecl field() {
    FieldDecl f;
    if (peek().kind == 2 && (peek().s == "optional" || peek().s == "required" || peek().s == "repeated")) f.label = next().s;
    if (peek().kind == 2 && peek().s == "group") throw HostOnly{"group"};
    if (peek().kind == 2 && peek().s == "map" && is_sym(peek2(), "<")) {  // map<K, V> name = N;  (language guide: no label; K an integral or string scalar, V anything but a map)
      if (!f.label.empty()) throw SyntaxError{"a map field takes no label"};
      next(); expect("<");
      f.map_key = ident(); expect(","); f.type = ident(); expect(">");
      f.is_map = true;
    } else f.type = ident();
    f.name = ident();
    expect("=");
    const Tok num = next();
    if (num.kind != 3) throw SyntaxError{"field number"};
    char *end = nullptr;
    f.number = std::strtoll(num.s.c_str(), &end, 0);
    if (!end || *end) throw SyntaxError{"field number"};
    if (is_sym(peek(), "[")) {
      int depth = 0;
      for (;;) {
        const Tok k = next();
        if (k.kind == 0) throw SyntaxError{"unexpected end"};
        if (k.kind == 2 && k.s == "default" && depth == 1) f.has_default = true;
        if (k.kind == 4 && std::strchr("[{(<", k.s[0])) depth++;
        else if (k.kind == 4 && std::strchr("]})>", k.s[0])) { if (--depth == 0) break; }
      }
    }
    expect(";");
    return f;
  }
};

static const std::map<std::string, int> SCALAR = {
    {"double", TFGPU_PB_DOUBLE}, {"float", TFGPU_PB_FLOAT}, {"int64", TFGPU_PB_INT64}, {"uint64", TFGPU_PB_UINT64}, {"int32", TFGPU_PB_INT32}, {"fixed64", TFGPU_PB_FIXED64},
    {"fixed32", TFGPU_PB_FIXED32}, {"bool", TFGPU_PB_BOOL}, {"string", TFGPU_PB_STRING}, {"bytes", TFGPU_PB_BYTES}, {"uint32", TFGPU_PB_UINT32}, {"sfixed32", TFGPU_PB_SFIXED32},
    {"sfixed64", TFGPU_PB_SFIXED64}, {"sint32", TFGPU_PB_SINT32}, {"sint64", TFGPU_PB_SINT64}};

struct Member { std::string name; int32_t number, ptype; };
struct Field { std::string name; int32_t number, ptype; std::vector<Member> members; bool repeated = false, map = false; int32_t oneof = 0; };

static void collect(const std::vector<std::unique_ptr<Msg>> &ms, std::map<std::string, const Msg *> &msgs, std::vector<std::string> &enums) {
  for (auto &m : ms) { msgs[m->full] = m.get(); for (auto &e : m->enums) enums.push_back(e); collect(m->messages, msgs, enums); }
}

struct Compiled { std::string record, ns, table; std::vector<Field> fields; };

static Compiled compile(const std::string &text, const std::string &policy, const std::string &manual, const std::string &message_name) {
  Parser p;
  p.t = tokens(text);
  p.file();
  for (auto &imp : p.imports) if (imp != "confluent/meta.proto" && imp != "confluent/type/decimal.proto") throw HostOnly{"import " + imp};
  if (p.messages.empty()) throw HostOnly{"no message in the file: the reference dereferences a nil descriptor"};
  std::map<std::string, const Msg *> msgs;
  std::vector<std::string> enums = p.enums;
  collect(p.messages, msgs, enums);
  const Msg *md = nullptr;
  if (!message_name.empty()) { auto it = msgs.find(message_name); if (it != msgs.end()) md = it->second; }
  if (!md) md = p.messages[0].get();  // getRecordName: the first message of the file
  Compiled out;
  out.record = md->full;
  if (!manual.empty()) out.table = manual;
  else if (policy == "debezium_style") {
    std::vector<std::string> parts;
    size_t a = 0;
    for (;;) { const size_t d = out.record.find('.', a); parts.push_back(out.record.substr(a, d == std::string::npos ? d : d - a)); if (d == std::string::npos) break; a = d + 1; }
    if (parts.size() != 4) throw SyntaxError{"Can't split recordName '" + out.record + "' into schema and table names"};
    out.ns = parts[1]; out.table = parts[2];
  } else if (policy == "message_name") out.table = out.record.substr(out.record.rfind('.') == std::string::npos ? 0 : out.record.rfind('.') + 1);
  else throw SyntaxError{"invalid ProtobufTableNamePolicy"};
  if (p.syntax != "proto3") throw HostOnly{"proto2: required / default / groups"};
  // protobuf's name resolution: the innermost scope outwards; a leading dot is fully qualified.  1 message, 2 enum, 3 built in
  auto resolve = [&](const std::string &scope, const std::string &typ, const Msg **m) {
    std::vector<std::string> cands;
    if (!typ.empty() && typ[0] == '.') cands.push_back(typ.substr(1));
    else {
      std::string s = scope;
      for (;;) {
        cands.push_back((s.empty() ? "" : s + ".") + typ);
        if (s.empty()) break;
        const size_t d = s.rfind('.');
        s = d == std::string::npos ? "" : s.substr(0, d);
      }
    }
    for (auto &c : cands) {
      auto it = msgs.find(c);
      if (it != msgs.end()) { *m = it->second; return 1; }
      if (std::find(enums.begin(), enums.end(), c) != enums.end()) return 2;
      if (c == "confluent.type.Decimal") return 3;
    }
    throw HostOnly{"type " + typ + " is not in this file (an import the device does not restate)"};
  };
  std::map<long long, bool> seen;
  for (auto &f : md->fields) {
    if (f.has_default) throw HostOnly{"default option"};
    if (f.number <= 0 || f.number > 536870911 || !seen.emplace(f.number, true).second) throw SyntaxError{"field number"};
    Field o;
    o.name = f.name; o.number = (int32_t)f.number; o.repeated = f.label == "repeated"; o.oneof = f.oneof;
    if (f.is_map) {
      // map<string, V>: on the wire a repeated entry message {K key = 1; V value = 2;}; the dynamic message holds a Go map, unpackRepeatedVal
      // (types_protobuf.go:57-71) takes string keys only and json.Marshal writes the map's keys in byte order — an `any` column
      // {"k":v,…}.  Other key types are an error in the reference ("not supported yet as a map key"): stock path.  Message values: one more level: stock path.
      if (f.map_key != "string") throw HostOnly{"map with a key type other than string"};
      auto vs = SCALAR.find(f.type);
      int vt;
      if (vs != SCALAR.end()) vt = vs->second;
      else { const Msg *r2 = nullptr; if (resolve(md->full, f.type, &r2) != 2) throw HostOnly{"map with message values"}; vt = TFGPU_PB_ENUM; }
      o.ptype = TFGPU_PB_MESSAGE; o.map = true;
      o.members = {{"key", 1, TFGPU_PB_STRING}, {"value", 2, vt}};
      out.fields.push_back(std::move(o));
      continue;
    }
    auto sc = SCALAR.find(f.type);
    if (sc != SCALAR.end()) { o.ptype = sc->second; out.fields.push_back(std::move(o)); continue; }
    const Msg *ref = nullptr;
    const int kind = resolve(md->full, f.type, &ref);
    if (kind == 2) { o.ptype = TFGPU_PB_ENUM; out.fields.push_back(std::move(o)); continue; }
    o.ptype = TFGPU_PB_MESSAGE;   // (a repeated message field: the array of its elements' maps, unpackRepeatedVal over *dynamic.Message elements)
    if (kind == 3) o.members = {{"value", 1, TFGPU_PB_BYTES}, {"precision", 2, TFGPU_PB_UINT32}, {"scale", 3, TFGPU_PB_INT32}};  // confluent/type/decimal.proto
    else {
      std::map<long long, bool> seen2;
      for (auto &g : ref->fields) {
        if (g.label == "repeated" || g.is_map || g.has_default || g.label == "required" || g.oneof) throw HostOnly{"a nested message the device does not walk"};
        if (g.number <= 0 || g.number > 536870911 || !seen2.emplace(g.number, true).second) throw SyntaxError{"field number"};
        auto s2 = SCALAR.find(g.type);
        if (s2 != SCALAR.end()) { o.members.push_back({g.name, (int32_t)g.number, s2->second}); continue; }
        const Msg *r2 = nullptr;
        if (resolve(ref->full, g.type, &r2) != 2) throw HostOnly{"messages nested deeper than one level"};
        o.members.push_back({g.name, (int32_t)g.number, TFGPU_PB_ENUM});
      }
    }
    std::sort(o.members.begin(), o.members.end(), [](const Member &a, const Member &b) { return a.name < b.name; });  // json.Marshal of a map: keys in byte order
    for (size_t k = 1; k < o.members.size(); k++) if (o.members[k].name == o.members[k - 1].name) throw SyntaxError{"a field name repeats"};
    out.fields.push_back(std::move(o));
  }
  for (size_t a = 0; a < out.fields.size(); a++) for (size_t b = a + 1; b < out.fields.size(); b++) if (out.fields[a].name == out.fields[b].name) throw SyntaxError{"a field name repeats"};
  return out;
}

}  // namespace pb
}  // namespace tf

using namespace tf;

struct tfgpu_pb_schema {
  int code = 0; std::string why;
  pb::Compiled c;
  std::vector<std::vector<tfgpu_pb_member>> cmembers;
  std::vector<tfgpu_pb_field> cfields;
  void seal() {
    cmembers.clear(); cfields.clear();
    for (auto &f : c.fields) {
      std::vector<tfgpu_pb_member> ms;
      for (auto &m : f.members) ms.push_back(tfgpu_pb_member{m.name.c_str(), m.number, m.ptype});
      cmembers.push_back(std::move(ms));
    }
    for (size_t i = 0; i < c.fields.size(); i++) cfields.push_back(tfgpu_pb_field{c.fields[i].name.c_str(), c.fields[i].number, c.fields[i].ptype, (int32_t)cmembers[i].size(), cmembers[i].empty() ? nullptr : cmembers[i].data(), c.fields[i].map ? 2 : (c.fields[i].repeated ? 1 : 0), c.fields[i].oneof});
  }
};

extern "C" {

int tfgpu_sr_compile_proto(const char *schema_text, uint64_t len, const char *policy, const char *manual_table_name, const char *message_name, tfgpu_pb_schema **out) {
  if (!out || (len && !schema_text)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_sr_compile_proto: null argument");
  try {
    auto s = std::make_unique<tfgpu_pb_schema>();
    try { s->c = pb::compile(std::string(schema_text ? schema_text : "", (size_t)len), policy && *policy ? policy : "debezium_style", manual_table_name ? manual_table_name : "", message_name ? message_name : ""); }
    catch (const pb::SyntaxError &e) { s->code = TFGPU_ROW_SR_PROTO; s->why = e.why; s->c = pb::Compiled(); }
    catch (const pb::HostOnly &e) { s->code = TFGPU_ROW_HOST_FALLBACK; s->why = e.why; s->c = pb::Compiled(); }
    s->seal();
    *out = s.release();
    return TFGPU_OK;
  } catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); }
}
int tfgpu_pb_schema_info(const tfgpu_pb_schema *s, int32_t *code, const tfgpu_pb_field **fields, int32_t *nfields, const char **table_ns, const char **table_name, const char **record, const char **why) {
  if (!s) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_pb_schema_info: null schema");
  if (code) *code = s->code;
  if (fields) *fields = s->cfields.data();
  if (nfields) *nfields = (int32_t)s->cfields.size();
  if (table_ns) *table_ns = s->c.ns.c_str();
  if (table_name) *table_name = s->c.table.c_str();
  if (record) *record = s->c.record.c_str();
  if (why) *why = s->why.c_str();
  return TFGPU_OK;
}
void tfgpu_pb_schema_free(tfgpu_pb_schema *s) { delete s; }

}  // extern "C"
