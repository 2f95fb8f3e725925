// tf_api.hip — the transformer half of the C ABI (include/tfgpu.h): plan
// lifecycle, Suitable / ResultSchema, and Apply over a chain of plans.
#include "tf_plan.hpp"

using namespace tf;
namespace tf { std::unique_ptr<tfgpu_dbatch> partition_rows(const tfgpu_dbatch &in, int nparts, int64_t *counts); }

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

// blank-import list of the reference: pkg/transformer/registry/registry.go:3-20
// (device-resident subset; see DESIGN.md for what stays on the host)
static const char *REGISTRY[] = {"mask_field", "rename_tables", "filter_columns", "skip_events", "filter_rows",
                                 "convert_to_string", "convert_to_datetime", "sharder_transformer", "replace_primary_key", "sql"};

static char *dup_cstr(const std::string &s) {
  char *r = (char *)std::malloc(s.size() + 1);
  std::memcpy(r, s.c_str(), s.size() + 1);
  return r;
}

namespace tf {
std::string type_name(int dtype) {
  static const char *const N[] = {"", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float", "double", "boolean", "string", "utf8", "date", "datetime",
                                  "timestamp", "interval", "any"};
  return dtype > 0 && dtype < TFGPU_T__COUNT ? N[dtype] : "?";
}
void plan_result_columns(const tfgpu_plan &p, std::vector<SchemaCol> &cols) {
  std::vector<SchemaCol> out;
  if (p.kind == PK_SQL) {  // clickhouse_local.go:351-421: the query's result columns, keys by name; a result without a key is refused
    bool has_key = false;
    for (const SqlOut &o : sql_resolve(p, cols)) { out.push_back(SchemaCol{o.name, o.yt, o.key ? (uint32_t)TFGPU_COL_KEY : 0u}); has_key |= o.key; }
    if (!has_key) throw Error(TFGPU_ERR_CONFIG, "sql: result table has no primary key");
    cols.swap(out);
    return;
  }
  if (p.kind == PK_REPLACE_PK) {  // replace_primary_key.go:108-131
    if (p.new_keys.size() == 1) { for (auto &c : cols) { if (p.is_new_key(c.name)) c.flags |= TFGPU_COL_KEY; else c.flags &= ~(uint32_t)TFGPU_COL_KEY; } return; }
    for (auto &k : p.new_keys) for (auto &c : cols) if (c.name == k) { SchemaCol o = c; o.flags |= TFGPU_COL_KEY; out.push_back(o); break; }
    for (auto &c : cols) if (!p.is_new_key(c.name)) { SchemaCol o = c; o.flags &= ~(uint32_t)TFGPU_COL_KEY; out.push_back(o); }
    cols.swap(out);
    return;
  }
  for (auto &c : cols) {
    SchemaCol o = c;
    switch (p.kind) {
      case PK_MASK: if (p.mask_has(c.name)) o.dtype = TFGPU_T_UTF8; break;
      case PK_TO_STRING: if (p.columns.match(c.name)) o.dtype = p.to_bytes ? TFGPU_T_BYTES : TFGPU_T_UTF8; break;
      case PK_TO_DATETIME: if (p.columns.match(c.name) && (c.dtype == TFGPU_T_INT32 || c.dtype == TFGPU_T_UINT32)) o.dtype = TFGPU_T_DATETIME; break;
      case PK_FILTER_COLUMNS: if (!p.columns.match(c.name)) continue; break;
      default: break;
    }
    out.push_back(std::move(o));
  }
  cols.swap(out);
}
}  // namespace tf

extern "C" {

int tfgpu_registry_count(void) { return (int)(sizeof REGISTRY / sizeof *REGISTRY); }
const char *tfgpu_registry_name(int i) { return (i >= 0 && i < tfgpu_registry_count()) ? REGISTRY[i] : nullptr; }

int tfgpu_plan_create(const char *type_name, const char *config_json, tfgpu_plan **out) {
  TF_API_BEGIN
  if (!type_name || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_plan_create: null argument");
  *out = make_plan(type_name, config_json ? config_json : "{}").release();
  return TFGPU_OK;
  TF_API_END
}
void tfgpu_plan_destroy(tfgpu_plan *p) { delete p; }
const char *tfgpu_plan_type(const tfgpu_plan *p) { return p ? p->type_name.c_str() : nullptr; }

int tfgpu_plan_description(const tfgpu_plan *p, char *buf, size_t cap) {
  TF_API_BEGIN
  if (!p || !buf || !cap) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_plan_description: null argument");
  std::string d = plan_description(*p);
  size_t n = std::min(cap - 1, d.size());
  std::memcpy(buf, d.data(), n); buf[n] = 0;
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_plan_suitable(const tfgpu_plan *p, const char *ns, const char *name, const tfgpu_schema *s, int *out) {
  TF_API_BEGIN
  if (!p || !s || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_plan_suitable: null argument");
  *out = plan_suitable(*p, ns ? ns : "", name ? name : "", *s) ? 1 : 0;
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_plan_result_schema(const tfgpu_plan *p, const tfgpu_schema *in, tfgpu_schema **out) {
  TF_API_BEGIN
  if (!p || !in || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_plan_result_schema: null argument");
  if (p->kind == PK_SQL) {  // a new table schema: names, YT types of the ClickHouse result types, OriginalType "ch:<type>", keys by name
    std::vector<SchemaCol> cols;
    for (int i = 0; i < in->ncols; i++) cols.push_back(SchemaCol{in->cols[i].name ? in->cols[i].name : "", in->cols[i].dtype, in->cols[i].flags});
    const std::vector<SqlOut> outs = sql_resolve(*p, cols);
    bool has_key = false;
    for (auto &o : outs) has_key |= o.key;
    if (!has_key) return tf::fail(TFGPU_ERR_CONFIG, "sql: result table has no primary key");
    static const char *const CH[] = {"", "Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16", "UInt32", "UInt64", "Float64", "String", "Date", "DateTime", "DateTime64(9)"};
    auto *rs = (tfgpu_schema *)std::calloc(1, sizeof(tfgpu_schema));
    rs->cols = (tfgpu_colschema *)std::calloc(std::max<size_t>(outs.size(), 1), sizeof(tfgpu_colschema));
    for (auto &o : outs) {
      tfgpu_colschema &c = rs->cols[rs->ncols++];
      c.name = dup_cstr(o.name); c.dtype = o.yt; c.flags = o.key ? TFGPU_COL_KEY : 0u;
      c.path = dup_cstr(""); c.original_type = dup_cstr(std::string("ch:") + CH[o.ch]);
      c.table_schema = dup_cstr(""); c.table_name = dup_cstr(""); c.expression = dup_cstr(""); c.properties_json = nullptr;
    }
    *out = rs;
    return TFGPU_OK;
  }
  auto *s = (tfgpu_schema *)std::calloc(1, sizeof(tfgpu_schema));
  s->cols = (tfgpu_colschema *)std::calloc((size_t)std::max(in->ncols, 1), sizeof(tfgpu_colschema));
  std::vector<int> order((size_t)in->ncols);
  std::vector<uint32_t> flags((size_t)in->ncols);
  for (int i = 0; i < in->ncols; i++) { order[(size_t)i] = i; flags[(size_t)i] = in->cols[i].flags; }
  if (p->kind == PK_REPLACE_PK) {  // replace_primary_key.go:108-131: keys first in the configured order (composite), flags rewritten
    std::vector<SchemaCol> cols;
    for (int i = 0; i < in->ncols; i++) cols.push_back(SchemaCol{in->cols[i].name ? in->cols[i].name : "", i, in->cols[i].flags});  // dtype slot carries the index
    plan_result_columns(*p, cols);
    for (size_t k = 0; k < cols.size(); k++) { order[k] = cols[k].dtype; flags[k] = cols[k].flags; }
  }
  for (int oi = 0; oi < in->ncols; oi++) {
    const int i = order[(size_t)oi];
    tfgpu_colschema c = in->cols[i];
    c.flags = flags[(size_t)oi];
    std::string name = c.name ? c.name : "";
    int dtype = c.dtype;
    std::string orig = c.original_type ? c.original_type : "";
    switch (p->kind) {
      case PK_MASK:  // hmac_hasher.go:35-46
        if (p->mask_has(name)) { dtype = TFGPU_T_UTF8; orig = ""; }
        break;
      case PK_TO_STRING:  // to_string.go:114-127
        if (p->columns.match(name)) dtype = p->to_bytes ? TFGPU_T_BYTES : TFGPU_T_UTF8;
        break;
      case PK_TO_DATETIME:  // to_datetime.go:125-133
        if (p->columns.match(name) && (dtype == TFGPU_T_INT32 || dtype == TFGPU_T_UINT32)) dtype = TFGPU_T_DATETIME;
        break;
      case PK_FILTER_COLUMNS:  // filter_columns_transformer.go:228-239
        if (!p->columns.match(name)) continue;
        break;
      default: break;  // rename, skip_events, filter_rows, sharder: schema unchanged
    }
    tfgpu_colschema &o = s->cols[s->ncols++];
    o.name = dup_cstr(name); o.dtype = dtype; o.flags = c.flags;
    o.path = dup_cstr(c.path ? c.path : ""); o.original_type = dup_cstr(orig);
    o.table_schema = dup_cstr(c.table_schema ? c.table_schema : ""); o.table_name = dup_cstr(c.table_name ? c.table_name : "");
    o.expression = dup_cstr(c.expression ? c.expression : ""); o.properties_json = c.properties_json ? dup_cstr(c.properties_json) : nullptr;
  }
  *out = s;
  return TFGPU_OK;
  TF_API_END
}

void tfgpu_schema_free(tfgpu_schema *s) {
  if (!s) return;
  for (int i = 0; i < s->ncols; i++) {
    const tfgpu_colschema &c = s->cols[i];
    std::free((void *)c.name); std::free((void *)c.path); std::free((void *)c.original_type);
    std::free((void *)c.table_schema); std::free((void *)c.table_name); std::free((void *)c.expression); std::free((void *)c.properties_json);
  }
  std::free(s->cols); std::free(s);
}

int tfgpu_apply(tfgpu_plan *const *plans, int nplans, const tfgpu_dbatch *in, tfgpu_dbatch **out, tfgpu_row_error *errs,
                int64_t errs_cap, int64_t *nerrs) {
  TF_API_BEGIN
  if (!in || !out || (nplans > 0 && !plans)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_apply: null argument");
  std::lock_guard<std::mutex> lk(ctx().mu);
  ApplyCtx ax;
  std::unique_ptr<tfgpu_dbatch> cur = std::make_unique<tfgpu_dbatch>(*in);
  // transformation.do (transformation.go:252-274): toApply = t.Apply(toApply).Transformed
  std::vector<std::vector<int>> hopped;
  const std::vector<int> seq = chain_sequence(plans, nplans, &hopped);  // a filter_rows in front of the mask_fields it does not read
  for (int q = 0; q < nplans; q++) {
    const int i = seq[(size_t)q];
    ax.step = i;
    size_t before = ax.errs.size();
    for (int m : hopped[(size_t)q]) mask_precheck(*plans[m], *cur);  // what those masks would have refused comes first, as in the configured order
    std::unique_ptr<tfgpu_dbatch> next = apply_plan(*plans[i], *cur, ax);
    // errors are reported against rows of the ORIGINAL input batch
    if (ax.errs.size() > before && cur->src_row) {
      std::vector<int32_t> sr((size_t)cur->nrows);
      d2h(sr.data(), cur->src_row->p, sr.size() * 4);
      tf::sync();
      for (size_t k = before; k < ax.errs.size(); k++) ax.errs[k].row = sr[(size_t)ax.errs[k].row];
    }
    cur = std::move(next);
  }
  if (nerrs) *nerrs = (int64_t)ax.errs.size();
  if (errs) for (int64_t k = 0; k < errs_cap && k < (int64_t)ax.errs.size(); k++) errs[k] = ax.errs[(size_t)k];
  *out = cur.release();
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_partition(const tfgpu_dbatch *in, int nparts, tfgpu_dbatch **out, int64_t *counts) {
  TF_API_BEGIN
  tf::dense(in, true);  // its rows may still be a selection (tfgpu_dbatch::pending)
  if (!in || !out || !counts || nparts < 1) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_partition: bad argument");
  std::lock_guard<std::mutex> lk(ctx().mu);
  *out = partition_rows(*in, nparts, counts).release();
  return TFGPU_OK;
  TF_API_END
}

}  // extern "C"
