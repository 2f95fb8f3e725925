// tf_plan.hpp — host-side transformer plans: the C++ image of the reference's
// transformer structs, built from the same JSON configs the Go factories take
// (pkg/transformer/registry.go:34-47).  Immutable after construction, so a plan
// can be used from any number of threads (transformation.go:131-135).
#pragma once
#include <array>
#include <map>
#include <memory>
#include <regex>
#include <string>
#include <vector>

#include "tf_common.hpp"

namespace tf {

// ---- tiny JSON reader (configs only) ---------------------------------------
struct Json {
  enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;
  const Json *get(const std::string &key) const;  // case-insensitive fallback like encoding/json
  std::string s(const std::string &key, const std::string &dflt = "") const;
  bool flag(const std::string &key, bool dflt = false) const;
  std::vector<std::string> strings(const std::string &key) const;
  static Json parse(const std::string &text);  // throws Error(TFGPU_ERR_CONFIG)
};

// ---- filter.Filter (pkg/transformer/registry/filter/filter.go:19-74) --------
struct NameFilter {
  std::vector<std::string> include_src, exclude_src;
  std::vector<std::regex> include, exclude;
  void init(const std::vector<std::string> &inc, const std::vector<std::string> &exc);
  bool match(const std::string &v) const;
  bool empty() const { return include_src.empty() && exclude_src.empty(); }
  // MatchAnyTableNameVariant (transformer_common.go:9-33)
  bool match_table(const std::string &ns, const std::string &name) const;
};

// ---- filter_rows predicate (library/go/yandex/cloud/filter) ------------------
enum FOp : int32_t { F_EQ, F_NE, F_LT, F_LE, F_GT, F_GE, F_IN, F_NOTIN, F_MATCH, F_NOTMATCH };
enum FVal : int32_t { FV_STRING, FV_TIME, FV_BOOL, FV_FLOAT, FV_INT, FV_NULL, FV_BADLIST };

struct FTerm {
  std::string attr;
  int32_t op = F_EQ;
  int32_t vtype = FV_INT;  // element type for lists
  bool is_list = false;
  std::vector<int64_t> ints;      // FV_INT / FV_BOOL(0,1) / FV_TIME (unix micro)
  std::vector<double> floats;     // FV_FLOAT
  std::vector<std::string> strs;  // FV_STRING
};
struct FExpr { std::vector<FTerm> terms; };
std::vector<FTerm> parse_filter(const std::string &src);  // throws Error(TFGPU_ERR_CONFIG)

}  // namespace tf

enum PlanKind { PK_MASK, PK_RENAME, PK_FILTER_COLUMNS, PK_SKIP_EVENTS, PK_FILTER_ROWS, PK_TO_STRING, PK_TO_DATETIME, PK_SHARDER, PK_REPLACE_PK, PK_SQL };

namespace tf {
// ---- `sql` transformer, device subset (tf_sql.cpp) ----------------------------------------------------------------------
enum SqlType : int32_t { SQL_PENDING = 0, SQL_I8, SQL_I16, SQL_I32, SQL_I64, SQL_U8, SQL_U16, SQL_U32, SQL_U64, SQL_F64, SQL_STRING, SQL_DATE, SQL_DATETIME, SQL_DATETIME64 };  // ClickHouse types
enum SqlItemKind : int32_t { SQL_STAR, SQL_COLUMN, SQL_CONST_INT, SQL_CONST_STR, SQL_INT_EXPR, SQL_TO_STRING, SQL_TO_DATETIME, SQL_EXPR };
// the general expression tree (SQL_EXPR items, a WHERE that is not an OR of ANDs of column-against-literal terms)
enum SqlNodeOp : int32_t { SN_COL, SN_INT, SN_STR, SN_ADD, SN_SUB, SN_MUL, SN_NEG, SN_CAST, SN_LEN, SN_CITY64, SN_LOWER, SN_UPPER, SN_TOSTR, SN_TODT,
                           SN_EQ, SN_NE, SN_LT, SN_LE, SN_GT, SN_GE, SN_AND, SN_OR, SN_NOT, SN_IN, SN_NOTIN };
struct SqlNode { int op = SN_INT; int ty = SQL_PENDING; int64_t ival = 0; std::string s; std::vector<int> kids; };  // ty: literals and casts; the rest is typed over a schema (sql_node_types)
struct SqlStep { bool is_cast; int ty; int64_t addend; bool minus; };  // toIntN(...) / ± integer literal (ty: the literal's type)
struct SqlItem {  // one entry of the select list as parsed
  int kind = SQL_COLUMN;
  std::string name;  // result column name (alias, or the column's own)
  std::string src;   // source column
  int cast = 0;      // ClickHouse type of a constant / toString / toDateTime result
  int64_t ival = 0; std::string sval;
  std::vector<SqlStep> steps;  // SQL_INT_EXPR: applied to the source column in order
  int root = -1;               // SQL_EXPR: node of tfgpu_plan::sql_nodes
};
struct SqlOp { int64_t addend; int ty; };  // v = wrap(v + addend, ty)
struct SqlOut {  // one result column over a concrete input schema
  int kind = SQL_COLUMN; std::string name; int src = -1; int ch = SQL_PENDING, yt = 0; bool key = false;
  int64_t ival = 0; std::string sval; std::vector<SqlOp> ops;
  int root = -1;  // SQL_EXPR
};
}  // namespace tf

struct tfgpu_plan {
  int kind = PK_MASK;
  std::string type_name;
  tf::NameFilter tables, columns;
  // mask_field
  std::string salt;
  std::vector<std::string> mask_cols;
  uint32_t ipad_state[8], opad_state[8];  // SHA-256 midstates after the key block
  // rename_tables
  std::vector<std::array<std::string, 4>> renames;  // from_ns, from_name, to_ns, to_name
  // skip_events
  bool skip[4] = {false, false, false, false};
  // filter_rows
  std::vector<tf::FExpr> exprs;
  // convert_to_string
  bool to_bytes = false, skip_utc = false;
  // sharder_transformer
  int64_t shards = 1;
  bool is_random = false;
  // sql (device subset: tf_sql.cpp); the WHERE clause is `exprs`, an OR of ANDs like filter_rows'
  std::string sql_query;
  std::vector<tf::SqlItem> sql_items;
  bool sql_has_where = false;
  std::vector<tf::SqlNode> sql_nodes;  // expression trees of SQL_EXPR items and of a general WHERE
  int sql_where_root = -1;             // >= 0: the WHERE is this tree (`exprs` is empty then)
  int sql_where_tree = -1;             // the WHERE's tree whatever its shape (a UInt64 column sends the filter_rows form to it too: sql_where_as_tree)
  // replace_primary_key
  std::vector<std::string> new_keys;
  bool is_new_key(const std::string &n) const {
    for (auto &c : new_keys) if (c == n) return true;
    return false;
  }

  bool mask_has(const std::string &n) const {
    for (auto &c : mask_cols) if (c == n) return true;
    return false;
  }
};

namespace tf {
struct SchemaCol { std::string name; int dtype; uint32_t flags; };  // the three ColSchema fields Suitable / ResultSchema read
// ResultSchema of one transformer over a column list, in place (hmac_hasher.go:35-46, to_string.go:114-127, ...)
void plan_result_columns(const tfgpu_plan &p, std::vector<SchemaCol> &cols);
void executor_shutdown();
std::unique_ptr<tfgpu_plan> make_plan(const std::string &type_name, const std::string &config_json);
bool plan_suitable(const tfgpu_plan &p, const std::string &ns, const std::string &name, const tfgpu_schema &s);
std::string plan_description(const tfgpu_plan &p);
bool is_system_table(const std::string &name);
void sha256_midstate(const uint8_t block[64], uint32_t out[8]);

void sql_parse(const std::string &query, tfgpu_plan &p);                                              // tf_sql.cpp
std::vector<SqlOut> sql_resolve(const tfgpu_plan &p, const std::vector<SchemaCol> &in);            // tf_sql.cpp
bool sql_where_as_tree(const tfgpu_plan &p, const std::vector<SchemaCol> &in);                        // the WHERE runs as the expression program over this schema
std::vector<int> sql_node_types(const tfgpu_plan &p, const std::vector<SchemaCol> &in);               // ClickHouse type of every node that is reachable from an item / the WHERE (SQL_PENDING elsewhere)
std::string type_name(int dtype);                                                                    // YT type name of a TFGPU_T_* code
// Apply one plan to a device batch (tf_transform.hip). `errs` collects row errors.
struct ApplyCtx {
  std::vector<tfgpu_row_error> errs;
  int step = 0;
};
std::unique_ptr<tfgpu_dbatch> apply_plan(const tfgpu_plan &p, const tfgpu_dbatch &in, ApplyCtx &ax);
void mask_precheck(const tfgpu_plan &p, const tfgpu_dbatch &in);  // throws what apply_mask would refuse for the whole batch
std::vector<int> chain_sequence(const tfgpu_plan *const *plans, int n, std::vector<std::vector<int>> *hopped);  // see tf_transform.hip
}  // namespace tf
