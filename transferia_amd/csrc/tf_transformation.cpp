// tf_transformation.cpp — transformation.Push around the Apply chain (pkg/transformer/transformation.go:46-282):
// the table plan per (TableID, schema) built from Suitable / ResultSchema and cached, the Apply loop of
// transformation.do that keeps the INPUT item of every TransformerError (what errorChangeItems :191-235 turns into
// `__transform_error` rows), MiddlewareTransformerStats — and an executor that runs pushes on the library's own lane
// threads behind a token (tfgpu_transformation_push_async / tfgpu_wait), so that the parsequeue's goroutines
// (pkg/parsequeue/parsequeue.go:57-154) submit and wait instead of each pinning an OS thread to a device lane.
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <thread>

#include "tf_plan.hpp"

using namespace tf;
namespace tf {
std::unique_ptr<tfgpu_dbatch> gather_rows(const tfgpu_dbatch &in, const Buf &sel, int64_t m);  // tf_transform.hip
void plan_result_columns(const tfgpu_plan &p, std::vector<SchemaCol> &cols);                    // tf_api.hip
}

#define TF_API_BEGIN try {
#define TF_API_END                                                        \
  }                                                                       \
  catch (const tf::Error &e) { return tf::fail(e.code, e.what()); }       \
  catch (const std::bad_alloc &) { return tf::fail(TFGPU_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return tf::fail(TFGPU_ERR_INVALID, e.what()); }

struct tfgpu_transformation {
  std::vector<const tfgpu_plan *> transformers;  // config order, then ExtraTransformers (middlewares/transformation.go:15-22)
  std::vector<tfgpu_plan *> owned;                // plans built here from the transfer's config (tfgpu_transformation_from_config)
  std::string errors_output;                      // Transformers.ErrorsOutput.Type: "sink" | "devnull" | ""
  ~tfgpu_transformation() { for (auto *p : owned) tfgpu_plan_destroy(p); }
  std::mutex mu;                                  // u.mutex: plans are prepared under it (transformation.go:93-95)
  std::map<std::string, std::vector<int>> plan;   // (TableID, schema) -> indices of the Suitable transformers, in order
  tfgpu_transformation_stats st{};
};

// the TableSchema a run is judged by: the one the shim passed, else the batch's own (TableSchema if uploaded, else columns)
static std::vector<SchemaCol> schema_of(const tfgpu_dbatch &b, const tfgpu_schema *s) {
  std::vector<SchemaCol> out;
  if (s) {
    for (int i = 0; i < s->ncols; i++) out.push_back(SchemaCol{s->cols[i].name ? s->cols[i].name : "", s->cols[i].dtype, s->cols[i].flags});
    return out;
  }
  auto is_key = [&](const std::string &n) { for (auto &k : b.key_names) if (k == n) return true; return false; };
  if (!b.schema.empty()) for (auto &c : b.schema) out.push_back(SchemaCol{c.first, c.second, is_key(c.first) ? (uint32_t)TFGPU_COL_KEY : 0u});
  else for (auto &c : b.cols) out.push_back(SchemaCol{c.name, c.dtype, is_key(c.name) ? (uint32_t)TFGPU_COL_KEY : 0u});
  return out;
}
// what stands in for TableSchema.Hash() (table_schema.go: a hash of the marshalled columns): equal schemas, equal keys
static std::string plan_key(const std::string &ns, const std::string &table, const std::vector<SchemaCol> &cols) {
  std::string k = ns; k.push_back('\0'); k += table; k.push_back('\0');
  for (auto &c : cols) { k += c.name; k.push_back('\1'); k += std::to_string(c.dtype); k.push_back('\1'); k += std::to_string(c.flags); k.push_back('\2'); }
  return k;
}
static bool suitable(const tfgpu_plan &p, const std::string &ns, const std::string &table, const std::vector<SchemaCol> &cols) {
  std::vector<tfgpu_colschema> cs(cols.size());
  for (size_t i = 0; i < cols.size(); i++) { cs[i].name = cols[i].name.c_str(); cs[i].dtype = cols[i].dtype; cs[i].flags = cols[i].flags; cs[i].path = ""; cs[i].original_type = ""; }
  tfgpu_schema s{(int32_t)cs.size(), cs.data()};
  return plan_suitable(p, ns, table, s);
}
// AddTablePlan (transformation.go:46-85): walk the transformers, each judged against the schema its predecessors left
static const std::vector<int> &table_plan(tfgpu_transformation &t, const std::string &ns, const std::string &table, std::vector<SchemaCol> cols) {
  std::lock_guard<std::mutex> lk(t.mu);
  const std::string key = plan_key(ns, table, cols);
  auto it = t.plan.find(key);
  if (it != t.plan.end()) return it->second;
  std::vector<int> plan;
  for (size_t i = 0; i < t.transformers.size(); i++) {
    if (!suitable(*t.transformers[i], ns, table, cols)) continue;
    plan.push_back((int)i);
    plan_result_columns(*t.transformers[i], cols);
  }
  t.st.plans_built++;
  return t.plan.emplace(key, std::move(plan)).first->second;
}

struct PushOut {
  std::unique_ptr<tfgpu_dbatch> transformed;
  std::vector<std::unique_ptr<tfgpu_dbatch>> failed;  // per failing transformer, in plan order: the rows it refused, as it saw them
  std::vector<int32_t> failed_step;                  // index of that transformer in the transformation's list
  std::vector<tfgpu_row_error> errs;                 // grouped like `failed`: the k-th error of step s is row k of its batch
};
static void push_run(tfgpu_transformation &t, const tfgpu_dbatch &in, const tfgpu_schema *schema, PushOut &o) {
  std::vector<tfgpu_row_error> &errs = o.errs;
  const auto t0 = std::chrono::steady_clock::now();
  const std::vector<int> plan = table_plan(t, in.ns, in.table, schema_of(in, schema));
  std::lock_guard<std::mutex> lk(ctx().mu);
  std::unique_ptr<tfgpu_dbatch> cur = std::make_unique<tfgpu_dbatch>(in);
  std::vector<const tfgpu_plan *> chain;
  for (int pi : plan) chain.push_back(t.transformers[(size_t)pi]);
  std::vector<std::vector<int>> hopped;
  const std::vector<int> seq = chain_sequence(chain.data(), (int)chain.size(), &hopped);  // a filter_rows in front of the mask_fields it does not read
  for (size_t q = 0; q < seq.size(); q++) {  // transformation.do :252-274
    const int pi = plan[(size_t)seq[q]];
    ApplyCtx ax;
    ax.step = pi;
    for (int m : hopped[q]) mask_precheck(*chain[(size_t)m], *cur);
    std::unique_ptr<tfgpu_dbatch> next = apply_plan(*t.transformers[(size_t)pi], *cur, ax);
    if (!ax.errs.empty()) {
      std::vector<int32_t> rows(ax.errs.size());
      for (size_t k = 0; k < ax.errs.size(); k++) rows[k] = (int32_t)ax.errs[k].row;
      Buf sel = upload_small(rows.data(), rows.size() * 4);
      std::unique_ptr<tfgpu_dbatch> refused = gather_rows(*cur, sel, (int64_t)rows.size());
      for (int m : hopped[q]) { ApplyCtx mx; refused = apply_plan(*chain[(size_t)m], *refused, mx); }  // as the configured order would have handed them over: masked
      o.failed.push_back(std::move(refused));  // TransformerError.Input = the item handed to THIS transformer
      o.failed_step.push_back(pi);
      if (cur->src_row) {  // errs[].row is reported against the ORIGINAL input
        std::vector<int32_t> sr((size_t)cur->nrows);
        d2h(sr.data(), cur->src_row->p, sr.size() * 4);
        tf::sync();
        for (auto &e : ax.errs) e.row = sr[(size_t)e.row];
      }
      errs.insert(errs.end(), ax.errs.begin(), ax.errs.end());
    }
    cur = std::move(next);
  }
  o.transformed = std::move(cur);
  tf::sync();
  const int64_t nout = o.transformed ? o.transformed->nrows : 0;
  std::lock_guard<std::mutex> sk(t.mu);
  t.st.pushes++; t.st.items_in += in.nrows; t.st.items_out += nout;
  if (in.nrows > nout) t.st.dropped += in.nrows - nout;  // sta.Dropped (:150-153)
  t.st.errors += (int64_t)errs.size();                   // sta.Errors
  t.st.elapsed_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

// ---- executor: pushes run on library threads, each bound to its own device lane ------------------------------------
struct tfgpu_token {
  std::future<int> done;
  PushOut out;
  std::string error;
  hipEvent_t ready = nullptr;  // everything the submitting lane had enqueued when the job was made
};
namespace {
struct Executor {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  std::vector<std::thread> workers;
  int alive = 0, failed = 0;  // workers bound to a lane / workers whose lane could not be bound (under mu)
  bool stop = false;
  void start(int n) {
    std::lock_guard<std::mutex> lk(mu);
    while ((int)workers.size() < n) {
      const int lane = 1 + (int)workers.size();  // lane 0 stays with the caller's synchronous calls
      workers.emplace_back([this, lane] {
        const bool bound = tfgpu_lane_use(lane % tfgpu_lane_count()) == TFGPU_OK;
        { std::lock_guard<std::mutex> lk2(mu); if (bound) alive++; else failed++; }
        cv.notify_all();
        if (!bound) return;  // (a job is never left waiting for this worker: submit() refuses when nobody is alive)
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lk2(mu);
            cv.wait(lk2, [this] { return stop || !q.empty(); });
            if (stop && q.empty()) return;
            job = std::move(q.front()); q.pop_front();
          }
          job();
        }
      });
    }
  }
  // false: no worker could bind its lane — the job would never run
  bool submit(std::function<void()> f) {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [this] { return alive > 0 || failed >= (int)workers.size(); });  // until the first worker reports in
      if (alive == 0) return false;
      q.push_back(std::move(f));
    }
    cv.notify_one();
    return true;
  }
  bool empty() { std::lock_guard<std::mutex> lk(mu); return workers.empty(); }
  void shutdown() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    for (auto &w : workers) if (w.joinable()) w.join();
    std::lock_guard<std::mutex> lk(mu);
    workers.clear(); stop = false; alive = 0; failed = 0;
  }
  ~Executor() { shutdown(); }
};
Executor g_exec;
}  // namespace
namespace tf { void executor_shutdown() { g_exec.shutdown(); } }

namespace {
struct RawJson {  // just enough of a scanner to cut member names and raw value spans out of a JSON text
  const char *p, *e;
  void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
  [[noreturn]] void bad(const char *m) { throw tf::Error(TFGPU_ERR_CONFIG, std::string("transformation config: ") + m); }
  std::string str() {
    if (p >= e || *p != '"') bad("expected a string");
    std::string o; p++;
    while (p < e && *p != '"') { if (*p == '\\' && p + 1 < e) { o += p[1]; p += 2; } else o += *p++; }
    if (p >= e) bad("unterminated string");
    p++;
    return o;
  }
  std::pair<const char *, const char *> value() {  // the span of one value
    ws();
    const char *a = p;
    if (p >= e) bad("unexpected end");
    if (*p == '"') { str(); return {a, p}; }
    if (*p == '{' || *p == '[') {
      int depth = 0;
      while (p < e) {
        if (*p == '"') { str(); continue; }
        if (*p == '{' || *p == '[') depth++;
        else if (*p == '}' || *p == ']') { depth--; if (!depth) { p++; return {a, p}; } }
        p++;
      }
      bad("unbalanced brackets");
    }
    while (p < e && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\r' && *p != '\t') p++;
    return {a, p};
  }
  // members of the object at p: f(name, value span)
  template <class F> void object(F f) {
    ws();
    if (p >= e || *p != '{') bad("expected an object");
    p++; ws();
    if (p < e && *p == '}') { p++; return; }
    for (;;) {
      ws(); std::string k = str(); ws();
      if (p >= e || *p != ':') bad("expected ':'");
      p++;
      auto v = value();
      f(k, v.first, v.second);
      ws();
      if (p < e && *p == ',') { p++; continue; }
      if (p < e && *p == '}') { p++; return; }
      bad("expected ',' or '}'");
    }
  }
};
// util.Snakify (pkg/util/snaker.go:8-17): "(.)([A-Z][a-z]+)" -> "$1_$2", then "([a-z0-9])([A-Z])" -> "$1_$2", then ToLower
std::string snakify(const std::string &in) {
  auto up = [](char c) { return c >= 'A' && c <= 'Z'; };
  auto lo = [](char c) { return c >= 'a' && c <= 'z'; };
  std::string a;
  for (size_t i = 0; i < in.size();) {
    if (i + 2 < in.size() + 0 && i + 1 < in.size() && up(in[i + 1]) && i + 2 < in.size() && lo(in[i + 2])) {
      size_t k = i + 2;
      while (k < in.size() && lo(in[k])) k++;
      a += in[i]; a += '_'; a.append(in, i + 1, k - (i + 1));
      i = k;
    } else a += in[i++];
  }
  std::string b;
  for (size_t i = 0; i < a.size();) {
    if (i + 1 < a.size() && (lo(a[i]) || (a[i] >= '0' && a[i] <= '9')) && up(a[i + 1])) { b += a[i]; b += '_'; b += a[i + 1]; i += 2; }
    else b += a[i++];
  }
  for (auto &c : b) if (up(c)) c = (char)(c - 'A' + 'a');
  return b;
}
}  // namespace

extern "C" {

int tfgpu_transformation_create(tfgpu_plan *const *transformers, int n, tfgpu_transformation **out) {
  TF_API_BEGIN
  if (!out || (n > 0 && !transformers)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_transformation_create: null argument");
  auto t = std::make_unique<tfgpu_transformation>();
  for (int i = 0; i < n; i++) { if (!transformers[i]) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_transformation_create: null transformer"); t->transformers.push_back(transformers[i]); }
  *out = t.release();
  return TFGPU_OK;
  TF_API_END
}
void tfgpu_transformation_destroy(tfgpu_transformation *t) { delete t; }

// ---- the chain as middlewares.Transformation builds it (pkg/middlewares/transformation.go:12-36) from the transfer's
//      transformer.Transformers value (pkg/transformer/abstract.go:21-66): {"debugMode":…, "transformers":[{"<type>":{config},
//      "transformerId":"…"}, …], "errorsOutput":{"Type":"sink"|"devnull", …}} -------------------------------------------------

int tfgpu_transformation_from_config(const char *transformers_json, tfgpu_plan *const *extra, int n_extra, tfgpu_transformation **out) {
  TF_API_BEGIN
  if (!transformers_json || !out || (n_extra > 0 && !extra)) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_transformation_from_config: null argument");
  auto t = std::make_unique<tfgpu_transformation>();
  RawJson top{transformers_json, transformers_json + std::strlen(transformers_json)};
  top.object([&](const std::string &k, const char *a, const char *b) {
    if (k == "transformers") {
      RawJson arr{a, b};
      arr.ws();
      if (arr.p < arr.e && *arr.p == 'n') return;  // null: no transformers
      if (arr.p >= arr.e || *arr.p != '[') arr.bad("\"transformers\" is not a list");
      arr.p++; arr.ws();
      if (arr.p < arr.e && *arr.p == ']') return;
      for (;;) {
        // Transformer.Type() / Config(): the one member that is not "transformerId" (abstract.go:23-49)
        std::string type, cfg;
        arr.object([&](const std::string &name, const char *va, const char *vb) { if (name != "transformerId" && type.empty()) { type = snakify(name); cfg.assign(va, vb); } });
        if (type.empty()) arr.bad("a transformer entry without a type");
        tfgpu_plan *plan = nullptr;
        const int rc = tfgpu_plan_create(type.c_str(), cfg.c_str(), &plan);
        if (rc) throw tf::Error(rc, "unable to init: " + type + ": " + tfgpu_last_error());  // middlewares/transformation.go:18-20
        t->owned.push_back(plan);
        t->transformers.push_back(plan);
        arr.ws();
        if (arr.p < arr.e && *arr.p == ',') { arr.p++; continue; }
        if (arr.p < arr.e && *arr.p == ']') break;
        arr.bad("expected ',' or ']' in \"transformers\"");
      }
    } else if (k == "errorsOutput") {
      RawJson eo{a, b};
      eo.ws();
      if (eo.p < eo.e && *eo.p == '{') eo.object([&](const std::string &name, const char *va, const char *vb) { if (name == "Type" || name == "type") { RawJson s{va, vb}; s.ws(); if (s.p < s.e && *s.p == '"') t->errors_output = s.str(); } });
    }
  });
  for (int i = 0; i < n_extra; i++) { if (!extra[i]) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_transformation_from_config: null extra transformer"); t->transformers.push_back(extra[i]); }  // Transformation.ExtraTransformers
  *out = t.release();
  return TFGPU_OK;
  TF_API_END
}
int tfgpu_transformation_size(const tfgpu_transformation *t) { return t ? (int)t->transformers.size() : 0; }
const char *tfgpu_transformation_plan_type(const tfgpu_transformation *t, int i) { return (t && i >= 0 && i < (int)t->transformers.size()) ? tfgpu_plan_type(t->transformers[(size_t)i]) : ""; }
const char *tfgpu_transformation_errors_output(const tfgpu_transformation *t) { return t ? t->errors_output.c_str() : ""; }

int tfgpu_transformation_table_plan(tfgpu_transformation *t, const char *ns, const char *table, const tfgpu_schema *schema, int32_t *idx, int32_t cap, int32_t *n) {
  TF_API_BEGIN
  if (!t || !schema || !n) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_transformation_table_plan: null argument");
  tfgpu_dbatch none;
  const std::vector<int> &p = table_plan(*t, ns ? ns : "", table ? table : "", schema_of(none, schema));
  *n = (int32_t)p.size();
  for (int32_t i = 0; idx && i < cap && i < *n; i++) idx[i] = p[(size_t)i];
  return TFGPU_OK;
  TF_API_END
}

static void hand_over(PushOut &o, tfgpu_dbatch **transformed, tfgpu_dbatch **error_batches, int32_t *error_steps, int32_t batches_cap, int32_t *n_error_batches,
                      tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs) {
  if (nerrs) *nerrs = (int64_t)o.errs.size();
  for (int64_t k = 0; errs && k < errs_cap && k < (int64_t)o.errs.size(); k++) errs[k] = o.errs[(size_t)k];
  *transformed = o.transformed.release();
  if (n_error_batches) *n_error_batches = (int32_t)o.failed.size();
  for (int32_t g = 0; error_batches && g < batches_cap && g < (int32_t)o.failed.size(); g++) {
    error_batches[g] = o.failed[(size_t)g].release();
    if (error_steps) error_steps[g] = o.failed_step[(size_t)g];
  }
}

int tfgpu_transformation_push(tfgpu_transformation *t, const tfgpu_dbatch *in, const tfgpu_schema *schema, tfgpu_dbatch **transformed, tfgpu_dbatch **error_batches,
                              int32_t *error_steps, int32_t batches_cap, int32_t *n_error_batches, tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs) {
  TF_API_BEGIN
  tf::dense(in, true);  // its rows may still be a selection (tfgpu_dbatch::pending); ABSENT cells: the plans say which of them read such rows (apply_plan)
  if (!t || !in || !transformed) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_transformation_push: null argument");
  PushOut o;
  push_run(*t, *in, schema, o);
  hand_over(o, transformed, error_batches, error_steps, batches_cap, n_error_batches, errs, errs_cap, nerrs);
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_transformation_get_stats(tfgpu_transformation *t, tfgpu_transformation_stats *out) {
  TF_API_BEGIN
  if (!t || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_transformation_get_stats: null argument");
  std::lock_guard<std::mutex> lk(t->mu);
  *out = t->st;
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_executor_start(int workers) {
  TF_API_BEGIN
  ctx();  // fails without a device
  if (workers < 1 || workers >= tfgpu_lane_count()) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_executor_start: workers must be 1 .. lanes - 1");
  g_exec.start(workers);
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_transformation_push_async(tfgpu_transformation *t, const tfgpu_dbatch *in, const tfgpu_schema *schema, tfgpu_token **token) {
  TF_API_BEGIN
  tf::dense(in, true);  // its rows may still be a selection (tfgpu_dbatch::pending); ABSENT cells: see tfgpu_transformation_push
  if (!t || !in || !token) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_transformation_push_async: null argument");
  if (g_exec.empty()) g_exec.start(2);
  auto tok = std::make_unique<tfgpu_token>();
  TF_HIP(hipEventCreateWithFlags(&tok->ready, hipEventDisableTiming));
  TF_HIP(hipEventRecord(tok->ready, ctx().stream));
  // the schema may die with the caller's frame: keep the three fields Suitable reads
  auto cols = std::make_shared<std::vector<SchemaCol>>();
  const bool has_schema = schema != nullptr;
  if (schema) { tfgpu_dbatch none; *cols = schema_of(none, schema); }
  auto prom = std::make_shared<std::promise<int>>();
  tok->done = prom->get_future();
  tfgpu_token *raw = tok.get();
  // the job holds its own copy of the batch (the column buffers are shared_ptr: a cheap copy): the caller may free its
  // handle before tfgpu_wait
  auto batch = std::make_shared<tfgpu_dbatch>(*in);
  const bool queued = g_exec.submit([t, batch, cols, has_schema, raw, prom] {
    const tfgpu_dbatch *in = batch.get();
    int rc = TFGPU_OK;
    try {
      TF_HIP(hipStreamWaitEvent(ctx().stream, raw->ready, 0));
      std::vector<tfgpu_colschema> cs(cols->size());
      for (size_t i = 0; i < cols->size(); i++) { cs[i].name = (*cols)[i].name.c_str(); cs[i].dtype = (*cols)[i].dtype; cs[i].flags = (*cols)[i].flags; cs[i].path = ""; cs[i].original_type = ""; }
      tfgpu_schema s{(int32_t)cs.size(), cs.data()};
      push_run(*t, *in, has_schema ? &s : nullptr, raw->out);
    } catch (const tf::Error &e) { rc = e.code; raw->error = e.what(); }
    catch (const std::exception &e) { rc = TFGPU_ERR_INVALID; raw->error = e.what(); }
    prom->set_value(rc);
  });
  if (!queued) { (void)hipEventDestroy(tok->ready); return tf::fail(TFGPU_ERR_DEVICE, "tfgpu_transformation_push_async: no executor worker could bind a device lane"); }
  *token = tok.release();
  return TFGPU_OK;
  TF_API_END
}

int tfgpu_wait(tfgpu_token *token, tfgpu_dbatch **transformed, tfgpu_dbatch **error_batches, int32_t *error_steps, int32_t batches_cap, int32_t *n_error_batches,
               tfgpu_row_error *errs, int64_t errs_cap, int64_t *nerrs) {
  TF_API_BEGIN
  if (!token || !transformed) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_wait: null argument");
  std::unique_ptr<tfgpu_token> tok(token);
  const int rc = tok->done.get();
  if (tok->ready) (void)hipEventDestroy(tok->ready);
  if (rc != TFGPU_OK) return tf::fail(rc, tok->error);
  hand_over(tok->out, transformed, error_batches, error_steps, batches_cap, n_error_batches, errs, errs_cap, nerrs);
  return TFGPU_OK;
  TF_API_END
}

}  // extern "C"
