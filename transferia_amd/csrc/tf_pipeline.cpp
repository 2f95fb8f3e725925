// tf_pipeline.cpp — the two scheduling components either side of the device path, as host code behind the C ABI:
//
//   tfgpu_parsequeue_*  pkg/parsequeue/parsequeue.go:16-217 — bounded-parallel parse, in-order push, in-order ack.  Add() starts
//                       the parse of a message at once on its own thread (a goroutine there) and blocks when `parallelism`
//                       parses are in flight (parallelism - 2 buffered tasks + the one the push loop waits for + the one Add is
//                       blocked on: the reference's own arithmetic, parsequeue.go:183-199); the push loop takes the tasks in
//                       Add order, waits for each result, hands it to the sink's AsyncPush and queues the ack; the ack loop
//                       waits for each push in order and acknowledges the message.  The first parse / push / ack error
//                       cancels the queue and is what Error() reports.  The parse callback receives a SLOT in
//                       [0, parallelism): the shim maps slots onto device lanes (tfgpu_lane_use), so that parses overlap
//                       on the GPU the way the goroutines overlap on cores.
//   tfgpu_bufferer_*    pkg/middlewares/synchronizer/bufferer/{bufferer,buffer}.go — pushes are collected until the item count,
//                       the Values size, the interval since the last flush or a non-row item triggers a flush (or Close does);
//                       one flush in flight, the next one waits for it (backpressure); Flush concatenates the buffered batches
//                       ONCE (buffer.go:33-45) — on the device: tfgpu_dbatch_concat — pushes the result to the sink and
//                       answers every buffered push with the sink's error.
//
// Neither touches the device itself (the bufferer's concat is a callback-free call into tf_shard.hip, and only when it holds
// more than one batch), so their reference tests run in the CPU suite against this very code (tests/test_pipeline.py).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "tf_common.hpp"

namespace tf {

// ---- ParseQueue -------------------------------------------------------------------------------------------------------
struct ParseTask {
  uint64_t msg = 0;
  int slot = 0;
  std::mutex mu; std::condition_variable cv;
  bool done = false; int err = 0; void *parsed = nullptr;
  bool handed = false;                // push() took `parsed` (under mu): it is the sink's from then on
  std::atomic<bool> finished{false};  // the parse thread has nothing left to do: it may be joined without waiting
  std::thread th;
};
struct AckTask { uint64_t msg; uint64_t ticket; int64_t push_start_ns; };

}  // namespace tf

struct tfgpu_parsequeue {
  int parallelism = 0;
  tfgpu_pq_parse_fn parse = nullptr; tfgpu_pq_push_fn push = nullptr; tfgpu_pq_wait_fn wait = nullptr; tfgpu_pq_ack_fn ack = nullptr;
  tfgpu_pq_release_fn release = nullptr;  // what becomes of a parse result nobody pushes (a cancelled queue): tfgpu_parsequeue_set_release
  void *user = nullptr;
  std::mutex mu; std::condition_variable cv;
  bool cancelled = false;
  std::deque<std::shared_ptr<tf::ParseTask>> push_q;   // capacity parallelism - 2
  std::deque<tf::AckTask> ack_q;
  std::vector<int> free_slots;
  std::deque<std::shared_ptr<tf::ParseTask>> all;       // tasks whose threads are not joined yet (reaped in Add order as they finish)
  int first_code = 0; std::string first_err;
  std::thread push_th, ack_th;

  void fail(int code, const std::string &m) {
    std::lock_guard<std::mutex> lk(mu);
    if (!first_code) { first_code = code ? code : TFGPU_ERR_INVALID; first_err = m; }
    cancelled = true;
    cv.notify_all();
  }
};

namespace tf {

static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void pq_push_loop(tfgpu_parsequeue *q) {
  for (;;) {
    std::shared_ptr<ParseTask> t;
    {
      std::unique_lock<std::mutex> lk(q->mu);
      q->cv.wait(lk, [&] { return q->cancelled || !q->push_q.empty(); });
      if (q->cancelled) return;
      t = q->push_q.front(); q->push_q.pop_front();
      q->cv.notify_all();  // a blocked Add may go on
    }
    {  // wait for the parse result — or the cancellation
      std::unique_lock<std::mutex> lk(t->mu);
      while (!t->done) {
        t->cv.wait_for(lk, std::chrono::milliseconds(20));
        if (!t->done) { std::lock_guard<std::mutex> g(q->mu); if (q->cancelled) return; }
      }
    }
    if (t->err) { q->fail(t->err, "parsing error: message " + std::to_string(t->msg)); return; }
    uint64_t ticket = 0;
    const int64_t st = now_ns();
    { std::lock_guard<std::mutex> lk(t->mu); t->handed = true; }
    const int rc = q->push(q->user, t->parsed, &ticket);
    if (rc) { q->fail(rc, "push error: message " + std::to_string(t->msg)); return; }
    std::lock_guard<std::mutex> lk(q->mu);
    if (q->cancelled) return;
    q->ack_q.push_back(AckTask{t->msg, ticket, st});
    q->cv.notify_all();
  }
}
static void pq_ack_loop(tfgpu_parsequeue *q) {
  for (;;) {
    AckTask a;
    {
      std::unique_lock<std::mutex> lk(q->mu);
      q->cv.wait(lk, [&] { return q->cancelled || !q->ack_q.empty(); });
      if (q->cancelled) return;
      a = q->ack_q.front(); q->ack_q.pop_front();
    }
    // the sink's answer to that push: polled so that a push that never finishes does not outlive Close (TestSinkNotBlocking)
    int rc = TFGPU_PQ_PENDING;
    while (rc == TFGPU_PQ_PENDING) {
      rc = q->wait(q->user, a.ticket, 20);
      if (rc == TFGPU_PQ_PENDING) { std::lock_guard<std::mutex> lk(q->mu); if (q->cancelled) return; }
    }
    if (rc) { q->fail(rc, "push error: message " + std::to_string(a.msg)); return; }
    const int ar = q->ack(q->user, a.msg, a.push_start_ns);
    if (ar) { q->fail(ar, "ack error: message " + std::to_string(a.msg)); return; }
  }
}

// ---- Bufferer ---------------------------------------------------------------------------------------------------------
struct BufTicket {
  std::mutex mu; std::condition_variable cv;
  bool done = false; int err = 0;
  void finish(int e) { std::lock_guard<std::mutex> lk(mu); done = true; err = e; cv.notify_all(); }
};
struct Buffer {
  std::vector<const tfgpu_dbatch *> batches;
  std::vector<int64_t> meta_rows;  // per batch: rows of the source batch its src_row counts in, or -1 (unknown)
  int64_t rows = 0; uint64_t values_size = 0;
  std::vector<std::shared_ptr<BufTicket>> tickets;
};

}  // namespace tf

struct tfgpu_bufferer {
  int64_t trig_count = 0; uint64_t trig_size = 0; int64_t trig_interval_ms = 0;
  int concat_on_device = 1;
  tfgpu_buf_flush_fn flush_fn = nullptr; void *user = nullptr;
  std::mutex mu; std::condition_variable cv;
  bool closed = false, closing = false;
  // the input "channel" (unbuffered: a push returns once run() has taken the item)
  struct Input { const tfgpu_dbatch *b; int64_t rows; uint64_t size; int non_row; int64_t meta_rows; std::shared_ptr<tf::BufTicket> t; };
  std::deque<Input> in;
  uint64_t taken = 0, offered = 0;
  std::thread run_th;
  // flush state (owned by run())
  std::unique_ptr<tf::Buffer> buf;
  std::thread flush_th; bool flush_active = false;
  std::chrono::steady_clock::time_point timer_at; bool timer_fired = true, timer_ticking = false;
  tfgpu_bufferer_stats stats{};
  // tickets whose answer has not been fetched yet (an answered and fetched ticket is gone: a replication pushes for days)
  std::unordered_map<uint64_t, std::shared_ptr<tf::BufTicket>> tickets;
  uint64_t next_ticket = 0;
};

namespace tf {

static void buf_flush(tfgpu_bufferer *b) {  // bufferer.flush: wait for the flush in flight, swap the buffer, start the next one
  b->stats.flush_all++;
  if (b->flush_th.joinable()) b->flush_th.join();
  std::unique_ptr<Buffer> todo = std::move(b->buf);
  b->buf = std::make_unique<Buffer>();
  b->flush_th = std::thread([b, t = std::shared_ptr<Buffer>(todo.release())] {
    int err = 0;
    if (t->rows > 0 || !t->batches.empty()) {
      const tfgpu_dbatch *one = nullptr; tfgpu_dbatch *merged = nullptr;
      if (t->batches.size() == 1 || !b->concat_on_device) one = t->batches.size() == 1 ? t->batches[0] : nullptr;  // "to not copy changeitems"
      else if (t->batches.size() > 1) {
        // src_row of the merged batch: with every part's source extent known the parts' source rows line up behind one another
        // (the sink concatenates its row metas in the same order); otherwise a merged src_row would collide across parts and is dropped
        bool known = true;
        std::vector<int64_t> base(t->batches.size(), 0);
        for (size_t g = 0; g < t->batches.size(); g++) { if (t->meta_rows[g] < 0) known = false; if (g + 1 < t->batches.size()) base[g + 1] = base[g] + std::max<int64_t>(t->meta_rows[g], 0); }
        err = tfgpu_dbatch_concat(t->batches.data(), (int)t->batches.size(), known ? base.data() : nullptr, &merged);  // the single concat copy (buffer.go:38-45)
        if (!err && !known) tf::dbatch_drop_src_row(merged);
        one = merged;
      }
      if (!err) err = b->flush_fn(b->user, one, t->batches.data(), (int)t->batches.size(), t->rows, t->values_size);
      if (merged) tfgpu_dbatch_free(merged);
    }
    for (auto &k : t->tickets) k->finish(err);
  });
  // the timer restarts at the START of a flush (bufferer.go:246-248)
  if (b->trig_interval_ms > 0) { b->timer_at = std::chrono::steady_clock::now() + std::chrono::milliseconds(b->trig_interval_ms); b->timer_ticking = true; b->timer_fired = false; }
  else { b->timer_fired = true; b->timer_ticking = false; }
}

static void buf_run(tfgpu_bufferer *b) {
  b->buf = std::make_unique<Buffer>();
  // the first push must happen AFTER the interval passes (bufferer.go:181-182)
  if (b->trig_interval_ms > 0) { b->timer_at = std::chrono::steady_clock::now() + std::chrono::milliseconds(b->trig_interval_ms); b->timer_ticking = true; b->timer_fired = false; }
  for (;;) {
    tfgpu_bufferer::Input it{};
    bool have = false, closing = false, tick = false;
    {
      std::unique_lock<std::mutex> lk(b->mu);
      auto pred = [&] { return !b->in.empty() || b->closing; };
      if (b->timer_ticking) {
        if (!b->cv.wait_until(lk, b->timer_at, pred)) tick = true;
      } else b->cv.wait(lk, pred);
      if (!b->in.empty()) { it = b->in.front(); b->in.pop_front(); b->taken++; have = true; b->cv.notify_all(); }
      else if (b->closing) closing = true;
    }
    if (tick && !have && !closing) {  // case <-Timer.C()
      b->timer_ticking = false; b->timer_fired = true;
      if (b->buf->rows > 0) { b->stats.flush_on_interval++; buf_flush(b); }
      continue;
    }
    if (b->timer_ticking && std::chrono::steady_clock::now() >= b->timer_at) { b->timer_ticking = false; b->timer_fired = true; }
    if (closing && !have) {  // Close: flush what is left and wait for it
      b->stats.flush_on_non_row++;
      buf_flush(b);
      if (b->flush_th.joinable()) b->flush_th.join();
      return;
    }
    b->buf->batches.push_back(it.b); b->buf->meta_rows.push_back(it.meta_rows); b->buf->rows += it.rows; b->buf->values_size += it.size; b->buf->tickets.push_back(it.t);
    if (b->trig_count > 0 && b->buf->rows >= b->trig_count) { b->stats.flush_on_count++; buf_flush(b); continue; }
    if (b->trig_size > 0 && b->buf->values_size >= b->trig_size) { b->stats.flush_on_size++; buf_flush(b); continue; }
    if (b->trig_interval_ms > 0 && b->timer_fired) { b->stats.flush_on_interval++; buf_flush(b); continue; }
    if (it.non_row) { b->stats.flush_on_non_row++; buf_flush(b); continue; }
  }
}

}  // namespace tf

using namespace tf;

extern "C" {

int tfgpu_parsequeue_create(int parallelism, tfgpu_pq_parse_fn parse, tfgpu_pq_push_fn push, tfgpu_pq_wait_fn wait, tfgpu_pq_ack_fn ack, void *user, tfgpu_parsequeue **out) {
  if (!parse || !push || !wait || !ack || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parsequeue_create: null argument");
  if (parallelism == 0) parallelism = 10;  // DefaultParallelism
  if (parallelism < 2) parallelism = 2;
  auto q = std::make_unique<tfgpu_parsequeue>();
  q->parallelism = parallelism; q->parse = parse; q->push = push; q->wait = wait; q->ack = ack; q->user = user;
  for (int s = parallelism - 1; s >= 0; s--) q->free_slots.push_back(s);
  q->push_th = std::thread(pq_push_loop, q.get());
  q->ack_th = std::thread(pq_ack_loop, q.get());
  *out = q.release();
  return TFGPU_OK;
}

int tfgpu_parsequeue_add(tfgpu_parsequeue *q, uint64_t msg) {
  if (!q) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parsequeue_add: null queue");
  auto t = std::make_shared<ParseTask>();
  t->msg = msg;
  {
    std::unique_lock<std::mutex> lk(q->mu);
    if (q->cancelled) return tf::fail(TFGPU_ERR_INVALID, "parse queue is already closed");
    // makeParseTask: the parse starts NOW (a slot is free whenever fewer than `parallelism` parses are unfinished or unconsumed)
    q->cv.wait(lk, [&] { return q->cancelled || !q->free_slots.empty(); });
    if (q->cancelled) return tf::fail(TFGPU_ERR_INVALID, "parse queue failed on sending parse task");
    t->slot = q->free_slots.back(); q->free_slots.pop_back();
    while (!q->all.empty() && q->all.front()->finished.load()) { q->all.front()->th.join(); q->all.pop_front(); }
    // the thread exists before the task is published (a Close on another thread must find it joinable), and it is created
    // while mu is held: its body takes mu only after the parse
    t->th = std::thread([q, t] {
      void *parsed = nullptr;
      const int rc = q->parse(q->user, t->msg, t->slot, &parsed);
      { std::lock_guard<std::mutex> lk(t->mu); t->done = true; t->err = rc; t->parsed = parsed; t->cv.notify_all(); }
      { std::lock_guard<std::mutex> lk(q->mu); q->free_slots.push_back(t->slot); q->cv.notify_all(); }
      t->finished.store(true);
    });
    q->all.push_back(t);
  }
  std::unique_lock<std::mutex> lk(q->mu);
  // the buffered channel of parallelism - 2 tasks: Add blocks here while it is full
  q->cv.wait(lk, [&] { return q->cancelled || (int)q->push_q.size() < std::max(q->parallelism - 2, 1); });
  if (q->cancelled) return tf::fail(TFGPU_ERR_INVALID, "parse queue failed on sending parse task");
  q->push_q.push_back(t);
  q->cv.notify_all();
  return TFGPU_OK;
}

int tfgpu_parsequeue_error(tfgpu_parsequeue *q, char *msg, size_t cap) {
  if (!q) return TFGPU_ERR_INVALID;
  std::lock_guard<std::mutex> lk(q->mu);
  if (msg && cap) { const std::string m = q->first_code ? "parse queue: " + q->first_err : ""; std::snprintf(msg, cap, "%s", m.c_str()); }
  return q->first_code;
}

int tfgpu_parsequeue_close(tfgpu_parsequeue *q) {
  if (!q) return TFGPU_OK;
  { std::lock_guard<std::mutex> lk(q->mu); q->cancelled = true; q->cv.notify_all(); }
  if (q->push_th.joinable()) q->push_th.join();
  if (q->ack_th.joinable()) q->ack_th.join();
  std::deque<std::shared_ptr<ParseTask>> all;
  { std::lock_guard<std::mutex> lk(q->mu); all.swap(q->all); }
  for (auto &t : all) if (t->th.joinable()) t->th.join();
  // results of parses that nobody pushed (the queue was cancelled, or closed with work in flight): theirs to release
  std::deque<std::shared_ptr<ParseTask>> pending;
  { std::lock_guard<std::mutex> lk(q->mu); pending.swap(q->push_q); }
  for (auto &t : all) { bool found = false; for (auto &x : pending) if (x == t) found = true; if (!found) pending.push_back(t); }
  for (auto &t : pending) {
    std::lock_guard<std::mutex> lk(t->mu);
    if (t->done && !t->handed && t->parsed && !t->err && q->release) q->release(q->user, t->parsed);
    t->handed = true;
  }
  return TFGPU_OK;
}
int tfgpu_parsequeue_set_release(tfgpu_parsequeue *q, tfgpu_pq_release_fn release) {
  if (!q) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_parsequeue_set_release: null queue");
  std::lock_guard<std::mutex> lk(q->mu);
  q->release = release;
  return TFGPU_OK;
}
void tfgpu_parsequeue_destroy(tfgpu_parsequeue *q) { if (q) { tfgpu_parsequeue_close(q); delete q; } }

static constexpr uint64_t BUF_TICKET_DONE = 1ull << 63;  // ids of pushes answered at creation: the low 32 bits are the answer

int tfgpu_bufferer_create(int64_t trigging_count, uint64_t trigging_size, int64_t trigging_interval_ms, int concat_on_device, tfgpu_buf_flush_fn flush, void *user, tfgpu_bufferer **out) {
  if (!flush || !out) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_bufferer_create: null argument");
  auto b = std::make_unique<tfgpu_bufferer>();
  b->trig_count = trigging_count; b->trig_size = trigging_size; b->trig_interval_ms = trigging_interval_ms; b->concat_on_device = concat_on_device;
  b->flush_fn = flush; b->user = user;
  b->run_th = std::thread(buf_run, b.get());
  *out = b.release();
  return TFGPU_OK;
}

int tfgpu_bufferer_async_push_meta(tfgpu_bufferer *b, const tfgpu_dbatch *batch, int64_t nrows, uint64_t values_size, int has_non_row_item, int64_t meta_rows, uint64_t *ticket) {
  if (!b || !ticket) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_bufferer_async_push: null argument");
  std::unique_lock<std::mutex> lk(b->mu);
  // A push that is answered at once (into a closed bufferer: AsyncPushConcurrencyErr rides on the ticket; or an empty batch) gets a
  // reserved id that carries its answer — nothing is registered, so a caller that never waits on such a ticket leaks nothing.
  if (b->closed || b->closing) { *ticket = BUF_TICKET_DONE | (uint32_t)TFGPU_ERR_INVALID; return TFGPU_OK; }
  if (nrows == 0) { *ticket = BUF_TICKET_DONE | (uint32_t)TFGPU_OK; return TFGPU_OK; }
  auto t = std::make_shared<BufTicket>();
  const uint64_t id = ++b->next_ticket;
  b->tickets.emplace(id, t);
  *ticket = id;
  b->in.push_back({batch, nrows, values_size, has_non_row_item, meta_rows, t});
  const uint64_t mine = ++b->offered;
  b->cv.notify_all();
  b->cv.wait(lk, [&] { return b->taken >= mine; });  // an unbuffered channel: the push returns when run() has the item
  return TFGPU_OK;
}
int tfgpu_bufferer_async_push(tfgpu_bufferer *b, const tfgpu_dbatch *batch, int64_t nrows, uint64_t values_size, int has_non_row_item, uint64_t *ticket) {
  return tfgpu_bufferer_async_push_meta(b, batch, nrows, values_size, has_non_row_item, -1, ticket);
}

int tfgpu_bufferer_wait(tfgpu_bufferer *b, uint64_t ticket, int64_t timeout_ms) {
  if (!b) return TFGPU_ERR_INVALID;
  if (ticket & BUF_TICKET_DONE) return (int)(int32_t)(uint32_t)ticket;  // answered when it was pushed
  std::shared_ptr<BufTicket> t;
  {
    std::lock_guard<std::mutex> lk(b->mu);
    auto it = b->tickets.find(ticket);
    if (it == b->tickets.end()) return tf::fail(TFGPU_ERR_INVALID, "tfgpu_bufferer_wait: unknown ticket (or its answer was fetched already)");
    t = it->second;
  }
  int err;
  {
    std::unique_lock<std::mutex> lk(t->mu);
    if (timeout_ms < 0) t->cv.wait(lk, [&] { return t->done; });
    else if (!t->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return t->done; })) return TFGPU_PQ_PENDING;
    err = t->err;
  }
  { std::lock_guard<std::mutex> lk(b->mu); b->tickets.erase(ticket); }  // a final answer is given once
  return err;
}

int tfgpu_bufferer_get_stats(tfgpu_bufferer *b, tfgpu_bufferer_stats *out) {
  if (!b || !out) return TFGPU_ERR_INVALID;
  *out = b->stats;  // (counters are written by run() only; a torn read of a counter is not an error for a metric)
  return TFGPU_OK;
}

int tfgpu_bufferer_close(tfgpu_bufferer *b) {
  if (!b) return TFGPU_OK;
  {
    std::lock_guard<std::mutex> lk(b->mu);
    if (b->closed) return TFGPU_OK;
    b->closing = true;
    b->cv.notify_all();
  }
  if (b->run_th.joinable()) b->run_th.join();
  std::lock_guard<std::mutex> lk(b->mu);
  b->closed = true;
  return TFGPU_OK;
}
void tfgpu_bufferer_destroy(tfgpu_bufferer *b) { if (b) { tfgpu_bufferer_close(b); delete b; } }

}  // extern "C"
