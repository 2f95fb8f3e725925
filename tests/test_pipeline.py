"""ParseQueue and Bufferer (tf_pipeline.cpp) against the reference's own tests — host scheduling code behind the C ABI, no device
needed: parsequeue_test.go (TestSinkNotBlocking, TestParseErrNotBlocking, TestAckOrder, TestGracefullyShutdown,
TestRandomParseDelay) and bufferer_test.go (all eight), with the same counts, orders and timings; callbacks are ctypes
functions, the "batches" the bufferer holds are opaque handles here (concat_on_device = 0: the sink sees the parts).  The device
half of the bufferer — ONE tfgpu_dbatch_concat per flush — is a gpu-marked test at the end."""
import ctypes as C
import random
import threading
import time

import numpy as np
import pytest

from transferia_amd import abi, lib

PARSE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p))
PUSH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64))
WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_int64)
ACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_int64)
FLUSH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_uint64)
PENDING = -1


class Queue:
    def __init__(self, parallelism, parse, push, wait, ack):
        self.L = lib.load()
        self.L.tfgpu_parsequeue_add.argtypes = [C.c_void_p, C.c_uint64]
        self.L.tfgpu_parsequeue_error.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        self.L.tfgpu_parsequeue_close.argtypes = [C.c_void_p]
        self.L.tfgpu_parsequeue_destroy.argtypes = [C.c_void_p]
        self.cb = (PARSE(parse), PUSH(push), WAIT(wait), ACK(ack))  # kept alive
        self.h = C.c_void_p()
        assert self.L.tfgpu_parsequeue_create(parallelism, *self.cb, None, C.byref(self.h)) == 0

    def add(self, msg):
        return self.L.tfgpu_parsequeue_add(self.h, msg)

    def error(self):
        buf = C.create_string_buffer(256)
        return self.L.tfgpu_parsequeue_error(self.h, buf, 256), buf.value.decode()

    def close(self):
        self.L.tfgpu_parsequeue_close(self.h)

    def __del__(self):
        self.L.tfgpu_parsequeue_destroy(self.h)


def wait_for(cond, seconds=1.0):
    t0 = time.time()
    while time.time() - t0 < seconds and not cond():
        time.sleep(0.001)


def test_parsequeue_sink_not_blocking():
    """TestSinkNotBlocking: pushes that never finish do not stop parsing and pushing; nothing is acked; Close returns"""
    cnt = {"parse": 0, "push": 0, "ack": 0}
    mu = threading.Lock()

    def bump(k):
        with mu:
            cnt[k] += 1
    q = Queue(10, lambda u, m, s, out: bump("parse") or 0, lambda u, p, t: bump("push") or 0, lambda u, t, ms: (time.sleep(ms / 1000), PENDING)[1],
              lambda u, m, st: bump("ack") or 0)
    for i in range(1000):
        assert q.add(i) == 0
    wait_for(lambda: cnt["push"] == 1000)
    assert (cnt["parse"], cnt["push"], cnt["ack"]) == (1000, 1000, 0)
    assert q.error()[0] == 0
    q.close()
    assert q.error()[0] == 0


def test_parsequeue_parse_error():
    """TestParseErrNotBlocking: a parse error cancels the queue — no push, no ack, Error() carries it, later Adds fail"""
    cnt = {"parse": 0, "push": 0, "ack": 0}
    mu = threading.Lock()

    def bump(k, rc=0):
        with mu:
            cnt[k] += 1
        return rc
    q = Queue(10, lambda u, m, s, out: bump("parse", 77), lambda u, p, t: bump("push"), lambda u, t, ms: 0, lambda u, m, st: bump("ack"))
    rcs = [q.add(i) for i in range(1000)]
    wait_for(lambda: q.error()[0] != 0)
    assert cnt["parse"] <= 1000 and cnt["push"] == 0 and cnt["ack"] == 0
    code, msg = q.error()
    assert code == 77 and msg.startswith("parse queue: parsing error")
    assert rcs[-1] != 0
    q.close()
    assert q.error()[0] == 77


def test_parsequeue_ack_order_and_parallelism_bound():
    """TestAckOrder: exactly `parallelism` parses run while the first ones block; acks come in Add order whatever order the parses
    finish in"""
    par = 10
    gates = {i: threading.Event() for i in range(par * 10)}
    res, mu = [], threading.Lock()
    inflight = [0]
    slots_seen = set()

    def parse(u, m, slot, out):
        with mu:
            inflight[0] += 1
            assert 0 <= slot < par and slot not in slots_seen
            slots_seen.add(slot)
        gates[m].wait()
        with mu:
            inflight[0] -= 1
            slots_seen.discard(slot)
        return 0

    def ack(u, m, st):
        with mu:
            res.append(int(m))
        return 0
    q = Queue(par, parse, lambda u, p, t: 0, lambda u, t, ms: 0, ack)
    th = threading.Thread(target=lambda: [q.add(i) for i in range(par * 10)])
    th.start()
    time.sleep(1.0)
    with mu:
        assert inflight[0] == par
        assert res == []
    order = list(gates)
    random.Random(1).shuffle(order)   # Go ranges over a map: an arbitrary order
    for i in order:
        gates[i].set()
    th.join()
    wait_for(lambda: len(res) == par * 10, 5.0)
    assert res == list(range(par * 10))
    q.close()
    assert q.error()[0] == 0


def test_parsequeue_graceful_shutdown():
    """TestGracefullyShutdown: Close while a producer keeps adding; no error"""
    res, mu = [], threading.Lock()
    q = Queue(5, lambda u, m, s, out: (time.sleep(0.001), 0)[1], lambda u, p, t: (time.sleep(0.002), 0)[1], lambda u, t, ms: 0,
              lambda u, m, st: (mu.acquire(), res.append(int(m)), mu.release(), 0)[3])
    stop = []

    def producer():
        i = 0
        while q.add(i) == 0:
            i += 1
            time.sleep(0.01)
        stop.append(i)
    th = threading.Thread(target=producer)
    th.start()
    time.sleep(1.0)
    q.close()
    th.join(5)
    assert stop and q.error()[0] == 0
    assert res == list(range(len(res))) and len(res) >= 20


def test_parsequeue_random_parse_delay():
    """TestRandomParseDelay: never more than `parallelism` parses at once, pushes and acks strictly in Add order"""
    par, n = 10, 100
    mu = threading.Lock()
    state = {"counter": 0, "peak": 0, "push": 0, "ack": 0, "bad": 0}
    rng = random.Random(7)
    payload = {}

    def parse(u, m, slot, out):
        with mu:
            state["counter"] += 1
            state["peak"] = max(state["peak"], state["counter"])
        time.sleep(rng.randrange(10) / 1000)
        with mu:
            state["counter"] -= 1
            payload[int(m)] = int(m)
        out[0] = int(m) + 1   # the parsed "batch": its own number (+1: not NULL)
        return 0

    def push(u, p, t):
        with mu:
            if (p or 0) - 1 != state["push"]:
                state["bad"] += 1
            state["push"] += 1
        t[0] = p or 0
        return 0

    def ack(u, m, st):
        with mu:
            if int(m) != state["ack"]:
                state["bad"] += 1
            state["ack"] += 1
        return 0
    q = Queue(par, parse, push, lambda u, t, ms: 0, ack)
    for i in range(n):
        assert q.add(i) == 0
    wait_for(lambda: state["ack"] == n, 10.0)
    q.close()
    assert state["ack"] == n and state["push"] == n and state["bad"] == 0 and state["peak"] <= par
    assert q.error()[0] == 0


# ---- bufferer ----------------------------------------------------------------------------------------------------------
INIT, DONE, ROW = "init_load", "done_load", "row"


class MockSinker:
    """NewMockSinker(errors): Push returns the next error of the list (then nil) and keeps the items of successful pushes"""

    def __init__(self, errors=(), delay=None):
        self.items, self.errors, self.i = [], list(errors), 0
        self.mu = threading.Lock()
        self.delay = delay   # an Event the push waits for first (a sink that takes its time)

    def push(self, items):
        if self.delay is not None:
            self.delay.wait(5)
        with self.mu:
            err = self.errors[self.i] if self.i < len(self.errors) else 0
            self.i += 1
            if not err:
                self.items += items
            return err


class Bufferer:
    def __init__(self, count, size, interval_ms, sink):
        self.L = lib.load()
        self.L.tfgpu_bufferer_async_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
        self.L.tfgpu_bufferer_wait.argtypes = [C.c_void_p, C.c_uint64, C.c_int64]
        self.L.tfgpu_bufferer_close.argtypes = [C.c_void_p]
        self.L.tfgpu_bufferer_destroy.argtypes = [C.c_void_p]
        self.L.tfgpu_bufferer_create.argtypes = [C.c_int64, C.c_uint64, C.c_int64, C.c_int, FLUSH, C.c_void_p, C.POINTER(C.c_void_p)]
        self.sink, self.held = sink, {}

        def flush(user, merged, parts, nparts, nrows, size):
            items = []
            for k in range(nparts):
                items += self.held[parts[k]]
            assert len(items) == nrows
            return self.sink.push(items)
        self.cb = FLUSH(flush)
        self.h = C.c_void_p()
        assert self.L.tfgpu_bufferer_create(count, size, interval_ms, 0, self.cb, None, C.byref(self.h)) == 0

    def async_push(self, items, sizes=None):
        handle = len(self.held) + 1   # an opaque, non-NULL "batch"
        self.held[handle] = list(items)
        t = C.c_uint64(0)
        non_row = any(i != ROW for i in items)
        assert self.L.tfgpu_bufferer_async_push(self.h, handle, len(items), sum(sizes or []), int(non_row), C.byref(t)) == 0
        return t.value

    def wait(self, ticket, timeout_ms=-1):
        return self.L.tfgpu_bufferer_wait(self.h, ticket, timeout_ms)

    def push(self, items, sizes=None):
        return self.wait(self.async_push(items, sizes))

    def close(self):
        assert self.L.tfgpu_bufferer_close(self.h) == 0

    def __del__(self):
        self.L.tfgpu_bufferer_destroy(self.h)


def test_bufferer_common():
    m = MockSinker()
    b = Bufferer(2, 0, 250, m)
    assert b.push([INIT]) == 0
    assert b.push([ROW] * 3) == 0
    assert len(m.items) == 4
    assert b.push([DONE]) == 0
    assert len(m.items) == 5
    b.close()
    assert len(m.items) == 5


def test_bufferer_periodic_with_error():
    m = MockSinker([0, 0, 9])
    b = Bufferer(5, 0, 250, m)
    assert b.push([INIT]) == 0
    assert b.push([ROW] * 3) == 0
    assert len(m.items) == 4
    assert b.push([DONE]) == 9
    assert len(m.items) == 4
    b.close()
    assert len(m.items) == 4


def test_bufferer_with_inflight():
    m = MockSinker()
    b = Bufferer(10, 0, 2000, m)
    assert b.push([INIT]) == 0
    ts = [b.async_push([ROW] * 2) for _ in range(4)]
    for k in (3, 0, 1, 2):
        assert b.wait(ts[k]) == 0
    assert len(m.items) == 9
    assert b.push([ROW] * 10) == 0
    assert len(m.items) == 19
    assert b.push([DONE]) == 0
    assert len(m.items) == 20
    b.close()
    assert len(m.items) == 20


def test_bufferer_with_inflight_error_no_init():
    m = MockSinker([0, 0, 9])
    b = Bufferer(3, 0, 2000, m)
    ts = [b.async_push([ROW] * 2), b.async_push([ROW] * 2)]
    assert [b.wait(t) for t in ts] == [0, 0]
    assert len(m.items) == 4
    ts = [b.async_push([ROW]) for _ in range(3)]
    assert [b.wait(t) for t in ts] == [0, 0, 0]
    assert len(m.items) == 7
    ts = [b.async_push([ROW]) for _ in range(2)]
    assert [b.wait(t) for t in ts] == [9, 9]   # flushed by the interval, two seconds after the previous flush started
    assert len(m.items) == 7
    b.close()
    assert len(m.items) == 7


def test_bufferer_with_interval_only():
    m = MockSinker()
    b = Bufferer(0, 0, 250, m)
    assert b.push([INIT]) == 0
    assert b.push([ROW] * 3) == 0
    assert b.push([ROW] * 3) == 0
    time.sleep(1.0)
    assert b.push([ROW] * 3) == 0
    assert len(m.items) == 10
    assert b.push([DONE]) == 0
    assert len(m.items) == 11
    b.close()
    assert len(m.items) == 11


def test_bufferer_with_count_only():
    m = MockSinker()
    b = Bufferer(2, 0, 0, m)
    assert b.push([INIT]) == 0
    assert b.push([ROW] * 2) == 0
    assert b.push([ROW] * 3) == 0
    time.sleep(0.3)
    assert b.push([ROW] * 4) == 0
    assert len(m.items) == 10
    assert b.push([DONE]) == 0
    assert len(m.items) == 11
    b.close()
    assert len(m.items) == 11


def test_bufferer_with_size_only():
    K = 1024
    m = MockSinker()
    b = Bufferer(0, 2 * K * K, 0, m)
    assert b.push([INIT]) == 0
    assert len(m.items) == 1
    t1 = b.async_push([ROW] * 2, [512 * K] * 2)
    assert b.wait(t1, 50) == PENDING and len(m.items) == 1     # buffered, not pushed
    t2 = b.async_push([ROW] * 2, [512 * K] * 2)
    assert b.wait(t1) == 0 and b.wait(t2) == 0
    assert len(m.items) == 5
    t3 = b.async_push([ROW] * 2, [256 * K] * 2)
    assert b.wait(t3, 50) == PENDING and len(m.items) == 5
    b.close()
    assert len(m.items) == 7 and b.wait(t3) == 0


def test_bufferer_without_any_triggers():
    m = MockSinker()
    b = Bufferer(0, 0, 0, m)
    assert b.push([INIT]) == 0
    assert len(m.items) == 1
    ts = [b.async_push([ROW] * 2), b.async_push([ROW] * 3)]
    time.sleep(0.3)
    ts.append(b.async_push([ROW] * 4))
    assert len(m.items) == 1
    b.close()
    assert len(m.items) == 10 and [b.wait(t) for t in ts] == [0, 0, 0]
    st = (C.c_int64 * 5)()
    b.L.tfgpu_bufferer_get_stats(b.h, st)
    assert list(st) == [2, 0, 0, 0, 2]   # the InitTableLoad push and Close, both counted as FlushOnNonRow like the reference


@pytest.mark.gpu
def test_bufferer_flush_is_one_device_concat():
    """the device half: batches buffered until the count trigger leave as ONE batch = tfgpu_dbatch_concat of them, rows in push order"""
    lib.init()
    L = lib.load()
    got = []

    def flush(user, merged, parts, nparts, nrows, size):
        view = lib.DeviceBatch(C.c_void_p(merged))
        b = view.download()
        view._h = None   # the handle stays the bufferer's: the wrapper must not free it
        got.append((nparts, nrows, [int(x) for x in b.cols[0].values]))
        return 0
    cb = FLUSH(flush)
    h = C.c_void_p()
    L.tfgpu_bufferer_create.argtypes = [C.c_int64, C.c_uint64, C.c_int64, C.c_int, FLUSH, C.c_void_p, C.POINTER(C.c_void_p)]
    L.tfgpu_bufferer_async_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    L.tfgpu_bufferer_wait.argtypes = [C.c_void_p, C.c_uint64, C.c_int64]
    assert L.tfgpu_bufferer_create(10, 0, 0, 1, cb, None, C.byref(h)) == 0
    schema = abi.Schema.of([["id", "int64", True], ["s", "utf8"]])
    dbs, tickets = [], []
    for k in range(4):
        rows = [[["int64", 100 * k + i], ["string", "r%d" % (100 * k + i)]] for i in range(3)]
        db = lib.DeviceBatch.upload(abi.batch_from_rows(schema, ["id", "s"], rows, "db", "t"))
        dbs.append(db)
        t = C.c_uint64(0)
        sz = int(lib.deepsizeof(db))   # Size.Values of the batch: the Measurer's accounting, on the device
        assert L.tfgpu_bufferer_async_push(h, db._h, 3, sz, 0, C.byref(t)) == 0
        tickets.append(t.value)
    assert [L.tfgpu_bufferer_wait(h, t, -1) for t in tickets] == [0] * 4
    assert got == [(4, 12, [0, 1, 2, 100, 101, 102, 200, 201, 202, 300, 301, 302])]
    L.tfgpu_bufferer_close.argtypes = [C.c_void_p]
    L.tfgpu_bufferer_destroy.argtypes = [C.c_void_p]
    L.tfgpu_bufferer_close(h)
    L.tfgpu_bufferer_destroy(h)


def test_bufferer_ticket_is_answered_once_and_then_forgotten():
    """a replication pushes for days: a ticket whose final answer was fetched is gone (the collector keeps no history);
    TFGPU_PQ_PENDING is not a final answer and keeps the ticket"""
    release = threading.Event()
    m = MockSinker(delay=release)
    b = Bufferer(1, 0, 0, m)
    t = b.async_push([ROW])
    assert b.wait(t, 20) == PENDING and b.wait(t, 20) == PENDING   # still there
    release.set()
    assert b.wait(t) == 0
    assert b.wait(t) != 0 and b.wait(t, 10) != PENDING               # unknown by now
    tickets = [b.async_push([ROW]) for _ in range(50)]
    assert len(set(tickets)) == 50 and [b.wait(x) for x in tickets] == [0] * 50
    b.close()


def test_parsequeue_releases_results_nobody_pushes():
    """a cancelled queue drops the results of parses that were already running: they go to the release callback at Close
    (tfgpu_parsequeue_set_release), each exactly once, and never a result that push() took"""
    RELEASE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
    mu = threading.Lock()
    parsed, pushed, released = set(), set(), []

    def parse(u, m, s, out):
        time.sleep(0.002)
        with mu:
            parsed.add(int(m) + 1)
        out[0] = int(m) + 1
        return 0

    def push(u, p, t):
        with mu:
            pushed.add(int(p))
        t[0] = int(p)
        return 55 if int(p) == 6 else 0   # the sixth push fails: the queue is cancelled with parses in flight

    def rel(u, p):
        with mu:
            released.append(int(p))
    q = Queue(6, parse, push, lambda u, t, ms: 0, lambda u, m, st: 0)
    cb = RELEASE(rel)
    q.L.tfgpu_parsequeue_set_release.argtypes = [C.c_void_p, RELEASE]
    assert q.L.tfgpu_parsequeue_set_release(q.h, cb) == 0
    for i in range(200):
        if q.add(i) != 0:
            break
    wait_for(lambda: q.error()[0] != 0)
    q.close()
    assert q.error()[0] == 55
    assert len(released) == len(set(released)) and not (set(released) & pushed)
    assert set(released) | pushed == parsed   # every parse result went to the sink or to the release callback


@pytest.mark.gpu
def test_bufferer_merged_src_row_lines_up_with_the_parts_sources():
    """batches that come out of a row filter carry src_row into THEIR source; merged, part g's src_row is shifted by the source rows
    of the parts in front of it when every push states them (async_push_meta), and dropped — never colliding — when one does not"""
    lib.init()
    L = lib.load()
    got = []

    def flush(user, merged, parts, nparts, nrows, size):
        view = lib.DeviceBatch(C.c_void_p(merged))
        b = view.download()
        has_src = bool(view.view().src_row)   # (download() reads a NULL src_row as the identity: ask the view)
        view._h = None
        got.append((nparts, [int(x) for x in b.src_row] if has_src else None, [int(x) for x in b.cols[0].values]))
        return 0
    cb = FLUSH(flush)
    L.tfgpu_bufferer_create.argtypes = [C.c_int64, C.c_uint64, C.c_int64, C.c_int, FLUSH, C.c_void_p, C.POINTER(C.c_void_p)]
    L.tfgpu_bufferer_async_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    L.tfgpu_bufferer_async_push_meta.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int, C.c_int64, C.POINTER(C.c_uint64)]
    L.tfgpu_bufferer_wait.argtypes = [C.c_void_p, C.c_uint64, C.c_int64]
    L.tfgpu_bufferer_close.argtypes = [C.c_void_p]
    L.tfgpu_bufferer_destroy.argtypes = [C.c_void_p]
    schema = abi.Schema.of([["id", "int64", True], ["s", "utf8"]])
    keep = lib.Transformer("filter_rows", {"filter": "id > 1"})
    for with_meta in (True, False):
        h = C.c_void_p()
        assert L.tfgpu_bufferer_create(4, 0, 0, 1, cb, None, C.byref(h)) == 0
        held, tickets = [], []
        for k in range(2):
            rows = [[["int64", 10 * k + i], ["string", "r%d" % i]] for i in range(4)]   # ids 0..3 and 10..13: the filter drops 0 and 1 of the first
            src = lib.DeviceBatch.upload(abi.batch_from_rows(schema, ["id", "s"], rows, "db", "t"))
            res = lib.apply_chain([keep], src)
            held += [src, res.transformed]
            t = C.c_uint64(0)
            n = res.transformed.nrows
            if with_meta:
                assert L.tfgpu_bufferer_async_push_meta(h, res.transformed._h, n, 64, 0, 4, C.byref(t)) == 0
            else:
                assert L.tfgpu_bufferer_async_push(h, res.transformed._h, n, 64, 0, C.byref(t)) == 0
            tickets.append(t.value)
        L.tfgpu_bufferer_close(h)
        assert [L.tfgpu_bufferer_wait(h, t, -1) for t in tickets] == [0, 0]
        L.tfgpu_bufferer_destroy(h)
        for x in held:
            x.free()
    (n1, src1, ids1), (n2, src2, ids2) = got
    assert n1 == n2 == 2 and ids1 == ids2 == [2, 3, 10, 11, 12, 13]
    assert src1 == [2, 3, 4, 5, 6, 7]   # the second part's rows 0..3 behind the first part's four source rows
    assert src2 is None


@pytest.mark.gpu
def test_parsequeue_parses_on_device_lanes_and_pushes_in_order():
    """the queue as the shim uses it: parse = tfgpu_csv_parse on the lane of the task's slot (parses of different messages overlap
    on the device), push = the transformer chain + hand-over in Add order, ack in Add order; the rows that come out are the rows
    of the messages in the order they were added"""
    lib.init()
    schema = abi.Schema.of([["id", "int64", True, "0"], ["s", "utf8", False, "1"]])
    cs = schema.to_c()
    opts = abi.csv_options()
    nmsg, per = 24, 500
    chunks = [b"".join(b"%d,v%d\n" % (m * per + i, m * per + i) for i in range(per)) for m in range(nmsg)]
    keep = lib.Transformer("filter_rows", {"filter": "id > 99"})
    parsed, mu, pushed, acked, lanes_used = {}, threading.Lock(), [], [], set()

    def parse(u, m, slot, out):
        try:
            lane = 1 + slot % 3
            lib.lane_use(lane)
            db, consumed, errs = lib.csv_parse(opts, cs, chunks[m])
            lib.synchronize()
            with mu:
                parsed[int(m)] = db
                lanes_used.add(lane)
            out[0] = int(m) + 1
            return 0
        except Exception:  # noqa: BLE001
            return 5

    def push(u, p, t):
        m = (p or 0) - 1
        try:
            db = parsed.pop(m).to_lane(0)        # the sink's lane
            lib.lane_use(0)
            res = lib.apply_chain([keep], db).transformed.download()
            with mu:
                pushed.append([int(x) for x in res.cols[0].values])
            t[0] = m + 1
            return 0
        except Exception:  # noqa: BLE001
            return 6
    q = Queue(4, parse, push, lambda u, t, ms: 0, lambda u, m, st: (acked.append(int(m)), 0)[1])
    for m in range(nmsg):
        assert q.add(m) == 0
    wait_for(lambda: len(acked) == nmsg, 60.0)
    q.close()
    assert q.error() == (0, "")
    assert acked == list(range(nmsg))
    assert [x for rows in pushed for x in rows] == list(range(100, nmsg * per))
    assert len(lanes_used) == 3
    lib.lane_use(0)
