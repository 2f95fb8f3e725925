"""BASELINE.json's full sizes (2^20-row hits CSV; 2^20-row CDC slice) through size-independent properties: the oracle
needs ~25 s per million rows, so whole-batch comparisons are replaced by (1) sampled windows against the oracle,
(2) concatenation (parse / mask of halves = halves of the whole), (3) a predicate and its complement splitting the
batch, (4) a serialize → parse round trip, (5) Collapse of key-partitioned shards = Collapse of the whole."""
import os
import sys

import numpy as np
import pytest

from transferia_amd import abi, workload
from test_gpu_transformers import assert_batches_equal

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu
N = 1 << 20


@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


@pytest.fixture(scope="module")
def hits(tf):
    """The bench's shard, staged in HBM exactly as bench.py does, parsed once."""
    import bench
    dbuf, total, hs = bench.stage_shard(tf, workload, 0, N)
    schema, opts = workload.hits_schema(), workload.hits_csv_options()
    db, consumed, errs = tf.csv_parse(opts, schema, dbuf)
    assert consumed == total and not errs and db.nrows == N
    return dict(dbuf=dbuf, total=total, schema=schema, opts=opts, db=db, full=db.download())


def rows_slice(b: abi.Batch, a: int, e: int) -> abi.Batch:
    cols = []
    for c in b.cols:
        if c.repr in abi.VAR_REPRS:
            o0, o1 = int(c.offsets[a]), int(c.offsets[e])
            cols.append(abi.Column(c.name, c.dtype, c.repr, offsets=(c.offsets[a:e + 1] - c.offsets[a]).astype(np.uint32), data=c.data[o0:o1],
                                   validity=None if c.validity is None else c.validity[a:e]))
        else:
            cols.append(abi.Column(c.name, c.dtype, c.repr, values=c.values[a:e], nanos=None if c.nanos is None else c.nanos[a:e],
                                   validity=None if c.validity is None else c.validity[a:e]))
    return abi.Batch(cols, e - a, b.table_ns, b.table_name)


def test_sampled_windows_match_oracle(tf, oracle, hits):
    """16 windows of 2048 rows spread over the million: the oracle parses exactly those lines."""
    full = hits["full"]
    for k in range(16):
        r0 = k * (N // 16) + 37 * k
        data = workload.hits_csv(2048, row0=r0)
        ref = oracle.csv_parse(hits["opts"], hits["schema"], data, "", "")
        assert not ref.errors
        assert_batches_equal(rows_slice(full, r0, r0 + 2048), ref.batch, "window %d" % k)


def test_parse_and_mask_of_halves(tf, hits):
    """parse(first half) ++ parse(second half) == parse(all); the same for mask_field (a per-row function)."""
    import bench
    full = hits["full"]
    plan = [tf.Transformer("mask_field", {"maskFunctionHash": {"userDefinedSalt": "clickbench-salt"}, "columns": ["clientip", "title"]})]
    masked = tf.apply_chain(plan, hits["db"]).transformed.download()
    for a, e in ((0, N // 2), (N // 2, N)):
        dbuf, total, _ = bench.stage_shard(tf, workload, a, e - a)
        db, consumed, errs = tf.csv_parse(hits["opts"], hits["schema"], dbuf)
        assert consumed == total and not errs
        assert_batches_equal(db.download(), rows_slice(full, a, e), "half %d" % a)
        assert_batches_equal(tf.apply_chain(plan, db).transformed.download(), rows_slice(masked, a, e), "masked half %d" % a)
        db.free(); dbuf.free()


def test_predicate_and_complement_split_the_batch(tf, hits):
    """eventdate >= d and eventdate < d partition the rows: counts add up, src_row sets are complementary and
    ordered, and every kept row equals its source row (text columns packed straight from the CSV text)."""
    full = hits["full"]
    hi = tf.apply_chain([tf.Transformer("filter_rows", {"filter": "eventdate >= 2013-07-15"})], hits["db"]).transformed.download()
    lo = tf.apply_chain([tf.Transformer("filter_rows", {"filter": "eventdate < 2013-07-15"})], hits["db"]).transformed.download()
    assert hi.nrows + lo.nrows == N and hi.nrows == 573840  # the bench line's rows_out_per_step
    both = np.concatenate([hi.src_row, lo.src_row])
    assert np.array_equal(np.sort(both), np.arange(N, dtype=np.int32))
    assert (np.diff(hi.src_row) > 0).all() and (np.diff(lo.src_row) > 0).all()
    ed = full.col("eventdate").values
    cut = 15901 * 86400  # 2013-07-15
    assert (ed[hi.src_row] >= cut).all() and (ed[lo.src_row] < cut).all()
    for part in (hi, lo):
        for c, f in zip(part.cols, full.cols):
            if c.repr in abi.VAR_REPRS:
                ln = (f.offsets[1:] - f.offsets[:-1])[part.src_row]
                assert np.array_equal(c.offsets[1:] - c.offsets[:-1], ln), c.name
                idx = np.flatnonzero(ln)[:: max(1, part.nrows // 3000)]  # payload of ~3000 cells per column
                for k in idx:
                    s = int(part.src_row[k])
                    assert c.get_bytes(int(k)) == f.get_bytes(s), (c.name, s)
            else:
                assert np.array_equal(c.values, f.values[part.src_row]), c.name


def test_csv_round_trip(tf, hits):
    """csvSerializer text of the integer and text columns parses back to the same columns — up to the one asymmetry of
    the reference pair itself: encoding/csv quotes a field for a LEADING space only (fieldNeedsQuotes), the reader
    trims both ends of an unquoted field (reader.go:275), so a cell ending in a space comes back without it."""
    keep = [c.name for c in hits["schema"].cols if c.dtype in ("int16", "int32", "int64", "utf8")]
    proj = tf.apply_chain([tf.Transformer("filter_columns", {"columns": {"includeColumns": ["^%s$" % n for n in keep]}})], hits["db"]).transformed
    text = tf.serialize(abi.FMT_CSV, proj)
    names = proj.column_names()
    assert sorted(names) == sorted(keep)
    dt = {c.name: c.dtype for c in hits["schema"].cols}
    schema2 = abi.Schema.of([[n, dt[n], False, str(i)] for i, n in enumerate(names)])
    back, consumed, errs = tf.csv_parse(abi.csv_options(), schema2, text)
    assert consumed == text.size and not errs and back.nrows == N
    got, want = back.download(), proj.download()
    trimmed = 0
    for g, w in zip(got.cols, want.cols):
        assert g.name == w.name and g.repr == w.repr
        if g.repr not in abi.VAR_REPRS:
            assert np.array_equal(g.values, w.values), g.name
            continue
        lg, lw = g.offsets[1:] - g.offsets[:-1], w.offsets[1:] - w.offsets[:-1]
        diff = np.flatnonzero(lg != lw)
        for r in diff[:2000]:
            o = w.get_bytes(int(r))
            assert o.strip(b" ") == g.get_bytes(int(r)) and o[:1] != b" " and not any(ch in o for ch in b',"\r\n'), (g.name, int(r), o)
        trimmed += len(diff)
        same = np.flatnonzero(lg == lw)
        for r in same[:: max(1, len(same) // 4000)]:
            assert g.get_bytes(int(r)) == w.get_bytes(int(r)), (g.name, int(r))
    assert 0 < trimmed < N // 5


def test_collapse_of_key_shards(tf):
    """configs[4]'s shape: rows hash-partitioned by primary key, Collapse per shard = Collapse of the whole stream
    (no primary-key changes, so a key's chain never leaves its shard)."""
    b, schema = workload.cdc_batch(N, p_pk_change=0.0)
    whole = tf.collapse(tf.DeviceBatch.upload(b)).download()
    ids = b.col("id").values
    got_src = []
    for part in range(4):
        idx = np.flatnonzero(ids % 4 == part)
        sb = abi.Batch([abi.Column(c.name, c.dtype, c.repr, values=None if c.values is None else c.values[idx],
                                   offsets=None if c.offsets is None else (np.arange(len(idx) + 1, dtype=np.uint32) * 24),
                                   data=None if c.data is None else c.data.reshape(-1, 24)[idx].reshape(-1)) for c in b.cols], len(idx), b.table_ns, b.table_name,
                       kind=b.kind[idx], src_row=idx.astype(np.int32))
        sb.schema = schema
        sb.old_keys = [abi.Column("id", "int64", abi.R_INT64, values=b.old_keys[0].values[idx], validity=b.old_keys[0].validity[idx])]
        sb.old_present = b.old_present[idx]
        out = tf.collapse(tf.DeviceBatch.upload(sb)).download()
        got_src.append(np.stack([out.src_row.astype(np.int64), out.col("ver").values, out.kind.astype(np.int64)], axis=1))
    got = np.concatenate(got_src)
    want = np.stack([whole.src_row.astype(np.int64), whole.col("ver").values, whole.kind.astype(np.int64)], axis=1)
    assert got.shape == want.shape
    assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])


# ---- Confluent-SR ingest and the queue serializers at 2^20 frames -------------------------------------------------------
SR_SCHEMA = ('{"title":"db.events","type":"object","properties":{"a":{"type":"array"},"b":{"oneOf":[{"type":"null"},{"type":"boolean"}]},'
             '"id":{"type":"integer"},"n":{"type":"number"},"s":{"type":"string"}},"required":["id","s"]}')


def _sr_payload(i):
    return (b'{"id":%d,"s":"row \\"%d\\" <%x>","n":%d.%03d,"b":%s,"a":[%d,"x"]}' % (i, i, i * 2654435761 & 0xFFFFFF, i % 1000, i % 997, (b"true", b"false", b"null")[i % 3], i % 7))


@pytest.fixture(scope="module")
def sr_events(tf):
    from transferia_amd import confluent_sr
    frames = [b"\0\0\0\0\x09" + _sr_payload(i) for i in range(N)]
    msgs, i = [], 0
    while i < N:  # one or two frames per Kafka message
        k = 2 if (i // 2) % 5 == 0 and i + 1 < N else 1
        msgs.append(b"".join(frames[i:i + k]))
        i += k
    data, cm = abi.messages(msgs, np.arange(len(msgs)), np.full(len(msgs), 1_700_000_000_000_000_000))
    o = confluent_sr.sr_json_options(9, SR_SCHEMA)
    res = tf.sr_json_parse(o, tf.DeviceBuffer.upload(data), cm)
    return o, data, cm, msgs, res


def test_sr_fullsize(tf, oracle, sr_events):
    """2^20 frames: every frame is a row, in order; sampled messages against the oracle; a message's frames share its index."""
    o, data, cm, msgs, res = sr_events
    assert not res.errors and res.device_batch.nrows == N
    b = res.batch
    assert np.array_equal(b.src_row, np.arange(N, dtype=np.int32))
    assert np.array_equal(b.col("id").values, np.arange(N, dtype=np.int64))
    assert np.all(np.diff(b.part_id.astype(np.int64)) >= 0) and int(b.part_id[-1]) == len(msgs) - 1
    bv = b.col("b")
    assert np.array_equal(bv.validity, np.arange(N) % 3 != 2) and np.array_equal(bv.values[bv.validity], (np.arange(N) % 3 == 0)[bv.validity].astype(np.uint8))
    for lo in (0, 5000, len(msgs) - 3000):  # windows of whole messages through the oracle
        sub = msgs[lo:lo + 3000]
        d2, m2 = abi.messages(sub)
        ref = oracle.sr_json_parse(o, d2, m2)
        first = int(np.searchsorted(b.part_id, lo))
        assert not ref.errors
        for c, rc in zip(b.cols, ref.batch.cols):
            for k in (0, 1, 17, ref.batch.nrows - 1):
                assert abi.norm_value(c.pyvalue(first + k)) == abi.norm_value(rc.pyvalue(k)), (c.name, lo, k)
        assert first + ref.batch.nrows <= N and int(b.col("id").values[first + ref.batch.nrows - 1]) == int(ref.batch.col("id").values[-1])


def test_sr_to_queue_json_roundtrip_fullsize(tf, sr_events):
    """SR parse → queue JSON serializer (1 MiB batches) → the generic JSON parser over the messages = the parsed batch."""
    o, data, cm, msgs, res = sr_events
    q = tf.queue_serialize(abi.queue_options(abi.QFMT_JSON, enabled=True, max_message_size=1 << 20), res.device_batch)
    sizes = np.diff(q.msg_start.astype(np.int64))
    assert len(q) > 50 and sizes.max() <= (1 << 20) and int(q.msg_row[-1]) == N and np.all(np.diff(q.msg_row) > 0)
    raw = q.values.download()
    assert raw.count(b"\n") == N - len(q)  # rows of one message are joined by "\n", messages lie back to back
    vals = [raw[int(q.msg_start[i]):int(q.msg_start[i + 1])] for i in range(len(q))]
    d2, m2 = abi.messages(vals)
    fields = abi.Schema.of([["a", "utf8"], ["b", "boolean"], ["id", "int64", True], ["n", "utf8"], ["s", "utf8"]])
    db, errs = tf.json_parse(abi.json_options(topic="t"), fields, d2, m2)
    assert not errs and db.nrows == N
    back, b = db.download(), res.batch
    assert np.array_equal(back.col("id").values, b.col("id").values)
    for name in ("s", "n", "a"):  # text columns: the same bytes (numbers and arrays as their JSON text)
        x, y = back.col(name), b.col(name)
        assert np.array_equal(x.offsets, y.offsets) and np.array_equal(x.data[: int(x.offsets[-1])], y.data[: int(y.offsets[-1])]), name
    bb, by = back.col("b"), b.col("b")
    assert np.array_equal(bb.validity if bb.validity is not None else np.ones(N, bool), by.validity)
    assert np.array_equal(bb.values[by.validity], by.values[by.validity])


def test_native_queue_fullsize(tf, sr_events):
    """2^20 rows through the native serializer: one ChangeItem per row in order, every message a JSON array within its limits."""
    import json
    o, data, cm, msgs, res = sr_events
    meta = abi.row_meta(N, ids=np.arange(N) % 1000, lsns=np.arange(N, dtype=np.uint64) + 10, commit_times=np.full(N, 1_700_000_000_000_000_000, np.uint64))
    q = tf.queue_serialize(abi.queue_options(abi.QFMT_NATIVE, enabled=True, max_change_items=1000, max_message_size=1 << 19, omit_table_schema=True), res.device_batch, meta)
    raw = q.values.download()
    assert raw.count(b'{"id":') == N and int(q.msg_row[-1]) == N
    rows = np.diff(q.msg_row)
    assert rows.max() <= 1000 and np.diff(q.msg_start.astype(np.int64)).max() <= (1 << 19)
    for m in (0, len(q) // 2, len(q) - 1):
        items = json.loads(raw[int(q.msg_start[m]):int(q.msg_start[m + 1])])
        r0 = int(q.msg_row[m])
        assert len(items) == int(rows[m])
        for k in (0, len(items) - 1):
            it, i = items[k], r0 + k
            assert (it["id"], it["nextlsn"], it["kind"], it["schema"], it["table"]) == (i % 1000, i + 10, "insert", "db", "events")
            assert it["columnnames"] == ["a", "b", "id", "n", "s"] and it["columnvalues"][2] == i and it["columnvalues"][4] == 'row "%d" <%x>' % (i, i * 2654435761 & 0xFFFFFF)
