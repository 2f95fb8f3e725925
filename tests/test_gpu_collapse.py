"""tfgpu_collapse (abstract.Collapse on device) against the oracle, through the C ABI.  Needs a real MI355X."""
import numpy as np
import pytest

from transferia_amd import abi
from collapse_cases import random_batch, rows_of
from test_collapse import CASES, SHAPES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


def check(tf, oracle, b, schema, ctx):
    got = tf.collapse(tf.DeviceBatch.upload(b)).download()
    ref = oracle.collapse(b, schema).batch
    assert got.nrows == ref.nrows, ctx
    assert [c.name for c in got.cols] == [c.name for c in ref.cols], ctx
    assert rows_of(got) == rows_of(ref), ctx
    return got


def test_reference_cases_uniform_columns(tf, oracle):
    """TestCollapse cases whose items share one ColumnNames list (the columnar form); the others are the host's."""
    ran = 0
    for case in CASES:
        items = case["items"]
        if len({tuple(it["names"]) for it in items}) != 1 or not items[0]["names"]:
            continue
        if any(v[0] == "json" for it in items for v in it["values"]):
            continue
        keys = items[0]["keys"]
        names = items[0]["names"]
        gotype = {nm: items[0]["values"][i][0] for i, nm in enumerate(names)}
        dt = {"int": "int64", "string": "utf8"}
        schema = abi.Schema([abi.ColSchema(nm, dt[gotype[nm]], nm in keys, "", "") for nm in names])
        b = abi.batch_from_rows(schema, names, [it["values"] for it in items], "", "t", [it["kind"] for it in items])
        b.schema = schema
        onames = next((it["old_names"] for it in items if it.get("old_names")), None)
        if onames:
            if any(it.get("old_names", onames) != onames for it in items):
                continue
            ogo = {nm: next(it["old_values"][i][0] for it in items if it.get("old_names")) for i, nm in enumerate(onames)}
            osch = abi.Schema([abi.ColSchema(nm, dt[ogo[nm]], False, "", "") for nm in onames])
            ob = abi.batch_from_rows(osch, onames, [it.get("old_values") or [["nil", None]] * len(onames) for it in items])
            b.old_keys, b.old_present = ob.cols, np.array([bool(it.get("old_names")) for it in items])
        got = check(tf, oracle, b, schema, case["name"])
        e = case["expect"]
        assert got.nrows == e["len"], case["name"]
        for k, kind in e.get("kinds", {}).items():
            assert ["insert", "update", "delete", "other"][int(got.kind[int(k)])] == kind if got.kind is not None else kind == "insert"
        ran += 1
    assert ran >= 6


@pytest.mark.parametrize("shape", range(len(SHAPES)))
def test_random_batches(tf, oracle, shape):
    for seed in range(4):
        b, schema = random_batch(1000 * shape + seed, **SHAPES[shape])
        check(tf, oracle, b, schema, (shape, seed))


def test_large_batches(tf, oracle):
    """A 200k-row CDC batch (30k keys x 6 strings, chains, primary-key changes): thousands of components walked in
    parallel; then one hot key — a single component, one lane replaying 20k rows in input order."""
    b, schema = random_batch(99, 200_000, domain=30_000, two_keys=True, nstrs=6)
    got = check(tf, oracle, b, schema, "large")
    assert 0 < got.nrows < b.nrows
    b2, schema2 = random_batch(5, 20_000, domain=1, weights=(1, 6, 2, 0))
    check(tf, oracle, b2, schema2, "hot key")


def test_passthrough_and_errors(tf, oracle):
    """len < 2, inserts only and no key column come back unchanged (change_item_collapse.go:49-66); a NaN key is refused."""
    b, schema = random_batch(3, 50, weights=(1, 0, 0, 0))
    assert rows_of(tf.collapse(tf.DeviceBatch.upload(b)).download()) == rows_of(b)
    b, schema = random_batch(4, 1)
    assert tf.collapse(tf.DeviceBatch.upload(b)).download().nrows == 1
    b, schema = random_batch(5, 50)
    b.schema = abi.Schema([abi.ColSchema(c.name, c.dtype, False, "", "") for c in schema.cols])
    assert rows_of(tf.collapse(tf.DeviceBatch.upload(b)).download()) == rows_of(b)
    sch = abi.Schema([abi.ColSchema("f", "double", True, "", ""), abi.ColSchema("v", "int64", False, "", "")])
    nb = abi.batch_from_rows(sch, ["f", "v"], [[["float64", 1.5], ["int64", 0]], [["float64", float("nan")], ["int64", 1]]], "", "t", ["insert", "update"])
    nb.schema = sch
    with pytest.raises(tf.TfgpuError):
        tf.collapse(tf.DeviceBatch.upload(nb))
    ok = abi.batch_from_rows(sch, ["f", "v"], [[["float64", 1.5], ["int64", 0]], [["float64", 1.5], ["int64", 1]], [["float64", -0.0], ["int64", 2]], [["float64", 0.0], ["int64", 3]]],
                             "", "t", ["insert", "update", "update", "update"])
    ok.schema = sch
    check(tf, oracle, ok, sch, "float keys")  # 1.5 merges; -0 and 0 print differently ("-0" / "0"): two keys


def test_cdc_workload(tf, oracle):
    """bench.py --workload collapse's stream at 2^18 rows: bit-identical to the oracle, columns compared as arrays."""
    from transferia_amd import workload
    from test_gpu_transformers import assert_batches_equal
    b, schema = workload.cdc_batch(1 << 18)
    got = tf.collapse(tf.DeviceBatch.upload(b)).download()
    ref = oracle.collapse(b, schema).batch
    assert_batches_equal(got, ref, "cdc")
    assert np.array_equal(got.kind, ref.kind) and np.array_equal(got.src_row, ref.src_row)
    assert np.array_equal(got.old_present, ref.old_present)
    assert np.array_equal(got.old_keys[0].values[got.old_present], ref.old_keys[0].values[ref.old_present])


def _one_item_batch(item):
    dt = {"int": "int64", "string": "utf8"}
    names = item["names"]
    schema = abi.Schema([abi.ColSchema(nm, dt[item["values"][i][0]], nm in item["keys"], "", "") for i, nm in enumerate(names)])
    b = abi.batch_from_rows(schema, names, [item["values"]], "", "t", [item["kind"]])
    b.schema = schema
    if item.get("old_names"):
        osch = abi.Schema([abi.ColSchema(nm, dt[item["old_values"][i][0]], False, "", "") for i, nm in enumerate(item["old_names"])])
        b.old_keys = abi.batch_from_rows(osch, item["old_names"], [item["old_values"]]).cols
        b.old_present = np.array([True])
    return b, schema


def test_keys_changed(tf, oracle):
    """tfgpu_keys_changed = ChangeItem.KeysChanged: the reference's TestPkeyChange cases, then random batches against
    the oracle, then the SplitUpdatedPKeys cut plan against the oracle's sublists."""
    from test_collapse import KC
    for c in KC["keys_changed"]:
        b, _ = _one_item_batch(c["item"])
        assert list(tf.keys_changed(tf.DeviceBatch.upload(b))) == [c["changed"]], c["ref"]
    for seed, kw in enumerate([dict(n=500, domain=4), dict(n=500, domain=3, two_keys=True, null_keys=0.2), dict(n=300, p_old=0.0),
                               dict(n=300, two_keys=True, bytes_key=True), dict(n=5000, domain=50, two_keys=True)]):
        b, schema = random_batch(400 + seed, **kw)
        db = tf.DeviceBatch.upload(b)
        want = oracle.keys_changed(b, schema)
        assert np.array_equal(tf.keys_changed(db), want), kw
        plan = tf.split_updated_pkeys(db)
        lens = [p[2] - p[1] if p[0] == "rows" else 2 for p in plan]
        ref_lens, cur = [], 0
        for ch in want:  # SplitUpdatedPKeys, utils.go:75-128
            if ch:
                if cur:
                    ref_lens.append(cur)
                ref_lens.append(2)
                cur = 0
            else:
                cur += 1
        if cur:
            ref_lens.append(cur)
        assert lens == ref_lens
    # different Go types under one key name are never deeply equal; NaN differs from NaN
    sch = abi.Schema([abi.ColSchema("f", "double", True, "", "")])
    fb = abi.batch_from_rows(sch, ["f"], [[["float64", 1.0]], [["float64", float("nan")]], [["float64", 0.0]]], "", "t", ["update"] * 3)
    fb.schema = sch
    fb.old_keys = abi.batch_from_rows(sch, ["f"], [[["float64", 1.0]], [["float64", float("nan")]], [["float64", -0.0]]]).cols
    fb.old_present = np.ones(3, bool)
    assert list(tf.keys_changed(tf.DeviceBatch.upload(fb))) == [False, True, False]
    fb.old_keys = abi.batch_from_rows(abi.Schema([abi.ColSchema("f", "int64", True, "", "")]), ["f"], [[["int64", 1]], [["int64", 0]], [["int64", 0]]]).cols
    assert list(tf.keys_changed(tf.DeviceBatch.upload(fb))) == [True, True, True]


@pytest.mark.gpu
def test_configs4_pipeline_single_rank(tf, oracle):
    """configs[4] on one GPU: sharder → tfgpu_partition → all-to-all over RCCL (world 1: every buffer, OldKeys included,
    makes the round trip through torch) → tfgpu_collapse → native queue serializer.  The messages are the oracle's for
    the same stream.  Own process, torch first: libtfgpu and torch must share the HIP runtime torch loads."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = textwrap.dedent("""
        import socket, sys
        sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
        import numpy as np
        import torch, torch.distributed as dist
        torch.cuda.set_device(0); torch.cuda.init()
        from transferia_amd import lib as tf, workload, partition, abi
        from oracle import oracle
        tf.init(0)
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% port, world_size=1, rank=0, device_id=torch.device("cuda", 0))
        N = 20000
        b, schema = workload.cdc_batch(N, p_pk_change=0.0)
        db = tf.DeviceBatch.upload(b)
        one = tf.Transformer("sharder_transformer", {"shardsCount": "1", "columns": {"includeColumns": ["^id$"]}}).apply(db).transformed
        grouped, counts = tf.partition(one, 1)
        assert counts == [N]
        back, recv = partition.exchange_device_batch(dist, tf, grouped, counts, schema)
        assert recv == [N]
        got = back.download()
        assert np.array_equal(got.old_present, b.old_present) and np.array_equal(got.old_keys[0].values[b.old_present], b.old_keys[0].values[b.old_present])
        col = tf.collapse(back)
        ref = oracle.collapse(b, schema).batch
        a = col.download()
        key = lambda x: sorted((int(x.col("id").values[i]), int(x.col("ver").values[i]), int(x.kind[i])) for i in range(x.nrows))
        assert key(a) == key(ref) and a.nrows < N
        meta = abi.row_meta(N, ids=np.arange(N) %% 97, lsns=np.arange(N, dtype=np.uint64) + 5, commit_times=np.full(N, 1700000000000000000, np.uint64))
        o = abi.queue_options(abi.QFMT_NATIVE, enabled=True, max_message_size=1 << 16, table_schema=schema)
        msgs = tf.queue_serialize(o, col, meta).messages()
        # the oracle serialises the device's collapsed rows (same order, same src_row → same row meta)
        a.schema = schema
        exp = oracle.queue_serialize(o, a, schema, meta)
        assert msgs == exp and len(msgs) > 10
        dist.destroy_process_group()
        print("PIPELINE_OK", a.nrows, len(msgs))
    """ % (root, root))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PIPELINE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_synchronize_items_count_as_inserts(tf, oracle):
    """InsertsOnly (change_item_collapse.go:37-44) takes SynchronizeKind for an insert: a batch of inserts with duplicate keys
    plus a synchronize item comes back unchanged — not deduplicated."""
    from transferia_amd import workload
    b, schema = workload.cdc_batch(500, keys=40, p_old=0.0)
    b.kind[:] = abi.K_INSERT
    b.kind[17] = abi.K_SYNCHRONIZE
    b.old_keys, b.old_present = None, None
    out = tf.collapse(tf.DeviceBatch.upload(b)).download()
    ref = oracle.collapse(b, schema).batch
    assert out.nrows == ref.nrows == 500
    b.kind[17] = abi.K_OTHER  # any other non-row kind: the duplicates collapse
    assert tf.collapse(tf.DeviceBatch.upload(b)).download().nrows == oracle.collapse(b, schema).batch.nrows < 500


@pytest.mark.gpu
def test_equal_hashes_are_settled_by_the_key_strings():
    """With TFGPU_COLLAPSE_WEAK_HASH=1 the 128-bit hash keeps two bits: nearly every pair of keys collides and the string
    compare in collapse_intern alone keeps distinct keys apart — the results must still be the oracle's (own process: the
    switch is read once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TFGPU_COLLAPSE_WEAK_HASH="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_collapse.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "reference_cases or random_batches or keys_that_print_alike"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.gpu
def test_keys_that_print_alike(tf, oracle):
    """Two byte strings that differ only in invalid UTF-8 print as the same JSON key (each bad byte becomes the escape \\ufffd,
    encoding/json encode.go): the reference's map files them together, and so does the device — by comparing the key TEXT
    when the raw values differ."""
    schema = abi.Schema.of([["k", "utf8", True], ["v", "int64", False]])
    rows = [[["string", b"a\xffb"], ["int64", 1]], [["string", b"a\xfeb"], ["int64", 2]], [["string", b"a\xef\xbf\xbdb"], ["int64", 3]], [["string", b"ab"], ["int64", 4]]]
    b = abi.batch_from_rows(schema, ["k", "v"], rows, "db", "t", kinds=["update", "update", "update", "update"])
    b.schema = schema
    out = tf.collapse(tf.DeviceBatch.upload(b)).download()
    ref = oracle.collapse(b, schema).batch
    # 0xFF and 0xFE both print as the escape \\ufffd: one key; a real U+FFFD prints as its three bytes: another
    assert out.nrows == ref.nrows == 3
    assert sorted(int(x) for x in out.col("v").values) == sorted(int(x) for x in ref.col("v").values) == [2, 3, 4]


# ---- compareColumns (change_item_collapse.go:7-35, :86-100): items whose ColumnNames differ, as ABSENT cells -----------------------
def _collapse_items(tf, items, names=None):
    from collapse_cases import batch_from_items, items_of
    made = batch_from_items(items, names=names)
    assert made is not None
    b, schema = made
    by_name = lambda its: [dict(it_, names=sorted(it_["names"]), values=[v for _, v in sorted(zip(it_["names"], it_["values"]))]) for it_ in its]
    assert by_name(items_of(b)) == by_name([dict(it_, src=i) for i, it_ in enumerate(_as_out(items))])  # the columnar form says what the items say (names in batch order)
    return items_of(tf.collapse(tf.DeviceBatch.upload(b)).download())


def _as_out(items):
    from transferia_amd import abi as _abi
    return [{"kind": it["kind"], "names": list(it["names"]), "values": [list(_abi.norm_value(v)) for v in it["values"]],
             "old": [[n, list(_abi.norm_value(v))] for n, v in zip(it.get("old_names") or [], it.get("old_values") or [])]} for it in items]


def test_reference_cases_diff_column_names(tf, oracle):
    """The TestCollapse runs whose items do NOT share one ColumnNames list (TOAST updates, deletes that list nothing): on the device,
    the merged rows equal to the oracle's row-wise restatement of the Go loop — names, values, kinds, OldKeys, source rows."""
    from collapse_cases import norm_items
    ran = 0
    for case in CASES:
        items = case["items"]
        if len({tuple(it["names"]) for it in items}) == 1:
            continue
        if any(v[0] == "json" for it in items for v in it["values"] + (it.get("old_values") or [])):
            continue
        want = norm_items(oracle.collapse_rows(items))
        got = _collapse_items(tf, items)
        assert got == want, case["name"]
        assert len(got) == case["expect"]["len"]
        for k, kind in case["expect"].get("kinds", {}).items():
            assert got[int(k)]["kind"] == kind
        ran += 1
    assert ran >= 8


def _soak_seed():
    import os
    return int(os.environ.get("TFGPU_TEST_SEED", "0")) != 0


@pytest.mark.parametrize("shape", [dict(n=2), dict(n=60, toastable=[4]), dict(n=400, domain=3), dict(n=400, domain=40, ncols=9), dict(n=300, p_absent=0.8, weights=(1, 8, 1, 0), toastable=[4]),
                                   dict(n=300, domain=2, weights=(2, 10, 0, 1), ncols=3, toastable=[2]), dict(n=300, domain=2, weights=(2, 10, 0, 1), ncols=3),
                                   dict(n=3000, domain=300, ncols=6, toastable=[5], p_absent=0.5, p_nokey=0.0)])
def test_random_toast_batches(tf, oracle, shape):
    """Random CDC streams whose Updates leave columns out: long merge chains, key changes, deletes that list only the key."""
    from collapse_cases import random_toast_items, norm_items
    names = ["id"] + ["c%d" % j for j in range(shape.get("ncols", 5))]
    merged = compared = reordered = 0
    for seed in range(8):
        items = random_toast_items(100 + seed, **shape)
        want = norm_items(oracle.collapse_rows(items))
        # (a chain that starts with a partial Update and later gains a column in front of one it has: the merged names leave batch order and the
        #  result carries every row's own order, tfgpu_batch.col_order)
        reordered += not all(r["names"] == [nm for nm in names if nm in r["names"]] for r in want)
        got = _collapse_items(tf, items, names)
        assert got == want, (shape, seed)
        compared += 1
        # a merged row mixes items: its cells ("r<item>.<column>" / item * 10 + column) name more than one
        merged += sum(1 for r in want if len({v[1].split(b".")[0][1:] if v[0] == "string" else b"%d" % (v[1] // 10)
                                              for nm, v in zip(r["names"], r["values"]) if nm != "id" and v[0] != "nil"}) > 1)
    assert compared == 8 and (merged or shape["n"] < 10), (compared, merged)
    assert reordered or "toastable" in shape or shape["n"] < 100 or _soak_seed(), shape   # (the committed streams' shape, not the product: a soak seed's streams may hold no such chain — seed 633)


def test_merged_names_out_of_batch_order(tf, oracle):
    """compareColumns APPENDS the names an Update brings; a chain whose merged ColumnNames are not in the batch's column order comes back with
    every row's own order (tfgpu_batch.col_order) — the oracle's names, in the oracle's order; such a batch goes to download, the native queue
    format and the Debezium emitter, and nowhere else."""
    from collapse_cases import norm_items, batch_from_items
    items = [{"kind": "update", "keys": ["id"], "names": ["id", "b"], "values": [["int64", 1], ["string", "b0"]]},
             {"kind": "update", "keys": ["id"], "names": ["id", "a"], "values": [["int64", 1], ["string", "a1"]]}]
    assert norm_items(oracle.collapse_rows(items))[0]["names"] == ["id", "b", "a"]
    assert _collapse_items(tf, items, ["id", "b", "a"])[0]["names"] == ["id", "b", "a"]  # batch order = merged order: no order array
    b, _ = batch_from_items(items, names=["id", "a", "b"])
    col = tf.collapse(tf.DeviceBatch.upload(b))
    host = col.download()
    assert host.col_order.tolist() == [[0, 2, 1]]
    assert _collapse_items(tf, items, ["id", "a", "b"])[0]["names"] == ["id", "b", "a"]
    for refuses in (lambda: tf.Transformer("mask_field", {"columns": ["a"], "maskFunctionHash": {"userDefinedSalt": "s"}}).apply(col), lambda: tf.DeviceBatch.upload(host),
                    lambda: tf.queue_serialize(abi.queue_options(abi.QFMT_JSON, enabled=True, max_message_size=1 << 12), col, None)):
        with pytest.raises(tf.TfgpuError, match="order|ABSENT"):
            refuses()
    # random streams whose Inserts leave columns out too: the oracle's rows either way
    from collapse_cases import random_toast_items
    names = ["id"] + ["c%d" % j for j in range(4)]
    seen = {True: 0, False: 0}
    for seed in range(24):
        items = random_toast_items(900 + seed, 40, ncols=4, domain=5, front_ok=True)
        want = norm_items(oracle.collapse_rows(items))
        seen[all(r["names"] == [nm for nm in names if nm in r["names"]] for r in want)] += 1
        assert _collapse_items(tf, items, names) == want, seed
    assert seen[True] and seen[False], seen


def test_absent_cells_travel_and_value_entries_refuse(tf, oracle):
    """ABSENT cells survive upload / view / download and tfgpu_partition; entries that compute on values say no by name."""
    from collapse_cases import random_toast_items, batch_from_items, items_of
    items = random_toast_items(7, 200)
    b, schema = batch_from_items(items, names=["id"] + ["c%d" % j for j in range(5)])
    db = tf.DeviceBatch.upload(b)
    assert items_of(db.download()) == items_of(b)
    grouped, counts = tf.partition(tf.Transformer("sharder_transformer", {"shardsCount": "3", "columns": {"includeColumns": ["^id$"]}}).apply(
        tf.DeviceBatch.upload(_without_absent(b))).transformed, 3)
    assert sum(counts) == 200
    with pytest.raises(tf.TfgpuError, match="ABSENT"):   # (mask_field and friends walk each row's own names since round 6: test_transformers_walk_each_rows_own_column_names)
        tf.apply_chain([tf.Transformer("replace_primary_key", {"keys": ["id"], "tables": {}}), tf.Transformer("sql", {"tables": {"include_tables": [".*"]}, "query": "select * from table where id >= 0"})], db)
    with pytest.raises(tf.TfgpuError, match="ABSENT"):
        tf.serialize(abi.FMT_CSV, db)
    # the changed-key flags read an absent key column as nil, like ChangeItem.KeysChanged reading ColumnValues by name
    assert list(tf.keys_changed(db)) == list(oracle.keys_changed_rows(items))


def test_upload_makes_an_absent_cell_read_nil(tf, oracle):
    """tfgpu_batch_upload establishes the invariant itself: an ABSENT cell's validity bit is cleared on the device even when
    the caller sends validity = NULL ("no nils") or leaves the bit set — the key kernels (hash, compare, sharder, keys_changed) read
    validity only, and an unlisted key must read nil like CurrentKeysString of a name the item does not list."""
    import dataclasses
    from collapse_cases import random_toast_items, batch_from_items, items_of
    items = random_toast_items(11, 300, p_nokey=0.3)
    names = ["id"] + ["c%d" % j for j in range(5)]
    b, schema = batch_from_items(items, names=names)
    assert any(c.absent is not None and c.absent.any() for c in b.cols)
    for variant in ("null", "ones"):
        cols = [dataclasses.replace(c, validity=None if variant == "null" else np.ones(b.nrows, bool)) if (c.absent is not None and c.validity is not None and not (~c.validity & ~c.absent).any()) else c
                for c in b.cols]
        sloppy = abi.Batch(cols, b.nrows, b.table_ns, b.table_name, b.kind, b.src_row, b.part_id)
        sloppy.schema = b.schema
        sloppy.old_keys, sloppy.old_present = getattr(b, "old_keys", None), getattr(b, "old_present", None)
        db = tf.DeviceBatch.upload(sloppy)
        got = db.download()
        for c in got.cols:
            if c.absent is not None and c.absent.any():
                assert c.validity is not None and not (c.validity & c.absent).any(), (variant, c.name)
        assert items_of(got) == items_of(b), variant
        assert list(tf.keys_changed(db)) == list(oracle.keys_changed_rows(items)), variant


def _without_absent(b):
    import dataclasses
    c = abi.Batch([dataclasses.replace(col, absent=None) for col in b.cols], b.nrows, b.table_ns, b.table_name, b.kind, b.src_row, b.part_id)
    c.schema = b.schema
    return c


def test_toast_rows_through_collapse_and_the_native_queue_format(tf, oracle):
    """A TOAST stream → tfgpu_collapse → NativeSerializer on the device: every message byte for byte the oracle's ChangeItem.MarshalJSON
    (change_item.go:568-616) of the same rows — each row writes its OWN columnnames, and columnvalues only when it lists a column."""
    from collapse_cases import random_toast_items, batch_from_items, items_of
    names = ["id"] + ["c%d" % j for j in range(5)]
    done, seen = 0, set()
    for seed in range(6):
        items = random_toast_items(300 + seed, 120, toastable=[3, 4], p_absent=0.5, weights=(3, 6, 2, 0))
        for it in items[::17]:                      # some rows list nothing at all (a Delete named by its OldKeys alone)
            if it["kind"] == "delete":
                it["names"], it["values"] = [], []
                it.setdefault("old_names", ["id"]); it.setdefault("old_values", [["int64", 1]])
        b, schema = batch_from_items(items, names=names)
        N = b.nrows
        meta = abi.row_meta(N, ids=np.arange(N) % 97, lsns=np.arange(N, dtype=np.uint64) + 5, commit_times=np.full(N, 1700000000000000000, np.uint64))
        o = abi.queue_options(abi.QFMT_NATIVE, enabled=True, max_message_size=1 << 12, table_schema=schema)
        db = tf.DeviceBatch.upload(b)
        for dev in (db, None):                      # the stream as it came, then what Collapse leaves of it
            if dev is None:
                dev = tf.collapse(db)
            host = dev.download()
            host.schema = schema
            want = oracle.queue_serialize(o, host, schema, meta)
            got = tf.queue_serialize(o, dev, meta).messages()
            assert got == want and got, seed
            seen.update(x for m in got for x in (b'"columnnames":["id","c0","c1","c2"]', b'"columnnames":["id","c0","c1","c2","c4"]', b'"columnnames":[],"table_schema"',
                                                 b'"columnnames":["id","c0","c1","c2","c3","c4"]') if x in m)
            done += 1
    assert done == 12 and len(seen) == 4, (done, seen)
    # the JSON queue format computes on values: it says no, by name
    with pytest.raises(tf.TfgpuError, match="ABSENT"):
        tf.queue_serialize(abi.queue_options(abi.QFMT_JSON, enabled=True, max_message_size=1 << 12), db, None)


def test_toast_rows_through_the_bufferers_concat_and_the_measurer(tf, oracle):
    """Size.Values of a TOAST batch (util.DeepSizeof(ColumnValues): a row's slice holds the columns it lists), then the Bufferer's flush — ONE
    tfgpu_dbatch_concat of the buffered batches (some with ABSENT cells, some without) — and a cut back into shards: every row keeps its
    ColumnNames, and Collapse over the merged batch is the oracle's over the concatenated items."""
    from collapse_cases import random_toast_items, batch_from_items, items_of, norm_items
    names = ["id"] + ["c%d" % j for j in range(5)]
    parts, items_all = [], []
    for k, (n, p_abs, w) in enumerate([(64, 0.4, (3, 6, 1, 0)), (40, 0.0, (3, 6, 0, 0)), (128, 0.6, (3, 6, 1, 0)), (8, 0.3, (3, 6, 1, 0))]):
        items = random_toast_items(500 + k, n, p_absent=p_abs, weights=w, p_nokey=0.0)
        assert p_abs or all(it["names"] == names for it in items)  # the second part lists everything: no bitmap of its own in the concat
        b, schema = batch_from_items(items, names=names)
        total, per = oracle.deepsizeof(b, schema)
        db = tf.DeviceBatch.upload(b)
        assert int(tf.deepsizeof(db)) == total
        parts.append(db)
        items_all += items
    merged = tf.DeviceBatch.concat(parts)
    whole, _ = batch_from_items(items_all, names=names)
    assert items_of(merged.download()) == [dict(x, src=x["src"]) for x in items_of(whole)] or \
        [dict(x, src=0) for x in items_of(merged.download())] == [dict(x, src=0) for x in items_of(whole)]
    got = [dict(x, src=0) for x in items_of(tf.collapse(merged).download())]
    want = [dict(x, src=0) for x in norm_items(oracle.collapse_rows(items_all))]
    assert got == want
    cut = merged.slice(64, 104)   # rows 64 .. 167: the all-listed part and the head of the third
    assert [dict(x, src=0) for x in items_of(cut.download())] == [dict(x, src=0) for x in items_of(whole)[64:168]]


def test_download_without_room_for_the_absent_bits_is_refused(tf):
    """A binding that knows nothing of tfgpu_column.absent must not read a TOAST batch as if its rows listed nils."""
    import ctypes as C
    from collapse_cases import random_toast_items, batch_from_items
    b, _ = batch_from_items(random_toast_items(3, 50), names=["id"] + ["c%d" % j for j in range(5)])
    db = tf.DeviceBatch.upload(b)
    v = db.view()
    n = int(v.nrows)
    carr = (abi.CColumn * v.ncols)()
    keep = []
    for i in range(v.ncols):
        c = v.cols[i]
        carr[i].name, carr[i].dtype, carr[i].repr = c.name, c.dtype, c.repr
        if c.repr in abi.VAR_REPRS:
            off, dat = np.zeros(n + 1, np.uint32), np.zeros(max(int(c.data_len), 1), np.uint8)
            keep += [off, dat]
            carr[i].offsets, carr[i].data, carr[i].data_len = off.ctypes.data, dat.ctypes.data, c.data_len
        else:
            vals = np.zeros(max(n, 1), abi.REPR_NP[c.repr])
            keep.append(vals)
            carr[i].values = vals.ctypes.data
    hb = abi.CBatch()
    hb.nrows, hb.ncols, hb.cols, hb.mem = n, v.ncols, carr, abi.MEM_HOST
    assert tf.load().tfgpu_dbatch_download(db._h, C.byref(hb)) != 0
    assert b"absent" in (tf.load().tfgpu_last_error() or b"")
    assert db.download().nrows == n   # the binding that brings the room gets the rows


def test_toast_rows_through_collapse_and_clickhouse_jsoneachrow(tf, oracle):
    """A Postgres-CDC batch with TOASTed Updates stays on the device all the way to the ClickHouse sink (round 6): the stream as it came and what
    tfgpu_collapse leaves of it → ClickHouse JSONEachRow on the device, byte for byte the oracle's MarshalCItoJSON over the same rows — it walks each row's
    OWN ColumnNames and skips nils (marshal.go:82-125), so a cell the row does not list prints like a nil one: nothing.  Rows that carry their own name
    ORDER (col_order) are refused by name; json / csv (they type the i-th value by the schema's i-th column) keep refusing ragged rows."""
    from collapse_cases import random_toast_items, batch_from_items
    names = ["id"] + ["c%d" % j for j in range(5)]
    done = refused = 0
    for seed in range(8):
        items = random_toast_items(1300 + seed, 160, toastable=[2, 3, 4], p_absent=0.5, weights=(3, 6, 0, 0), p_nokey=0.0)
        b, schema = batch_from_items(items, names=names)
        assert any(c.absent is not None and c.absent.any() for c in b.cols)
        db = tf.DeviceBatch.upload(b)
        for dev in (db, tf.collapse(db)):
            host = dev.download()
            host.schema = schema
            if getattr(host, "col_order", None) is not None:
                with pytest.raises(tf.TfgpuError, match="ABSENT"):
                    tf.serialize(abi.FMT_CH_JSON_EACH_ROW, dev)
                refused += 1
                continue
            want = oracle.serialize(abi.FMT_CH_JSON_EACH_ROW, host, schema)
            got = tf.serialize(abi.FMT_CH_JSON_EACH_ROW, dev).download()
            assert want is not None and bytes(got) == bytes(want), seed
            if any(c.absent is not None and c.absent.any() for c in host.cols):
                done += 1
                with pytest.raises(tf.TfgpuError, match="ABSENT"):
                    tf.serialize(abi.FMT_JSON, dev)
    assert done >= 8, (done, refused)


def test_transformers_walk_each_rows_own_column_names(tf, oracle):
    """mask_field / convert_to_string / convert_to_datetime leave a cell the row does not list as it is (hmac_hasher.go:56-63: the loop runs over
    item.ColumnNames), filter_rows fails a row that does not list a filtered column ("Unable to find column", filter_rows.go:147-154) unless it lists
    nothing at all, filter_columns / skip_events carry the bitmaps — the chain over Inserts that leave columns out, device against the oracle's row-wise
    transformers: kept rows, names, values, row errors; then on to the ClickHouse sink."""
    from collapse_cases import random_toast_items, batch_from_items, items_of
    names = ["id"] + ["c%d" % j for j in range(5)]
    chains = [
        [("mask_field", {"maskFunctionHash": {"userDefinedSalt": "s"}, "columns": ["c0", "c3"]})],
        [("convert_to_string", {"columns": {"includeColumns": ["^c1$", "^c3$"]}, "tables": {}})],
        [("filter_rows", {"filter": "c1 > 50"})],
        [("filter_rows", {"filters": ["c1 > 50", "c3 > 20"]}), ("mask_field", {"maskFunctionHash": {"userDefinedSalt": "s"}, "columns": ["c2"]})],
        [("mask_field", {"maskFunctionHash": {"userDefinedSalt": "s"}, "columns": ["c0"]}), ("filter_columns", {"columns": {"excludeColumns": ["^c4$"]}, "tables": {}}),
         ("convert_to_string", {"columns": {"includeColumns": ["^c1$"]}, "tables": {}})],
    ]
    checked = 0
    for seed in range(4):
        items = random_toast_items(1700 + seed, 200, p_absent=0.4, weights=(1, 0, 0, 0), front_ok=True)
        if seed == 3:
            items[5]["names"], items[5]["values"] = [], []          # a row that lists nothing: filter_rows' name loop never runs for it
        b, schema = batch_from_items(items, names=names)
        assert any(c.absent is not None and c.absent.any() for c in b.cols)
        for chain in chains:
            ref = oracle.apply_chain([oracle.Transformer(t, c) for t, c in chain], b, schema)
            res = tf.apply_chain([tf.Transformer(t, c) for t, c in chain], tf.DeviceBatch.upload(b))
            got = res.transformed.download()
            assert items_of(got) == items_of(ref.batch), (seed, chain[0][0])
            assert sorted((e[0], e[1]) for e in res.errors) == sorted((e[0], abi.ROWERR[e[1]]) for e in ref.errors), (seed, chain[0][0])
            got.schema = ref.schema
            want = oracle.serialize(abi.FMT_CH_JSON_EACH_ROW, ref.batch, ref.schema)
            if want is not None and got.nrows:
                assert bytes(tf.serialize(abi.FMT_CH_JSON_EACH_ROW, res.transformed).download()) == bytes(want), (seed, chain[0][0])
            checked += 1
    assert checked == 4 * len(chains)
    with pytest.raises(tf.TfgpuError, match="ABSENT"):   # `sql` serializes whole rows for clickhouse-local: stock path
        tf.apply_chain([tf.Transformer("replace_primary_key", {"keys": ["id"], "tables": {}}), tf.Transformer("sql", {"tables": {"include_tables": [".*"]}, "query": "select * from table where id >= 0"})],
                       tf.DeviceBatch.upload(b))
