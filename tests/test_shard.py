"""One process, several GPUs: row-range shards of a device batch on lanes, per-lane transform, ordered concat
== the single-lane result.  On the GPU box the lanes share device 0 (or its devices, if it has several); under tools/hipemu a
subprocess runs the same check over two FAKE devices (HIPEMU_DEVICES=2), which exercises the lane -> device mapping and the
re-homing copies' bookkeeping — peer copies on real xGMI stay unmeasured until a multi-GPU box exists."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from transferia_amd import abi
from test_gpu_transformers import assert_batches_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED0 = int(os.environ.get("TFGPU_TEST_SEED", "0"))


@pytest.fixture(scope="module")
def tf():
    from transferia_amd import lib
    lib.init()
    return lib


SCHEMA = abi.Schema.of([["id", "int64", True], ["name", "utf8"], ["ts", "timestamp"], ["opt", "int32"], ["flag", "boolean"]])
NAMES = ["id", "name", "ts", "opt", "flag"]


def make(n, seed=1, old=True):
    rng = np.random.default_rng(SEED0 + seed)
    rows = []
    for i in range(n):
        k = int(rng.integers(0, 1 << 40))
        rows.append([["int64", k], ["nil", None] if k % 11 == 0 else ["string", ("n%d-" % k) * (k % 5)],
                     ["time", (1_600_000_000 + k % 100000, k % 1000)], ["nil", None] if k % 5 == 0 else ["int32", k % 100000], ["bool", bool(k & 1)]])
    kinds = [("insert", "update", "delete")[int(rng.integers(0, 3))] for _ in range(n)]
    b = abi.batch_from_rows(SCHEMA, NAMES, rows, "db", "t", kinds)
    b.schema = SCHEMA
    if old and n:
        present = rng.integers(0, 4, n) != 0
        b.old_keys = [abi.Column("id", "int64", abi.R_INT64, values=rng.integers(0, 1 << 40, n).astype(np.int64), validity=present.copy())]
        b.old_present = present
    return b


def extras_equal(a, b, ctx=""):
    assert (a.table_ns, a.table_name) == (b.table_ns, b.table_name), ctx
    for f in ("kind", "src_row", "part_id"):
        x, y = getattr(a, f), getattr(b, f)
        if f == "src_row":  # NULL = identity (include/tfgpu.h); a download spells it out
            x = np.arange(a.nrows, dtype=np.int32) if x is None else x
            y = np.arange(b.nrows, dtype=np.int32) if y is None else y
        assert (x is None) == (y is None) and (x is None or np.array_equal(x, y)), (ctx, f)
    assert len(a.old_keys) == len(b.old_keys), ctx
    pa = a.old_present if a.old_present is not None else np.ones(a.nrows, bool)
    pb = b.old_present if b.old_present is not None else np.ones(b.nrows, bool)
    if a.old_keys:
        assert np.array_equal(pa, pb), (ctx, "old_present")
    for ca, cb in zip(a.old_keys, b.old_keys):
        va = ca.validity if ca.validity is not None else np.ones(a.nrows, bool)
        vb = cb.validity if cb.validity is not None else np.ones(b.nrows, bool)
        assert ca.name == cb.name and np.array_equal(va, vb) and np.array_equal(ca.values[va], cb.values[vb]), (ctx, ca.name)


@pytest.mark.gpu
@pytest.mark.parametrize("n,g", [(0, 2), (1, 3), (63, 2), (64, 2), (1000, 3), (4099, 8)])
def test_shard_concat_is_identity(tf, n, g):
    host = make(n, seed=n + g)
    db = tf.DeviceBatch.upload(host)
    parts, row0 = db.shard_rows(g, lanes=list(range(g)))
    assert row0[0] == 0 and all(r % 64 == 0 for r in row0) and sum(p.nrows for p in parts) == n
    assert max(p.nrows for p in parts) - min(p.nrows for p in parts) <= 64 + (n % 64)
    back = tf.DeviceBatch.concat(parts).download()
    assert_batches_equal(back, host, "n=%d g=%d" % (n, g))
    if n:
        extras_equal(back, host)
    # a slice is the rows it names
    if n >= 16:
        s = db.slice(8, n - 11).download()
        assert s.nrows == n - 11
        for cs, ch in zip(s.cols, host.cols):
            for i in (0, 1, s.nrows - 1):
                assert cs.pyvalue(i) == ch.pyvalue(8 + i), (cs.name, i)
    with pytest.raises(tf.TfgpuError):
        db.slice(3, 1)


def sharded_chain(tf, db, chain, lanes):
    """shard -> one host thread per lane applies the chain -> concat with the shards' first rows: what one Go worker with a
    goroutine per device lane does"""
    parts, row0 = db.shard_rows(len(lanes), lanes=lanes)
    outs, errs = [None] * len(lanes), []

    def work(k):
        try:
            tf.lane_use(lanes[k])
            plans = [tf.Transformer(t, c) for t, c in chain]
            outs[k] = tf.apply_chain(plans, parts[k]).transformed
            tf.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work, args=(k,)) for k in range(len(lanes))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    return tf.DeviceBatch.concat(outs, row_base=row0)


CHAIN = [("mask_field", {"columns": ["name"], "maskFunctionHash": {"userDefinedSalt": "s"}}),
         ("filter_rows", {"filter": "opt > 50000 AND flag = true"}),
         ("convert_to_string", {"columns": {"includeColumns": ["opt"]}})]


@pytest.mark.gpu
@pytest.mark.parametrize("n,g", [(1000, 2), (5000, 4)])
def test_sharded_chain_equals_single_lane(tf, n, g):
    host = make(n, seed=7 * g, old=False)
    host.kind = None
    db = tf.DeviceBatch.upload(host)
    single = tf.apply_chain([tf.Transformer(t, c) for t, c in CHAIN], db).transformed.download()
    got = sharded_chain(tf, db, CHAIN, list(range(g))).download()
    tf.lane_use(0)
    assert_batches_equal(got, single, "g=%d" % g)
    assert single.src_row is not None and np.array_equal(got.src_row, single.src_row)


WORKER = r"""
import os, sys
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from transferia_amd import lib
if os.environ.get("TFGPU_TEST_EMU_LIB"):
    lib._LIBPATH = os.environ["TFGPU_TEST_EMU_LIB"]
import numpy as np
import test_shard as T
from test_gpu_transformers import assert_batches_equal
ndev = int(sys.argv[2])
devs = lib.init_devices(list(range(ndev)))
assert [lib.lane_device(k) for k in range(2 * ndev)] == devs + devs
host = T.make(3000, seed=99, old=False)
host.kind = None
db = lib.DeviceBatch.upload(host)                  # lane 0, device 0
single = lib.apply_chain([lib.Transformer(t, c) for t, c in T.CHAIN], db).transformed.download()
got = T.sharded_chain(lib, db, T.CHAIN, list(range(ndev))).download()   # lane k on device k
lib.lane_use(0)
assert_batches_equal(got, single, "devices")
assert np.array_equal(got.src_row, single.src_row)
# a batch re-homed on another device's lane and brought back is the same batch
there = db.to_lane(ndev - 1)
lib.lane_use(ndev - 1)
lib.synchronize()
back = there.to_lane(0)
lib.lane_use(0)
assert_batches_equal(back.download(), host, "round trip")
try:
    lib.init_devices([0])
    raise SystemExit("a second device list was accepted")
except lib.TfgpuError:
    pass
print("OK", ndev)
"""


@pytest.mark.gpu
def test_several_devices_in_one_process(tf):
    """Needs >= 2 devices: real ones, or the emulator's fake ones (HIPEMU_DEVICES, set here for the emulated run)."""
    env = dict(os.environ)
    if env.get("TFGPU_TEST_EMU_LIB"):
        env["HIPEMU_DEVICES"] = "2"
        ndev = 2
    else:
        import ctypes as C
        n = C.c_int(0)
        tf.load().tfgpu_device_count(C.byref(n))
        ndev = min(n.value, 4)
        if ndev < 2:
            pytest.skip("one device visible: the several-devices path runs under tools/hipemu with fake devices, and is unmeasured on hardware")
    r = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(ndev)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "OK %d" % ndev in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
