"""Every kernel of libtfgpu runs as plain C++ in this GPU-less container (tools/hipemu: one fiber per thread, wavefront
operations and barriers exchanged in lockstep) and must already agree with the oracle byte for byte — a pre-flight for
the `-m gpu` run, not a substitute for it: it proves the kernels' logic, not the gfx950 build.  Runs in a subprocess so that the product
binding of THIS process never points at anything but libtfgpu.so."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_emulated_kernels_match_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hipemu", "run_gpu_tests.py")], capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


EXCHANGE_WORKER = r"""
import json, os, sys, time
import numpy as np
root, rank, world, case, idfile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from transferia_amd import abi, lib
if os.environ.get("TFGPU_TEST_EMU_LIB"):   # the CPU pre-flight; without it: the real library, one GPU per rank (LOCAL_RANK)
    lib._LIBPATH = os.environ["TFGPU_TEST_EMU_LIB"]
from oracle import oracle as ora
ora.build()
lib.init()

SCHEMA = abi.Schema.of([["id", "int64", True], ["name", "utf8"], ["ts", "timestamp"], ["opt", "int32"]])
NAMES = ["id", "name", "ts", "opt"]


def make(r):
    # rows of rank r as (host batch, python rows, kinds, old key values or None): deterministic, every rank can rebuild every other's
    n = {"plain": 300 + 41 * r, "ragged": 300 + 41 * r, "toast": 300 + 41 * r, "empty_rank": 0 if r == 1 else 257, "mismatch": 50, "unnamed_old": 60}[case]
    lo = 1000 * r
    lean = case in ("ragged", "unnamed_old") and r == 1   # this rank has no nil anywhere, whole seconds only, no OldKeys, no kinds
    rows = []
    for i in range(n):
        k = lo + i
        rows.append([["int64", k * 7919 % 100003], ["nil", None] if (k % 11 == 0 and not lean) else ["string", ("n%d-" % k) * (k % 5)],
                     ["time", (1_600_000_000 + k, 0 if lean else k % 1000)], ["nil", None] if (k % 5 == 0 and not lean) else ["int32", k]])
    kinds = None if lean else [("insert", "update", "delete")[(lo + i) % 3] for i in range(n)]
    schema = SCHEMA
    if case == "unnamed_old" and r == 1:  # … and no key in its TableSchema either: it cannot name the OldKeys the others carry
        schema = abi.Schema.of([["id", "int64"], ["name", "utf8"], ["ts", "timestamp"], ["opt", "int32"]])
    if case == "mismatch" and r == 1:
        schema = abi.Schema.of([["id", "int64", True], ["name", "utf8"], ["ts", "timestamp"], ["opt", "int64"]])
        for row in rows:
            if row[3][0] == "int32":
                row[3][0] = "int64"
    if n:
        b = abi.batch_from_rows(schema, NAMES, rows, "db", "t", kinds)
    else:  # a rank whose rows were all filtered away still holds the table's typed (empty) columns
        z = np.zeros(0, np.int64)
        b = abi.Batch([abi.Column("id", "int64", abi.R_INT64, values=z), abi.Column("name", "utf8", abi.R_STRING, offsets=np.zeros(1, np.uint32), data=np.zeros(0, np.uint8)),
                       abi.Column("ts", "timestamp", abi.R_TIME, values=z, nanos=np.zeros(0, np.int32)), abi.Column("opt", "int32", abi.R_INT32, values=np.zeros(0, np.int32))],
                      0, "db", "t", kind=np.zeros(0, np.uint8))
    b.schema = schema
    if case == "toast" and r != 1:  # rows that do not list `opt` / `name` (tfgpu_column.absent): rank 1 has none, its peers do
        for nm, mod in (("opt", 7), ("name", 5)):
            c = b.col(nm)
            ab = np.array([(lo + i) % mod == 3 for i in range(n)], bool)
            c.absent = ab
            c.validity = (np.ones(n, bool) if c.validity is None else c.validity) & ~ab
            for i in np.nonzero(ab)[0]:
                rows[int(i)][NAMES.index(nm)] = ["absent", None]
        if b.col("name").data is not None:  # an absent text cell holds no bytes
            nmcol = abi.batch_from_rows(schema, NAMES, [[v if v[0] != "absent" else ["nil", None] for v in row] for row in rows], "db", "t", kinds).col("name")
            nmcol.absent, nmcol.validity = b.col("name").absent, b.col("name").validity
            b.cols[NAMES.index("name")] = nmcol
    if lean:
        for c in b.cols:
            c.validity = None
        b.col("ts").nanos = None
        old = None
    else:
        present = np.array([(lo + i) % 3 != 0 and (lo + i) % 4 != 1 for i in range(n)], bool)
        old = [(lo + i) * 3 if present[i] else None for i in range(n)]
        b.old_keys = [abi.Column("id", "int64", abi.R_INT64, values=np.array([v or 0 for v in old], np.int64), validity=present.copy())]
        b.old_present = present
    return b, rows, kinds, old


shard = {"shardsCount": str(world), "columns": {"includeColumns": ["^id$"]}}
mine, _, _, _ = make(rank)
db = lib.DeviceBatch.upload(mine)
one = lib.Transformer("sharder_transformer", shard).apply(db).transformed
grouped, counts = lib.partition(one, world)
if rank == 0:
    uid = lib.Comm.unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 120
        time.sleep(0.02)
    uid = open(idfile, "rb").read()
comm = lib.Comm.create(uid, rank, world)
if case in ("mismatch", "unnamed_old"):
    try:
        comm.exchange(grouped, counts)
        print(json.dumps({"rank": rank, "error": None}))
    except lib.TfgpuError as e:
        print(json.dumps({"rank": rank, "error": e.code == lib.ERR_INVALID and ("representation" if case == "mismatch" else "cannot name") in str(e)}))
    comm.close()
    sys.exit(0)
for _ in range(2):   # a communicator carries any number of exchanges
    back, recv = comm.exchange(grouped, counts)
got = back.download()

# what the reference's sharder says every rank sends where (oracle), source ranks in order, row order kept
exp_rows, exp_kinds, exp_old, exp_src, exp_recv = [], [], [], [], []
for s in range(world):
    hb, rows, kinds, old = make(s)
    part = ora.Transformer("sharder_transformer", shard).apply(hb, SCHEMA).batch.part_id if hb.nrows else np.zeros(0, np.uint32)
    sel = [i for i in range(hb.nrows) if int(part[i]) == rank]
    exp_recv.append(len(sel))
    exp_rows += [rows[i] for i in sel]
    exp_kinds += [(kinds[i] if kinds else "insert") for i in sel]
    exp_old += [(old[i] if old else None) for i in sel]
    exp_src += sel
assert recv == exp_recv, (recv, exp_recv)
assert got.nrows == len(exp_rows)
assert (got.table_ns, got.table_name) == ("db", "t")
for j, name in enumerate(NAMES):
    c = got.col(name)
    for i in range(got.nrows):
        g, e = c.pyvalue(i), exp_rows[i][j]
        if e[0] == "absent":  # the row does not list the column: the bit travelled with it
            assert c.absent is not None and c.absent[i] and g == ["nil", None], (name, i)
            continue
        assert c.absent is None or not c.absent[i], (name, i)
        if e[0] == "string":
            e = ["string", e[1].encode()]
        if e[0] == "time":
            e = ["time", tuple(e[1])]
        assert g == e, (name, i, g, e)
if got.nrows:
    kid = {"insert": abi.K_INSERT, "update": abi.K_UPDATE, "delete": abi.K_DELETE}
    assert [int(k) for k in got.kind] == [kid[k] for k in exp_kinds]
    ok = got.old_keys[0]
    assert ok.name == "id" and ok.repr == abi.R_INT64
    pres = got.old_present if got.old_present is not None else np.ones(got.nrows, bool)
    for i in range(got.nrows):
        assert bool(pres[i]) == (exp_old[i] is not None), i
        if exp_old[i] is not None:
            assert ok.is_valid(i) and int(ok.values[i]) == exp_old[i]
    assert [int(x) for x in got.part_id] == [rank] * got.nrows
# and the consumer after the exchange takes it as is: Collapse on the received rows == the oracle's on the same rows
col = lib.collapse(back).download()
print(json.dumps({"rank": rank, "rows": got.nrows, "collapsed": col.nrows, "recv": recv, "sent": counts}))
comm.close()
"""


EXCHANGE_CASES = [("plain", 2), ("ragged", 2), ("empty_rank", 2), ("mismatch", 2), ("unnamed_old", 2), ("toast", 2), ("toast", 3)]


@pytest.fixture(scope="module")
def exchange_runs(tmp_path_factory):
    """Every case's ranks, all cases side by side (a rank is mostly start-up — the interpreter, numpy, the two libraries; one case after
    another they were 150 s of the CPU suite): {(case, world): [(returncode, stdout, stderr) per rank]}."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "hipemu"))
    import build as emu_build
    so = emu_build.build()
    fake = os.path.join(os.path.dirname(so), "libfakerccl.so")
    env = dict(os.environ, TFGPU_TEST_EMU_LIB=so, TFGPU_RCCL_LIB=fake)
    procs = {}
    for case, world in EXCHANGE_CASES:
        d = tmp_path_factory.mktemp("xchg_%s_%d" % (case, world))
        script = d / "worker.py"
        script.write_text(EXCHANGE_WORKER)
        procs[(case, world)] = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), case, str(d / "uid")], stdout=subprocess.PIPE,
                                                 stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    out = {}
    for key, ps in procs.items():
        res = []
        for p in ps:
            try:
                o, e = p.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                p.kill()
                o, e = p.communicate()
                e += "\n(timed out)"
            res.append((p.returncode, o, e))
        out[key] = res
    return out


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("case,world", EXCHANGE_CASES)
def test_emulated_exchange_ranks(exchange_runs, case, world):
    """tfgpu_exchange on world_size 2 / 3 without a GPU: one process per rank, the emulated kernels, RCCL replaced by the
    socket stand-in (tools/hipemu/fake_rccl.cpp) — the descriptor agreement, the grouped send/recv plan and the rebuild of
    offsets / bitmaps / OldKeys are the product code.  Expected rows come from the oracle's sharder."""
    import json
    outs = []
    for rc, o, e in exchange_runs[(case, world)]:
        assert rc == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    if case in ("mismatch", "unnamed_old"):
        assert [o["error"] for o in outs] == [True] * world  # every rank refuses (the one that cannot name the OldKeys AND its peers), nobody hangs
        return
    assert sum(o["rows"] for o in outs) == sum(sum(o["sent"]) for o in outs)
    for r, o in enumerate(outs):
        assert o["recv"] == [outs[s]["sent"][r] for s in range(world)]


def _gpu_count():
    """Devices as the library itself sees them.  Deliberately NOT through torch: this runs at collection time, and importing
    torch there loads torch's own bundled HIP / HSA / RCCL copies into the test process before libtfgpu.so — the RCCL the
    library then binds (dlopen by soname finds the bundled one) fails its init with 'no ROCm-capable device is detected'."""
    try:
        import ctypes
        from transferia_amd import lib
        n = ctypes.c_int(0)
        return int(n.value) if lib.load().tfgpu_device_count(ctypes.byref(n)) == 0 else 0
    except Exception:  # noqa: BLE001
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: tfgpu_exchange between two processes over real RCCL / xGMI")
@pytest.mark.parametrize("case", ["plain", "ragged", "empty_rank", "mismatch", "toast"])
def test_exchange_two_gpus(tmp_path, case):
    """The same worker as the emulated world_size-2 test, on two MI355X: one process per GPU, librccl, the rendezvous id through a
    file.  Runs only where two GPUs are visible (the build box has one)."""
    import json
    script = tmp_path / "worker.py"
    script.write_text(EXCHANGE_WORKER)
    procs = []
    for r in range(2):
        env = {k: v for k, v in os.environ.items() if k not in ("TFGPU_TEST_EMU_LIB", "TFGPU_RCCL_LIB")}
        env.update(LOCAL_RANK=str(r), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", case, str(tmp_path / "uid")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=env))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([ln for ln in o.strip().splitlines() if ln.startswith("{")][-1]))
    outs.sort(key=lambda d: d["rank"])
    if case == "mismatch":
        assert [o["error"] for o in outs] == [True, True]
        return
    for r, o in enumerate(outs):
        assert o["recv"] == [outs[s]["sent"][r] for s in range(2)]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_bench_devices_mode_runs_end_to_end_on_fake_devices():
    """`bench.py --devices 0,1` — ONE process driving several GPUs, a lane and a row-range shard each (tfgpu_init_devices) — has never had two real
    devices under it (the builder's and the driver's boxes hold one).  Here the whole command line runs on the emulator's two fake devices: argument
    handling, the per-device staging threads, the timed region, the JSON line with its process_model block.  A structural test: the line says so."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools", "hipemu"))
    import build as emu_build
    so = emu_build.build()
    env = dict(os.environ, TFGPU_TEST_EMU_LIB=so, HIPEMU_DEVICES="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--devices", "0,1", "--rows", "1536", "--steps", "2", "--passes", "1", "--warmup", "1", "--prof-steps", "1",
                        "--cpu-rows", "0", "--other-configs", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["process_model"]["devices"] == [0, 1] and d["process_model"]["row_range_shards"] == 2
    assert d["config"]["rows_per_pass"] == 1536 and d["value"] > 0 and "emulated" in d
    assert d["kernels"].get("csv_parse_regular") or d["kernels"].get("csv_parse_tiles_general") or d["kernels"].get("csv_parse_listed")
