"""The N>1 path on CPU: two processes over gloo (127.0.0.1) exercise exactly what bench.py uses for multi-GPU —
row-range sharding with no data-path collective, the barrier, and the MAX-over-ranks of the wall time — and check that
the per-rank shards of the synthetic input concatenate to the single-process input (same order as one GPU)."""
import os
import socket
import subprocess
import sys
import textwrap

from transferia_amd import dist as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_in_order():
    for total in (0, 1, 7, 1 << 20, 100_000_007):
        for world in (1, 2, 3, 8):
            rs = [tdist.shard_range(total, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(hi - lo for lo, hi in rs) - min(hi - lo for lo, hi in rs) <= 1
    assert tdist.weak_shard(1 << 20, 3) == (3 << 20, 4 << 20)


WORKER = textwrap.dedent("""
    import hashlib, json, os, sys, time
    sys.path.insert(0, {root!r})
    from transferia_amd import dist as tdist, workload
    g = tdist.Group("gloo")
    assert g.world == 2
    total = 3001
    lo, hi = tdist.shard_range(total, g.world, g.rank)
    shard = workload.hits_csv(hi - lo, row0=lo, header=False)       # this rank's contiguous row range
    g.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (g.rank + 1))                                  # rank 1 is the slow one
    dt = time.perf_counter() - t0
    mx = g.max_seconds(dt)
    rows = g.sum_int(shard.count(b"\\n"))
    g.barrier()
    per_rank = g.all_gather_float(dt)   # what bench.py prints per rank (multi_gpu.per_rank_ms_per_step)
    seen = g.sum_int(1)
    print(json.dumps({{"rank": g.rank, "lo": lo, "hi": hi, "sha": hashlib.sha256(shard).hexdigest(), "len": len(shard), "dt": dt, "max": mx, "rows": rows, "per_rank": per_rank, "seen": seen}}))
    g.close()
""")


def test_two_ranks_over_gloo(tmp_path):
    from transferia_amd import workload
    import hashlib
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [o["rank"] for o in outs] == [0, 1]
    assert outs[0]["hi"] == outs[1]["lo"] and outs[0]["lo"] == 0 and outs[1]["hi"] == 3001
    # every rank sees the slowest rank's time; the job's rows are the sum of the shards
    assert abs(outs[0]["max"] - outs[1]["max"]) < 1e-9 and outs[0]["max"] >= outs[1]["dt"] - 1e-9 and outs[0]["max"] > outs[0]["dt"]
    assert outs[0]["rows"] == outs[1]["rows"] == 3001
    # every rank's own time reaches every rank, in rank order; the ranks count themselves
    assert outs[0]["per_rank"] == outs[1]["per_rank"] == [outs[0]["dt"], outs[1]["dt"]] and outs[0]["seen"] == outs[1]["seen"] == 2
    # shards in rank order == the single-process input
    whole = workload.hits_csv(3001, header=False)
    a = workload.hits_csv(outs[0]["hi"], row0=0, header=False)
    assert hashlib.sha256(a).hexdigest() == outs[0]["sha"]
    assert outs[0]["len"] + outs[1]["len"] == len(whole)
    assert hashlib.sha256(whole[outs[0]["len"]:]).hexdigest() == outs[1]["sha"]


EXCHANGE_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    sys.path.insert(0, os.path.join({root!r}, "tests"))
    import torch.distributed as dist
    from transferia_amd import abi, dist as tdist, partition
    from oracle import oracle as ora
    g = tdist.Group("gloo")
    rank, world = g.rank, g.world
    # every rank owns a contiguous range of a CDC-like table, hashes it to a destination rank with the reference's
    # sharder (oracle), groups by destination, exchanges
    schema = abi.Schema.of([["id", "int64", True], ["name", "utf8"], ["ts", "timestamp"], ["opt", "int32"]])
    n = 500 + 37 * rank
    lo = 1000 * rank
    rows = [[["int64", lo + i], ["string", ("name-%d" % (lo + i)) * ((lo + i) % 4)], ["time", (1_600_000_000 + lo + i, (lo + i) % 1000)],
             ["nil", None] if (lo + i) % 5 == 0 else ["int32", lo + i]] for i in range(n)]
    b = abi.batch_from_rows(schema, ["id", "name", "ts", "opt"], rows, "db", "t")
    sh = ora.Transformer("sharder_transformer", {{"shardsCount": str(world), "columns": {{"includeColumns": ["^id$"]}}}})
    part = sh.apply(b, schema).batch.part_id.astype(np.int64)
    order = np.argsort(part, kind="stable")
    grouped = abi.batch_from_rows(schema, ["id", "name", "ts", "opt"], [rows[i] for i in order], "db", "t")
    grouped.src_row = order.astype(np.int32)
    counts = [int((part == d).sum()) for d in range(world)]
    out = partition.exchange_host_batch(dist, grouped, counts)
    ids = out.col("id").values.tolist()
    names = [out.col("name").get_bytes(i).decode() for i in range(out.nrows)]
    opt = [None if not out.col("opt").is_valid(i) else int(out.col("opt").values[i]) for i in range(out.nrows)]
    nan = out.col("ts").nanos.tolist()
    print(json.dumps({{"rank": rank, "ids": ids, "names": names, "opt": opt, "nanos": nan, "recv": out.recv_counts, "sent": counts}}))
    g.close()
""")


def test_hash_partition_exchange_over_gloo(tmp_path, oracle):
    """configs[4]'s exchange step on CPU: two ranks hash their rows with the reference's sharder, group by destination
    and run the all-to-all; every row lands on the rank its PartID names, once, in source-rank then original order."""
    import json
    import zlib
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "xworker.py"
    script.write_text(EXCHANGE_WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    all_ids = [list(range(0, 500)), list(range(1000, 1537))]
    for r, o in enumerate(outs):
        # destination rule: CRC32_IEEE(SerializeToString(id)) % 2 (sharder.go:130-145)
        exp = [i for src in all_ids for i in src if zlib.crc32(str(i).encode()) % 2 == r]
        assert o["ids"] == exp
        assert o["names"] == [("name-%d" % i) * (i % 4) for i in exp]
        assert o["opt"] == [None if i % 5 == 0 else i for i in exp]
        assert o["nanos"] == [i % 1000 for i in exp]
        assert o["recv"] == [outs[0]["sent"][r], outs[1]["sent"][r]]
    assert sorted(outs[0]["ids"] + outs[1]["ids"]) == sorted(all_ids[0] + all_ids[1])


CDC_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    sys.path.insert(0, os.path.join({root!r}, "tests"))
    import torch.distributed as dist
    from transferia_amd import abi, dist as tdist, partition, workload
    from oracle import oracle as ora
    g = tdist.Group("gloo")
    rank, world = g.rank, g.world
    # one CDC stream (OldKeys on most updates / deletes, no PK changes: a key never moves between shards), every rank
    # owning a contiguous slice of it in stream order
    N = 4000
    whole, schema = workload.cdc_batch(N, p_pk_change=0.0)
    lo, hi = rank * N // world, (rank + 1) * N // world
    def rows_of(b, idx):
        out = abi.Batch([], len(idx), b.table_ns, b.table_name)
        for c in b.cols:
            if c.repr in abi.VAR_REPRS:
                parts = [c.get_bytes(int(i)) for i in idx]
                off = np.concatenate([[0], np.cumsum([len(x) for x in parts])]).astype(np.uint32)
                out.cols.append(abi.Column(c.name, c.dtype, c.repr, offsets=off, data=np.frombuffer(b"".join(parts) or b"\\0", np.uint8).copy()))
            else:
                out.cols.append(abi.Column(c.name, c.dtype, c.repr, values=c.values[idx].copy()))
        out.kind = b.kind[idx].copy()
        k = b.old_keys[0]
        out.old_keys = [abi.Column(k.name, k.dtype, k.repr, values=k.values[idx].copy(), validity=k.validity[idx].copy())]
        out.old_present = b.old_present[idx].copy()
        out.schema = schema
        return out
    mine = rows_of(whole, np.arange(lo, hi))
    sh = ora.Transformer("sharder_transformer", {{"shardsCount": str(world), "columns": {{"includeColumns": ["^id$"]}}}})
    part = sh.apply(mine, schema).batch.part_id.astype(np.int64)
    order = np.argsort(part, kind="stable")
    grouped = rows_of(mine, order)
    counts = [int((part == d).sum()) for d in range(world)]
    got = partition.exchange_host_batch(dist, grouped, counts)
    col = ora.collapse(got, schema).batch
    rows = sorted((int(col.col("id").values[i]), int(col.col("ver").values[i]), int(col.kind[i]), col.col("payload").get_bytes(i).decode()) for i in range(col.nrows))
    print(json.dumps({{"rank": rank, "rows": rows, "received": got.nrows, "old_present": int(got.old_present.sum())}}))
    g.close()
""")


def test_exchange_then_collapse_over_gloo(tmp_path, oracle):
    """configs[4] end to end on CPU: hash-partition → all-to-all (OldKeys travel with their rows) → Collapse per rank.
    The union of the per-rank results is the Collapse of the whole stream: rows of one key meet on one rank, in stream order."""
    import json
    import zlib

    from transferia_amd import workload
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "cdcworker.py"
    script.write_text(CDC_WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    whole, schema = workload.cdc_batch(4000, p_pk_change=0.0)
    ref = oracle.collapse(whole, schema).batch
    exp = sorted((int(ref.col("id").values[i]), int(ref.col("ver").values[i]), int(ref.kind[i]), ref.col("payload").get_bytes(i).decode()) for i in range(ref.nrows))
    got = sorted([tuple(r) for o in outs for r in o["rows"]])
    assert got == exp and len(exp) > 500
    for r, o in enumerate(outs):  # every row sits on the rank its key hashes to (sharder.go:130-145)
        assert all(zlib.crc32(str(row[0]).encode()) % 2 == r for row in o["rows"])
    assert outs[0]["received"] + outs[1]["received"] == 4000
    assert outs[0]["old_present"] + outs[1]["old_present"] == int(whole.old_present.sum()) > 0
