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
    print(json.dumps({{"rank": g.rank, "lo": lo, "hi": hi, "sha": hashlib.sha256(shard).hexdigest(), "len": len(shard), "dt": dt, "max": mx, "rows": rows}}))
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
    # shards in rank order == the single-process input
    whole = workload.hits_csv(3001, header=False)
    a = workload.hits_csv(outs[0]["hi"], row0=0, header=False)
    assert hashlib.sha256(a).hexdigest() == outs[0]["sha"]
    assert outs[0]["len"] + outs[1]["len"] == len(whole)
    assert hashlib.sha256(whole[outs[0]["len"]:]).hexdigest() == outs[1]["sha"]
