"""bench/cpu_workers.py — the host side of the cpu_baseline leg: which cores this process may use, persistent oracle workers on every one of
them (the reference's shape for several tables: one sink pipeline each), threads around ctypes calls."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def host_info():
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"nproc": os.cpu_count() or 1, "usable_cores": usable_cores(), "cpu_model": model}


def usable_cores():
    """Hardware threads this process may actually run on: the scheduler affinity mask capped by the cgroup CPU quota (a container on
    a 256-thread host is typically given a handful; 256 workers on an 8-CPU quota measure the quota, not the reference)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts and parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(q / int(f.read().split()[0]))))
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            pass
    return max(n, 1)


ALL_CORES_HELPER = r"""
# Persistent workers: each builds its own sample, loads the oracle and runs ONE warm-up pipeline before the clock starts
# (page faults, allocator growth and imports are not the reference's throughput), then all start together at a barrier and
# run `reps` parse+mask+filter pipelines each.  Prints the wall time of the common region and every worker's own seconds.
import json, multiprocessing as mp, sys, time
sys.path.insert(0, sys.argv[1])
nc, per, reps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
CHAIN = json.loads(sys.argv[5])
def worker(i, go, done, q):
    from transferia_amd import workload
    from oracle import oracle as ora
    schema, opts = workload.hits_schema(), workload.hits_csv_options()
    sample = workload.hits_csv(per, row0=i * per)
    def one():
        chain = [ora.Transformer(t, c) for t, c in CHAIN]
        r = ora.csv_parse(opts, schema, sample, "", "")
        ora.apply_chain(chain, r.batch, r.schema)
    one()  # warm-up
    go.wait()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    dt = time.perf_counter() - t0
    q.put(dt)
    done.wait()
ctx = mp.get_context("fork")
go, done, q = ctx.Barrier(nc + 1), ctx.Barrier(nc + 1), ctx.Queue()
ps = [ctx.Process(target=worker, args=(i, go, done, q)) for i in range(nc)]
for p in ps: p.start()
go.wait()
t0 = time.perf_counter()
secs = [q.get() for _ in range(nc)]
wall = time.perf_counter() - t0
done.wait()
for p in ps: p.join()
print(json.dumps({"wall": wall, "mean_worker_s": sum(secs) / len(secs), "max_worker_s": max(secs)}))
"""


def all_cores_csv(nc, per, reps, single_thread_rows_per_s, chain):
    """Every host core runs its own parse+mask+filter pipeline over its own rows (the reference's shape for several snapshot
    parts / tables: one sink pipeline each, load_snapshot.go:962).  The aggregate is printed only when a worker keeps at least
    half of the single-thread rate — otherwise the number measures fork / page-fault / SMT overhead, not the reference."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", ALL_CORES_HELPER, ROOT, str(nc), str(per), str(reps), json.dumps(chain)], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-300:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    per_worker = per * reps / d["mean_worker_s"]
    out = {"unit": "rows/s", "cores": nc, "per_worker_rows_per_s": round(per_worker, 1), "single_thread_rows_per_s": round(single_thread_rows_per_s, 1),
           "per_worker_vs_single_thread": round(per_worker / max(single_thread_rows_per_s, 1e-9), 3),
           "sample": f"{nc} persistent processes x {reps} passes x {per} rows after one warm-up pass each, one parse+mask+filter pipeline per hardware thread "
                     f"({d['wall']:.2f}s wall, {d['mean_worker_s']:.2f}s mean / {d['max_worker_s']:.2f}s max per worker)"}
    if per_worker >= 0.5 * single_thread_rows_per_s:
        out["value"] = round(per * reps * nc / d["wall"], 1)
    else:
        out["value"] = None
        out["refused"] = "a worker runs at less than half the single-thread rate: the aggregate would measure host contention (SMT, memory bandwidth, page faults), not the reference path"
        out["aggregate_if_printed"] = round(per * reps * nc / d["wall"], 1)
    return out


def run_threads(fn, parts):
    """fn(part) on one thread per part (the oracle runs inside ctypes calls, which drop the GIL); returns wall seconds."""
    errs = []

    def w(p):
        try:
            fn(p)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=w, args=(p,)) for p in parts]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    if errs:
        raise errs[0]
    return dt

