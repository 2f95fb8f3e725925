#!/usr/bin/env python3
"""bench.py — the transform-stage hot path on ClickBench-hits-shaped batches, one MI355X per rank.

Default workload = BASELINE.json configs[1] ("ClickBench hits-1M CSV parse → mask(ip)+filter(EventDate) → devnull,
1×MI355X"): one *step* is one pass of that path over one HBM-resident batch of 2^20 synthetic hits rows per GPU.
Rows shard by range across ranks with no data-path collective (weak scaling: every rank works on its own batch);
torch.distributed (nccl = RCCL) is used for the barrier and the max-over-ranks timing — and, in configs[4] only,
for the one real exchange of the path (hash-partition all-to-all).

    python bench.py [--workload csv|configs2|configs3|configs4|json|sr|collapse] --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (DESIGN.md "Measurement" explains every field).  Every workload reports
`roofline` (dominant kernel: algorithmic bytes per launch ÷ its HIP-event launch time, against the 8 TB/s HBM peak)
and `cpu_baseline` (the oracle — a C restatement of the Go reference — on a bounded sample of the same input).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
INT32_PEAK_TOPS = 39.3    # 256 CUs x 64 lanes x 2.4 GHz, one VALU INSTRUCTION per lane per clock (same guide): an issue peak, so the work is priced in instructions
# mask_field per value: HMAC-SHA256 with cached ipad / opad midstates = 2 compressions.  The fewest gfx950 VALU instructions that
# compute one (v_alignbit_b32 = a rotate, v_xor3_b32, v_bfi_b32, v_add3_u32 each fold two or three two-input operations): a round is
# Sigma1 (3 rotates + xor3 = 4) + Ch (bfi = 1) + Sigma0 (4) + Maj (xor + bfi = 2) + t1 (two add3 = 2) + new e, new a (2) = 15; the
# message schedule of 48 rounds sigma0 (2 rotates + shift + xor3 = 4) + sigma1 (4) + add3 + add (2) = 10; 8 feed-forward adds:
# 64 x 15 + 48 x 10 + 8 = 1 448 per compression, 2 896 per value, plus ~250 for the decimal text, the padding, the byte swaps and
# the 64 hex characters.  (Until round 4 the tally counted two-input OPERATIONS, 4 400 — against an instruction peak that can exceed 1.)
# The kernel's MEASURED count is in profiles/pmc_traffic.json (issue_frac below): 3 900.
MASK_INT_OPS_PER_VALUE = 3150

MASK = ("mask_field", {"maskFunctionHash": {"userDefinedSalt": "clickbench-salt"}, "columns": ["clientip"]})
CHAIN = [MASK, ("filter_rows", {"filter": "eventdate >= 2013-07-15"})]
JSON_CHAIN = [MASK, ("filter_rows", {"filter": "regionid >= 40"})]


# ----------------------------------------------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------------------------------------------
def stage_shard(lib, workload, row0, nrows, chunk_rows=1 << 15):
    """Generate rows [row0,row0+nrows) chunk by chunk into one reused host buffer and assemble the CSV (header +
    rows) in HBM.  Returns (DeviceBuffer, nbytes, stream)."""
    hs = workload.HitsStream(chunk_rows=chunk_rows)
    sizes, r = [], row0
    while r < row0 + nrows:  # first pass: sizes only (the generator is deterministic)
        k = min(chunk_rows, row0 + nrows - r)
        sizes.append((r, k, hs.chunk(r, k)[1]))
        r += k
    total = len(hs.header) + sum(s[2] for s in sizes)
    dbuf = lib.DeviceBuffer.alloc(total)
    hdr = np.frombuffer(hs.header, dtype=np.uint8).copy()
    dbuf.write(0, hdr, len(hdr))
    off = len(hdr)
    for (r, k, n) in sizes:
        buf, n2 = hs.chunk(r, k)
        assert n2 == n
        dbuf.write(off, buf, n)
        off += n
    return dbuf, total, hs


def json_fields(workload, abi):
    """The hits schema as a generic-parser field list: date → utf8 text, timestamps → datetime (epoch seconds)."""
    out = []
    for name, typ, key in workload.hits_columns():
        t = {"date": "utf8", "timestamp": "datetime"}.get(typ, typ)
        out.append([name, t, bool(key)])
    return abi.Schema.of(out)


def make_messages(workload, nrows, row0=0):
    """One flat JSON object per hits row (ints as numbers, timestamps as epoch seconds, the rest as text)."""
    import calendar
    import csv
    import datetime
    import io
    cols = workload.hits_columns()
    hs = workload.HitsStream(chunk_rows=min(max(nrows, 1), 1 << 15))
    vals, r = [], row0
    while r < row0 + nrows:
        k = min(hs.chunk_rows, row0 + nrows - r)
        buf, n = hs.chunk(r, k)
        rd = csv.reader(io.StringIO(bytes(buf[:n]).decode("utf-8")))
        for row in rd:
            doc = {}
            for (name, typ, _), cell in zip(cols, row):
                if typ in ("int16", "int32", "int64"):
                    doc[name] = int(cell)
                elif typ == "timestamp":  # epoch seconds, the form extractTimeValue takes without dateparse
                    if cell.lstrip("-").isdigit():
                        doc[name] = int(cell)
                    else:
                        doc[name] = calendar.timegm(datetime.datetime.strptime(cell[:19].replace("T", " "), "%Y-%m-%d %H:%M:%S").timetuple())
                else:
                    doc[name] = cell
            vals.append(json.dumps(doc, ensure_ascii=False, separators=(",", ":")).encode("utf-8"))
        r += k
    return vals


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


def all_cores_csv(nc, per, reps, single_thread_rows_per_s):
    """Every host core runs its own parse+mask+filter pipeline over its own rows (the reference's shape for several snapshot
    parts / tables: one sink pipeline each, load_snapshot.go:962).  The aggregate is printed only when a worker keeps at least
    half of the single-thread rate — otherwise the number measures fork / page-fault / SMT overhead, not the reference."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", ALL_CORES_HELPER, ROOT, str(nc), str(per), str(reps), json.dumps(CHAIN)], capture_output=True, text=True, timeout=600)
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


# ----------------------------------------------------------------------------------------------------------------
# workloads: setup() stages inputs in HBM (untimed), step() is one pass, alg() the algorithmic bytes per step of each
# kernel (SURVEY §8d per-row figures x rows), cpu() the oracle on a bounded sample
# ----------------------------------------------------------------------------------------------------------------

# ----------------------------------------------------------------------------------------------------------------
# post-run parity: every workload line re-checks its own configuration against the oracle on a bounded sample (untimed; the
# product never calls the oracle — this is the cpu_baseline leg using its by-product)
# ----------------------------------------------------------------------------------------------------------------
def _cells_same(a, b):
    if a == b:
        return True
    if a[0] == "json" and b[0] in ("string", "bool", "jsonnum"):  # an `any` column holds json.Marshal's text of the oracle's Go value
        want = (b'"' + b[1] + b'"') if b[0] == "string" else (b"true" if b[1] else b"false") if b[0] == "bool" else b[1]
        return a[1] == want
    return a[0] == b[0] and a[0] in ("float32", "float64") and a[1] != a[1] and b[1] != b[1]  # NaN


def _rows_diff(abi, got_rows, want_rows):
    """index of the first row that differs (cells as abi.norm_value), or -1"""
    if len(got_rows) != len(want_rows):
        return min(len(got_rows), len(want_rows))
    for i, (g, w) in enumerate(zip(got_rows, want_rows)):
        if len(g) != len(w) or not all(_cells_same(x, y) for x, y in zip(g, w)):
            return i
    return -1


def _batch_diff(abi, dev, ref):
    """None when two host batches hold the same columns and cells, else what differs first"""
    if [c.name for c in dev.cols] != [c.name for c in ref.cols]:
        return "column names"
    if [c.repr for c in dev.cols] != [c.repr for c in ref.cols]:
        return "column representations"
    d = _rows_diff(abi, abi.batch_rows(dev), abi.batch_rows(ref))
    return None if d < 0 else "row %d" % d


def _parity(k, what, err=None, **kw):
    out = {"identical": err is None, "checked_input_rows": k, "checked": what}
    if err is not None:
        out["error"] = str(err)[:300]
    out.update(kw)
    return out


def _test_helpers(name):
    """the GPU tests' own comparison helpers (tests/ travels with the tree): the bench lines check themselves with the code the suite uses"""
    t = os.path.join(ROOT, "tests")
    if t not in sys.path:
        sys.path.insert(0, t)
    import importlib
    return importlib.import_module(name)


def _guard_parity(fn):
    try:
        return fn()
    except Exception as ex:  # noqa: BLE001
        return {"identical": False, "error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}

class ctypes_void:
    """a void* out-parameter for the few raw C-ABI calls bench.py makes itself"""

    def __init__(self):
        import ctypes
        self._p = ctypes.c_void_p()
        self.ref = ctypes.byref(self._p)

    @property
    def value(self):
        return self._p.value


class Base:
    metric = ""
    scaling = "weak"

    def __init__(self, args, env):
        self.args, self.env, self.state = args, env, {}

    def extra(self):
        return {}

    def timed(self, steps):
        """K steps bracketed by barrier + device sync on both sides; returns wall seconds of THIS rank."""
        e = self.env
        e.sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        e.sync_all()
        return time.perf_counter() - t0

    def int_roofline(self, kernels):
        return None


class CsvWorkload(Base):
    """BASELINE.json configs[1]."""
    metric = "ChangeItems/sec through CSV parse -> mask(ClientIP)+filter(EventDate) -> devnull, ClickBench hits, HBM-resident input (the PCIe-inclusive rate is `pcie_inclusive`)"
    default_rows = 1 << 20

    def setup(self):
        e, a = self.env, self.args
        self.schema = e.workload.hits_schema()
        self.cschema = self.schema.to_c()  # the tfgpu_schema, converted once like a Go caller's (0.5 ms of Python per call otherwise)
        self.opts = e.workload.hits_csv_options()
        self.plans = [e.lib.Transformer(t, c) for t, c in CHAIN]
        self.row0 = a.rows * e.rank
        self.dbuf, self.csv_bytes, _ = stage_shard(e.lib, e.workload, self.row0, a.rows)
        self.nl = 1

    def step(self, keep=False, buf=None, plans=None, dense=False):
        lib = self.env.lib
        db, consumed, errs = lib.csv_parse(self.opts, self.cschema, buf if buf is not None else self.dbuf)
        res = lib.apply_chain(plans if plans is not None else self.plans, db)
        # the devnull sink: counts the items it is pushed and drops them.  filter_rows hands its kept rows on as a SELECTION over the
        # parsed batch (tfgpu_dbatch::pending) and nothing here reads a column of them, so no dense copy of the kept rows is made;
        # `dense` (the dense_sink side measurement) is a sink that asks for one
        self.sunk = res.transformed.nrows
        if dense:
            res.transformed.dense()
        if keep:
            s = self.state
            s["parsed_rows"], s["parsed_bytes"] = db.nrows, db.payload_bytes()
            v = db.view()
            fixed = strb = 0
            for i in range(v.ncols):
                c = v.cols[i]
                if c.repr in (12, 13, 14, 15):
                    strb += int(c.data_len)
                else:
                    fixed += int(v.nrows) * (np.dtype(lib.abi.REPR_NP[c.repr]).itemsize + (4 if c.nanos else 0))
            s["fixed_bytes"], s["str_bytes"], s["nstr"] = fixed, strb, sum(1 for i in range(v.ncols) if v.cols[i].repr in (12, 13, 14, 15))
            s["out_rows"], s["out_bytes"] = res.transformed.nrows, res.transformed.payload_bytes()
            s["errors"] = len(errs) + len(res.errors)
        res.transformed.free()
        db.free()  # devnull sink

    def rows(self):
        return self.state["parsed_rows"]

    # the K timed steps, spread round-robin over `lanes` host threads, each bound to its own device lane (stream + HBM
    # cache) and reading its own HBM-resident copy of the shard; lanes = 1 is the strictly serial bench line
    def timed_devices(self, steps):
        """One process, several devices (tfgpu_init_devices): lane k lives on the k-th device of --devices, stages ITS row-range
        shard there and runs every one of the K steps over it — the same weak-scaling job as one process per GPU, driven by one
        worker with a thread per device.  No data-path collective; the lanes meet at two host barriers."""
        e, a = self.env, self.args
        nl = len(e.devices)
        go, done = threading.Barrier(nl + 1), threading.Barrier(nl + 1)
        errs = []

        def lane_main(k):
            try:
                e.lib.lane_use(k)
                buf = self.dbuf if k == 0 else stage_shard(e.lib, e.workload, self.row0 + a.rows * k, a.rows)[0]
                plans = self.plans if k == 0 else [e.lib.Transformer(t, c) for t, c in CHAIN]
                for _ in range(max(min(a.warmup, 2), 1)):
                    self.step(buf=buf, plans=plans)
                e.lib.synchronize()
                go.wait()
                for _ in range(steps):
                    self.step(buf=buf, plans=plans)
                e.lib.synchronize()
                done.wait()
            except Exception as ex:  # noqa: BLE001
                errs.append(ex)
                go.abort(); done.abort()
        threads = [threading.Thread(target=lane_main, args=(k,)) for k in range(nl)]
        for t in threads:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        done.wait()
        dt = time.perf_counter() - t0
        for t in threads:
            t.join()
        if errs:
            raise errs[0]
        e.lib.lane_use(0)
        return dt

    def timed(self, steps, nlanes=None, host_bufs=None):
        e, a = self.env, self.args
        if getattr(e, "devices", None) and nlanes is None and host_bufs is None:
            return self.timed_devices(steps)
        nl = max(1, min(nlanes or a.lanes, steps, e.lib.lane_count()))
        go, done = threading.Barrier(nl + 1), threading.Barrier(nl + 1)
        lane_err = []

        def lane_main(k):
            try:
                e.lib.lane_use(k)
                if host_bufs is not None:
                    buf = host_bufs[k]
                else:
                    buf = self.dbuf if k == 0 else stage_shard(e.lib, e.workload, self.row0, a.rows)[0]
                if k or host_bufs is not None:
                    for _ in range(max(min(a.warmup, 2), 1)):
                        self.step(buf=buf)
                e.lib.synchronize()
                go.wait()
                for i in range(steps):
                    if i % nl == k:
                        self.step(buf=buf)
                e.lib.synchronize()
                done.wait()
            except Exception as ex:  # noqa: BLE001
                lane_err.append(ex)
                go.abort(); done.abort()

        threads = [threading.Thread(target=lane_main, args=(k,)) for k in range(nl)]
        for t in threads:
            t.start()
        e.sync_all()
        go.wait()
        t0 = time.perf_counter()
        done.wait()
        e.sync_all()
        dt = time.perf_counter() - t0
        for t in threads:
            t.join()
        if lane_err:
            raise lane_err[0]
        e.lib.lane_use(0)
        self.nl = nl if host_bufs is None and nlanes is None else self.nl
        return dt

    def mask_rows(self, kernels=None):
        """Rows the mask kernel is LAUNCHED on: the library's own count (tfgpu_prof_get_units) — behind a hoisted filter_rows
        (tf_transform.hip chain_sequence) that is the kept rows, not the parsed rows."""
        k = (kernels or getattr(self, "kernels", None) or {}).get("mask_hmac_sha256") or {}
        if not k.get("units_per_step"):
            raise RuntimeError("the library reported no row count for mask_hmac_sha256: its roofline cannot be priced")
        return k["units_per_step"]

    def alg(self):
        s = self.state
        rows, sel = s["parsed_rows"], s["out_rows"] / max(s["parsed_rows"], 1)
        # compaction: fixed-width values, and for the late-materialised text columns 8 bytes per cell (length + position)
        gather = int((1 + sel) * (s["fixed_bytes"] + 8 * s["nstr"] * rows))
        return {"csv_count_newlines": self.csv_bytes, "csv_parse_regular": self.csv_bytes + s["fixed_bytes"],  # read every input byte once, write every fixed-width value once
                "csv_parse_rows": self.csv_bytes + s["fixed_bytes"], "csv_copy_words": 2 * s["str_bytes"], "scan_u32_segments": 12 * rows * s["nstr"],
                "mask_hmac_sha256": 72 * self.mask_rows(),      # 4 B in + 64 B hex + 4 B offset per value it is launched on
                "filter_rows_eval": int((8 + 4 + 0.125) * rows), "compact_gather": gather, "scan_u32": 12 * rows}

    def alg_views(self):
        s = self.state
        return {"csv_parse_regular": 8 * s["nstr"] * s["parsed_rows"], "csv_parse_rows": 8 * s["nstr"] * s["parsed_rows"]}

    def int_roofline(self, kernels):
        k = kernels.get("mask_hmac_sha256")
        if not k:
            return None
        mrows = self.mask_rows(kernels)
        ach = MASK_INT_OPS_PER_VALUE * mrows / max(k["launches_per_step"], 1) / (k["avg_ms"] * 1e-3) / 1e12
        out = {"kernel": "mask_hmac_sha256", "bound": "int32_valu", "achieved": round(ach, 2), "peak": INT32_PEAK_TOPS, "unit": "Top/s",
               "frac": round(ach / INT32_PEAK_TOPS, 4), "int_ops_per_value": MASK_INT_OPS_PER_VALUE, "values_per_step": mrows,
               "note": "mask is ALU-bound (2 SHA-256 compressions per 72 algorithmic bytes): its HBM fraction is legitimately low.  int_ops_per_value is the FEWEST gfx950 VALU "
                       "instructions that compute one value (v_add3 / v_xor3 / v_bfi / v_alignbit counted as one each), `peak` one VALU instruction per lane per clock: `frac` is the share "
                       "of the issue slots that minimum would fill.  `issue_frac` prices the kernel's MEASURED instruction count instead (when profiles/pmc_traffic.json is of this build)"}
        # the same fraction from the kernel's MEASURED VALU instruction count per value (rocprofv3 --pmc SQ_INSTS_VALU / SQ_WAVES of this
        # very source, tools/gpu_visit.sh evidence) instead of the algorithmic tally: instructions issued, against one per lane per cycle
        try:
            import hashlib
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f).get("mask_hmac_sha256")
            with open(os.path.join(ROOT, "transferia_amd", "csrc", t["source_file"]), "rb") as f:
                if hashlib.sha256(f.read()).hexdigest() == t["source_sha256"]:
                    per = t["valu_instructions_per_value"]
                    issued = per * mrows / max(k["launches_per_step"], 1) / (k["avg_ms"] * 1e-3) / 1e12
                    out["measured_valu_instructions_per_value"] = per
                    out["issued"] = round(issued, 2)
                    out["issue_frac"] = round(issued / INT32_PEAK_TOPS, 4)
        except (OSError, ValueError, KeyError, TypeError):
            pass
        return out

    def config(self):
        s, e = self.state, self.env
        return {"workload": "ClickBench hits-1M CSV parse -> mask(ip)+filter(EventDate) -> devnull (BASELINE.json configs[1])",
                "rows_per_gpu_per_step": s["parsed_rows"], "csv_bytes_per_gpu_per_step": self.csv_bytes, "columns": len(self.schema.cols),
                "chain": [t for t, _ in CHAIN], "filter_selectivity": round(s["out_rows"] / max(s["parsed_rows"], 1), 4),
                "sink": "devnull: counts the pushed rows (tfgpu_dbatch_nrows) and drops them; the kept rows stay a selection over the parsed batch, no dense copy is made (side measurement dense_sink: a sink that asks for one)",
                "parallelism": f"row-range shard x{len(e.devices) if getattr(e, 'devices', None) else e.world}, no collective"}

    def extra(self):
        s, e, a = self.state, self.env, self.args
        out = {"lanes": self.nl}
        if getattr(self, "dt", None):
            out["gib_per_s_csv_in"] = round(self.csv_bytes * e.world * self.total_passes / self.dt / 2**30, 3)
            out["gib_per_s_deepsizeof"] = round((s["parsed_bytes"] + 16 * len(self.schema.cols) * s["parsed_rows"]) * e.world * self.total_passes / self.dt / 2**30, 3)
        return out

    def side_measurements(self):
        """Never `value`: the same steps over several device lanes, and starting from pinned host memory (PCIe inside)."""
        e, a = self.env, self.args
        out = {}
        k = max(min(getattr(self, 'total_passes', a.steps), 200), 3)
        if a.overlap_lanes > 1 and a.overlap_lanes != a.lanes:
            dt2 = e.group.max_seconds(self.timed(k, nlanes=a.overlap_lanes))
            out["overlapped_lanes"] = {"lanes": min(a.overlap_lanes, e.lib.lane_count()), "steps": k, "rows_per_s": round(a.rows * e.world * k / dt2, 1), "ms_per_step": round(dt2 / k * 1e3, 3),
                                       "note": "the same steps spread over several device lanes (parse of batch N+1 beside the transform of batch N); "
                                               "kernels of different lanes share the GPU, so per-kernel spans are not comparable"}
        if e.rank == 0:  # a sink that reads every column: the kept rows are gathered (what every step did until round 5)
            e.lib.synchronize()
            kk = max(min(k, 50), 3)
            for _ in range(2):
                self.step(dense=True)
            e.lib.synchronize()
            t0 = time.perf_counter()
            for _ in range(kk):
                self.step(dense=True)
            e.lib.synchronize()
            dtd = time.perf_counter() - t0
            out["dense_sink"] = {"steps": kk, "ms_per_step": round(dtd / kk * 1e3, 3), "rows_per_s": round(a.rows * kk / dtd, 1),
                                 "note": "the same pass with a sink that asks for the kept rows as dense columns (tfgpu_dbatch_dense: compact_gather over all 105 columns, text cells still "
                                         "positions in the CSV) — the devnull sink of configs[1] does not"}
        if a.pcie_steps > 0 and e.rank == 0:
            raw = self.dbuf.download()
            res = {}
            for nl in sorted({1, max(1, min(a.pcie_lanes, e.lib.lane_count()))}):
                hosts = [e.lib.HostBuffer(raw) for _ in range(nl)]
                kk = a.pcie_steps * nl
                dth = self.timed(kk, nlanes=nl, host_bufs=hosts)
                res[f"lanes_{nl}"] = {"rows_per_s": round(a.rows * kk / dth, 1), "gb_per_s_h2d": round(self.csv_bytes * kk / dth / 1e9, 2), "ms_per_step": round(dth / kk * 1e3, 3)}
                for h in hosts:
                    h.free()
            res["note"] = ("input in pinned host memory (hipHostMalloc), hipMemcpyAsync on each lane's stream inside the step: with several lanes the H2D of "
                           "one batch runs beside the kernels of another; PCIe Gen5 x16 bounds this at ~55-60 GB/s")
            out["pcie_inclusive"] = res
        if a.pcie_steps > 0 and e.rank == 0 and e.world == 1 and not a.no_pull_push:
            # configs[1]'s sink is devnull: nothing to push.  The pull AND the push overlapped is configs[2]'s job ("async double-buffer"), measured
            # here too so that the default line carries it: that workload's three-stage pipeline (one puller, two transform lanes, one pusher)
            try:
                import copy
                a2 = copy.copy(a)
                a2.rows = 0
                W2 = Configs2Workload(a2, e)
                a2.rows = W2.default_rows
                W2.setup()
                W2.step(); W2.step(keep=True)
                e.lib.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    W2.step()
                e.lib.synchronize()
                W2.dt, W2.total_passes = time.perf_counter() - t0, 5
                pp = W2.side_measurements(only_pipeline=True).get("pull_push_concurrent", {})
                out["configs2_pull_push"] = {"workload": "configs[2]: SR wire bytes -> replace_primary_key + sql -> JSONEachRow, %d rows a batch" % W2.n,
                                             "hbm_resident_ms_per_step": round(W2.dt / 5 * 1e3, 3), "copies_alone_concurrent": pp.get("copies_alone_concurrent"),
                                             "pipeline_3_stage": pp.get("pipeline_3_stage")}
            except Exception as ex:  # noqa: BLE001
                out["configs2_pull_push"] = {"error": str(ex)[:200]}
        return out

    def parity_against(self, ref, n1):
        """One device step over the bench's shard; its output rows whose source row is below n1 against `ref`, the oracle's
        parse + mask + filter of the first n1 rows of the same CSV."""
        lib, abi = self.env.lib, self.env.abi
        db, _, errs = lib.csv_parse(self.opts, self.schema, self.dbuf)
        res = lib.apply_chain(self.plans, db)
        out = res.transformed.download()
        res.transformed.free(); db.free()
        src = out.src_row if out.src_row is not None else np.arange(out.nrows, dtype=np.int32)
        m = int(np.searchsorted(src, n1))  # kept rows are in input order
        if m != ref.nrows or not np.array_equal(src[:m], ref.src_row):
            return {"identical": False, "checked_input_rows": n1, "error": "kept rows differ: %d vs %d" % (m, ref.nrows)}
        for a, b in zip(out.cols, ref.cols):
            ok = a.name == b.name and a.repr == b.repr
            if ok and a.repr in abi.VAR_REPRS:
                end = int(a.offsets[m])
                ok = np.array_equal(a.offsets[:m + 1], b.offsets[:m + 1]) and bytes(a.data[:end]) == bytes(b.data[:end])
            elif ok:
                ok = np.array_equal(a.values[:m], b.values[:m]) and (a.nanos is None or b.nanos is None or np.array_equal(a.nanos[:m], b.nanos[:m]))
            if ok and (a.validity is not None or b.validity is not None):
                va = a.validity[:m] if a.validity is not None else np.ones(m, bool)
                vb = b.validity[:m] if b.validity is not None else np.ones(m, bool)
                ok = np.array_equal(va, vb)
            if not ok:
                return {"identical": False, "checked_input_rows": n1, "error": "column %s differs" % a.name}
        return {"identical": True, "checked_input_rows": n1, "compared_output_rows": m, "columns": len(out.cols), "row_errors": len(errs) + len(res.errors)}

    def cpu(self):
        e, a = self.env, self.args
        from oracle import oracle as ora
        n1 = a.cpu_rows
        sample = e.workload.hits_csv(n1)
        ochain = [ora.Transformer(t, c) for t, c in CHAIN]
        r1 = ora.csv_parse(self.opts, self.schema, sample, "", "")
        r2 = ora.apply_chain(ochain, r1.batch, r1.schema)
        secs = r1.seconds + r2.seconds
        out = {"value": round(n1 / secs, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{n1} rows of the same synthetic hits CSV: oracle csv parse+strictify ({r1.seconds:.2f}s) + mask+filter ({r2.seconds:.2f}s), "
                         "single thread = the reference's shape for one table (transformation.go:131-135: one goroutine per table)",
               "note": "C restatement of the Go reference (row-oriented boxed values), not the Go binary; expect the Go binary to be ~2-3x faster per core "
                       "(BenchmarkTextFetcher: a 105-column row in ~6 us including parse)"}
        out.update(host_info())
        # The oracle's result over these n1 rows is also the checker of the bench's own output: one more (untimed) device step,
        # its kept rows that come from the first n1 input rows compared value for value (the product never calls the oracle;
        # this is bench.py's cpu_baseline leg using its by-product).
        try:
            out["parity"] = self.parity_against(r2.batch, n1)
        except Exception as ex:  # noqa: BLE001
            out["parity"] = {"identical": False, "error": str(ex)[:300]}
        # leg (ii): every host core, the reference's parallelism shape for SEVERAL tables / snapshot parts (one sink pipeline each,
        # load_snapshot.go:962): independent slices, one oracle pipeline per core, each in its own process (a clean interpreter
        # forks the workers: no GIL between them, no HIP state inherited)
        nc = out["usable_cores"]
        if nc > 1 and a.cpu_all_rows > 0:
            reps = 4
            per = 16384  # 65 536 rows per worker over the timed passes
            try:
                out["all_cores"] = all_cores_csv(nc, per, reps, out["value"])
            except Exception as ex:  # noqa: BLE001
                out["all_cores"] = {"error": str(ex)[:200]}
        return out


class _Prepared(Base):
    """Helpers shared by the side workloads: messages staged once, roofline bytes recorded by a keep step."""
    default_rows = 1 << 18

    def rows(self):
        return self.n


class JsonWorkload(_Prepared):
    metric = "ChangeItems/sec through Kafka JSON parse -> mask(ClientIP)+filter -> ClickHouse JSONEachRow, ClickBench hits"

    def setup(self):
        e, a = self.env, self.args
        abi, lib = e.abi, e.lib
        self.fields = json_fields(e.workload, abi)
        self.vals = make_messages(e.workload, a.rows, a.rows * e.rank)
        self.n = len(self.vals)
        self.data, self.msgs = abi.messages(self.vals, list(range(self.n)), [1_700_000_000_000_000_000 + i for i in range(self.n)])
        self.opts = abi.json_options(topic="hits", add_dedupe_keys=True, add_rest=True, partition='{"partition":0,"topic":"hits"}')
        self.dbuf = lib.DeviceBuffer.upload(self.data)
        self.plans = [lib.Transformer(t, c) for t, c in JSON_CHAIN]

    def step(self, keep=False):
        lib, abi = self.env.lib, self.env.abi
        db, errs = lib.json_parse(self.opts, self.fields, self.dbuf, self.msgs)
        res = lib.apply_chain(self.plans, db)
        out = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, res.transformed)
        if keep:
            self.state.update(parsed_bytes=db.payload_bytes(), out_rows=res.transformed.nrows, out_bytes=out.size, kept_bytes=res.transformed.payload_bytes(),
                              errors=len(errs) + len(res.errors))
        out.free(); res.transformed.free(); db.free()

    def alg(self):
        s = self.state
        return {"json_parse_quick": len(self.data) + s["parsed_bytes"], "json_parse_tiles": len(self.data) + s["parsed_bytes"], "json_parse_lines": len(self.data) + s["parsed_bytes"], "csv_count_newlines": len(self.data), "ser_chunk_write": s["kept_bytes"] + s["out_bytes"], "ser_chunk_len": s["kept_bytes"], "ser_cell_write": s["kept_bytes"] + s["out_bytes"],
                "ser_cell_len": s["kept_bytes"], "json_copy_cells": 2 * s["parsed_bytes"]}

    def config(self):
        return {"workload": "Kafka JSON (one flat hits object per message) -> generic parser -> mask+filter -> JSONEachRow (BASELINE.json configs[2] shape, generic-parser flavour)",
                "rows_per_gpu_per_step": self.n, "json_bytes_per_step": len(self.data), "columns": len(self.fields.cols)}

    def extra(self):
        s = self.state
        return {"rows_out_per_step": s["out_rows"], "text_out_bytes_per_step": s["out_bytes"]}

    def cpu(self):
        from oracle import oracle as ora
        abi = self.env.abi
        k = min(self.args.cpu_rows, self.n, 1 << 13)
        d2, m2 = abi.messages(self.vals[:k], list(range(k)), [0] * k)
        r1 = ora.json_parse(self.opts, self.fields, d2, m2, want_rows=False)
        out = {"value": round(k / r1.seconds, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} of the same messages through the oracle's generic JSON parser only ({r1.seconds:.2f}s), single thread",
               "note": "C restatement of the Go reference (fastjson value tree, boxed values per row), not the Go binary"}
        out.update(host_info())
        out["parity"] = _guard_parity(lambda: self.parity(min(k, 4096)))
        return out

    def parity(self, k):
        """parse of the first k messages against the oracle's rows; then mask + filter + JSONEachRow of that batch, device against oracle"""
        from oracle import oracle as ora
        lib, abi = self.env.lib, self.env.abi
        d2, m2 = abi.messages(self.vals[:k], list(range(k)), [1_700_000_000_000_000_000 + i for i in range(k)])
        ref = ora.json_parse(self.opts, self.fields, d2, m2)
        db, errs = lib.json_parse(self.opts, self.fields, d2, m2)
        host = db.download()
        if errs or host.nrows != ref.nrows:
            return _parity(k, "parse", "rows %d vs %d, %d device errors" % (host.nrows, ref.nrows, len(errs)))
        d = _rows_diff(abi, abi.batch_rows(host), [[abi.norm_value(v) for v in r] for r in ref.rows])
        if d >= 0:
            return _parity(k, "parse", "row %d differs" % d)
        res = lib.apply_chain(self.plans, db)
        text = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, res.transformed).download()
        host.schema = ref.schema
        r2 = ora.apply_chain([ora.Transformer(t, c) for t, c in JSON_CHAIN], host, ref.schema)
        want = ora.serialize(abi.FMT_CH_JSON_EACH_ROW, r2.batch, r2.schema)
        if bytes(text) != bytes(want):
            return _parity(k, "chain + JSONEachRow", "text differs (%d vs %d bytes)" % (len(text), len(want)))
        return _parity(k, "json parse (cell for cell) + mask + filter + JSONEachRow (byte for byte)", compared_output_rows=r2.batch.nrows, compared_output_bytes=len(want))


def sr_inputs(e, a):
    """Kafka messages in the Schema Registry wire format: 0x00 | BE schema id | one flat hits object (JSON schema of hits)."""
    from transferia_amd import confluent_sr
    vals = make_messages(e.workload, a.rows, a.rows * e.rank)
    jt = {"int16": "integer", "int32": "integer", "int64": "integer", "timestamp": "integer"}
    props = {name: {"type": jt.get(typ, "string")} for name, typ, _ in e.workload.hits_columns()}
    schema_text = json.dumps({"type": "object", "title": "default.hits", "properties": props, "required": ["watchid"]})
    sid = 42
    frames = [b"\0" + sid.to_bytes(4, "big") + v for v in vals]
    data, msgs = e.abi.messages(frames, list(range(len(frames))), [1_700_000_000_000_000_000 + i for i in range(len(frames))])
    return frames, data, msgs, confluent_sr.sr_json_options(sid, schema_text), len(props)


class SrWorkload(_Prepared):
    metric = "ChangeItems/sec through Confluent-SR JSON parse -> queue JSON serializer (batched), ClickBench hits"

    def setup(self):
        e, a = self.env, self.args
        self.frames, self.data, self.msgs, self.opts, self.ncols = sr_inputs(e, a)
        self.n = len(self.frames)
        self.qopts = e.abi.queue_options(e.abi.QFMT_JSON, enabled=True, max_message_size=1 << 20)
        self.dbuf = e.lib.DeviceBuffer.upload(self.data)

    def step(self, keep=False):
        lib = self.env.lib
        res = lib.sr_json_parse(self.opts, self.dbuf, self.msgs)
        out = lib.queue_serialize(self.qopts, res.device_batch)
        if keep:
            self.state.update(parsed=res.device_batch.nrows, parsed_bytes=res.device_batch.payload_bytes(), out_bytes=out.values.size, messages=len(out), errors=len(res.errors))
        out.values.free(); res.device_batch.free()

    def alg(self):
        s = self.state
        return {"sr_parse_quick": len(self.data), "sr_parse_tiles": len(self.data), "sr_parse_frames": len(self.data), "sr_cell_values": len(self.data) + s["parsed_bytes"], "sr_cell_text": 2 * s["parsed_bytes"],
                "ser_cell_write": s["parsed_bytes"] + s["out_bytes"], "ser_cell_len": s["parsed_bytes"],
                "ser_chunk_write": s["parsed_bytes"] + s["out_bytes"], "ser_chunk_len": s["parsed_bytes"]}

    def config(self):
        return {"workload": "Kafka messages in the Schema Registry wire format (one flat hits object each) -> SR JSON parser -> queue JSON serializer, 1 MiB batches "
                            "(the ingest of BASELINE.json configs[2] and the sink half of configs[4])",
                "rows_per_gpu_per_step": self.n, "wire_bytes_per_step": len(self.data), "columns": self.ncols}

    def extra(self):
        s = self.state
        return {"rows_out_per_step": s["parsed"], "messages_out_per_step": s["messages"], "text_out_bytes_per_step": s["out_bytes"]}

    def cpu(self):
        from oracle import oracle as ora
        abi = self.env.abi
        k = min(self.args.cpu_rows, self.n, 1 << 13)
        d2, m2 = abi.messages(self.frames[:k], list(range(k)), [0] * k)
        r1 = ora.sr_json_parse(self.opts, d2, m2)
        ora.queue_serialize(self.qopts, r1.batch, r1.schema)
        sec = r1.seconds + ora.queue_serialize.seconds
        out = {"value": round(k / sec, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} of the same messages through the oracle's SR JSON parser ({r1.seconds:.2f}s) + queue JSON serializer ({ora.queue_serialize.seconds:.2f}s), single thread",
               "note": "C restatement of the Go reference (decoded value tree, boxed values per row), not the Go binary"}
        out.update(host_info())
        out["parity"] = _guard_parity(lambda: self.parity(min(k, 4096)))
        return out

    def parity(self, k):
        from oracle import oracle as ora
        lib, abi = self.env.lib, self.env.abi
        d2, m2 = abi.messages(self.frames[:k], list(range(k)), [1_700_000_000_000_000_000 + i for i in range(k)])
        ref = ora.sr_json_parse(self.opts, d2, m2)
        res = lib.sr_json_parse(self.opts, d2, m2)
        host = res.device_batch.download()
        if res.errors or ref.errors:
            return _parity(k, "parse", "row errors: %d device, %d oracle" % (len(res.errors), len(ref.errors)))
        why = _batch_diff(abi, host, ref.batch)
        if why:
            return _parity(k, "parse", why)
        got = lib.queue_serialize(self.qopts, res.device_batch)
        want = ora.queue_serialize(self.qopts, ref.batch, ref.schema)
        text = bytes(got.values.download())
        if want is None or text != b"".join(want) or len(got) != len(want):
            return _parity(k, "queue JSON serializer", "messages differ")
        return _parity(k, "SR JSON parse (cell for cell) + queue JSON serializer (byte for byte)", compared_messages=len(want), compared_output_bytes=len(text))


class Configs2Workload(_Prepared):
    """BASELINE.json configs[2] as ONE chain: Kafka JSON under a Confluent-SR JSON schema → the `sql` transformer (predicate +
    casts) → ClickHouse JSONEachRow.  The reference's sql transformer shells out to clickhouse-local
    (clickhouse_local.go:97-143) and needs a primary key in the table (ResultSchema :417-419), which a Confluent-SR JSON table
    does not have: replace_primary_key (the reference's own transformer for that, transformation_test.go:29-111) puts one in
    front.  The query stays inside the device subset documented in transferia_amd/csrc/tf_sql.cpp."""
    metric = "ChangeItems/sec through Confluent-SR JSON parse -> replace_primary_key + sql (predicate + casts) -> ClickHouse JSONEachRow, ClickBench hits"
    QUERY = ("select *, toString(userid) as userid_s, toString(counterid) as counterid_s, toInt32(regionid) as region32, toDateTime(eventtime) as eventtime_dt "
             "from table where regionid >= 40")
    CH = [("replace_primary_key", {"keys": ["watchid"], "tables": {}}), ("sql", {"tables": {"include_tables": [".*"]}, "query": QUERY})]
    # the CPU leg's chain: the oracle restates the sql subset in Python (oracle/ora_sql.py), so the timed C leg runs the same
    # predicate and casts through the stock transformers with the same row semantics (filter_rows + convert_to_string)
    CPU_CH = [("filter_rows", {"filter": "regionid >= 40"}), ("convert_to_string", {"columns": {"includeColumns": ["^userid$", "^counterid$"]}, "tables": {}})]

    def setup(self):
        e, a = self.env, self.args
        self.frames, self.data, self.msgs, self.opts, self.ncols = sr_inputs(e, a)
        self.n = len(self.frames)
        self.dbuf = e.lib.DeviceBuffer.upload(self.data)
        self.plans = [e.lib.Transformer(t, c) for t, c in self.CH]
        self.stage = e.lib.Transformation(self.plans)  # transformation.Push: the table plan = the Suitable transformers (transformation.go:46-85)

    def step(self, keep=False):
        lib, abi = self.env.lib, self.env.abi
        res = lib.sr_json_parse(self.opts, self.dbuf, self.msgs)
        db = res.device_batch
        tr = self.stage.push_run(db)
        out = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, tr.transformed)
        if keep:
            v = db.view()
            sch = abi.Schema.of([[v.cols[i].name.decode(), abi.DTYPES[v.cols[i].dtype], False] for i in range(v.ncols)])
            self.state.update(parsed_bytes=db.payload_bytes(), out_rows=tr.transformed.nrows, kept_bytes=tr.transformed.payload_bytes(), out_bytes=out.size,
                              errors=len(res.errors) + len(tr.errors), table_plan=[self.CH[i][0] for i in self.stage.table_plan((v.table_ns or b"").decode(), (v.table_name or b"").decode(), sch)])
        for _, eb in tr.error_batches:
            eb.free()
        out.free(); tr.transformed.free(); db.free()

    def alg(self):
        s = self.state
        return {"sr_parse_quick": len(self.data), "sr_parse_tiles": len(self.data), "sr_parse_frames": len(self.data), "sr_cell_values": len(self.data) + s["parsed_bytes"], "sr_cell_text": 2 * s["parsed_bytes"],
                "ser_chunk_write": s["kept_bytes"] + s["out_bytes"], "ser_chunk_len": s["kept_bytes"], "ser_cell_write": s["kept_bytes"] + s["out_bytes"], "ser_cell_len": s["kept_bytes"], "compact_gather": int(s["parsed_bytes"] + s["kept_bytes"])}

    def config(self):
        return {"workload": "Kafka JSON (confluent_sr JSON schema, one flat hits object per message) -> replace_primary_key + sql transformer (predicate + casts) "
                            "-> ClickHouse JSONEachRow (BASELINE.json configs[2])", "query": self.QUERY, "rows_per_gpu_per_step": self.n, "wire_bytes_per_step": len(self.data),
                "columns": self.ncols, "chain": [t for t, _ in self.CH], "table_plan": self.state.get("table_plan")}

    def extra(self):
        s = self.state
        return {"rows_out_per_step": s["out_rows"], "text_out_bytes_per_step": s["out_bytes"]}

    def side_measurements(self, only_pipeline=False):
        """Never `value`: configs[2] says "async double-buffer" and north_star "overlapping Kafka pull and ClickHouse push" — the
        pull AND the push inside the step, together.  Every step takes its Kafka bytes from PINNED host memory (hipMemcpyAsync H2D on
        the lane's stream: the pull), parses and transforms them, serializes JSONEachRow and copies the text back into pinned host
        memory (D2H: what httpuploader would POST).  On one lane the three legs follow each other; on several lanes (the parsequeue's
        shape: parse of batch N+1 beside the push of batch N, parsequeue.go:118-154) the H2D of one batch, the kernels of another and
        the D2H of a third share the GPU and both directions of the PCIe link.  overlap_efficiency = the longest of the three legs
        alone (H2D at the measured one-lane copy rate, kernels = the HBM-resident step, D2H likewise) over the measured step."""
        e, a = self.env, self.args
        lib, abi = e.lib, e.abi
        if a.pcie_steps <= 0 or e.rank != 0:
            return {}
        cap = int(self.state["out_bytes"] * 1.05) + (1 << 20)
        res = {}
        kernels_ms = getattr(self, "dt", 0) / max(getattr(self, "total_passes", 1), 1) * 1e3  # the HBM-resident step of the timed region
        # each direction alone, one lane: what the link gives this message size
        hin = lib.HostBuffer(self.data)
        lib.lane_use(0)
        t0 = time.perf_counter()
        import ctypes
        for _ in range(3):
            hd = ctypes.c_void_p()
            lib._check(lib.load().tfgpu_dbuf_upload(ctypes.c_void_p(hin.ptr), hin.size, ctypes.byref(hd)))
            lib.synchronize()
            lib.DeviceBuffer(hd).free()
        h2d_ms = (time.perf_counter() - t0) / 3 * 1e3
        # what the link gives when BOTH directions run and nothing else does: two lanes, one re-uploading the input, one re-downloading a
        # buffer of the output's size — the ceiling of any pull / push overlap on this box
        try:
            dev_out = lib.DeviceBuffer.alloc(cap)
            hout = lib.HostBuffer.__new__(lib.HostBuffer)
            pp = ctypes_void()
            lib._check(lib.load().tfgpu_host_alloc(dev_out.size, pp.ref))
            hout.ptr, hout.size = pp.value, dev_out.size
            go2, nrep = threading.Barrier(3), 6
            tms = {}

            def up():
                lib.lane_use(1)
                go2.wait()
                t0_ = time.perf_counter()
                for _ in range(nrep):
                    hd_ = ctypes.c_void_p()
                    lib._check(lib.load().tfgpu_dbuf_upload(ctypes.c_void_p(hin.ptr), hin.size, ctypes.byref(hd_)))
                    lib.synchronize()
                    lib.DeviceBuffer(hd_).free()
                tms["h2d"] = time.perf_counter() - t0_

            def down():
                lib.lane_use(2)
                go2.wait()
                t0_ = time.perf_counter()
                for _ in range(nrep):
                    lib._check(lib.load().tfgpu_dbuf_download(dev_out._h, hout.ptr, dev_out.size))
                tms["d2h"] = time.perf_counter() - t0_
            ths2 = [threading.Thread(target=up), threading.Thread(target=down)]
            for t in ths2:
                t.start()
            go2.wait()
            for t in ths2:
                t.join()
            lib.lane_use(0)
            res["copies_alone_concurrent"] = {"gb_per_s_h2d": round(hin.size * nrep / tms["h2d"] / 1e9, 2), "gb_per_s_d2h": round(dev_out.size * nrep / tms["d2h"] / 1e9, 2),
                                              "note": "both directions of the link at once, no kernels: the ceiling of the pull / push overlap here"}
            hout.free(); dev_out.free()
        except Exception as ex:  # noqa: BLE001
            res["copies_alone_concurrent"] = {"error": str(ex)[:200]}
        # lanes: 1 (the three legs follow each other), --pcie-lanes, and twice that — two batches in flight per pull / transform / push
        # stage, so that a lane waiting for its copy never leaves a direction of the link idle (the double buffer of configs[2])
        for nl in ([] if only_pipeline else sorted({1, max(1, min(a.pcie_lanes, lib.lane_count())), max(1, min(2 * a.pcie_lanes, lib.lane_count()))})):
            ins, outs = [lib.HostBuffer(self.data) for _ in range(nl)], []
            for _ in range(nl):
                hb = lib.HostBuffer.__new__(lib.HostBuffer)
                pp = ctypes_void()
                lib._check(lib.load().tfgpu_host_alloc(cap, pp.ref))
                hb.ptr, hb.size = pp.value, cap
                outs.append(hb)
            kk = max(a.pcie_steps, 2) * nl
            go, done = threading.Barrier(nl + 1), threading.Barrier(nl + 1)
            errs = []

            def lane_main(k, ins=ins, outs=outs, nl=nl, kk=kk, go=go, done=done):
                try:
                    lib.lane_use(k)
                    plans = self.plans if k == 0 else [lib.Transformer(t, c) for t, c in self.CH]
                    stage = self.stage if k == 0 else lib.Transformation(plans)

                    def one():
                        res_ = lib.sr_json_parse(self.opts, ins[k], self.msgs)          # pull: H2D from pinned memory inside the call
                        tr = stage.push_run(res_.device_batch)
                        out = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, tr.transformed)
                        lib._check(lib.load().tfgpu_dbuf_download(out._h, outs[k].ptr, out.size))  # push: D2H into pinned memory
                        for _, eb in tr.error_batches:
                            eb.free()
                        out.free(); tr.transformed.free(); res_.device_batch.free()
                    one()
                    lib.synchronize()
                    go.wait()
                    for i in range(kk):
                        if i % nl == k:
                            one()
                    lib.synchronize()
                    done.wait()
                except Exception as ex:  # noqa: BLE001
                    errs.append(ex); go.abort(); done.abort()
            ths = [threading.Thread(target=lane_main, args=(k,)) for k in range(nl)]
            for t in ths:
                t.start()
            try:
                go.wait()
                t0 = time.perf_counter()
                done.wait()
                dt = time.perf_counter() - t0
            except threading.BrokenBarrierError:
                dt = float("nan")
            for t in ths:
                t.join()
            lib.lane_use(0)
            for hb in ins + outs:
                hb.free()
            if errs:
                res[f"lanes_{nl}"] = {"error": str(errs[0])[:200]}
                continue
            step_ms = dt / kk * 1e3
            d2h_ms = self.state["out_bytes"] / 52e9 * 1e3  # the D2H leg alone at the link's measured one-direction rate (configs3 d2h_inclusive: 52 GB/s)
            res[f"lanes_{nl}"] = {"rows_per_s": round(self.n * kk / dt, 1), "ms_per_step": round(step_ms, 3),
                                  "gb_per_s_h2d": round(len(self.data) * kk / dt / 1e9, 2), "gb_per_s_d2h": round(self.state["out_bytes"] * kk / dt / 1e9, 2),
                                  "legs_alone_ms": {"h2d": round(h2d_ms, 3), "kernels": round(kernels_ms, 3), "d2h": round(d2h_ms, 3)},
                                  "overlap_efficiency": round(max(h2d_ms, kernels_ms, d2h_ms) / step_ms, 3)}
            dup = res.get("copies_alone_concurrent") or {}
            if dup.get("gb_per_s_h2d") and dup.get("gb_per_s_d2h"):
                # with both directions busy the link itself gives each less than it gives alone: the step no overlap can beat is the
                # longest leg at the rates measured for the two copies running side by side
                h2d_dup, d2h_dup = len(self.data) / dup["gb_per_s_h2d"] / 1e6, self.state["out_bytes"] / dup["gb_per_s_d2h"] / 1e6
                res[f"lanes_{nl}"]["legs_duplex_ms"] = {"h2d": round(h2d_dup, 3), "d2h": round(d2h_dup, 3)}
                res[f"lanes_{nl}"]["overlap_efficiency_vs_duplex_ceiling"] = round(max(h2d_dup, d2h_dup, kernels_ms) / step_ms, 3)
        # ---- the same work as a three-stage pipeline: ONE puller (H2D back to back on its own lane), transform lanes that take device-resident
        #      batches, ONE pusher (D2H back to back on its own lane), bounded queues of two batches between the stages — the double buffer of
        #      configs[2] spelled out: neither direction of the link ever waits for a lane to finish its other two legs ----
        try:
            import queue as _queue
            ncomp = 2
            kk = max(a.pcie_steps, 2) * 6
            hins = [lib.HostBuffer(self.data) for _ in range(2)]
            houts = []
            for _ in range(2):
                hb = lib.HostBuffer.__new__(lib.HostBuffer)
                pp = ctypes_void()
                lib._check(lib.load().tfgpu_host_alloc(cap, pp.ref))
                hb.ptr, hb.size = pp.value, cap
                houts.append(hb)
            q_in, q_out = _queue.Queue(maxsize=2), _queue.Queue(maxsize=2)
            errs = []
            go = threading.Barrier(ncomp + 3)
            tdone = {}

            def puller():
                try:
                    lib.lane_use(ncomp + 1)
                    go.wait()
                    for i in range(kk):
                        hd_ = ctypes.c_void_p()
                        lib._check(lib.load().tfgpu_dbuf_upload(ctypes.c_void_p(hins[i % 2].ptr), hins[i % 2].size, ctypes.byref(hd_)))
                        lib.synchronize()   # the batch is in HBM: another lane may read it
                        q_in.put(lib.DeviceBuffer(hd_))
                    for _ in range(ncomp):
                        q_in.put(None)
                except Exception as ex:  # noqa: BLE001
                    errs.append(ex); go.abort()
                    for _ in range(ncomp):
                        q_in.put(None)

            def transformer(k):
                try:
                    lib.lane_use(k)
                    plans = self.plans if k == 0 else [lib.Transformer(t, c) for t, c in self.CH]
                    stage = self.stage if k == 0 else lib.Transformation(plans)
                    warm = lib.DeviceBuffer.upload(self.data)
                    r0 = lib.sr_json_parse(self.opts, warm, self.msgs); t0_ = stage.push_run(r0.device_batch); o0 = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, t0_.transformed)
                    for _, eb in t0_.error_batches:
                        eb.free()
                    o0.free(); t0_.transformed.free(); r0.device_batch.free(); warm.free()
                    lib.synchronize()
                    go.wait()
                    while True:
                        din = q_in.get()
                        if din is None:
                            break
                        res_ = lib.sr_json_parse(self.opts, din, self.msgs)     # device-resident bytes: no copy inside the call
                        tr = stage.push_run(res_.device_batch)
                        out = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, tr.transformed)
                        for _, eb in tr.error_batches:
                            eb.free()
                        tr.transformed.free(); res_.device_batch.free(); din.free()
                        lib.synchronize()
                        q_out.put(out)
                    q_out.put(None)
                except Exception as ex:  # noqa: BLE001
                    errs.append(ex); go.abort(); q_out.put(None)

            def pusher():
                try:
                    lib.lane_use(ncomp + 2)
                    go.wait()
                    ends, i = 0, 0
                    while ends < ncomp:
                        out = q_out.get()
                        if out is None:
                            ends += 1
                            continue
                        lib._check(lib.load().tfgpu_dbuf_download(out._h, houts[i % 2].ptr, out.size))   # returns when the text is in pinned memory
                        out.free(); i += 1
                    tdone["n"] = i
                    tdone["t"] = time.perf_counter()
                except Exception as ex:  # noqa: BLE001
                    errs.append(ex); go.abort()
            ths = [threading.Thread(target=puller), threading.Thread(target=pusher)] + [threading.Thread(target=transformer, args=(k,)) for k in range(ncomp)]
            for t in ths:
                t.start()
            try:
                go.wait()
                t0 = time.perf_counter()
            except threading.BrokenBarrierError:
                t0 = float("nan")
            for t in ths:
                t.join()
            lib.lane_use(0)
            for hb in hins + houts:
                hb.free()
            if errs or tdone.get("n") != kk:
                res["pipeline_3_stage"] = {"error": str(errs[0])[:200] if errs else "batches lost"}
            else:
                dt = tdone["t"] - t0
                step_ms = dt / kk * 1e3
                ent = {"rows_per_s": round(self.n * kk / dt, 1), "ms_per_step": round(step_ms, 3), "transform_lanes": ncomp, "batches": kk,
                       "gb_per_s_h2d": round(len(self.data) * kk / dt / 1e9, 2), "gb_per_s_d2h": round(self.state["out_bytes"] * kk / dt / 1e9, 2),
                       "overlap_efficiency": round(max(h2d_ms, kernels_ms, self.state["out_bytes"] / 52e9 * 1e3) / step_ms, 3),
                       "note": "one puller, %d transform lanes, one pusher, queues of two batches between them; the first batch's pull and the last batch's push are inside the time" % ncomp}
                dup = res.get("copies_alone_concurrent") or {}
                if dup.get("gb_per_s_h2d") and dup.get("gb_per_s_d2h"):
                    h2d_dup, d2h_dup = len(self.data) / dup["gb_per_s_h2d"] / 1e6, self.state["out_bytes"] / dup["gb_per_s_d2h"] / 1e6
                    ent["overlap_efficiency_vs_duplex_ceiling"] = round(max(h2d_dup, d2h_dup, kernels_ms) / step_ms, 3)
                res["pipeline_3_stage"] = ent
        except Exception as ex:  # noqa: BLE001
            res["pipeline_3_stage"] = {"error": str(ex)[:200]}
        hin.free()
        res["note"] = ("every step pulls its %.2f GB of Kafka bytes from pinned host memory (H2D), runs parse + replace_primary_key + sql + JSONEachRow, and pushes the "
                       "%.2f GB of text back into pinned host memory (D2H); with several lanes the three legs of different batches overlap: H2D of batch N+1, "
                       "kernels of N, D2H of N-1 (PCIe Gen5 x16 is full duplex, ~52-54 GB/s per direction measured)" % (len(self.data) / 1e9, self.state["out_bytes"] / 1e9))
        return {"pull_push_concurrent": res}

    def cpu(self):
        from oracle import oracle as ora
        abi = self.env.abi
        k = min(self.args.cpu_rows, self.n, 1 << 13)
        d2, m2 = abi.messages(self.frames[:k], list(range(k)), [0] * k)
        r1 = ora.sr_json_parse(self.opts, d2, m2)
        r2 = ora.apply_chain([ora.Transformer(t, c) for t, c in self.CPU_CH], r1.batch, r1.schema)
        t0 = time.perf_counter()
        ora.serialize(abi.FMT_CH_JSON_EACH_ROW, r2.batch, r2.schema)
        ts = time.perf_counter() - t0
        sec = r1.seconds + r2.seconds + ts
        out = {"value": round(k / sec, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} of the same messages: oracle SR parse ({r1.seconds:.2f}s) + the query's predicate and casts as filter_rows + convert_to_string ({r2.seconds:.2f}s) + JSONEachRow ({ts:.2f}s), single thread",
               "note": "C restatement of the Go reference, not the Go binary, and without the reference's clickhouse-local fork/exec and double JSON round trip "
                       "(clickhouse_local.go:145-210), i.e. faster than the reference's sql transformer; the reference parses with GOMAXPROCS goroutines (generic_parser.go:406-438)"}
        out.update(host_info())
        out["parity"] = _guard_parity(lambda: self.parity(min(k, 2048)))
        return out

    def parity(self, k):
        """SR parse of the first k messages against the oracle; replace_primary_key + sql on that batch against oracle/ora_sql.py (the
        restatement of ClickHouse's documented typing — `sql` parity is unpinned by construction: no clickhouse-local here); JSONEachRow
        of the result against the oracle's serializer"""
        from oracle import oracle as ora, ora_sql
        lib, abi = self.env.lib, self.env.abi
        d2, m2 = abi.messages(self.frames[:k], list(range(k)), [1_700_000_000_000_000_000 + i for i in range(k)])
        ref = ora.sr_json_parse(self.opts, d2, m2)
        res = lib.sr_json_parse(self.opts, d2, m2)
        host = res.device_batch.download()
        why = _batch_diff(abi, host, ref.batch)
        if why or res.errors:
            return _parity(k, "parse", why or "device row errors")
        tr = self.stage.push_run(res.device_batch)
        out = tr.transformed.download()
        keyed = [(c.name, c.dtype, c.name == "watchid") for c in ref.schema.cols]
        rows = [{"kind": "insert", "src": i, "values": {c.name: c.pyvalue(i) for c in host.cols}} for i in range(host.nrows)]
        exp = ora_sql.apply(self.QUERY, rows, keyed)
        if out.nrows != len(exp):
            return _parity(k, "replace_primary_key + sql", "rows %d vs %d" % (out.nrows, len(exp)))
        for i, e in enumerate(exp):
            got = [c.pyvalue(i) for c in out.cols]
            if [[g[0], g[1] if not isinstance(g[1], tuple) else tuple(g[1])] for g in got] != [[v[0], v[1]] for v in e["values"]]:
                return _parity(k, "replace_primary_key + sql", "row %d differs" % i)
        text = bytes(lib.serialize(abi.FMT_CH_JSON_EACH_ROW, tr.transformed).download())
        rs = ora_sql.resolve(self.QUERY, keyed)
        osch = abi.Schema.of([[r[0], r[3], bool(r[4])] for r in rs])
        out.schema = osch
        want = bytes(ora.serialize(abi.FMT_CH_JSON_EACH_ROW, out, osch))
        if text != want:
            return _parity(k, "JSONEachRow", "text differs (%d vs %d bytes)" % (len(text), len(want)))
        return _parity(k, "SR JSON parse (cell for cell) + replace_primary_key + sql against oracle/ora_sql.py (cell for cell; sql parity is unpinned: no clickhouse-local here) + JSONEachRow (byte for byte)",
                       compared_output_rows=out.nrows, compared_output_bytes=len(want))


class Configs3Workload(Base):
    """BASELINE.json configs[3]: columns delivered as if decoded from Parquet (typed columns resident in HBM; SURVEY §8d allows
    exactly this) → mask(ClientIP) + sharder-hash(UserID) + casts → ClickHouse JSONEachRow, row-sharded, no collective."""
    metric = "ChangeItems/sec through mask(ClientIP) + sharder(UserID) + convert_to_string + convert_to_datetime -> ClickHouse JSONEachRow on resident hits columns"
    default_rows = 1 << 20
    CH = [MASK, ("sharder_transformer", {"shardsCount": "8", "columns": {"includeColumns": ["^userid$"]}, "tables": {}}),
          ("convert_to_string", {"columns": {"includeColumns": ["^regionid$", "^counterid$"]}, "tables": {}}),
          ("convert_to_datetime", {"columns": {"includeColumns": ["^ipnetworkid$"]}, "tables": {}})]

    def setup(self):
        e, a = self.env, self.args
        lib = e.lib
        dbuf, self.csv_bytes, _ = stage_shard(lib, e.workload, a.rows * e.rank, a.rows)
        db, _, errs = lib.csv_parse(e.workload.hits_csv_options(), e.workload.hits_schema(), dbuf)
        assert not errs
        host = db.download()  # packs every column; re-upload = plain resident columns, no reference to the CSV text
        db.free(); dbuf.free()
        self.host_copy = host  # (the lanes of the d2h side measurement upload their own resident copy)
        self.db = lib.DeviceBatch.upload(host)
        self.n = self.db.nrows
        self.plans = [lib.Transformer(t, c) for t, c in self.CH]

    def rows(self):
        return self.n

    def step(self, keep=False):
        lib, abi = self.env.lib, self.env.abi
        tr = lib.apply_chain(self.plans, self.db)
        out = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, tr.transformed)
        if keep:
            self.state.update(in_bytes=self.db.payload_bytes(), out_bytes=out.size, kept_bytes=tr.transformed.payload_bytes(), errors=len(tr.errors))
        out.free(); tr.transformed.free()

    def alg(self):
        s = self.state
        return {"mask_hmac_sha256": 72 * self.n, "ser_chunk_write": s["kept_bytes"] + s["out_bytes"], "ser_chunk_len": s["kept_bytes"], "ser_cell_write": s["kept_bytes"] + s["out_bytes"], "ser_cell_len": s["kept_bytes"], "sharder_crc32": (8 + 4) * self.n,
                "tostring_write": 2 * 18 * self.n, "todatetime": 12 * self.n}

    def int_roofline(self, kernels):
        k = kernels.get("mask_hmac_sha256")
        if not k:
            return None
        mrows = k.get("units_per_step") or self.n  # the library's own count of the rows the kernel was launched on
        ach = MASK_INT_OPS_PER_VALUE * mrows / max(k["launches_per_step"], 1) / (k["avg_ms"] * 1e-3) / 1e12
        return {"kernel": "mask_hmac_sha256", "bound": "int32_valu", "achieved": round(ach, 2), "peak": INT32_PEAK_TOPS, "unit": "Top/s", "frac": round(ach / INT32_PEAK_TOPS, 4),
                "values_per_step": mrows}

    def config(self):
        return {"workload": "hits columns resident in HBM as if decoded from Parquet -> mask + sharder + casts -> ClickHouse JSONEachRow (BASELINE.json configs[3], per-GPU shard)",
                "rows_per_gpu_per_step": self.n, "columns": 105, "chain": [t for t, _ in self.CH], "parallelism": f"row-range shard x{self.env.world}, no collective"}

    def extra(self):
        return {"text_out_bytes_per_step": self.state["out_bytes"]}

    def parquet_source(self):
        """Never `value`: the same step from configs[3]'s REAL source format.  The resident columns are written as one Parquet
        object (pyarrow, uncompressed — the codec the device path takes — dictionary encoding as pyarrow chooses it, one row group)
        into pinned host memory; a step is tfgpu_parquet_read (upload of the object + decode on the device) + the chain + JSONEachRow."""
        e, a = self.env, self.args
        lib, abi = e.lib, e.abi
        try:
            import io
            import pyarrow as pa
            import pyarrow.parquet as pq
        except Exception as ex:  # noqa: BLE001
            return {"parquet_source": {"skipped": "pyarrow: %s" % ex}}
        h = self.host_copy
        arrays, names = [], []
        for c in h.cols:
            valid = c.validity
            vbuf = pa.py_buffer(np.packbits(valid, bitorder="little").tobytes()) if valid is not None else None
            if c.repr in abi.VAR_REPRS:
                off = c.offsets.astype(np.int32)
                arr = pa.Array.from_buffers(pa.string() if c.repr == abi.R_STRING else pa.binary(), h.nrows, [vbuf, pa.py_buffer(off.tobytes()), pa.py_buffer(bytes(c.data[: int(off[-1])]))])
            elif c.repr == abi.R_TIME:
                arr = pa.array(c.values.astype(np.int64), pa.int64(), mask=None if valid is None else ~valid)   # (epoch seconds as INT64: the decode cost of a time column)
            else:
                arr = pa.array(c.values, mask=None if valid is None else ~valid)
            arrays.append(arr); names.append(c.name)
        buf = io.BytesIO()
        pq.write_table(pa.table(arrays, names=names), buf, compression="NONE", row_group_size=h.nrows)
        data = buf.getvalue()
        pinned = lib.HostBuffer(data)
        schema = abi.Schema.of([[c.name, c.dtype] for c in h.cols])
        cs = schema.to_c()
        import ctypes as C

        def one(plans=None, read_only=False):
            out = C.c_void_p()
            lib._check(lib.load().tfgpu_parquet_read(C.c_void_p(pinned.ptr), C.c_uint64(len(data)), abi.MEM_HOST, C.byref(cs), b"", b"hits", C.byref(out)))
            db = lib.DeviceBatch(out)
            if not read_only:
                tr = lib.apply_chain(plans if plans is not None else self.plans, db)
                o = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, tr.transformed)
                o.free(); tr.transformed.free()
            db.free()
        one(); lib.synchronize()
        k = 5
        t0 = time.perf_counter()
        for _ in range(k):
            one()
        lib.synchronize()
        dt = (time.perf_counter() - t0) / k
        t0 = time.perf_counter()
        for _ in range(k):
            one(read_only=True)
        lib.synchronize()
        dt_read = (time.perf_counter() - t0) / k
        lib.prof_reset(); lib.prof_enable(True)
        one()
        lib.prof_enable(False)
        prof = {n: round(ms / max(l, 1), 4) for n, l, ms in lib.prof_get() if n.startswith("pq_")}
        lib.prof_reset()
        # the same step on three lanes: the upload of one object beside the decode / chain / serializer of another
        nl, kk = 3, 9
        go, done, errs = threading.Barrier(nl + 1), threading.Barrier(nl + 1), []

        def lane_main(j):
            try:
                lib.lane_use(j)
                plans = self.plans if j == 0 else [lib.Transformer(t, c) for t, c in self.CH]
                one(plans); lib.synchronize()
                go.wait()
                for i in range(kk):
                    if i % nl == j:
                        one(plans)
                lib.synchronize()
                done.wait()
            except Exception as ex:  # noqa: BLE001
                errs.append(ex); go.abort(); done.abort()
        ths = [threading.Thread(target=lane_main, args=(j,)) for j in range(nl)]
        for t in ths:
            t.start()
        try:
            go.wait()
            t0 = time.perf_counter()
            done.wait()
            dt3 = (time.perf_counter() - t0) / kk
        except threading.BrokenBarrierError:
            dt3 = float("nan")
        for t in ths:
            t.join()
        lib.lane_use(0)
        # ---- the same as a pull / decode pipeline: ONE puller brings the objects into HBM back to back on its own lane (into staging buffers
        #      of tfgpu_parquet_staging_size bytes), two lanes decode device-resident objects (tfgpu_parquet_read_staged: footer and page
        #      headers walked in the host copy, no upload in front of the kernels), run the chain and the serializer ----
        pipe = None
        try:
            import queue as _queue
            need = lib.parquet_staging_size(pinned)
            ndec, kkp = 2, 12
            q_in, errs2, gop, tdone = _queue.Queue(maxsize=2), [], threading.Barrier(ndec + 2), {}

            def puller():
                try:
                    lib.lane_use(ndec + 1)
                    gop.wait()
                    for _ in range(kkp):
                        st_ = lib.DeviceBuffer.alloc(need)
                        lib._check(lib.load().tfgpu_dbuf_write(st_._h, C.c_uint64(0), C.c_void_p(pinned.ptr), C.c_uint64(len(data))))
                        lib.synchronize()
                        q_in.put(st_)
                    for _ in range(ndec):
                        q_in.put(None)
                except Exception as ex:  # noqa: BLE001
                    errs2.append(ex); gop.abort()
                    for _ in range(ndec):
                        q_in.put(None)

            def decoder(j):
                try:
                    lib.lane_use(j)
                    plans = self.plans if j == 0 else [lib.Transformer(t, c) for t, c in self.CH]
                    one(plans); lib.synchronize()
                    gop.wait()
                    while True:
                        st_ = q_in.get()
                        if st_ is None:
                            break
                        db = lib.parquet_read_staged(pinned, st_, schema, "", "hits")
                        st_.free()
                        tr = lib.apply_chain(plans, db)
                        o = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, tr.transformed)
                        o.free(); tr.transformed.free(); db.free()
                        lib.synchronize()
                        tdone[j] = time.perf_counter()
                except Exception as ex:  # noqa: BLE001
                    errs2.append(ex); gop.abort()
            thp = [threading.Thread(target=puller)] + [threading.Thread(target=decoder, args=(j,)) for j in range(ndec)]
            for t in thp:
                t.start()
            try:
                gop.wait()
                t0p = time.perf_counter()
            except threading.BrokenBarrierError:
                t0p = float("nan")
            for t in thp:
                t.join()
            lib.lane_use(0)
            if errs2 or not tdone:
                pipe = {"error": str(errs2[0])[:200] if errs2 else "no object decoded"}
            else:
                dtp = (max(tdone.values()) - t0p) / kkp
                pipe = {"ms_per_step": round(dtp * 1e3, 3), "gb_per_s_parquet_in": round(len(data) / dtp / 1e9, 2), "rows_per_s": round(h.nrows / dtp, 1), "decode_lanes": ndec, "objects": kkp,
                        "note": "one puller (tfgpu_dbuf_write from pinned memory, back to back), two lanes that decode staged objects (tfgpu_parquet_read_staged), run the chain and serialize; "
                                "the first object's pull and the last one's decode are inside the time"}
        except Exception as ex:  # noqa: BLE001
            pipe = {"error": str(ex)[:200]}
        pinned.free()
        return {"parquet_source": {"object_bytes": len(data), "pipeline_pull_decode": pipe, "rows_per_s": round(h.nrows / dt, 1), "ms_per_step": round(dt * 1e3, 3), "gb_per_s_parquet_in": round(len(data) / dt / 1e9, 2),
                                   "read_only_ms": round(dt_read * 1e3, 3), "read_only_gb_per_s": round(len(data) / dt_read / 1e9, 2),
                                   "lanes_3": {"ms_per_step": round(dt3 * 1e3, 3), "gb_per_s_parquet_in": round(len(data) / dt3 / 1e9, 2), "rows_per_s": round(h.nrows / dt3, 1)} if not errs else {"error": str(errs[0])[:200]},
                                   "decode_kernels_avg_ms": prof,
                                   "note": "uncompressed Parquet written by pyarrow from the same columns (time columns as INT64), in pinned host memory; a step uploads the object (PCIe inside), "
                                           "decodes it on the device, then runs the chain and the serializer (read_only: upload + decode alone; lanes_3: whole steps of three lanes side by side); "
                                           "the decoder's parity: tests/test_parquet.py (pyarrow's reading of the same bytes) and tests/test_parquet_canon.py (the reference's reader canon)"}}

    def side_measurements(self):
        """Never `value`: the sink leg.  Every step ends with the JSONEachRow bytes copied to PINNED host memory
        (hipMemcpyAsync D2H on the lane's stream, the bytes httpuploader would POST: marshal.go:82-125) — on one lane, and on
        several, where the D2H of one batch runs beside the kernels of the next (parsequeue.go:118-154: push beside parse)."""
        e, a = self.env, self.args
        lib, abi = e.lib, e.abi
        if a.pcie_steps <= 0 or e.rank != 0:
            return {}
        cap = int(self.state["out_bytes"] * 1.05) + (1 << 20)
        res = {}
        for nl in sorted({1, max(1, min(a.pcie_lanes, lib.lane_count()))}):
            pinned = []
            for _ in range(nl):
                hb = lib.HostBuffer.__new__(lib.HostBuffer)
                pp = ctypes_void()
                lib._check(lib.load().tfgpu_host_alloc(cap, pp.ref))
                hb.ptr, hb.size = pp.value, cap
                pinned.append(hb)
            kk = max(a.pcie_steps, 2) * nl
            go, done = threading.Barrier(nl + 1), threading.Barrier(nl + 1)
            errs = []

            def lane_main(k, pinned=pinned, nl=nl, kk=kk, go=go, done=done):
                try:
                    lib.lane_use(k)
                    db = self.db if k == 0 else lib.DeviceBatch.upload(self.host_copy)
                    def one():
                        tr = lib.apply_chain(self.plans, db)
                        out = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, tr.transformed)
                        lib._check(lib.load().tfgpu_dbuf_download(out._h, pinned[k].ptr, out.size))  # D2H into pinned memory, then the lane's sync
                        out.free(); tr.transformed.free()
                    one()
                    lib.synchronize()
                    go.wait()
                    for i in range(kk):
                        if i % nl == k:
                            one()
                    lib.synchronize()
                    done.wait()
                    if k:
                        db.free()
                except Exception as ex:  # noqa: BLE001
                    errs.append(ex); go.abort(); done.abort()
            ths = [threading.Thread(target=lane_main, args=(k,)) for k in range(nl)]
            for t in ths:
                t.start()
            go.wait()
            t0 = time.perf_counter()
            done.wait()
            dt = time.perf_counter() - t0
            for t in ths:
                t.join()
            lib.lane_use(0)
            for hb in pinned:
                hb.free()
            if errs:
                res[f"lanes_{nl}"] = {"error": str(errs[0])[:200]}
            else:
                res[f"lanes_{nl}"] = {"rows_per_s": round(self.n * kk / dt, 1), "gb_per_s_d2h": round(self.state["out_bytes"] * kk / dt / 1e9, 2), "ms_per_step": round(dt / kk * 1e3, 3)}
        res["note"] = ("every step's JSONEachRow text (%.2f GB) is copied to pinned host memory inside the step; with several lanes the D2H of one batch runs beside "
                       "the kernels of another; PCIe Gen5 x16 bounds the copy at ~55-60 GB/s" % (self.state["out_bytes"] / 1e9))
        out = {"d2h_inclusive": res}
        try:
            out.update(self.parquet_source())
        except Exception as ex:  # noqa: BLE001
            out["parquet_source"] = {"error": str(ex)[:300]}
        return out

    def cpu(self):
        from oracle import oracle as ora
        e, abi = self.env, self.env.abi
        k = min(self.args.cpu_rows, self.n, 1 << 16)
        r1 = ora.csv_parse(e.workload.hits_csv_options(), e.workload.hits_schema(), e.workload.hits_csv(k), "", "")
        r2 = ora.apply_chain([ora.Transformer(t, c) for t, c in self.CH], r1.batch, r1.schema)
        t0 = time.perf_counter()
        ora.serialize(abi.FMT_CH_JSON_EACH_ROW, r2.batch, r2.schema)
        ts = time.perf_counter() - t0
        out = {"value": round(k / (r2.seconds + ts), 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} rows: oracle chain ({r2.seconds:.2f}s) + JSONEachRow ({ts:.2f}s) on already-typed rows, single thread (one table = one goroutine, transformation.go:131-135)",
               "note": "C restatement of the Go reference, not the Go binary"}
        out.update(host_info())
        out["parity"] = _guard_parity(lambda: self.parity(min(k, 8192), r1))
        return out

    def parity(self, k, r1):
        """the first k rows of the resident table (= the oracle's parse of the same CSV rows, checked) through the chain + JSONEachRow,
        device against oracle, byte for byte"""
        from oracle import oracle as ora
        lib, abi = self.env.lib, self.env.abi
        k = (k // 8) * 8
        head = self.db.slice(0, k)
        host = head.download()
        rb = r1.batch
        ref_head = abi.Batch([abi.Column(c.name, c.dtype, c.repr, values=None if c.values is None else c.values[:k], nanos=None if c.nanos is None else c.nanos[:k],
                                         offsets=None if c.offsets is None else c.offsets[:k + 1].copy(), data=None if c.data is None else c.data[:int(c.offsets[k])],
                                         validity=None if c.validity is None else c.validity[:k]) for c in rb.cols], k, rb.table_ns, rb.table_name)
        why = _batch_diff(abi, host, ref_head)
        if why:
            return _parity(k, "resident columns", why)
        tr = lib.apply_chain(self.plans, head)
        text = bytes(lib.serialize(abi.FMT_CH_JSON_EACH_ROW, tr.transformed).download())
        r2 = ora.apply_chain([ora.Transformer(t, c) for t, c in self.CH], ref_head, r1.schema)
        want = bytes(ora.serialize(abi.FMT_CH_JSON_EACH_ROW, r2.batch, r2.schema))
        if text != want:
            return _parity(k, "chain + JSONEachRow", "text differs (%d vs %d bytes)" % (len(text), len(want)))
        return _parity(k, "resident columns (cell for cell) + mask + sharder + casts + JSONEachRow (byte for byte)", compared_output_rows=r2.batch.nrows, compared_output_bytes=len(want))


class Configs0Workload(Base):
    """BASELINE.json configs[0] (SURVEY §8d "Config 1"): the plumbing case — 1 M rows of the 4-column table
    (id int64, name utf8, ip int32, ts timestamp) through rename_tables + mask_field(ip), nothing parsed and nothing serialized
    (devnull -> devnull).  The reference runs it on the CPU only; here the same two transformers run on resident columns and
    the oracle's single-thread time for the same chain on the same rows is printed beside it."""
    metric = "ChangeItems/sec through rename_tables + mask_field(ip), devnull -> devnull, on a resident 4-column table"
    default_rows = 1 << 20
    CH = [("rename_tables", {"renameTables": [{"originalName": {"nameSpace": "public", "name": "users"}, "newName": {"nameSpace": "bench", "name": "users_masked"}}]}),
          ("mask_field", {"maskFunctionHash": {"userDefinedSalt": "clickbench-salt"}, "columns": ["ip"]})]

    def table(self, row0, n):
        abi = self.env.abi
        rng = np.random.default_rng(0x5EEDC11C + row0)
        lens = rng.integers(3, 24, n)
        off = np.zeros(n + 1, np.uint32); off[1:] = np.cumsum(lens)
        data = rng.integers(97, 123, int(off[-1])).astype(np.uint8)
        cols = [abi.Column("id", "int64", abi.R_INT64, values=np.arange(row0, row0 + n, dtype=np.int64)),
                abi.Column("name", "utf8", abi.R_STRING, offsets=off, data=data),
                abi.Column("ip", "int32", abi.R_INT32, values=rng.integers(-2**31, 2**31, n).astype(np.int32)),
                abi.Column("ts", "timestamp", abi.R_TIME, values=rng.integers(1372636800, 1375315200, n), nanos=np.zeros(n, np.int32))]
        return abi.Batch(cols, n, "public", "users"), abi.Schema.of([["id", "int64", True], ["name", "utf8", False], ["ip", "int32", False], ["ts", "timestamp", False]])

    def setup(self):
        e, a = self.env, self.args
        host, _ = self.table(a.rows * e.rank, a.rows)
        self.db = e.lib.DeviceBatch.upload(host)
        self.n = self.db.nrows
        self.plans = [e.lib.Transformer(t, c) for t, c in self.CH]

    def rows(self):
        return self.n

    def step(self, keep=False):
        tr = self.env.lib.apply_chain(self.plans, self.db)
        if keep:
            assert tr.transformed.table_id() == ("bench", "users_masked"), tr.transformed.table_id()
            self.state.update(in_bytes=self.db.payload_bytes(), out_bytes=0, kept_bytes=tr.transformed.payload_bytes(), errors=len(tr.errors))
        tr.transformed.free()

    def alg(self):
        return {"mask_hmac_sha256": 68 * self.n}  # 4 B of int32 in, 64 B of hex out

    def int_roofline(self, kernels):
        return Configs3Workload.int_roofline(self, kernels)

    def side_measurements(self):
        """Never `value`: Apply([]ChangeItem) as transformation.do would call it (transformation.go:252-257) — the rows start as boxed
        []interface{} items on the host, are fanned out into column buffers, cross the C ABI once and are fanned back in.  A C++
        model of the Go data and of INTEGRATION.md §2's binding (tools/fanout/fanout_harness.cpp: there is no Go toolchain here)."""
        e, a = self.env, self.args
        if not getattr(a, "from_rows", False) or e.rank != 0:
            return {}
        import ctypes as C
        so = os.path.join(ROOT, "tools", "fanout", "libfanout.so")
        if not os.path.exists(so):
            return {"from_rows": {"error": "tools/fanout/libfanout.so is not built (__graft_entry__.build())"}}
        H = C.CDLL(so)
        H.fanout_run.argtypes = [C.c_char_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_char_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_size_t]

        def run(n, cols, touch, plans, ns, table, reps=3):
            names = (C.c_char_p * len(cols))(*[c[0].encode() for c in cols]); tags = (C.c_int32 * len(cols))(*[c[1] for c in cols])
            t = (C.c_int32 * len(touch))(*touch)
            pt = (C.c_char_p * len(plans))(*[p[0].encode() for p in plans]); pc = (C.c_char_p * len(plans))(*[json.dumps(p[1]).encode() for p in plans])
            out = C.create_string_buffer(4096)
            H.fanout_run(e.lib._LIBPATH.encode(), n, len(cols), names, tags, t, len(touch), pt, pc, len(plans), ns.encode(), table.encode(), reps, out, 4096)
            d = json.loads(out.value.decode())
            if "ms" in d:
                tot = sum(d["ms"].values())
                d["ms_total"] = round(tot, 3)
                d["rows_per_s"] = round(n / (tot * 1e-3), 1)
            return d
        four = [("id", 1), ("name", 4), ("ip", 2), ("ts", 5)]
        tagof = {"int16": 3, "int32": 2, "int64": 1, "utf8": 4, "timestamp": 5, "date": 6, "any": 4}
        hits = [(n, tagof[t]) for n, t, _ in e.workload.hits_columns()]
        hidx = {n: i for i, (n, _) in enumerate(hits)}
        n0, nh = min(self.n, 1 << 20), 1 << 18
        res = {"configs0_all_columns": run(n0, four, [0, 1, 2, 3], self.CH, "public", "users"),
               "configs0_touched_columns_only": run(n0, four, [2], self.CH, "public", "users"),
               "hits_mask_filter_2_of_105_columns": run(nh, hits, [hidx["clientip"], hidx["eventdate"]], CHAIN, "default", "hits"),
               "note": "per call of Apply(items): fan_out = one type switch per cell of the fanned-out columns into pinned staging; upload / apply / download = the one C-ABI crossing "
                       "(tfgpu_batch_upload, tfgpu_apply, tfgpu_dbatch_download); fan_in = a new item per kept row, untouched cells re-use the input's boxed values by src_row, the "
                       "rewritten column is boxed.  `touched_columns_only` fans out just what the chain reads (SURVEY 7: only materialise columns the chain touches).  A C++ model "
                       "of the Go data (16-byte interface words → heap boxes), not Go: no garbage collector, no write barriers — read it as a LOWER bound on the Go binding's host cost"}
        return {"from_rows": res}

    def config(self):
        return {"workload": "4-column table (id int64, name utf8, ip int32, ts timestamp) resident in HBM -> rename_tables + mask_field(ip) -> devnull (BASELINE.json configs[0]: the plumbing case, CPU-only in the reference)",
                "rows_per_gpu_per_step": self.n, "columns": 4, "chain": [t for t, _ in self.CH], "parallelism": f"row-range shard x{self.env.world}, no collective"}

    def cpu(self):
        from oracle import oracle as ora
        k = min(self.args.cpu_rows, self.n)
        b, schema = self.table(0, k)
        r = ora.apply_chain([ora.Transformer(t, c) for t, c in self.CH], b, schema)
        out = {"value": round(k / r.seconds, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} rows: oracle rename_tables + mask_field ({r.seconds:.2f}s) on the same typed rows, single thread (one table = one goroutine, transformation.go:131-135)",
               "note": "C restatement of the Go reference, not the Go binary"}
        out.update(host_info())

        def check():
            kk = (min(k, 1 << 16) // 8) * 8
            tr = self.env.lib.apply_chain(self.plans, self.env.lib.DeviceBatch.upload(b).slice(0, kk))   # the oracle's own sample (the resident table is another draw of the generator)
            got = tr.transformed.download()
            rb = r.batch
            want = self.env.abi.Batch([self.env.abi.Column(c.name, c.dtype, c.repr, values=None if c.values is None else c.values[:kk], nanos=None if c.nanos is None else c.nanos[:kk],
                                                           offsets=None if c.offsets is None else c.offsets[:kk + 1].copy(), data=None if c.data is None else c.data[:int(c.offsets[kk])],
                                                           validity=None if c.validity is None else c.validity[:kk]) for c in rb.cols], kk, rb.table_ns, rb.table_name)
            why = _batch_diff(self.env.abi, got, want)
            if why is None and (got.table_ns, got.table_name) != ("bench", "users_masked"):
                why = "table id"
            return _parity(kk, "rename_tables + mask_field, cell for cell", why, compared_output_rows=kk)
        out["parity"] = _guard_parity(check)
        return out


class Configs4Workload(Base):
    """BASELINE.json configs[4], from the decoded CDC batch on: hash-partition by key (sharder CRC32 % world → tfgpu_partition →
    all-to-all over RCCL/xGMI) → Collapse (PK-keyed dedup) → native queue serializer (Kafka-ready messages)."""
    metric = "ChangeItems/sec through hash-partition (RCCL all-to-all) -> Collapse -> native queue serializer, CDC stream"
    default_rows = 1 << 20

    def setup(self):
        e, a = self.env, self.args
        lib, abi = e.lib, e.abi
        b, self.schema = e.workload.cdc_batch(a.rows, seed=0x5EED + e.rank)
        self.n = a.rows
        self.db = lib.DeviceBatch.upload(b)
        self.shard = lib.Transformer("sharder_transformer", {"shardsCount": str(e.world), "columns": {"includeColumns": ["^id$"]}, "tables": {}})
        self.qopts = abi.queue_options(abi.QFMT_NATIVE, enabled=True, max_message_size=1 << 20, table_schema=self.schema)
        self.debezium = getattr(a, "sink", "native") == "debezium"
        if self.debezium:
            # the same table as a Postgres source describes it (OriginalType per column), the serializer's settings as the Kafka sink passes them
            self.pg_schema = abi.Schema([abi.ColSchema(c.name, c.dtype, c.key, "", "pg:text" if c.dtype == "utf8" else "pg:bigint") for c in self.schema.cols])
            self.dbz_params = {"database.dbname": "db", "topic.prefix": "srv", "dt.source.type": "pg"}
            self.dopts = abi.dbz_emit_options(self.dbz_params, self.pg_schema)
            self.metric = "ChangeItems/sec through hash-partition (RCCL all-to-all) -> Collapse -> Debezium emitter (key + value, inline schemas), CDC stream"
            tb, _ = e.workload.cdc_batch(64, seed=1)
            tiny = lib.debezium_emit(self.dopts, lib.DeviceBatch.upload(tb), abi.row_meta(64, ids=np.arange(64), lsns=np.arange(64, dtype=np.uint64), commit_times=np.full(64, 1, np.uint64)))
            k0, v0 = next((k, v) for k, v in tiny.messages() if v is not None)
            self.const_key, self.const_val = len(k0) - k0.index(b',"schema":'), len(v0) - v0.index(b',"schema":')   # the constant tails dbz_fill_const writes
        # ChangeItem.ID / LSN / CommitTime by input row: inputs like the columns, so resident in HBM when the timed region starts (tfgpu_row_meta.mem = DEVICE); until r05k the
        # bench handed them over as pageable host arrays and their 20 MB upload sat inside every step (0.4 ms of it)
        self.meta = abi.CRowMeta()
        self._meta_keep = [lib.DeviceBuffer.upload(np.ascontiguousarray(a, dt).tobytes()) for a, dt in ((np.arange(self.n) % 97, np.uint32), (np.arange(self.n, dtype=np.uint64) + 5, np.uint64),
                                                                                                      (np.full(self.n, 1700000000000000000, np.uint64), np.uint64))]
        self.meta.id, self.meta.lsn, self.meta.commit_time = (b.ptr for b in self._meta_keep)
        self.meta.n, self.meta.mem = self.n, abi.MEM_DEVICE
        if e.world > 1:
            import torch.distributed as dist
            from transferia_amd import partition
            self.comm = partition.device_comm(dist, lib)  # torch.distributed carries the 128-byte rendezvous id, nothing else
        elif a.exchange:
            self.comm = lib.Comm.create(lib.Comm.unique_id(), 0, 1)
        else:
            self.comm = None

    def rows(self):
        return self.n

    def step(self, keep=False):
        e = self.env
        lib = e.lib
        one = self.shard.apply(self.db).transformed
        grouped, counts = lib.partition(one, e.world)
        if self.comm is not None:
            back, recv = self.comm.exchange(grouped, counts)
            grouped.free()
        else:
            back = grouped
        col = lib.collapse(back)
        if self.debezium:
            out = lib.debezium_emit(self.dopts, col, self.meta if e.world == 1 else None)
            if keep:
                nv = int(len(out) - out.val_null.sum())
                self.state.update(in_bytes=self.db.payload_bytes(), out_rows=col.nrows, out_bytes=out.values.size + out.keys.size, messages=len(out),
                                  const_bytes=len(out) * self.const_key + nv * self.const_val, tombstones=int(out.val_null.sum()))
            out.keys.free()
        else:
            out = lib.queue_serialize(self.qopts, col, self.meta if e.world == 1 else None)
            if keep:
                self.state.update(in_bytes=self.db.payload_bytes(), out_rows=col.nrows, out_bytes=out.values.size, messages=len(out))
        out.values.free(); col.free(); back.free(); one.free()

    def alg(self):
        s = self.state
        if self.debezium:   # dbz_fill_const writes the schema halves (their source is a few KB, cache-resident); the cell pass reads the columns and writes the payload halves
            cells = s["in_bytes"] + s["out_bytes"] - s["const_bytes"]
            return {"dbz_fill_const": s["const_bytes"], "dbz_walk_write": cells, "dbz_cell_write": cells, "compact_gather": 2 * s["in_bytes"], "sharder_crc32": 12 * self.n}
        return {"ser_cell_write": s["in_bytes"] + s["out_bytes"], "compact_gather": 2 * s["in_bytes"], "sharder_crc32": 12 * self.n}

    def config(self):
        e = self.env
        return {"workload": "CDC slice (35% insert / 45% update / 20% delete over rows/4 keys) -> sharder CRC32 % world -> tfgpu_partition -> all-to-all -> Collapse -> "
                            + ("Debezium emitter: Emitter.EmitKV per row, key + value, PackerIncludeSchema" if self.debezium else "native queue serializer") +
                            " (BASELINE.json configs[4] from the decoded batch on)", "rows_per_gpu_per_step": self.n, "sink": "debezium" if self.debezium else "native",
                "exchange": "tfgpu_exchange: one grouped RCCL send/recv over all column buffers" if self.comm is not None else "none (1 rank: every row stays)",
                "parallelism": f"hash-partition x{e.world}"}

    def extra(self):
        s = self.state
        return {"rows_out_per_step": s["out_rows"], "messages_out_per_step": s["messages"], "text_out_bytes_per_step": s["out_bytes"]}

    def cpu(self):
        from oracle import oracle as ora
        e = self.env
        k = min(max(self.args.cpu_rows, 1 << 16), self.n, 1 << 18)
        b2, s2 = e.workload.cdc_batch(k)
        r1 = ora.collapse(b2, s2)
        a = r1.batch
        a.schema = s2
        meta = e.abi.row_meta(k, ids=np.arange(k) % 97, lsns=np.arange(k, dtype=np.uint64) + 5, commit_times=np.full(k, 1700000000000000000, np.uint64))
        if self.debezium:
            import time
            from oracle import dbz_emitter as E
            em = E.Emitter(self.dbz_params, "1.1.2.Final")
            cols = [E.Col(c.name, c.dtype, c.key, c.original_type) for c in self.pg_schema.cols]
            m = min(a.nrows, 1 << 13)   # the emitter's restatement is Python: a bounded slice of the collapsed rows
            names = [c.name for c in a.cols]
            t0 = time.perf_counter()
            nmsg = 0
            emitted = []
            for r in range(m):
                has = bool(a.old_present[r]) if getattr(a, "old_present", None) is not None else bool(getattr(a, "old_keys", None))
                sr_ = int(a.src_row[r]) if a.src_row is not None else r
                it = E.Item({0: "insert", 1: "update", 2: "delete"}.get(int(a.kind[r]), "other") if a.kind is not None else "insert", "public", "events", cols, names,
                            [tuple(c.pyvalue(r)) for c in a.cols], [c.name for c in a.old_keys] if has else [], [tuple(c.pyvalue(r)) for c in a.old_keys] if has else [],
                            int(sr_ % 97), sr_ + 5, 1700000000000000000)
                kv = em.emit_kv(it)
                emitted += kv
                nmsg += len(kv)
            t_emit = time.perf_counter() - t0
            sec = r1.seconds + t_emit * (a.nrows / max(m, 1))
            out = {"value": round(k / sec, 1), "unit": "rows/s", "cores": 1, "kind": "port",
                   "sample": f"a {k}-row slice of the same stream: oracle Collapse ({r1.seconds:.2f}s) + the emitter's Python restatement on {m} of its {a.nrows} surviving rows "
                             f"({t_emit:.2f}s, {nmsg} messages; scaled to the slice), single thread",
                   "note": "the emitter leg is a PYTHON restatement (oracle/dbz_emitter.py): a parity checker, far slower than the Go emitter — not a performance baseline"}
            out.update(host_info())

            def check_dbz():
                lib, abi = e.lib, e.abi
                head = abi.Batch([abi.Column(c.name, c.dtype, c.repr, values=None if c.values is None else c.values[:m], nanos=None if c.nanos is None else c.nanos[:m],
                                             offsets=None if c.offsets is None else c.offsets[:m + 1].copy(), data=None if c.data is None else c.data[:int(c.offsets[m])],
                                             validity=None if c.validity is None else c.validity[:m]) for c in a.cols], m, a.table_ns, a.table_name,
                                 kind=None if a.kind is None else a.kind[:m], src_row=None if a.src_row is None else a.src_row[:m])
                head.schema = s2
                if a.old_keys:
                    head.old_keys = [abi.Column(c.name, c.dtype, c.repr, values=None if c.values is None else c.values[:m], offsets=None if c.offsets is None else c.offsets[:m + 1].copy(),
                                                data=None if c.data is None else c.data[:int(c.offsets[m])], validity=None if c.validity is None else c.validity[:m]) for c in a.old_keys]
                    head.old_present = None if a.old_present is None else a.old_present[:m]
                got = lib.debezium_emit(self.dopts, lib.DeviceBatch.upload(head), meta).messages()
                want = [(kk_, v_) for kk_, v_ in emitted]
                if [(bytes(x) if x is not None else None, bytes(y) if y is not None else None) for x, y in got] != [(bytes(x) if x is not None else None, bytes(y) if y is not None else None) for x, y in want]:
                    return _parity(m, "Debezium emitter", "messages differ (%d vs %d)" % (len(got), len(want)))
                return _parity(m, "Debezium emitter: every key and value message (inline schemas, tombstones) of the first %d collapsed rows against oracle/dbz_emitter.py, byte for byte" % m, compared_messages=len(want))
            out["parity"] = _guard_parity(check_dbz)
            return out
        want = ora.queue_serialize(self.qopts, a, s2, meta)
        sec = r1.seconds + ora.queue_serialize.seconds
        out = {"value": round(k / sec, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"a {k}-row slice of the same stream: oracle Collapse ({r1.seconds:.2f}s) + native serializer ({ora.queue_serialize.seconds:.2f}s), single thread",
               "note": "C restatement of the Go reference (json.Marshal key strings, string-keyed maps), not the Go binary; context: the reference's own debezium parser "
                       "benchmark tops out at ~26 k msg/s on 10 cores (multithreadig_test.md)"}
        out.update(host_info())

        def check():
            lib, abi = e.lib, e.abi
            one = self.shard.apply(lib.DeviceBatch.upload(b2)).transformed
            grouped, counts = lib.partition(one, e.world)
            back = self.comm.exchange(grouped, counts)[0] if self.comm is not None else grouped
            col = lib.collapse(back)
            host = col.download()
            why = _batch_diff(abi, host, a)
            if why is None and not np.array_equal(host.src_row, a.src_row):
                why = "kept rows (src_row)"
            if why is None and host.kind is not None and a.kind is not None and not np.array_equal(host.kind, a.kind):
                why = "kinds"
            if why:
                return _parity(k, "sharder + partition + exchange + Collapse", why)
            got = lib.queue_serialize(self.qopts, col, meta)
            if want is None or bytes(got.values.download()) != b"".join(want) or len(got) != len(want):
                return _parity(k, "native queue serializer", "messages differ")
            return _parity(k, "sharder + partition + exchange (1 rank) + Collapse (cell for cell, kept rows, kinds) + native queue serializer (byte for byte)",
                           compared_output_rows=host.nrows, compared_messages=len(want))
        out["parity"] = _guard_parity(check)
        return out


class Configs4DebeziumWorkload(Base):
    """BASELINE.json configs[4] end to end from its real source format: Debezium-envelope bytes (Postgres CDC, inline schema) →
    tfgpu_debezium_unpack / parse → sharder CRC32 % world → tfgpu_partition → tfgpu_exchange → Collapse → native queue serializer
    (Kafka-ready messages).  Everything between the Kafka bytes in and the Kafka bytes out runs on the device."""
    metric = "messages/sec from Debezium-envelope bytes through parse -> hash-partition (RCCL) -> Collapse -> native queue serializer"
    default_rows = 1 << 17

    def setup(self):
        e, a = self.env, self.args
        lib, abi = e.lib, e.abi
        from transferia_amd import debezium
        self.n = a.rows
        msgs = e.workload.debezium_cdc_messages(self.n, seed=0x5EED + e.rank)
        self.first = msgs[0]
        data, self.msgs = abi.messages(msgs)
        self.nbytes = len(data)
        self.dbuf = lib.DeviceBuffer.upload(data)
        self.parser = debezium.Parser(lib)
        self.shard = lib.Transformer("sharder_transformer", {"shardsCount": str(e.world), "columns": {"includeColumns": ["^id$"]}, "tables": {}})
        self.qopts = None
        if e.world > 1:
            import torch.distributed as dist
            from transferia_amd import partition
            self.comm = partition.device_comm(dist, lib)
        else:
            self.comm = lib.Comm.create(lib.Comm.unique_id(), 0, 1) if a.exchange else None

    def rows(self):
        return self.n

    def step(self, keep=False):
        e = self.env
        lib, abi = e.lib, e.abi
        parsed, errors = self.parser.parse(self.dbuf, self.msgs, host_bytes=self.first)
        p = parsed[0]
        if self.qopts is None:
            self.qopts = abi.queue_options(abi.QFMT_NATIVE, enabled=True, max_message_size=1 << 20, table_schema=p.schema, old_key_types=["int64"])
        meta = None
        if e.world == 1:  # ID / LSN / CommitTime / names_form ride on src_row = message index
            ids, lsns, cts, nf = p.meta()   # the receiver lays them out by message index (tfgpu_dbz_receive_group_meta)
            meta = abi.row_meta(self.n, ids=ids, lsns=lsns, commit_times=cts, names_form=nf)
        one = self.shard.apply(p.batch).transformed
        grouped, counts = lib.partition(one, e.world)
        if self.comm is not None:
            back, _recv = self.comm.exchange(grouped, counts)
            grouped.free()
        else:
            back = grouped
        col = lib.collapse(back)
        out = lib.queue_serialize(self.qopts, col, meta)
        if keep:
            self.state.update(in_rows=p.batch.nrows, out_rows=col.nrows, out_bytes=out.values.size, messages=len(out), errors=len(errors))
        out.values.free(); col.free(); back.free(); one.free(); p.batch.free()

    def alg(self):
        return {"dbz_unpack": self.nbytes, "dbz_parse": self.nbytes // 6, "dbz_parse_quick": self.nbytes // 6}

    def config(self):
        e = self.env
        return {"workload": "Postgres CDC in Debezium's JSON envelope (inline Kafka Connect schema, ~2.3 KB per message; 35% c / 45% u / 20% d over n/4 keys) -> "
                            "Debezium receiver -> sharder -> tfgpu_partition -> tfgpu_exchange -> Collapse -> native queue serializer (BASELINE.json configs[4], end to end)",
                "messages_per_gpu_per_step": self.n, "bytes_per_step": self.nbytes, "parallelism": f"hash-partition x{e.world}"}

    def extra(self):
        s = self.state
        return {"rows_parsed_per_step": s["in_rows"], "rows_out_per_step": s["out_rows"], "messages_out_per_step": s["messages"], "text_out_bytes_per_step": s["out_bytes"],
                "gb_per_s_in": round(self.nbytes * self.total_passes / self.dt / 1e9, 2)}

    def cpu(self):
        from oracle import oracle as ora
        e = self.env
        k = max(256, min(self.n, self.args.cpu_rows // 32))
        data, msgs = e.abi.messages(e.workload.debezium_cdc_messages(k, seed=0x5EED))
        ora.debezium_parse.want_items = False
        ora.debezium_parse(data, msgs)
        sec = ora.debezium_parse.seconds
        out = {"value": round(k / sec, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} messages of the same stream through the oracle's Debezium receiver only ({sec:.2f}s) — the parse half; Collapse + native serializer add "
                         f"~1.2 us per row (bench.py --workload configs4)",
               "note": "C restatement of the Go reference, not the Go binary; published Go figures for the parse half: 3.1 k (1 thread) … 14.8 k (64 threads) msg/s on an M1 Pro"}
        out.update(host_info())

        def check():
            td = _test_helpers("test_debezium")
            lib, abi = e.lib, e.abi
            kk = min(k, 2048)
            mlist = e.workload.debezium_cdc_messages(kk, seed=0x5EED)
            d2, m2 = abi.messages(mlist)
            ora.debezium_parse.want_items = True
            exp_items, codes = ora.debezium_parse(d2, m2)
            ora.debezium_parse.want_items = False
            parsed, errors = self.parser.parse(d2, m2)
            if len(parsed) != 1 or errors:
                return _parity(kk, "parse", "%d batches, %d errors" % (len(parsed), len(errors)))
            p = parsed[0]
            got = td.device_items(lib, p)
            for it in exp_items:
                td.assert_same_items(got[it["src"]], it, it["src"])
            one = self.shard.apply(p.batch).transformed
            grouped, counts = lib.partition(one, e.world)
            back = self.comm.exchange(grouped, counts)[0] if self.comm is not None else grouped
            col = lib.collapse(back)
            a = col.download()
            host = p.batch.download(); host.schema = p.schema
            ref = ora.collapse(host, p.schema).batch
            key = lambda b: sorted((int(b.kind[i]), int(b.src_row[i])) for i in range(b.nrows))
            if key(a) != key(ref):
                return _parity(kk, "Collapse", "surviving rows differ")
            ids, lsns, cts, nf = p.meta()
            meta = abi.row_meta(kk, ids=ids, lsns=lsns, commit_times=cts, names_form=nf)
            o = abi.queue_options(abi.QFMT_NATIVE, enabled=True, max_message_size=1 << 20, table_schema=p.schema, old_key_types=["int64"])
            outm = lib.queue_serialize(o, col, meta).messages()
            a.schema = p.schema
            if outm != ora.queue_serialize(o, a, p.schema, meta):
                return _parity(kk, "native queue serializer", "messages differ")
            return _parity(kk, "Debezium receiver (item for item) + sharder + partition + exchange (1 rank) + Collapse (surviving rows) + native queue serializer (byte for byte)",
                           compared_output_rows=a.nrows, compared_messages=len(outm))
        out["parity"] = _guard_parity(check)
        return out


class CollapseWorkload(Base):
    metric = "ChangeItems/sec through abstract.Collapse (PK-keyed dedup of a CDC batch)"
    default_rows = 1 << 20

    def setup(self):
        e, a = self.env, self.args
        self.toast = float(getattr(a, "toast", 0.0) or 0.0)
        b, self.schema = e.workload.cdc_batch(a.rows, toast=self.toast)
        self.n = a.rows
        self.db = e.lib.DeviceBatch.upload(b)

    def rows(self):
        return self.n

    def step(self, keep=False):
        out = self.env.lib.collapse(self.db)
        if keep:
            self.state.update(out_rows=out.nrows)
        out.free()

    def alg(self):
        return {}

    def config(self):
        return {"workload": "CDC slice (35% insert / 45% update / 20% delete over rows/4 keys, 60% of U/D with OldKeys, 15% of those change the PK) -> Collapse"
                            + (" — %.0f%% of the Updates leave the text column out (TOAST): compareColumns merges on the device" % (100 * self.toast) if self.toast else ""),
                "rows_per_gpu_per_step": self.n, "rows_out_per_step": self.state["out_rows"], "toast": self.toast}

    def cpu(self):
        from oracle import oracle as ora
        if self.toast:
            return self.cpu_toast(ora)
        k = min(max(self.args.cpu_rows, 1 << 18), self.n)
        b2, s2 = self.env.workload.cdc_batch(k)
        r1 = ora.collapse(b2, s2)
        out = {"value": round(k / r1.seconds, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"a {k}-row slice of the same stream through the oracle's Collapse ({r1.seconds:.2f}s), single thread",
               "note": "C restatement of the Go reference (json.Marshal key strings, string-keyed maps, boxed values), not the Go binary"}
        out.update(host_info())

        def check():
            e = self.env
            got = e.lib.collapse(e.lib.DeviceBatch.upload(b2)).download()
            why = _batch_diff(e.abi, got, r1.batch)
            if why is None and not (np.array_equal(got.src_row, r1.batch.src_row) and np.array_equal(got.kind, r1.batch.kind)):
                why = "kept rows / kinds"
            if why is None and (got.old_keys or r1.batch.old_keys):
                ga, gb = got.old_keys or [], r1.batch.old_keys or []
                pa_, pb_ = got.old_present, r1.batch.old_present
                if [c.name for c in ga] != [c.name for c in gb] or not np.array_equal(pa_ if pa_ is not None else np.ones(got.nrows, bool), pb_ if pb_ is not None else np.ones(got.nrows, bool)):
                    why = "OldKeys"
            return _parity(k, "Collapse: surviving rows, their order, kinds, cells and OldKeys presence against the oracle", why, compared_output_rows=got.nrows)
        out["parity"] = _guard_parity(check)
        return out


def _collapse_cpu_toast(self, ora):
    """The TOAST stream through the oracle's ROW-WISE Collapse (items with their own ColumnNames, as the Go loop sees them) on a bounded slice,
    and the device's rows for the same slice against it: names, their order, values, kinds, OldKeys, source rows."""
    cc = _test_helpers("collapse_cases")
    k = min(self.n, 1 << 16)
    b2, s2 = self.env.workload.cdc_batch(k, toast=self.toast)
    b2.schema = s2
    items = [dict(it, keys=["id"], old_names=[o[0] for o in it["old"]], old_values=[o[1] for o in it["old"]]) for it in cc.items_of(b2)]
    for it in items:
        it["values"] = [[v[0], v[1].decode("latin-1") if isinstance(v[1], (bytes, bytearray)) else v[1]] for v in it["values"]]
    t0 = time.perf_counter()
    want = ora.collapse_rows(items)
    sec = time.perf_counter() - t0
    out = {"value": round(k / sec, 1), "unit": "rows/s", "cores": 1, "kind": "port",
           "sample": f"a {k}-row slice of the same stream through the oracle's row-wise Collapse ({sec:.2f}s INCLUDING the JSON hand-over of the items to the C oracle), single thread",
           "note": "a parity checker's timing, not a performance baseline (the items travel as JSON text)"}
    out.update(host_info())

    def check():
        e = self.env
        got = cc.items_of(e.lib.collapse(e.lib.DeviceBatch.upload(b2)).download())
        exp = cc.norm_items(want)
        why = None
        if len(got) != len(exp):
            why = "row count %d != %d" % (len(got), len(exp))
        else:
            for i, (g, w) in enumerate(zip(got, exp)):
                if g != w:
                    why = "row %d differs" % i
                    break
        return _parity(k, "Collapse over TOAST rows: every surviving row's ColumnNames (and their order), values, kind, OldKeys and source row against the oracle's row-wise Go loop", why, compared_output_rows=len(got))
    out["parity"] = _guard_parity(check)
    return out


CollapseWorkload.cpu_toast = _collapse_cpu_toast


class DebeziumWorkload(Base):
    """The ingest half of configs[4] from real envelope bytes, in the shape of the reference's own benchmark
    (BenchmarkParsingViaMultithreading, pkg/parsers/registry/debezium/engine/bench/parser_bench_test.go:18-40): ONE 13.6 KB
    Postgres event with its inline schema (engine/parser_test.jsonl, carried in tests/golden/debezium.json) replicated per
    batch, every message → one ChangeItem of 61 columns."""
    metric = "messages/sec through the Debezium parser (inline schema, 13.6 KB Postgres event replicated per batch) -> device ChangeItem columns"
    default_rows = 1 << 17

    def setup(self):
        e, a = self.env, self.args
        lib, abi = e.lib, e.abi
        from transferia_amd import debezium
        with open(os.path.join(ROOT, "tests", "golden", "debezium.json")) as f:
            self.msg = [c for c in json.load(f)["cases"] if c["name"] == "TestParser"][0]["message"].encode("utf-8")
        self.n = a.rows
        data, self.msgs = abi.messages([self.msg] * self.n)
        self.nbytes = len(data)
        self.dbuf = lib.DeviceBuffer.upload(data)
        self.parser = debezium.Parser(lib)

    def rows(self):
        return self.n

    def step(self, keep=False):
        parsed, errors = self.parser.parse(self.dbuf, self.msgs, host_bytes=self.msg)
        if keep:
            self.state.update(out_rows=sum(p.batch.nrows for p in parsed), out_bytes=sum(p.batch.payload_bytes() for p in parsed), errors=len(errors))
        for p in parsed:
            p.batch.free()

    def alg(self):
        pb = self.payload_bytes() + self.state.get("out_bytes", 0)
        return {"dbz_parse": pb, "dbz_parse_quick": pb, "dbz_cell_values": pb, "dbz_cell_text": pb, "dbz_unpack": self.nbytes, "dbz_prefix_same": self.nbytes}

    def payload_bytes(self):
        i = self.msg.index(b'"payload":')
        return (len(self.msg) - i) * self.n  # what dbz_parse walks: the payload member; the schema was hashed by dbz_unpack

    def config(self):
        return {"workload": "Debezium envelope JSON with inline schema, one 13 606-byte Postgres event (61 columns, every pg type) replicated per batch — the shape of the "
                            "reference's BenchmarkParsingViaMultithreading; unpack + schema grouping + receive on device, the schema compiled once on the host",
                "messages_per_gpu_per_step": self.n, "bytes_per_message": len(self.msg), "parallelism": f"message-range shard x{self.env.world}, no collective",
                "reference_cpu_context": "published: 3 056 msg/s (1 thread) … 14 838 msg/s (64 threads), Apple M1 Pro, schema cache disabled "
                                         "(pkg/parsers/registry/debezium/engine/bench/multithreadig_test.md) — other hardware, not vs_baseline"}

    def extra(self):
        return {"gb_per_s_in": round(self.nbytes * self.total_passes / self.dt / 1e9, 2), "column_bytes_out_per_step": self.state["out_bytes"]}

    def cpu(self):
        from oracle import oracle as ora
        abi = self.env.abi
        k = max(64, min(self.n, self.args.cpu_rows // 64))
        data, msgs = abi.messages([self.msg] * k)
        ora.debezium_parse.want_items = False
        items, codes = ora.debezium_parse(data, msgs)
        sec = ora.debezium_parse.seconds
        out = {"value": round(k / sec, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} messages x {len(self.msg)} B: oracle Receive per message, schema re-compiled every message (the reference's benchmark disables its cache too) ({sec:.2f}s)",
               "note": "C restatement of the Go reference, not the Go binary"}
        out.update(host_info())

        def check():
            td = _test_helpers("test_debezium")
            kk = min(k, 256)
            d2, m2 = abi.messages([self.msg] * kk)
            ora.debezium_parse.want_items = True
            exp_items, codes = ora.debezium_parse(d2, m2)
            ora.debezium_parse.want_items = False
            parsed, errors = self.parser.parse(d2, m2)
            if errors or sum(p.batch.nrows for p in parsed) != len(exp_items):
                return _parity(kk, "parse", "%d device errors, %d rows vs %d items" % (len(errors), sum(p.batch.nrows for p in parsed), len(exp_items)))
            got = {}
            for p in parsed:
                got.update(td.device_items(self.env.lib, p))
            for it in exp_items:
                td.assert_same_items(got[it["src"]], it, it["src"])
            return _parity(kk, "Debezium receiver: every message's item (kind, table, schema, 61 column values, OldKeys, row meta) against the oracle", compared_output_rows=len(exp_items))
        out["parity"] = _guard_parity(check)
        return out


class DebeziumSrWorkload(DebeziumWorkload):
    """The same Postgres event as `debezium`, in the wire form a schema registry gives it (NewDebeziumImpl with a registry client): the
    Kafka message is 0x00 | schema id | payload — 1.4 KB instead of 13.6 KB — and the schema is the registry's ConfluentJSONSchema text,
    converted and compiled once per id (tfgpu_dbz_receiver_add_registry_schema); per batch tfgpu_dbz_receive_registry."""
    metric = "messages/sec through the Debezium parser (schema-registry framed: 0x00 | id | payload of the same Postgres event) -> device ChangeItem columns"

    def setup(self):
        e, a = self.env, self.args
        lib, abi = e.lib, e.abi
        from transferia_amd import debezium
        with open(os.path.join(ROOT, "tests", "golden", "debezium.json")) as f:
            inline = [c for c in json.load(f)["cases"] if c["name"] == "TestParser"][0]["message"].encode("utf-8")
        self.schema_text, self.msg = e.workload.registry_framed(inline, 101)
        self.n = a.rows
        data, self.msgs = abi.messages([self.msg] * self.n)
        self.nbytes = len(data)
        self.dbuf = lib.DeviceBuffer.upload(data)
        self.parser = debezium.Parser(lib)
        self.parser.add_registry_schema(101, self.schema_text)

    def step(self, keep=False):
        parsed, errors, events = self.parser.parse_registry(self.dbuf, self.msgs)
        if keep:
            self.state.update(out_rows=sum(p.batch.nrows for p in parsed), out_bytes=sum(p.batch.payload_bytes() for p in parsed), errors=len(errors))
        for p in parsed:
            p.batch.free()

    def alg(self):
        pb = self.nbytes + self.state.get("out_bytes", 0)
        return {"dbz_parse": pb, "dbz_parse_quick": pb, "dbz_cell_values": pb, "dbz_cell_text": pb, "dbz_registry_frames": self.nbytes, "sr_frames": self.nbytes}

    def config(self):
        return {"workload": "Debezium events framed by a schema registry: 0x00 | schema id | payload of one Postgres event (61 columns, every pg type) replicated per batch; "
                            "the registry's ConfluentJSONSchema converted + compiled once on the host, framing + payload spans + receive on device",
                "messages_per_gpu_per_step": self.n, "bytes_per_message": len(self.msg), "parallelism": f"message-range shard x{self.env.world}, no collective"}

    def cpu(self):
        from oracle import oracle as ora
        abi = self.env.abi
        k = max(64, min(self.n, self.args.cpu_rows // 64))
        data, msgs = abi.messages([self.msg] * k)
        ora.debezium_parse.want_items = False
        t0 = time.perf_counter()
        ora.debezium_parse_sr(data, msgs, {101: self.schema_text})
        sec = time.perf_counter() - t0
        out = {"value": round(k / sec, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} events x {len(self.msg)} B: the oracle's DoBatch with a registry (Python cut + the C Receive per event, schema compiled per event) ({sec:.2f}s)",
               "note": "C / Python restatement of the Go reference, not the Go binary"}
        out.update(host_info())

        def check():
            td = _test_helpers("test_debezium")
            kk = min(k, 256)
            d2, m2 = abi.messages([self.msg] * kk)
            ora.debezium_parse.want_items = True
            exp_events, exp_items, codes = ora.debezium_parse_sr(d2, m2, {101: self.schema_text})
            ora.debezium_parse.want_items = False
            parsed, errors, events = self.parser.parse_registry(d2, m2)
            if errors or sum(p.batch.nrows for p in parsed) != len(exp_items):
                return _parity(kk, "parse", "%d device errors, %d rows vs %d items" % (len(errors), sum(p.batch.nrows for p in parsed), len(exp_items)))
            got = {}
            for p in parsed:
                got.update(td.device_items(self.env.lib, p))
            for ev, it in exp_items.items():
                td.assert_same_items(got[ev], it, ev)
            return _parity(kk, "registry-framed Debezium receiver: framing + every event's item against the oracle", compared_output_rows=len(exp_items))
        out["parity"] = _guard_parity(check)
        return out


class SrProtoWorkload(DebeziumWorkload):
    """Confluent-SR wire bytes with a PROTOBUF schema: the reference's own 60-column test message (engine/testdata/test_protobuf_1.bin with schema 6
    of test_schemas.json — every pg type through the Confluent protobuf converter, nested Point / VariableScaleDecimal / Decimal messages) replicated
    per batch, every message → one ChangeItem (tfgpu_sr_proto_parse; the schema compiled once: tfgpu_sr_compile_proto)."""
    metric = "messages/sec through the Confluent-SR parser, PROTOBUF schema (the reference's 784-byte 60-column test message replicated per batch) -> device ChangeItem columns"
    default_rows = 1 << 18

    def setup(self):
        e, a = self.env, self.args
        lib, abi = e.lib, e.abi
        import base64
        from transferia_amd import confluent_sr
        with open(os.path.join(ROOT, "tests", "golden", "sr_protobuf.json")) as f:
            case = json.load(f)["cases"][1]
        self.msg, self.sid, self.text = base64.b64decode(case["message_b64"]), int(case["schema_id"]), case["schema"].encode()
        self.n = a.rows
        data, self.msgs = abi.messages([self.msg] * self.n)
        self.nbytes = len(data)
        self.dbuf = lib.DeviceBuffer.upload(data)
        self.schema = confluent_sr.ProtoSchema(lib, self.text)
        assert self.schema.code == abi.ROW_OK, self.schema.why

    def step(self, keep=False):
        batch, errors = self.schema.parse(self.sid, self.dbuf, self.msgs)
        if keep:
            self.state.update(out_rows=batch.nrows, out_bytes=batch.payload_bytes(), errors=len(errors))
        batch.free()

    def alg(self):
        pb = self.nbytes + self.state.get("out_bytes", 0)
        return {"pb_decode": self.nbytes, "pb_cells": pb, "pb_text": pb}

    def config(self):
        return {"workload": "Confluent-SR wire bytes, PROTOBUF schema: 0x00 | schema id | 0x00 | protobuf of the reference's 60-column test message, replicated per batch; "
                            "wire format decoded on device, the .proto text compiled once on the host",
                "messages_per_gpu_per_step": self.n, "bytes_per_message": len(self.msg), "parallelism": f"message-range shard x{self.env.world}, no collective"}

    def cpu(self):
        from oracle import ora_protobuf as P
        k = max(64, min(self.n, self.args.cpu_rows // 256))
        t0 = time.perf_counter()
        P.parse_messages([self.msg] * k, {self.sid: self.text})
        sec = time.perf_counter() - t0
        out = {"value": round(k / sec, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} messages x {len(self.msg)} B: the oracle's PYTHON restatement of the protobuf branch ({sec:.2f}s)",
               "note": "a Python parity checker, far slower than the Go parser: not a performance baseline"}
        out.update(host_info())

        def check():
            import struct
            abi = self.env.abi
            kk = min(k, 256)
            want = P.parse_messages([self.msg] * kk, {self.sid: self.text})
            d2, m2 = abi.messages([self.msg] * kk)
            batch, errors = self.schema.parse(self.sid, d2, m2)
            b = batch.download()
            if errors or b.nrows != kk:
                return _parity(kk, "parse", "%d device errors, %d rows" % (len(errors), b.nrows))
            for r, (kind, it) in enumerate(want):
                if kind not in ("item", "row", "ok") and not isinstance(it, dict):
                    return _parity(kk, "parse", "the oracle did not parse message %d (%s)" % (r, kind))
                for c, w in zip(b.cols, it["values"]):
                    g = c.pyvalue(r)
                    same = (g[0] == "nil") if w[0] == "nil" else (g[0] == w[0] and (struct.pack("<d", g[1]) == struct.pack("<d", w[1]) or (g[1] != g[1] and w[1] != w[1]))) if w[0] in ("float32", "float64") \
                        else (bytes(g[1]) == w[1]) if w[0] == "json" else abi.norm_value(g) == abi.norm_value(w)
                    if not same:
                        return _parity(kk, "parse", "message %d column %s differs" % (r, c.name))
            return _parity(kk, "Confluent-SR protobuf parser: every message's 60 column values against the oracle", compared_output_rows=kk)
        out["parity"] = _guard_parity(check)
        return out


WORKLOADS = {"configs0": Configs0Workload, "debezium": DebeziumWorkload, "debezium_sr": DebeziumSrWorkload, "sr_proto": SrProtoWorkload, "configs4d": Configs4DebeziumWorkload, "csv": CsvWorkload, "json": JsonWorkload, "sr": SrWorkload, "configs2": Configs2Workload, "configs3": Configs3Workload,
             "configs4": Configs4Workload, "collapse": CollapseWorkload}


class Env:
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="csv",
                    help="csv = BASELINE.json configs[1] (the bench line); configs0 = the plumbing case; configs2/3/4 = the other GPU configs; json / sr / collapse = component benchmarks")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: enough for a timed region of >= 1 s)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU per step (default: 2^20 for csv / configs3 / configs4, 2^18 for the message workloads)")
    ap.add_argument("--cpu-rows", type=int, default=1 << 19, help="rows of the single-thread CPU-baseline sample (0 = skip); 2^19 hits rows = about 11 s of oracle time")
    ap.add_argument("--cpu-all-rows", type=int, default=1, help="1 = also time the oracle on every hardware thread (persistent workers, csv workload), 0 = skip")
    ap.add_argument("--prof-steps", type=int, default=10, help="passes of the per-kernel HIP-event measurement (median per kernel)")
    ap.add_argument("--passes", type=int, default=0, help="passes of the path per step (0 = as many as make a step of about --step-ms)")
    ap.add_argument("--step-ms", type=float, default=50.0)
    ap.add_argument("--lanes", type=int, default=1, help="device lanes (host threads) the steps are spread over (1 = strictly serial, the bench line)")
    ap.add_argument("--overlap-lanes", type=int, default=3, help="side measurement (never `value`): the same steps spread over this many lanes (0 = skip)")
    ap.add_argument("--no-pull-push", action="store_true", help="csv workload: skip the configs[2] pull || push side measurement")
    ap.add_argument("--pcie-steps", type=int, default=2, help="side measurement (never `value`): passes per lane that start from PINNED HOST memory (0 = skip)")
    ap.add_argument("--pcie-lanes", type=int, default=3)
    ap.add_argument("--devices", default="", help="ONE process driving several devices (csv workload): comma-separated HIP device ids, lane k on the k-th "
                    "(tfgpu_init_devices); without it --gpus N > 1 expects one process per GPU under torch.distributed.run")
    ap.add_argument("--sink", default="native", choices=["native", "debezium"], help="configs4: the queue sink's format — the native serializer, or the Debezium emitter "
                    "(queue.DebeziumSerializer, key + value with inline schemas: tfgpu_debezium_emit)")
    ap.add_argument("--from-rows", action="store_true", help="configs0: also cross the boundary the way the reference would — boxed []interface{} rows fanned out into column buffers, "
                    "one crossing, fanned back in (tools/fanout/fanout_harness.cpp, INTEGRATION.md §2), timed per leg")
    ap.add_argument("--toast", type=float, default=0.0, help="collapse workload: this share of the Updates leaves the text column out of its ColumnNames (TOAST): Collapse runs the compareColumns merge")
    ap.add_argument("--exchange", type=int, default=1, help="configs4 at 1 rank: 1 = every buffer still makes the RCCL round trip, 0 = skip the collective")
    args = ap.parse_args()

    import torch  # first: libtfgpu and torch must share the HIP runtime torch loads
    from transferia_amd import abi, dist as tdist, lib, workload
    if os.environ.get("TFGPU_LIB_VARIANT"):  # measurement only: an A/B build of the same sources (tools/build_variant.sh); the binding itself knows one library
        lib._LIBPATH = os.path.join(os.path.dirname(lib._LIBPATH), "variants", "libtfgpu_%s.so" % os.environ["TFGPU_LIB_VARIANT"])
    rank, local_rank, world = tdist.env_rank()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    e = Env()
    e.rank, e.local_rank, e.world = rank, local_rank, world
    e.device = torch.device("cuda", local_rank)
    e.group = tdist.Group("nccl", e.device)  # RCCL: barrier + MAX of the wall time (+ the configs4 exchange)
    e.lib, e.abi, e.workload = lib, abi, workload
    e.devices = [int(x) for x in args.devices.split(",") if x.strip() != ""] if args.devices else None
    if e.devices:
        if world != 1 or args.workload != "csv":
            raise SystemExit("--devices is the one-process mode of the csv workload: run it without torch.distributed.run")
        lib.init_devices(e.devices)
        args.overlap_lanes = 0; args.pcie_steps = 0
    else:
        lib.init(local_rank)

    def sync_all():
        lib.synchronize()
        e.group.barrier()
        torch.cuda.synchronize()
    e.sync_all = sync_all

    W = WORKLOADS[args.workload](args, e)
    if args.rows <= 0:
        args.rows = W.default_rows
    W.setup()
    for _ in range(args.warmup):
        W.step()
    W.step(keep=True)  # one extra untimed pass records sizes
    # One *step* of the headline loop is `passes` passes of the path (each over the whole resident batch): a pass takes ~1.5 ms,
    # and a driver that asks for --steps 20 would otherwise time a 30 ms region no utilisation sampler can see.
    passes = args.passes
    if passes <= 0:
        lib.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            W.step()
        lib.synchronize()
        per = (time.perf_counter() - t0) / 3
        passes = max(1, int(np.ceil(args.step_ms * 1e-3 / max(per, 1e-6))))
        if e.group.dist is not None:  # every rank must time the same number of passes
            t = torch.tensor([passes], dtype=torch.int64, device=e.device)
            e.group.dist.all_reduce(t, op=e.group.dist.ReduceOp.MAX)
            passes = int(t.item())
    if args.steps <= 0:  # no K given: a timed region of >= 1 s
        args.steps = 24
    W.total_passes = args.steps * passes
    if os.environ.get("TFGPU_BENCH_HOSTPROF") == "1":  # where the HOST spends a pass (stderr; the JSON line is unaffected): cProfile over a few untimed passes
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            W.step()
        lib.synchronize()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
    my_dt = W.timed(args.steps * passes)
    dt = e.group.max_seconds(my_dt)
    W.dt = dt
    # the N-GPU run validates itself: every rank's own time and rows arrive on rank 0 through the collective library
    rank_ms = e.group.all_gather_float(my_dt / args.steps * 1e3)
    rank_rows = e.group.all_gather_float(float(W.rows() * passes))
    ranks_seen = e.group.sum_int(1)
    ms_per_pass = dt / (args.steps * passes) * 1e3

    # ---- per-kernel device time: HIP events on the library stream, IMMEDIATELY after the timed region (before any side
    #      measurement touches clocks, lanes or PCIe), 2 warm-up passes, then >= 10 passes read back one by one: the figure kept
    #      per kernel is the MEDIAN over passes of (sum of its launches in the pass) ----
    lib.prof_reset(); lib.prof_enable(True)
    for _ in range(2):
        W.step()
    lib.prof_reset()
    per_pass, units = {}, {}
    nprof = max(args.prof_steps, 1)
    for i in range(nprof):
        W.step()
        for n, l, ms in lib.prof_get():
            if l:
                per_pass.setdefault(n, []).append((l, ms))
        for n, u in lib.prof_units().items():
            if u:
                units.setdefault(n, []).append(u)
        lib.prof_reset()
    lib.prof_enable(False)
    kernels = {}
    for n, samples in per_pass.items():
        if len(samples) < nprof:  # not launched in every pass: averaged over all passes
            ms_step = sum(ms for _, ms in samples) / nprof
            lps = sum(l for l, _ in samples) / nprof
        else:
            ms_step = float(np.median([ms for _, ms in samples]))
            lps = float(np.median([l for l, _ in samples]))
        kernels[n] = {"launches_per_step": lps, "avg_ms": ms_step / max(lps, 1e-9), "ms_per_step": ms_step}
        if units.get(n):  # the rows the kernel's launches were issued over, as the library counted them (tfgpu_prof_get_units)
            kernels[n]["units_per_step"] = int(np.median(units[n]))
    kernel_sum_ms = sum(v["ms_per_step"] for v in kernels.values())

    side = W.side_measurements() if hasattr(W, "side_measurements") else {}

    W.kernels = kernels
    alg = W.alg()
    dom = max(kernels.items(), key=lambda kv: kv[1]["ms_per_step"])[0] if kernels else None
    roofline = None
    if dom and alg.get(dom):
        per_launch = alg[dom] / max(kernels[dom]["launches_per_step"], 1)
        # The kernels of a pass cannot take longer than the pass that contains them.  When the event pass says otherwise (another
        # clock state, a noisy box) the dominant kernel is bounded from the TIMED region instead: pass time minus the other kernels.
        consistent = kernel_sum_ms <= 1.05 * ms_per_pass
        launch_ms, source = kernels[dom]["avg_ms"], "hip_events_median"
        if not consistent:
            others = kernel_sum_ms - kernels[dom]["ms_per_step"]
            bound_ms = max(ms_per_pass - others * ms_per_pass / kernel_sum_ms, 1e-6) if others >= ms_per_pass else ms_per_pass - others
            launch_ms, source = bound_ms / max(kernels[dom]["launches_per_step"], 1), "upper bound: timed pass minus the other kernels (event pass inconsistent with the timed region)"
        achieved = per_launch / (launch_ms * 1e-3) / 1e9
        # HBM traffic per launch cannot be read inside this process: it comes from the rocprofv3 --pmc passes of tools/gpu_visit.sh
        # (FETCH_SIZE + WRITE_SIZE with the guide's gfx950 corrections), which stamp profiles/pmc_traffic.json with the sha256 of
        # the kernel's source file — a number measured on another build of the kernel (or another launch shape) is not printed.
        traffic = None
        try:
            import hashlib
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f).get(dom)
            if t and t["rows_per_launch"] == W.rows() and t.get("workload", "csv") == args.workload:
                with open(os.path.join(ROOT, "transferia_amd", "csrc", t["source_file"]), "rb") as f:
                    fresh = hashlib.sha256(f.read()).hexdigest() == t["source_sha256"]
                traffic = t["bytes_per_launch"] if fresh else None
        except (OSError, ValueError, KeyError):
            pass
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "algorithmic_bytes_per_launch": int(per_launch), "avg_launch_ms": round(launch_ms, 4), "launch_ms_source": source,
                    "consistent": bool(consistent), "kernels_ms_per_pass": round(kernel_sum_ms, 4), "timed_ms_per_pass": round(ms_per_pass, 4), "event_passes": nprof}
        # `achieved` counts the conservative figure (input bytes + fixed-width values).  The CSV parse kernel also writes a (length, position)
        # pair per text cell — the late-materialised form of SURVEY 8(d)'s "every output column byte", 8 bytes per cell instead of offset + text:
        # stated next to it, never instead of it
        extra = W.alg_views().get(dom) if hasattr(W, "alg_views") else None
        if extra:
            extra = extra / max(kernels[dom]["launches_per_step"], 1)
            roofline["text_view_bytes_per_launch"] = int(extra)
            roofline["frac_with_text_views"] = round((per_launch + extra) / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    int_roof = W.int_roofline(kernels)
    for k, v in kernels.items():
        b = alg.get(k)
        if b:
            v["alg_gb_s"] = round(b / max(v["launches_per_step"], 1) / (v["avg_ms"] * 1e-3) / 1e9, 2)
        v["avg_ms"], v["ms_per_step"] = round(v["avg_ms"], 4), round(v["ms_per_step"], 4)

    cpu = W.cpu() if (rank == 0 and world == 1 and args.cpu_rows > 0) else None  # the oracle, rank 0, N = 1 only

    if rank == 0:
        rows = W.rows()
        shards = len(e.devices) if e.devices else world
        value = rows * shards * args.steps * passes / dt
        out = {"metric": W.metric, "value": round(value, 1), "unit": "rows/s", "n_gpus": len(set(e.devices)) if e.devices else world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": W.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": W.config()}
        out["passes_per_step"] = passes
        out["config"]["rows_per_pass"] = rows
        out["config"]["rows_per_gpu_per_step"] = rows * passes
        for key in [k for k in out["config"] if k.endswith("_bytes_per_gpu_per_step")]:  # setup() knows one pass; a step is `passes` of them
            out["config"][key.replace("_per_gpu_per_step", "_per_pass")] = out["config"][key]
            out["config"][key] = out["config"][key] * passes
        out.update(W.extra())
        if e.devices:
            out["process_model"] = {"mode": "one process, one host thread and one lane per device (tfgpu_init_devices)", "devices": e.devices, "row_range_shards": shards}
        out["multi_gpu"] = {"rccl_ranks_seen": ranks_seen, "world_size": world, "per_rank_ms_per_step": [round(x, 3) for x in rank_ms],
                            "per_rank_rows_per_step": [int(x) for x in rank_rows],
                            "n1_equivalent": "every rank runs exactly the N=1 step over its own row-range shard (rows %d.. of the same synthetic table): value = sum of per_rank_rows_per_step x steps / max time" % (rows * (world - 1)),
                            "measured_beyond_one_gpu": "nothing beyond N=1 had been measured on hardware when this was written (README.md): at N>1 this line is the first evidence" if world > 1 else None}
        out["row_errors"] = W.state.get("errors", 0)
        out["roofline"] = roofline
        if int_roof:
            out["int_roofline"] = int_roof
        out["cpu_baseline"] = cpu
        out["kernels"] = kernels
        out.update(side)
        # A fraction above 1 means the arithmetic around a kernel is wrong (or the kernel is not doing the work it is priced for):
        # such a line is not printed.
        bad = [(n, r.get("kernel"), f) for n, r in (("roofline", roofline), ("int_roofline", int_roof)) if r
               for f in (r.get("frac"), r.get("issue_frac"), r.get("frac_with_text_views")) if f is not None and f > 1.0]
        for n, v in kernels.items():  # a secondary kernel priced against bytes it does not see (a hand-over path that ran on a few rows): its figure is withheld, not printed
            if v.get("alg_gb_s", 0) > HBM_PEAK_GBS and n != dom:
                v["alg_gb_s"] = None
                v["alg_note"] = "launched on a fraction of the batch: not priced"
        if bad:
            raise SystemExit("bench.py refuses to print a roofline fraction above 1: %r" % bad)
        if cpu and isinstance(cpu.get("parity"), dict):
            out["parity_checked_rows"] = cpu["parity"].get("checked_input_rows", 0) if cpu["parity"].get("identical") else 0
            out["parity"] = {k: v for k, v in cpu["parity"].items() if k in ("identical", "checked", "checked_input_rows", "error")}
        # the comparison that means something end to end: input from pinned host memory (PCIe inside) against the CPU path on
        # every host core; the HBM-resident `value` over one CPU core is a ratio of two different jobs and is not printed
        pc = (side.get("pcie_inclusive") or {}) if isinstance(side, dict) else {}
        best = max((v.get("rows_per_s", 0) for k, v in pc.items() if isinstance(v, dict)), default=0)
        if cpu and best and isinstance(cpu.get("all_cores"), dict) and cpu["all_cores"].get("value"):
            out["pcie_inclusive_vs_cpu_all_cores"] = round(best / cpu["all_cores"]["value"], 1)
            out["pcie_inclusive_vs_cpu_all_cores_caveat"] = ("the CPU side is a C restatement of the Go reference, by its own note ~2-3x slower per core than the Go binary would be: "
                                                             "against the Go binary read this ratio as roughly a third to a half of what is printed")
        try:  # RCCL prints its version banner through C stdio: flush that first, the JSON line is the last one
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()
    e.group.close()
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return  # under rocprofv3 the tool writes its files at normal interpreter exit
    os._exit(0)  # nothing may print after the JSON line


if __name__ == "__main__":
    main()
