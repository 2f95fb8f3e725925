"""bench/common.py — what every workload of bench.py shares: constants (peaks, the chains of BASELINE.json's configurations), the synthetic
inputs, the parity helpers (the cpu_baseline leg's by-product: each line re-checks its own output against the oracle) and the Base class."""
import argparse  # noqa: F401
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from .cpu_workers import host_info, usable_cores, all_cores_csv, run_threads  # noqa: E402,F401

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


def _columns_diff(abi, dev, ref):
    """None when two host batches hold the same columns BIT FOR BIT — names, representations, validity, offsets + bytes, values (floats by
    their bit patterns), nanoseconds — compared as arrays (the form a 65 536-row parity leg can afford), else what differs first"""
    if dev.nrows != ref.nrows:
        return "rows %d vs %d" % (dev.nrows, ref.nrows)
    if dev.nrows == 0:
        return None
    if [c.name for c in dev.cols] != [c.name for c in ref.cols]:
        return "column names"
    n = dev.nrows
    for a, b in zip(dev.cols, ref.cols):
        if a.repr != b.repr:
            if a.repr == abi.R_JSON:   # an `any` column holds json.Marshal's text of the oracle's Go value: cell by cell (see _cells_same)
                for i in range(n):
                    if not _cells_same(abi.norm_value(a.pyvalue(i)), abi.norm_value(b.pyvalue(i))):
                        return "column %s (any): row %d" % (a.name, i)
                continue
            return "column %s: representation %d vs %d" % (a.name, a.repr, b.repr)
        va = a.validity if a.validity is not None else np.ones(n, bool)
        vb = b.validity if b.validity is not None else np.ones(n, bool)
        if not np.array_equal(va, vb):
            return "column %s: validity (first at row %d)" % (a.name, int(np.flatnonzero(va != vb)[0]))
        if a.repr in abi.VAR_REPRS:
            if not np.array_equal(a.offsets, b.offsets):
                return "column %s: offsets (first at row %d)" % (a.name, int(np.flatnonzero(np.asarray(a.offsets) != np.asarray(b.offsets))[0]) - 1)
            m = int(a.offsets[-1])
            if bytes(a.data[:m]) != bytes(b.data[:m]):
                return "column %s: text bytes" % a.name
        else:
            x, y = np.ascontiguousarray(a.values)[va], np.ascontiguousarray(b.values)[vb]
            if x.dtype.kind == "f":
                x, y = x.view("u%d" % x.dtype.itemsize), y.view("u%d" % y.dtype.itemsize)
            if not np.array_equal(x, y):
                return "column %s: values (first at valid row %d)" % (a.name, int(np.flatnonzero(x != y)[0]))
            if a.repr == abi.R_TIME:
                na = a.nanos if a.nanos is not None else np.zeros(n, np.int32)
                nb = b.nanos if b.nanos is not None else np.zeros(n, np.int32)
                if not np.array_equal(na[va], nb[vb]):
                    return "column %s: nanoseconds" % a.name
    return None


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

    def overlapped_lanes(self, nl, steps):
        """Never `value`: the same steps dealt round-robin to `nl` device lanes, each with its own copy of the workload (its own staged input, parser /
        plan objects and host thread) — the parsequeue's shape (parse of batch N+1 beside the transform of batch N, parsequeue.go:118-154): what one lane
        leaves idle between its kernels (read-backs, the host's share of a call) is filled by another lane's kernels."""
        import threading
        e = self.env
        nl = max(1, min(nl, e.lib.lane_count()))
        if nl < 2:
            return None
        go, done = threading.Barrier(nl + 1), threading.Barrier(nl + 1)
        errs = []

        def lane_main(k):
            try:
                e.lib.lane_use(k)
                w = self
                if k:
                    w = type(self)(self.args, self.env)
                    w.setup()
                for _ in range(6):   # the lane's block cache has to have seen every size the step asks for: a hipMalloc inside the timed region stalls EVERY lane
                    w.step()
                e.lib.synchronize()
                go.wait()
                for i in range(steps):
                    if i % nl == k:
                        w.step()
                e.lib.synchronize()
                done.wait()
            except Exception as ex:  # noqa: BLE001
                errs.append(ex)
                go.abort(); done.abort()
        threads = [threading.Thread(target=lane_main, args=(k,)) for k in range(nl)]
        for t in threads:
            t.start()
        try:
            go.wait()
            t0 = time.perf_counter()
            done.wait()
            dt = time.perf_counter() - t0
        except threading.BrokenBarrierError:
            dt = None
        for t in threads:
            t.join()
        e.lib.lane_use(0)
        if errs or dt is None:
            return {"error": str(errs[0])[:200] if errs else "a lane stopped"}
        return {"lanes": nl, "steps": steps, "rows_per_s": round(self.rows() * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3),
                "note": "the same steps dealt to several device lanes, each with its own host thread, staged input and parser / plan objects (the parsequeue's shape): "
                        "one lane's read-backs and host work run beside another lane's kernels; never `value`"}

