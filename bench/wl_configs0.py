"""bench/wl_configs0.py — BASELINE.json configs[0], the plumbing case: rename_tables + mask_field on a four-column table; --from-rows crosses the boundary with boxed rows."""
from .common import *  # noqa: F401,F403
from .common import _cells_same, _rows_diff, _batch_diff, _parity, _test_helpers, _guard_parity, _columns_diff  # noqa: F401


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
        from .wl_configs3 import Configs3Workload
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
            kk = (min(k, max(1 << 16, self.args.parity_rows)) // 8) * 8
            tr = self.env.lib.apply_chain(self.plans, self.env.lib.DeviceBatch.upload(b).slice(0, kk))   # the oracle's own sample (the resident table is another draw of the generator)
            got = tr.transformed.download()
            rb = r.batch
            want = self.env.abi.Batch([self.env.abi.Column(c.name, c.dtype, c.repr, values=None if c.values is None else c.values[:kk], nanos=None if c.nanos is None else c.nanos[:kk],
                                                           offsets=None if c.offsets is None else c.offsets[:kk + 1].copy(), data=None if c.data is None else c.data[:int(c.offsets[kk])],
                                                           validity=None if c.validity is None else c.validity[:kk]) for c in rb.cols], kk, rb.table_ns, rb.table_name)
            why = _columns_diff(self.env.abi, got, want)
            if why is None and (got.table_ns, got.table_name) != ("bench", "users_masked"):
                why = "table id"
            return _parity(kk, "rename_tables + mask_field, cell for cell", why, compared_output_rows=kk)
        out["parity"] = _guard_parity(check)
        return out


