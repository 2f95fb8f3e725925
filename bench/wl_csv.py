"""bench/wl_csv.py — BASELINE.json configs[1], the bench line: hits CSV -> mask(ClientIP) + filter(EventDate) -> devnull."""
from .common import *  # noqa: F401,F403
from .common import _cells_same, _rows_diff, _batch_diff, _parity, _test_helpers, _guard_parity, _columns_diff  # noqa: F401


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
                from .wl_configs2 import Configs2Workload
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
                                             "pipeline_3_stage": pp.get("pipeline_3_stage"), "pipeline_shapes": pp.get("pipeline_shapes")}
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
                out["all_cores"] = all_cores_csv(nc, per, reps, out["value"], CHAIN)
            except Exception as ex:  # noqa: BLE001
                out["all_cores"] = {"error": str(ex)[:200]}
        return out


