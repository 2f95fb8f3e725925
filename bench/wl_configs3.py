"""bench/wl_configs3.py — BASELINE.json configs[3]: resident hits columns (or a Parquet object) -> mask + sharder + casts -> ClickHouse JSONEachRow."""
from .common import *  # noqa: F401,F403
from .common import _cells_same, _rows_diff, _batch_diff, _parity, _test_helpers, _guard_parity, _columns_diff  # noqa: F401


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
        # ---- the same object as real writers leave it: SNAPPY pages (pyarrow's default codec; parquet-go inflates per page in its reader,
        #      reader_parquet.go:137-283).  The compressed object crosses PCIe, pq_inflate (one wave a page) inflates the data pages on the device;
        #      beside it the same read with every page inflated on the host (TFGPU_PQ_DEVICE_INFLATE=0: what rounds 4-5 did).  Rates are quoted in
        #      UNCOMPRESSED bytes (the NONE object's size): what the decoder behind the codec sees. ----
        compressed = {}
        for codec in ("snappy", "lz4_raw", "zstd"):
            try:
                bufc = io.BytesIO()
                pq.write_table(pa.table(arrays, names=names), bufc, compression=codec if codec != "lz4_raw" else "LZ4_RAW", row_group_size=h.nrows)
                datac = bufc.getvalue()
                pinc = lib.HostBuffer(datac)

                def onec(read_only=True, plans=None):
                    out = C.c_void_p()
                    lib._check(lib.load().tfgpu_parquet_read(C.c_void_p(pinc.ptr), C.c_uint64(len(datac)), abi.MEM_HOST, C.byref(cs), b"", b"hits", C.byref(out)))
                    db = lib.DeviceBatch(out)
                    if not read_only:
                        tr = lib.apply_chain(plans if plans is not None else self.plans, db)
                        o = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, tr.transformed)
                        o.free(); tr.transformed.free()
                    db.free()
                onec(); lib.synchronize()
                t0 = time.perf_counter()
                for _ in range(k):
                    onec()
                lib.synchronize()
                dtc = (time.perf_counter() - t0) / k
                lib.prof_reset(); lib.prof_enable(True)
                onec()
                lib.prof_enable(False)
                profc = {n: round(ms / max(l, 1), 4) for n, l, ms in lib.prof_get() if n.startswith("pq_")}
                lib.prof_reset()
                go3, done3, errs3 = threading.Barrier(nl + 1), threading.Barrier(nl + 1), []

                def lane_c(j):
                    try:
                        lib.lane_use(j)
                        onec(); lib.synchronize()
                        go3.wait()
                        for i in range(kk):
                            if i % nl == j:
                                onec()
                        lib.synchronize()
                        done3.wait()
                    except Exception as ex:  # noqa: BLE001
                        errs3.append(ex); go3.abort(); done3.abort()
                thc = [threading.Thread(target=lane_c, args=(j,)) for j in range(nl)]
                for t in thc:
                    t.start()
                try:
                    go3.wait()
                    t0 = time.perf_counter()
                    done3.wait()
                    dtc3 = (time.perf_counter() - t0) / kk
                except threading.BrokenBarrierError:
                    dtc3 = float("nan")
                for t in thc:
                    t.join()
                lib.lane_use(0)
                # every page on the HOST: on all the cores this process may use (pages inflated side by side ahead of the walk), then on one thread
                # (what rounds 4-5 did)
                dth = dth1 = dtd = float("nan")
                os.environ["TFGPU_PQ_DEVICE_INFLATE"] = "1"   # every eligible page on the device, whatever its shape
                try:
                    onec(); lib.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(2):
                        onec()
                    lib.synchronize()
                    dtd = (time.perf_counter() - t0) / 2
                finally:
                    os.environ.pop("TFGPU_PQ_DEVICE_INFLATE", None)
                os.environ["TFGPU_PQ_DEVICE_INFLATE"] = "0"
                try:
                    onec(); lib.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(2):
                        onec()
                    lib.synchronize()
                    dth = (time.perf_counter() - t0) / 2
                    os.environ["TFGPU_PQ_INFLATE_THREADS"] = "1"
                    t0 = time.perf_counter()
                    onec(); lib.synchronize()
                    dth1 = time.perf_counter() - t0
                finally:
                    os.environ.pop("TFGPU_PQ_DEVICE_INFLATE", None); os.environ.pop("TFGPU_PQ_INFLATE_THREADS", None)
                compressed[codec] = {"object_bytes": len(datac), "uncompressed_object_bytes": len(data), "read_only_ms": round(dtc * 1e3, 3),
                                     "read_only_gb_per_s_uncompressed": round(len(data) / dtc / 1e9, 2), "read_only_gb_per_s_pcie": round(len(datac) / dtc / 1e9, 2),
                                     "lanes_3": {"ms_per_object": round(dtc3 * 1e3, 3), "gb_per_s_uncompressed": round(len(data) / dtc3 / 1e9, 2)} if not errs3 else {"error": str(errs3[0])[:200]},
                                     "policy": "auto: small pages (<= 128 KiB inflated) and stored ones on the device, the others on the host's cores ahead of the walk",
                                     "device_all_pages_read_only_ms": round(dtd * 1e3, 3), "host_inflate_all_cores_read_only_ms": round(dth * 1e3, 3), "host_inflate_all_cores_gb_per_s_uncompressed": round(len(data) / dth / 1e9, 2),
                                     "host_inflate_one_thread_read_only_ms": round(dth1 * 1e3, 3), "usable_cores": usable_cores(),
                                     "data_pages_inflated_on": "device (pq_inflate)" if "pq_inflate" in profc else "host", "decode_kernels_avg_ms": profc}
                pinc.free()
            except Exception as ex:  # noqa: BLE001
                compressed[codec] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
        pinned.free()
        return {"parquet_source": {"object_bytes": len(data), "compressed": compressed, "pipeline_pull_decode": pipe, "rows_per_s": round(h.nrows / dt, 1), "ms_per_step": round(dt * 1e3, 3), "gb_per_s_parquet_in": round(len(data) / dt / 1e9, 2),
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
        k = max(min(self.args.cpu_rows, self.n, 1 << 16), min(self.n, self.args.parity_rows))
        r1 = ora.csv_parse(e.workload.hits_csv_options(), e.workload.hits_schema(), e.workload.hits_csv(k), "", "")
        r2 = ora.apply_chain([ora.Transformer(t, c) for t, c in self.CH], r1.batch, r1.schema)
        t0 = time.perf_counter()
        ora.serialize(abi.FMT_CH_JSON_EACH_ROW, r2.batch, r2.schema)
        ts = time.perf_counter() - t0
        out = {"value": round(k / (r2.seconds + ts), 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} rows: oracle chain ({r2.seconds:.2f}s) + JSONEachRow ({ts:.2f}s) on already-typed rows, single thread (one table = one goroutine, transformation.go:131-135)",
               "note": "C restatement of the Go reference, not the Go binary"}
        out.update(host_info())
        out["parity"] = _guard_parity(lambda: self.parity(min(k, self.args.parity_rows), r1))
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
        why = _columns_diff(abi, host, ref_head)
        if why:
            return _parity(k, "resident columns", why)
        tr = lib.apply_chain(self.plans, head)
        text = bytes(lib.serialize(abi.FMT_CH_JSON_EACH_ROW, tr.transformed).download())
        r2 = ora.apply_chain([ora.Transformer(t, c) for t, c in self.CH], ref_head, r1.schema)
        want = bytes(ora.serialize(abi.FMT_CH_JSON_EACH_ROW, r2.batch, r2.schema))
        if text != want:
            return _parity(k, "chain + JSONEachRow", "text differs (%d vs %d bytes)" % (len(text), len(want)))
        return _parity(k, "resident columns (cell for cell) + mask + sharder + casts + JSONEachRow (byte for byte)", compared_output_rows=r2.batch.nrows, compared_output_bytes=len(want))


