"""bench/wl_configs4.py — BASELINE.json configs[4]: CDC rows -> sharder -> partition -> exchange (RCCL) -> Collapse -> queue sink; from Debezium envelope bytes (configs4d); Collapse alone."""
from .common import *  # noqa: F401,F403
from .common import _cells_same, _rows_diff, _batch_diff, _parity, _test_helpers, _guard_parity, _columns_diff  # noqa: F401


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
            m = min(a.nrows, max(self.args.parity_rows, 1 << 13))   # the emitter's restatement is Python: a bounded slice of the collapsed rows (17 k rows/s)
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
            why = _columns_diff(abi, host, a)
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
            kk = min(self.n, max(self.args.parity_rows, 2048))
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
            why = _columns_diff(e.abi, got, r1.batch)
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


