"""bench/wl_configs2.py — BASELINE.json configs[2]: Confluent-SR JSON -> replace_primary_key + sql -> ClickHouse JSONEachRow; its pull || push side measurement lives in bench/pipeline.py."""
from .common import *  # noqa: F401,F403
from .common import _cells_same, _rows_diff, _batch_diff, _parity, _test_helpers, _guard_parity, _columns_diff  # noqa: F401
from .wl_messages import _Prepared, sr_inputs


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
        from .pipeline import pull_push
        return pull_push(self, only_pipeline)

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
        out["parity"] = _guard_parity(lambda: self.parity(min(self.n, self.args.parity_rows)))
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
        why = _columns_diff(abi, host, ref.batch)
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


