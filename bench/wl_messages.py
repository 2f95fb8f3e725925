"""bench/wl_messages.py — the message-shaped component lines: generic JSON parser and Confluent-SR JSON, each into a serializer."""
from .common import *  # noqa: F401,F403
from .common import _cells_same, _rows_diff, _batch_diff, _parity, _test_helpers, _guard_parity, _columns_diff  # noqa: F401


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
        out["parity"] = _guard_parity(lambda: self.parity(min(self.n, self.args.parity_rows)))
        return out

    def parity(self, k):
        """parse of the first k messages against the oracle's rows; then mask + filter + JSONEachRow of that batch, device against oracle"""
        from oracle import oracle as ora
        lib, abi = self.env.lib, self.env.abi
        d2, m2 = abi.messages(self.vals[:k], list(range(k)), [1_700_000_000_000_000_000 + i for i in range(k)])
        ref = ora.json_parse(self.opts, self.fields, d2, m2, want_rows=False, want_batch=True)
        db, errs = lib.json_parse(self.opts, self.fields, d2, m2)
        host = db.download()
        if errs or host.nrows != ref.nrows:
            return _parity(k, "parse", "rows %d vs %d, %d device errors" % (host.nrows, ref.nrows, len(errs)))
        why = _columns_diff(abi, host, ref.batch)
        if why:
            return _parity(k, "parse", why)
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
        out["parity"] = _guard_parity(lambda: self.parity(min(self.n, self.args.parity_rows)))
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
        why = _columns_diff(abi, host, ref.batch)
        if why:
            return _parity(k, "parse", why)
        got = lib.queue_serialize(self.qopts, res.device_batch)
        want = ora.queue_serialize(self.qopts, ref.batch, ref.schema)
        text = bytes(got.values.download())
        if want is None or text != b"".join(want) or len(got) != len(want):
            return _parity(k, "queue JSON serializer", "messages differ")
        return _parity(k, "SR JSON parse (cell for cell) + queue JSON serializer (byte for byte)", compared_messages=len(want), compared_output_bytes=len(text))


