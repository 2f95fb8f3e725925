"""bench/wl_debezium.py — the envelope parsers as component lines: Debezium with inline schemas, registry-framed Debezium, Confluent-SR protobuf."""
from .common import *  # noqa: F401,F403
from .common import _cells_same, _rows_diff, _batch_diff, _parity, _test_helpers, _guard_parity, _columns_diff  # noqa: F401


def _replicas_diff(abi, batch, k0):
    """The replicated-message lines: rows k0.. of a device batch whose INPUT messages are byte-identical to message 0 must equal row 0, column by
    column and bit for bit (values, validity, nanoseconds; text cells the same bytes at a constant stride) — None, or what differs first.  Rows
    0..k0 are compared item by item against the oracle's items by the caller; together that is every row against the oracle's output for its input."""
    n = batch.nrows
    for c in list(batch.cols) + list(getattr(batch, "old_keys", None) or []):
        v = c.validity if c.validity is not None else np.ones(n, bool)
        if not (v == v[0]).all():
            return "column %s: validity of row %d" % (c.name, int(np.flatnonzero(v != v[0])[0]))
        for name in ("absent",):
            ab = getattr(c, name, None)
            if ab is not None and not (ab == ab[0]).all():
                return "column %s: %s" % (c.name, name)
        if c.repr in abi.VAR_REPRS:
            off = np.asarray(c.offsets, dtype=np.int64)
            ln = int(off[1] - off[0])
            if not np.array_equal(off, off[0] + ln * np.arange(n + 1, dtype=np.int64)):
                return "column %s: cell lengths" % c.name
            if ln:
                d = np.asarray(c.data[int(off[0]):int(off[0]) + ln * n]).reshape(n, ln)
                bad = np.flatnonzero((d != d[0]).any(axis=1))
                if len(bad):
                    return "column %s: text of row %d" % (c.name, int(bad[0]))
        elif v[0]:
            x = np.ascontiguousarray(c.values)
            x = x.view("u%d" % x.dtype.itemsize) if x.dtype.kind == "f" else x
            if not (x == x[0]).all():
                return "column %s: value of row %d" % (c.name, int(np.flatnonzero(x != x[0])[0]))
            if c.nanos is not None and not (np.asarray(c.nanos) == c.nanos[0]).all():
                return "column %s: nanoseconds" % c.name
    for name in ("kind",):   # (src_row / part_id follow the message's index)
        arr = getattr(batch, name, None)
        if arr is not None and not (np.asarray(arr) == arr[0]).all():
            return name
    return None


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
            kk, kall = min(k, 256), min(self.n, max(self.args.parity_rows, 256))
            d2, m2 = abi.messages([self.msg] * kk)
            ora.debezium_parse.want_items = True
            exp_items, codes = ora.debezium_parse(d2, m2)
            ora.debezium_parse.want_items = False
            d3, m3 = abi.messages([self.msg] * kall)
            parsed, errors = self.parser.parse(d3, m3)
            if errors or len(parsed) != 1 or parsed[0].batch.nrows != kall:
                return _parity(kall, "parse", "%d device errors, %d batches, %d rows of %d" % (len(errors), len(parsed), sum(p.batch.nrows for p in parsed), kall))
            got = td.device_items(self.env.lib, parsed[0], rows=kk)
            for it in exp_items:
                td.assert_same_items(got[it["src"]], it, it["src"])
            why = _replicas_diff(abi, parsed[0].batch.download(), kk)
            if why:
                return _parity(kall, "replicated rows", why)
            return _parity(kall, "Debezium receiver: the items of the first %d messages (kind, table, schema, 61 column values, OldKeys, row meta) against the oracle's, item by item; "
                                 "the other %d messages are byte-identical inputs and their rows equal row 0 column by column, bit for bit" % (kk, kall - kk), compared_output_rows=kall)
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
            kall = min(self.n, max(self.args.parity_rows, kk))
            d3, m3 = abi.messages([self.msg] * kall)
            parsed, errors, events = self.parser.parse_registry(d3, m3)
            if errors or len(parsed) != 1 or parsed[0].batch.nrows != kall:
                return _parity(kall, "parse", "%d device errors, %d batches, %d rows of %d" % (len(errors), len(parsed), sum(p.batch.nrows for p in parsed), kall))
            why = _replicas_diff(abi, parsed[0].batch.download(), kk)
            if why:
                return _parity(kall, "replicated rows", why)
            return _parity(kall, "registry-framed Debezium receiver: framing + the items of the first %d events against the oracle's, item by item; the other %d events are "
                                 "byte-identical inputs and their rows equal row 0 column by column, bit for bit" % (kk, kall - kk), compared_output_rows=kall)
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
            kall = min(self.n, max(self.args.parity_rows, kk))
            d3, m3 = abi.messages([self.msg] * kall)
            batch, errors = self.schema.parse(self.sid, d3, m3)
            b3 = batch.download()
            if errors or b3.nrows != kall:
                return _parity(kall, "parse", "%d device errors, %d rows of %d" % (len(errors), b3.nrows, kall))
            why = _replicas_diff(abi, b3, kk)
            if why:
                return _parity(kall, "replicated rows", why)
            return _parity(kall, "Confluent-SR protobuf parser: the 60 column values of the first %d messages against the oracle's, cell by cell; the other %d messages are "
                                 "byte-identical inputs and their rows equal row 0 column by column, bit for bit" % (kk, kall - kk), compared_output_rows=kall)
        out["parity"] = _guard_parity(check)
        return out


