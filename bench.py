#!/usr/bin/env python3
"""bench.py — ClickBench-hits CSV parse → mask(ClientIP) + filter(EventDate) → devnull.

The workload is BASELINE.json configs[1] ("ClickBench hits-1M CSV parse →
mask(ip)+filter(EventDate) → devnull, 1×MI355X").  One *step* is one pass of
that hot path over one HBM-resident batch of 2^20 synthetic hits rows per GPU.
Rows shard by range across ranks with no data-path collective (weak scaling:
every rank parses its own 2^20-row batch); torch.distributed is used only for
the barrier and the max-over-ranks timing.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

CHAIN = [
    ("mask_field", {"maskFunctionHash": {"userDefinedSalt": "clickbench-salt"}, "columns": ["clientip"]}),
    ("filter_rows", {"filter": "eventdate >= 2013-07-15"}),
]


def stage_shard(lib, workload, row0, nrows, chunk_rows=1 << 15):
    """Generate rows [row0,row0+nrows) chunk by chunk into one reused host buffer and
    assemble the CSV (header + rows) in HBM.  Returns (DeviceBuffer, nbytes)."""
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


JSON_CHAIN = [("mask_field", {"maskFunctionHash": {"userDefinedSalt": "clickbench-salt"}, "columns": ["clientip"]}),
         ("filter_rows", {"filter": "regionid >= 40"})]


def json_fields(workload, abi):
    """The hits schema as a generic-parser field list: date → utf8 text, timestamps → datetime (epoch seconds)."""
    out = []
    for name, typ, key in workload.hits_columns():
        t = {"date": "utf8", "timestamp": "datetime"}.get(typ, typ)
        out.append([name, t, bool(key)])
    return abi.Schema.of(out)


def make_messages(workload, abi, nrows):
    import csv
    import io
    cols = workload.hits_columns()
    data = workload.hits_csv(nrows)
    vals = []
    rd = csv.reader(io.StringIO(data.decode("utf-8")))
    next(rd)  # header
    for row in rd:
        doc = {}
        for (name, typ, _), cell in zip(cols, row):
            if typ in ("int16", "int32", "int64"):
                doc[name] = int(cell)
            elif typ == "timestamp":  # epoch seconds, the form extractTimeValue takes without dateparse
                if cell.lstrip("-").isdigit():
                    doc[name] = int(cell)
                else:
                    import calendar
                    import datetime
                    doc[name] = calendar.timegm(datetime.datetime.strptime(cell[:19].replace("T", " "), "%Y-%m-%d %H:%M:%S").timetuple())
            else:
                doc[name] = cell
        vals.append(json.dumps(doc, ensure_ascii=False, separators=(",", ":")).encode("utf-8"))
    return vals


def json_main(args):
    """--workload json: BASELINE.json configs[2] shape as a SIDE benchmark (the driver's line is the CSV workload):
    Kafka-style JSON messages (one flat hits object per message) → generic JSON parser on device → mask + filter →
    ClickHouse JSONEachRow text, all HBM-resident; the oracle's generic parser on a bounded sample as cpu_baseline."""
    import csv
    import io
    args.rows = min(args.rows, 1 << 18)
    args.cpu_rows = min(args.cpu_rows, 1 << 13)
    from transferia_amd import abi, lib, workload
    lib.init(0)
    fields = json_fields(workload, abi)
    vals = make_messages(workload, abi, args.rows)
    rows = len(vals)
    data, msgs = abi.messages(vals, list(range(rows)), [1_700_000_000_000_000_000 + i for i in range(rows)])
    opts = abi.json_options(topic="hits", add_dedupe_keys=True, add_rest=True, partition='{"partition":0,"topic":"hits"}')
    dbuf = lib.DeviceBuffer.upload(data)
    plans = [lib.Transformer(t, c) for t, c in JSON_CHAIN]
    state = {}

    def step(keep=False):
        db, errs = lib.json_parse(opts, fields, dbuf, msgs)
        res = lib.apply_chain(plans, db)
        out = lib.serialize(abi.FMT_CH_JSON_EACH_ROW, res.transformed)
        if keep:
            state.update(parsed=db.nrows, parsed_bytes=db.payload_bytes(), out_rows=res.transformed.nrows, out_bytes=out.size, errors=len(errs) + len(res.errors))
        out.free(); res.transformed.free(); db.free()

    for _ in range(args.warmup):
        step()
    step(keep=True)
    lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    lib.synchronize()
    dt = time.perf_counter() - t0
    lib.prof_reset(); lib.prof_enable(True)
    for _ in range(3):
        step()
    lib.prof_enable(False)
    prof = lib.prof_get()
    kernels = {n: {"launches_per_step": l / 3, "avg_ms": round(ms / l, 4), "ms_per_step": round(ms / 3, 4)} for n, l, ms in prof if l}
    alg = {"json_parse_lines": len(data) + state["parsed_bytes"], "csv_count_newlines": len(data)}
    dom = max(kernels.items(), key=lambda kv: kv[1]["ms_per_step"])[0]
    roofline = None
    if dom in alg:
        ach = alg[dom] / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                    "traffic": None, "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": kernels[dom]["avg_ms"]}
    cpu = None
    if args.cpu_rows > 0:
        from oracle import oracle as ora
        k = min(args.cpu_rows, rows)
        d2, m2 = abi.messages(vals[:k], list(range(k)), [0] * k)
        r1 = ora.json_parse(opts, fields, d2, m2, want_rows=False)
        cpu = {"value": round(k / r1.seconds, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} of the same messages through the oracle's generic JSON parser only ({r1.seconds:.2f}s), single thread",
               "note": "C restatement of the Go reference (fastjson value tree, boxed values per row), not the Go binary"}
    print(json.dumps({
        "metric": "ChangeItems/sec through Kafka JSON parse -> mask(ClientIP)+filter -> ClickHouse JSONEachRow, ClickBench hits", "value": round(rows * args.steps / dt, 1),
        "unit": "rows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "Kafka JSON (one flat hits object per message) -> generic parser -> mask+filter -> JSONEachRow (BASELINE.json configs[2] shape)",
                   "rows_per_step": rows, "json_bytes_per_step": len(data), "columns": len(fields.cols)},
        "gib_per_s_json_in": round(len(data) * args.steps / dt / 2**30, 3), "rows_out_per_step": state["out_rows"], "text_out_bytes_per_step": state["out_bytes"],
        "row_errors": state["errors"], "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels}))


def sr_main(args):
    """--workload sr: the Confluent-SR half of BASELINE.json configs[2] / [4] as a SIDE benchmark: Kafka messages in the
    Schema Registry wire format (0x00 | schema id | one flat hits object, JSON schema of the hits table) → tfgpu_sr_json_parse
    → queue JSON serializer with batching (the "→ Kafka sink" half), all HBM-resident; the oracle on a bounded sample."""
    args.rows = min(args.rows, 1 << 18)
    args.cpu_rows = min(args.cpu_rows, 1 << 13)
    from transferia_amd import abi, confluent_sr, lib, workload
    lib.init(0)
    vals = make_messages(workload, abi, args.rows)
    rows = len(vals)
    jt = {"int16": "integer", "int32": "integer", "int64": "integer", "timestamp": "integer"}
    props = {name: {"type": jt.get(typ, "string")} for name, typ, _ in workload.hits_columns()}
    schema_text = json.dumps({"type": "object", "title": "default.hits", "properties": props, "required": ["watchid"]})
    sid = 42
    frames = [b"\0" + sid.to_bytes(4, "big") + v for v in vals]
    data, msgs = abi.messages(frames, list(range(rows)), [1_700_000_000_000_000_000 + i for i in range(rows)])
    opts = confluent_sr.sr_json_options(sid, schema_text)
    qopts = abi.queue_options(abi.QFMT_JSON, enabled=True, max_message_size=1 << 20)
    dbuf = lib.DeviceBuffer.upload(data)
    state = {}

    def step(keep=False):
        res = lib.sr_json_parse(opts, dbuf, msgs)
        out = lib.queue_serialize(qopts, res.device_batch)
        if keep:
            state.update(parsed=res.device_batch.nrows, parsed_bytes=res.device_batch.payload_bytes(), out_bytes=out.values.size, messages=len(out), errors=len(res.errors))
        out.values.free(); res.device_batch.free()

    for _ in range(args.warmup):
        step()
    step(keep=True)
    lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    lib.synchronize()
    dt = time.perf_counter() - t0
    lib.prof_reset(); lib.prof_enable(True)
    for _ in range(3):
        step()
    lib.prof_enable(False)
    prof = lib.prof_get()
    kernels = {n: {"launches_per_step": l / 3, "avg_ms": round(ms / l, 4), "ms_per_step": round(ms / 3, 4)} for n, l, ms in prof if l}
    alg = {"sr_parse_frames": len(data), "sr_cell_values": len(data) + state["parsed_bytes"], "sr_cell_text": 2 * state["parsed_bytes"],
           "ser_cell_write": state["parsed_bytes"] + state["out_bytes"], "ser_cell_len": state["parsed_bytes"]}
    dom = max(kernels.items(), key=lambda kv: kv[1]["ms_per_step"])[0]
    roofline = None
    if dom in alg:
        ach = alg[dom] / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                    "traffic": None, "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": kernels[dom]["avg_ms"]}
    cpu = None
    if args.cpu_rows > 0:
        from oracle import oracle as ora
        k = min(args.cpu_rows, rows)
        d2, m2 = abi.messages(frames[:k], list(range(k)), [0] * k)
        r1 = ora.sr_json_parse(opts, d2, m2)
        ora.queue_serialize(qopts, r1.batch, r1.schema)
        sec = r1.seconds + ora.queue_serialize.seconds
        cpu = {"value": round(k / sec, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{k} of the same messages through the oracle's SR JSON parser ({r1.seconds:.2f}s) + queue JSON serializer ({ora.queue_serialize.seconds:.2f}s), single thread",
               "note": "C restatement of the Go reference (decoded value tree, boxed values per row), not the Go binary"}
    print(json.dumps({
        "metric": "ChangeItems/sec through Confluent-SR JSON parse -> queue JSON serializer (batched), ClickBench hits", "value": round(rows * args.steps / dt, 1),
        "unit": "rows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "Kafka messages in the Schema Registry wire format (one flat hits object each) -> SR JSON parser -> queue JSON serializer, 1 MiB batches "
                               "(the ingest of BASELINE.json configs[2] and the sink half of configs[4])",
                   "rows_per_step": rows, "wire_bytes_per_step": len(data), "columns": len(props)},
        "gib_per_s_wire_in": round(len(data) * args.steps / dt / 2**30, 3), "rows_out_per_step": state["parsed"], "messages_out_per_step": state["messages"],
        "text_out_bytes_per_step": state["out_bytes"], "row_errors": state["errors"], "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels}))


def collapse_main(args):
    """--workload collapse: abstract.Collapse (the PK-keyed dedup of BASELINE.json configs[4]) as a SIDE benchmark: a CDC
    slice (workload.cdc_batch) resident in HBM → tfgpu_collapse; the oracle's restatement of the Go maps as cpu_baseline."""
    from transferia_amd import lib, workload
    lib.init(0)
    b, schema = workload.cdc_batch(args.rows)
    db = lib.DeviceBatch.upload(b)
    state = {}

    def step(keep=False):
        out = lib.collapse(db)
        if keep:
            state.update(out_rows=out.nrows)
        out.free()

    for _ in range(args.warmup):
        step()
    step(keep=True)
    lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    lib.synchronize()
    dt = time.perf_counter() - t0
    lib.prof_reset(); lib.prof_enable(True)
    for _ in range(3):
        step()
    lib.prof_enable(False)
    kernels = {n: {"launches_per_step": l / 3, "avg_ms": round(ms / l, 4), "ms_per_step": round(ms / 3, 4)} for n, l, ms in lib.prof_get() if l}
    cpu = None
    if args.cpu_rows > 0:
        from oracle import oracle as ora
        k = min(max(args.cpu_rows, 1 << 18), args.rows)
        b2, s2 = workload.cdc_batch(k)
        r1 = ora.collapse(b2, s2)
        cpu = {"value": round(k / r1.seconds, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"a {k}-row slice of the same stream through the oracle's Collapse ({r1.seconds:.2f}s), single thread",
               "note": "C restatement of the Go reference (json.Marshal key strings, string-keyed maps, boxed values), not the Go binary"}
    print(json.dumps({
        "metric": "ChangeItems/sec through abstract.Collapse (PK-keyed dedup of a CDC batch)", "value": round(args.rows * args.steps / dt, 1),
        "unit": "rows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "CDC slice (35% insert / 45% update / 20% delete over rows/4 keys, 60% of U/D with OldKeys, 15% of those change the PK) -> Collapse",
                   "rows_per_step": args.rows, "rows_out_per_step": state["out_rows"]},
        "roofline": None, "cpu_baseline": cpu, "kernels": kernels}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["csv", "json", "collapse", "sr"], default="csv",
                    help="csv = BASELINE.json configs[1] (the bench line); json = configs[2] shape, collapse = configs[4]'s dedup: side benchmarks")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=1 << 20, help="rows per GPU per step")
    ap.add_argument("--cpu-rows", type=int, default=1 << 19, help="rows of the CPU-baseline sample (0 = skip); 2^19 rows = about 11 s of oracle time")
    ap.add_argument("--prof-steps", type=int, default=3)
    ap.add_argument("--lanes", type=int, default=1, help="device lanes (host threads) the steps are spread over: batch N+1 is parsed "
                    "beside the transform of batch N, the way the parsequeue's workers overlap (1 = strictly serial)")
    ap.add_argument("--overlap-lanes", type=int, default=3, help="side measurement (never `value`): the same steps spread over this many lanes (0 = skip)")
    ap.add_argument("--pcie-steps", type=int, default=2, help="extra untimed-for-`value` passes that start from PINNED HOST memory (0 = skip)")
    args = ap.parse_args()
    if args.workload == "json":
        return json_main(args)
    if args.workload == "collapse":
        return collapse_main(args)
    if args.workload == "sr":
        return sr_main(args)

    import torch
    from transferia_amd import dist as tdist
    rank, local_rank, world = tdist.env_rank()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    group = tdist.Group("nccl", torch.device("cuda", local_rank))  # RCCL: barrier + MAX of the wall time only

    from transferia_amd import lib, workload
    lib.init(local_rank)

    barrier = group.barrier

    schema = workload.hits_schema()
    opts = workload.hits_csv_options()
    plans = [lib.Transformer(t, c) for t, c in CHAIN]

    # ---- stage this rank's shard in HBM (untimed) ----
    row0, _ = tdist.weak_shard(args.rows, rank)
    dbuf, csv_bytes, hs = stage_shard(lib, workload, row0, args.rows)

    state = {}

    def step(keep=False):
        db, consumed, errs = lib.csv_parse(opts, schema, dbuf)
        res = lib.apply_chain(plans, db)
        if keep:
            state["parsed_rows"] = db.nrows
            state["parsed_bytes"] = db.payload_bytes()
            v = db.view()
            fixed = 0
            str_bytes = 0
            for i in range(v.ncols):
                c = v.cols[i]
                if c.repr in (12, 13, 14, 15):
                    str_bytes += int(c.data_len)
                else:
                    fixed += int(v.nrows) * (np.dtype(lib.abi.REPR_NP[c.repr]).itemsize + (4 if c.nanos else 0))
            state["fixed_bytes"], state["str_bytes"] = fixed, str_bytes
            state["out_rows"] = res.transformed.nrows
            state["out_bytes"] = res.transformed.payload_bytes()
            state["errors"] = len(errs) + len(res.errors)
            state["consumed"] = consumed
        res.transformed.free()
        db.free()  # devnull sink

    for _ in range(args.warmup):
        step()
    step(keep=True)  # one extra untimed pass to record sizes

    # ---- the timed region: exactly args.steps steps, spread round-robin over `lanes` host threads, each bound
    #      to its own device lane (stream + HBM cache) and reading its own HBM-resident copy of the shard ----
    import threading

    def timed_region(nlanes):
      nl = max(1, min(nlanes, args.steps, lib.lane_count()))
      lane_bufs = [dbuf]
      go, done = threading.Barrier(nl + 1), threading.Barrier(nl + 1)
      lane_err = []

      def lane_main(k):
          try:
              lib.lane_use(k)
              if k:  # stage this lane's copy of the input (untimed), then warm the lane's caches
                  lane_bufs.append(None)
              buf = dbuf if k == 0 else stage_shard(lib, workload, row0, args.rows)[0]
              def lstep():
                  db, consumed, errs = lib.csv_parse(opts, schema, buf)
                  res = lib.apply_chain(plans, db)
                  res.transformed.free()
                  db.free()
              if k:
                  for _ in range(max(args.warmup, 1)):
                      lstep()
              lib.synchronize()
              go.wait()
              for i in range(args.steps):
                  if i % nl == k:
                      lstep()
              lib.synchronize()
              done.wait()
          except Exception as e:  # noqa: BLE001
              lane_err.append(e)
              go.abort(); done.abort()

      threads = [threading.Thread(target=lane_main, args=(k,)) for k in range(nl)]
      for t in threads:
          t.start()
      barrier()
      torch.cuda.synchronize()
      go.wait()
      t0 = time.perf_counter()
      done.wait()
      torch.cuda.synchronize()
      barrier()
      dt = time.perf_counter() - t0
      for t in threads:
          t.join()
      if lane_err:
          raise lane_err[0]
      lib.lane_use(0)
      return dt, nl

    dt, nl = timed_region(args.lanes)
    overlapped = None
    if args.overlap_lanes > 1 and args.overlap_lanes != args.lanes:
        dt2, nl2 = timed_region(args.overlap_lanes)
        dt2 = group.max_seconds(dt2)
        overlapped = {"lanes": nl2, "rows_per_s": round(args.rows * world * args.steps / dt2, 1), "ms_per_step": round(dt2 / args.steps * 1e3, 3),
                      "note": "side measurement, never `value`: the same K steps spread over several device lanes (parse of batch N+1 beside "
                              "the transform of batch N); kernels of different lanes share the GPU, so per-kernel spans are not comparable"}

    # ---- PCIe-inclusive side measurement (never `value`): the same step starting from pinned host memory ----
    pcie = None
    if args.pcie_steps > 0 and rank == 0:
        host = lib.HostBuffer(dbuf.download())
        def hstep():
            db, consumed, errs = lib.csv_parse(opts, schema, host)
            res = lib.apply_chain(plans, db)
            res.transformed.free(); db.free()
        hstep()
        lib.synchronize()
        th = time.perf_counter()
        for _ in range(args.pcie_steps):
            hstep()
        lib.synchronize()
        dth = time.perf_counter() - th
        pcie = {"rows_per_s": round(args.rows * args.pcie_steps / dth, 1), "gb_per_s_h2d": round(csv_bytes * args.pcie_steps / dth / 1e9, 2),
                "ms_per_step": round(dth / args.pcie_steps * 1e3, 3), "note": "input in pinned host memory, one lane, H2D inside the step"}
        host.free()
    dt = group.max_seconds(dt)

    # ---- per-kernel device time, HIP events on the library stream (separate, untimed pass) ----
    lib.prof_reset()
    lib.prof_enable(True)
    for _ in range(max(args.prof_steps, 1)):
        step()
    lib.prof_enable(False)
    prof = lib.prof_get()
    lib.prof_reset()

    rows, fixed_b, str_b = state["parsed_rows"], state["fixed_bytes"], state["str_bytes"]
    sel = state["out_rows"] / max(rows, 1)
    # ALGORITHMIC bytes per launch (DESIGN.md "Kernels"; SURVEY.md §8d per-row figures × rows per launch)
    alg = {
        "csv_count_newlines": csv_bytes,
        "csv_line_index": csv_bytes + 4 * rows,
        "csv_parse_rows": csv_bytes + fixed_b,            # read every input byte once, write every fixed-width value once
        "csv_parse_tiles": csv_bytes + fixed_b,           # same work, tile-cooperative kernel
        "csv_copy_strings": 2 * str_b,                    # string payload: one read + one write
        "csv_copy_tiles": 2 * str_b,
        "csv_copy_words": 2 * str_b,
        "scan_u32_segments": 12 * rows * 28,
        "mask_hmac_sha256": 72 * rows,                    # 4 B in + 64 B hex + 4 B offset per value
        "filter_rows_eval": int((8 + 4 + 0.125) * rows),  # time value + keep flag
        "compact_gather": int((1 + sel) * state["parsed_bytes"]),
        "scan_u32": 12 * rows,
    }
    kernels = {}
    for name, launches, ms in prof:
        if launches:
            kernels[name] = {"launches_per_step": launches / max(args.prof_steps, 1), "avg_ms": ms / launches,
                             "ms_per_step": ms / max(args.prof_steps, 1)}
    dom = max(kernels.items(), key=lambda kv: kv[1]["ms_per_step"])[0] if kernels else None
    roofline = None
    if dom:
        per_launch = alg.get(dom, 0) / max(kernels[dom]["launches_per_step"], 1)
        achieved = per_launch / (kernels[dom]["avg_ms"] * 1e-3) / 1e9 if per_launch else 0.0
        # HBM traffic per launch cannot be read inside this process: it comes from the rocprofv3 --pmc passes whose summary
        # is committed under profiles/ (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md "HBM"); null when the
        # recorded launch shape is not this run's.
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f).get(dom)
            if t and t["rows_per_launch"] == rows:
                traffic = t["bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "algorithmic_bytes_per_launch": int(per_launch),
                    "avg_launch_ms": round(kernels[dom]["avg_ms"], 4)}
    for k, v in kernels.items():
        b = alg.get(k)
        if b:
            v["alg_gb_s"] = round(b / max(v["launches_per_step"], 1) / (v["avg_ms"] * 1e-3) / 1e9, 2)
        v["avg_ms"], v["ms_per_step"] = round(v["avg_ms"], 4), round(v["ms_per_step"], 4)

    # ---- CPU baseline: the oracle (C restatement of the Go reference), rank 0, N=1 only ----
    cpu = None
    if rank == 0 and world == 1 and args.cpu_rows > 0:
        from oracle import oracle as ora
        sample = workload.hits_csv(args.cpu_rows)
        ochain = [ora.Transformer(t, c) for t, c in CHAIN]
        r1 = ora.csv_parse(opts, schema, sample, "", "")
        r2 = ora.apply_chain(ochain, r1.batch, r1.schema)
        secs = r1.seconds + r2.seconds
        cpu = {"value": round(args.cpu_rows / secs, 1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{args.cpu_rows} rows of the same synthetic hits CSV: oracle csv parse+strictify ({r1.seconds:.2f}s) + mask+filter ({r2.seconds:.2f}s), "
                         "single thread = the reference's shape for one table (transformation.go:131-135)",
               "note": "C restatement of the Go reference (row-oriented boxed values), not the Go binary"}

    if rank == 0:
        total_rows = rows * world * args.steps
        value = total_rows / dt
        out = {
            "metric": "ChangeItems/sec through CSV parse -> mask(ClientIP)+filter(EventDate) -> devnull, ClickBench hits",
            "value": round(value, 1), "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "ClickBench hits-1M CSV parse -> mask(ip)+filter(EventDate) -> devnull (BASELINE.json configs[1])",
                       "rows_per_gpu_per_step": rows, "csv_bytes_per_gpu_per_step": csv_bytes, "columns": len(schema.cols),
                       "chain": [t for t, _ in CHAIN], "filter_selectivity": round(sel, 4), "parallelism": f"row-range shard x{world}, no collective"},
            "gib_per_s_csv_in": round(csv_bytes * world * args.steps / dt / 2**30, 3),
            "gib_per_s_deepsizeof": round((state["parsed_bytes"] + 16 * len(schema.cols) * rows) * world * args.steps / dt / 2**30, 3),
            "rows_out_per_step": state["out_rows"], "row_errors": state["errors"],
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels, "lanes": nl, "overlapped_lanes": overlapped, "pcie_inclusive": pcie,
        }
        if cpu:
            out["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
        print(json.dumps(out))
    group.close()


if __name__ == "__main__":
    main()
