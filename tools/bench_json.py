#!/usr/bin/env python3
"""tools/bench_json.py — BASELINE.json configs[2] shape as a side benchmark (NOT the driver's bench line):
Kafka-style JSON messages (one flat `hits` object per message) → generic JSON parser on device →
mask(clientip) + filter_rows(regionid) → ClickHouse JSONEachRow text, all HBM-resident.

Prints one JSON line: rows/s, GiB/s of JSON in, per-kernel device time (HIP events), roofline of the dominant
kernel against its algorithmic bytes (B_json + B_bin per row), and the oracle (C restatement of the
Go reference) timed on a bounded sample of the same messages on one host core.

    python tools/bench_json.py --rows 131072 --steps 5 --warmup 2
"""
import argparse
import csv
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0
CHAIN = [("mask_field", {"maskFunctionHash": {"userDefinedSalt": "clickbench-salt"}, "columns": ["clientip"]}),
         ("filter_rows", {"filter": "regionid >= 40"})]


def json_fields(workload, abi):
    """The hits schema as a generic-parser field list: date → utf8 text, timestamps → datetime (epoch seconds)."""
    out = []
    for name, typ, key in workload.hits_columns():
        t = {"date": "utf8", "timestamp": "datetime"}.get(typ, typ)
        out.append([name, t, bool(key)])
    return abi.Schema.of(out)


def make_messages(workload, abi, nrows):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 17)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-rows", type=int, default=1 << 13)
    args = ap.parse_args()
    from transferia_amd import abi, lib, workload
    lib.init(0)
    fields = json_fields(workload, abi)
    vals = make_messages(workload, abi, args.rows)
    rows = len(vals)
    data, msgs = abi.messages(vals, list(range(rows)), [1_700_000_000_000_000_000 + i for i in range(rows)])
    opts = abi.json_options(topic="hits", add_dedupe_keys=True, add_rest=True, partition='{"partition":0,"topic":"hits"}')
    dbuf = lib.DeviceBuffer.upload(data)
    plans = [lib.Transformer(t, c) for t, c in CHAIN]
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


if __name__ == "__main__":
    main()
