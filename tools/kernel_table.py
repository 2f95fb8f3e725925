#!/usr/bin/env python3
"""Rewrites DESIGN.md's kernel table (between the KERNEL-TABLE markers) from ONE visit's bench lines under profiles/.
usage: python tools/kernel_table.py TAG        (reads profiles/TAG_bench_*.json: the last line of each is the bench's JSON)"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
ORDER = ["csv", "configs0", "configs2", "configs3", "configs4", "configs4_debezium", "configs4d", "json", "sr", "sr_proto", "collapse", "debezium", "debezium_sr"]
lines = {}
for f in glob.glob(os.path.join(ROOT, "profiles", tag + "_bench_*.json")):
    name = os.path.basename(f)[len(tag) + 7:-5]
    try:
        lines[name] = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        continue
out = ["Visit `%s` (`profiles/%s_bench_*.json`, one MI355X, `tools/gpu_visit.sh evidence %s`).  ms = median of ten HIP-event passes of that kernel, summed over its launches in a pass; "
       "GB/s = its algorithmic bytes ÷ that time; %% = of the 8 TB/s HBM peak (a dash: launched on a fraction of the batch, or not priced).\n" % (tag, tag, tag),
       "| workload (`bench.py --workload`) | rows/s, ms per pass | kernel | launches | ms | alg GB/s | % HBM |", "|---|---|---|---|---|---|---|"]
for w in [x for x in ORDER if x in lines] + sorted(x for x in lines if x not in ORDER):
    d = lines[w]
    passes = d.get("passes_per_step") or 1
    head = "**%s** — %.3g %s, %.3f ms" % (w, d["value"], d["unit"], d["ms_per_step"] / passes)
    par = d.get("parity") or {}
    if par.get("identical"):
        head += ", parity ✓ (%s rows)" % par.get("checked_input_rows")
    ks = sorted(d.get("kernels", {}).items(), key=lambda kv: -kv[1]["ms_per_step"])
    first = True
    for k, v in ks:
        if v["ms_per_step"] < 0.02 and not first:
            continue
        g = v.get("alg_gb_s")
        out.append("| %s | %s | `%s` | %g | %.3f | %s | %s |" % (head if first else "", "" if not first else "", k, round(v["launches_per_step"], 1), v["ms_per_step"],
                                                              ("%.0f" % g) if g else "—", ("%.1f" % (g / 80.0)) if g else "—"))
        first = False
    r = d.get("roofline") or {}
    if r:
        out.append("| | | *roofline*: `%s` %.1f %% of HBM peak, traffic %s | | | | |" % (r.get("kernel"), 100 * (r.get("frac") or 0), ("%.2f × algorithmic" % (r["traffic"] / r["algorithmic_bytes_per_launch"])) if r.get("traffic") else "n/a"))
    ir = d.get("int_roofline") or {}
    if ir:
        out.append("| | | *int roofline*: `%s` %.0f %% of the VALU issue peak (minimum instruction count)%s | | | | |" % (ir.get("kernel"), 100 * ir.get("frac", 0), (", %.0f %% by measured instructions" % (100 * ir["issue_frac"])) if ir.get("issue_frac") else ""))
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a, b = s.index("<!-- KERNEL-TABLE-BEGIN"), s.index("<!-- KERNEL-TABLE-END -->")
a = s.index("\n", a) + 1
open(p, "w").write(s[:a] + "\n".join(out) + "\n" + s[b:])
print("wrote %d table lines from %d bench lines" % (len(out), len(lines)))
