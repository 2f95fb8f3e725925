#!/usr/bin/env python3
"""One pass of a workload as a timeline: every kernel dispatch of rocprofv3's kernel trace between two launches of an anchor kernel,
with its start offset, duration and the idle gap in front of it — where a step's wall time goes that the kernels' sum does not explain.

    python tools/timeline.py TRACE.csv ANCHOR_KERNEL [pass_index]
"""
import csv, sys

def main():
    path, anchor = sys.argv[1], sys.argv[2]
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("tf::", "").replace("(anonymous namespace)::", "")))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(starts) < 3:
        print("anchor", anchor, "seen", len(starts), "times"); return
    a, b = starts[which], starts[which + 1]
    t0 = rows[a][0]
    busy = 0; prev_end = t0; gaps = 0
    print("%8s %8s %8s  %s" % ("at us", "dur us", "gap us", "kernel"))
    for s, e, n in rows[a:b]:
        gap = s - prev_end
        if gap > 0: gaps += gap
        print("%8.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, n))
        busy += e - s; prev_end = max(prev_end, e)
    wall = rows[b][0] - t0
    print("pass: wall %.1f us (anchor to anchor), kernels %.1f us, gaps between dispatches %.1f us, tail to next anchor %.1f us" % (wall / 1e3, busy / 1e3, gaps / 1e3, (rows[b][0] - prev_end) / 1e3))

main()
