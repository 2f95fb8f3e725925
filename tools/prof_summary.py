#!/usr/bin/env python3
"""rocprofv3 rocpd database (or *_kernel_stats.csv) -> a small per-kernel summary
that is committed under profiles/ (gpurun_out/ is scratch)."""
import csv
import sqlite3
import sys


def from_db(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    return [(n, int(c), float(t) * 1e3, float(a) * 1e3, float(p)) for n, c, t, a, p in rows]  # µs -> ns


def from_csv(path):
    out = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"])))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    rows = from_db(src) if src.endswith(".db") else from_csv(src)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), round(r[4], 3)])
    print("wrote", dst, len(rows), "kernels")


if __name__ == "__main__":
    main()
