#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats sqlite database (rocpd) as CSV:
kernel name, calls, total / average duration.  Usage:
    python tools/rocprof_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_x.csv
"""
import csv
import sqlite3
import sys


def main(path, limit=40):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, total, avg, pct in rows[:limit]:
        w.writerow([name[:160], calls, "%.2f" % total, "%.3f" % avg, "%.2f" % pct])


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
