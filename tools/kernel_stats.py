#!/usr/bin/env python3
"""Per-kernel launch statistics from a rocprofv3 --kernel-trace [--stats] output directory (kernel-trace CSV, or the rocpd
sqlite database rocprofv3 writes by default -- its per-dispatch table, so min / max are available too; this replaces the
former tools/rocprof_summary.py, which read the database's top_kernels view):
    python tools/kernel_stats.py gpurun_out/prof_r02 > profiles/r02_kernel_stats.csv"""
import csv
import glob
import os
import re
import sqlite3
import sys
from collections import defaultdict


def main(d):
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    if not dur:
        for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            c = sqlite3.connect(f)
            tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
            t = next((x for x in tabs if x.startswith("kernels")), None) or next((x for x in tabs if "kernel_dispatch" in x), None)
            cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
            name = "name" if "name" in cols else next(x for x in cols if "name" in x)
            for n, s, e in c.execute("select %s, start, end from %s" % (name, t)):
                dur[n].append((e - s) / 1e3)
    total = sum(sum(v) for v in dur.values())
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:40]:
        m = re.search(r"(k_[a-z0-9_]+)", n)
        w.writerow([(m.group(1) if m else n)[:120], len(v), "%.2f" % sum(v), "%.3f" % (sum(v) / len(v)), "%.3f" % min(v),
                    "%.3f" % max(v), "%.2f" % (100 * sum(v) / total)])


if __name__ == "__main__":
    main(sys.argv[1])
