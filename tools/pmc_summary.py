#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output directories (one pass per directory) per kernel and counter:
    python tools/pmc_summary.py [--last=N] gpurun_out/pmc_r02_* > profiles/r02_pmc_summary.csv
Rows: counter, kernel (short name), dispatches, mean, min, max over the dispatches -- the first
dispatch of every kernel is dropped (cold caches / first-touch page faults)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else None


def main(dirs):
    last = 0
    if dirs and dirs[0].startswith("--last="):        # keep only the last N dispatches of every kernel (steady state of a long run)
        last = int(dirs[0].split("=", 1)[1])
        dirs = dirs[1:]
    acc = defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = defaultdict(int)
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if not k:
                    continue
                key = (row["Counter_Name"], k)
                seen[key] += 1
                if seen[key] > 1:
                    acc[key].append(float(row["Counter_Value"]))
    w = csv.writer(sys.stdout)
    w.writerow(["counter", "kernel", "dispatches", "mean", "min", "max"])
    for (c, k), v in sorted(acc.items()):
        if last:
            v = v[-last:]
        w.writerow([c, k, len(v), "%.3f" % (sum(v) / len(v)), "%.3f" % min(v), "%.3f" % max(v)])


if __name__ == "__main__":
    main(sys.argv[1:])
