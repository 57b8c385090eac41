"""Compare two directories of fixtures written by make_golden.py (e.g. the committed ones, generated under NumPy 2.x,
with a regeneration under NumPy 1.x: CKR_GOLDEN_OUT=/tmp/np1 /opt/conda/bin/python3.9 make_golden.py).

    python tests/golden/compare_fixtures.py tests/golden /tmp/np1

Prints one line per file: identical arrays / values, or the keys that differ (count, max difference)."""
import json
import os
import sys

import numpy as np


def compare_npz(a, b):
    x, y = np.load(a, allow_pickle=True), np.load(b, allow_pickle=True)
    bad = []
    for k in sorted(set(x.files) | set(y.files)):
        if k not in x.files or k not in y.files:
            bad.append("%s: only in one file" % k)
            continue
        u, v = x[k], y[k]
        if u.shape != v.shape or u.dtype != v.dtype:
            bad.append("%s: %s %s vs %s %s" % (k, u.dtype, u.shape, v.dtype, v.shape))
        elif u.dtype.kind == "f":
            if not np.array_equal(u.view(np.uint8), v.view(np.uint8)):
                bad.append("%s: %d of %d values differ in their bits, max |diff| %.3g" % (k, int((u != v).sum()), u.size, float(np.abs(u.astype(np.float64) - v.astype(np.float64)).max())))
        elif not np.array_equal(u, v):
            bad.append("%s: %d of %d differ" % (k, int((u != v).sum()), u.size))
    return len(x.files), bad


def main():
    da, db = sys.argv[1], sys.argv[2]
    status = 0
    for name in sorted(os.listdir(da)):
        pa, pb = os.path.join(da, name), os.path.join(db, name)
        if not os.path.exists(pb):
            continue
        if name.endswith(".npz"):
            n, bad = compare_npz(pa, pb)
            print("%-20s %3d arrays: %s" % (name, n, "bit-identical" if not bad else "; ".join(bad)))
            status |= bool(bad)
        elif name.endswith(".json"):
            x, y = json.load(open(pa, encoding="utf-8")), json.load(open(pb, encoding="utf-8"))
            keys = [k for k in sorted(set(x) | set(y)) if x.get(k) != y.get(k)] if isinstance(x, dict) and isinstance(y, dict) else ([] if x == y else ["*"])
            # ln_table is an INPUT (this host's np.log, last-bit differences between NumPy builds: VALIDATION.md), not a result
            print("%-20s %s" % (name, "identical" if not keys else "DIFFERENT in " + ", ".join(keys)))
            status |= bool([k for k in keys if k != "ln_table"])
    return status


if __name__ == "__main__":
    sys.exit(main())
