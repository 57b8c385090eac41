#!/bin/bash
# K2 k_children: L2 <-> fabric bytes per launch (separate --pmc passes for FETCH_SIZE and WRITE_SIZE, as MI355X_MICROARCH.md prescribes).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_k2_pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/$c -o p -- python $R/tools/k2_probe.py > $O/$c.log 2>&1
done
cd $R
python - $O <<'PY'
import sys, glob, os, sqlite3, csv
o = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(o, c)
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "k_children" in k or "k_movegen" in k:
                rows.setdefault(("k_children" if "k_children" in k else "k_movegen", r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    if not rows:
        for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            cn = sqlite3.connect(f)
            tabs = [r[0] for r in cn.execute("select name from sqlite_master where type in ('table','view')")]
            t = next((x for x in tabs if x.startswith("counters_collection")), None)
            if t:
                cols = [r[1] for r in cn.execute("pragma table_info(%s)" % t)]
                kc = "kernel_name" if "kernel_name" in cols else next(x for x in cols if "kernel" in x and "name" in x)
                for kname, name, val in cn.execute("select %s, counter_name, value from %s" % (kc, t)):
                    if "k_children" in kname or "k_movegen" in kname:
                        rows.setdefault(("k_children" if "k_children" in kname else "k_movegen", name), []).append(float(val))
    for (k, name), v in sorted(rows.items()):
        v = sorted(v)
        print(c, k, name, "launches", len(v), "median", v[len(v) // 2], "max", v[-1])
PY
