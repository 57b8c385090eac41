#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_wino_pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in 0 3; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $O/v${v}_a -o p -- python $R/tools/winograd_pmc.py $v 7 > $O/v${v}_a.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY --kernel-trace -d $O/v${v}_b -o p -- python $R/tools/winograd_pmc.py $v 7 > $O/v${v}_b.log 2>&1
done
cd $R
python - $O <<'PY'
import sys, glob, os, sqlite3, csv
o = sys.argv[1]
for d in sorted(glob.glob(os.path.join(o, "v*_[ab]"))):
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "wino" in r.get("Kernel_Name", ""):
                rows.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if not rows:
        for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            c = sqlite3.connect(f)
            tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
            print(os.path.basename(d), "db tables:", [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()][:8])
            t = next((x for x in tabs if x.startswith("counters_collection")), None)
            if t:
                cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
                print(cols)
                for r in c.execute("select * from %s limit 3" % t): print(r)
                try:
                    for name, val in c.execute("select counter_name, avg(value) from %s where kernel_name like '%%wino%%' group by counter_name" % t):
                        rows.setdefault(name, []).append(val)
                except Exception as e:
                    print("query failed", e)
    print(os.path.basename(d), {k: round(sum(v) / len(v)) for k, v in sorted(rows.items())})
PY
