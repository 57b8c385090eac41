#!/usr/bin/env python3
"""Kernel timeline of the self-play step under several HIP streams, from a `rocprofv3 --output-format csv --kernel-trace`
directory of bench.py (or of any SplitRunner job):

    python tools/step_timeline.py <trace dir> [--steps 50] [--print-steps 3] > profiles/rNN_step_timeline_<what>.txt

Takes the last `--steps` steps of every part before the run's end (or before the first kernel that is not part of a step) and
reports, from the dispatch records alone (start, end, hardware queue, stream, grid):
  * which stream / hardware queue every part's kernels went to;
  * per kernel: launches, mean / min / max span, and the span's share spent beside 0 / 1 / 2 ... conv-stack launches of OTHER queues;
  * how many conv-stack launches were in flight over the window's wall time (the overlap the streams actually reach);
  * per queue: busy time, and the idle gaps between consecutive kernels of the chain;
  * the first `--print-steps` steps as a table: start (us), span (us), queue, kernel, grid.
"""
import argparse
import csv
import glob
import os
import re
import sys
from collections import defaultdict

STEP_KERNELS = ("k_step", "k_conv_stack", "k_policy_head", "k_prefetch_consume", "k_value_mlp", "k_partition", "k_sort", "k_net")


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "0"
            wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "1"
            rows.append(dict(s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]), q=r.get("Queue_Id", "?"),
                             st=r.get("Stream_Id", "?"), k=short(r["Kernel_Name"]), wgs=int(grid) // max(1, int(wg)),
                             vgpr=r.get("VGPR_Count", "?"), lds=r.get("LDS_Block_Size", "?")))
    rows.sort(key=lambda r: r["s"])
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--print-steps", type=int, default=3)
    ap.add_argument("--skip-tail", type=int, default=0, help="ignore this many step-kernel dispatches at the end of the trace")
    a = ap.parse_args()
    rows = load(a.dir)
    if not rows:
        raise SystemExit("no *kernel_trace.csv under %s" % a.dir)
    step_rows = [r for r in rows if r["k"].startswith(STEP_KERNELS)]
    if a.skip_tail:
        step_rows = step_rows[:-a.skip_tail]
    # the window: the last `steps` k_step launches of every queue that carries k_step
    by_q = defaultdict(list)
    for r in step_rows:
        if r["k"] == "k_step":
            by_q[r["q"]].append(r)
    # queues of the steady state = those of the last 3 x 50 k_step launches
    last = [r for r in step_rows if r["k"] == "k_step"][-a.steps * 8:]
    qs = sorted({r["q"] for r in last})
    t0 = max(by_q[q][-min(a.steps, len(by_q[q]))]["s"] for q in qs)          # every queue has `steps` (or all of its) launches after t0
    t1 = max(r["e"] for r in step_rows)
    win = [r for r in step_rows if r["s"] >= t0]
    n_step = {q: sum(1 for r in win if r["q"] == q and r["k"] == "k_step") for q in qs}
    wall = (t1 - t0) / 1e3
    print("# window: %.1f us, %d dispatches, k_step launches per queue %s -> %.1f us per step of all parts"
          % (wall, len(win), dict(n_step), wall / max(1, max(n_step.values()))))
    print("# queue <- stream(s): kernels")
    for q in sorted({r["q"] for r in win}):
        ks = defaultdict(int)
        for r in win:
            if r["q"] == q:
                ks[r["k"]] += 1
        print("#   queue %s <- stream %s: %s" % (q, ",".join(sorted({r["st"] for r in win if r["q"] == q})), dict(ks)))
    conv = [r for r in win if r["k"].startswith("k_conv_stack")]

    def conv_beside(r):
        """us of r's span with 0, 1, 2 ... conv launches of OTHER queues in flight"""
        ev = []
        for c in conv:
            if c["q"] != r["q"] and c["e"] > r["s"] and c["s"] < r["e"]:
                ev.append((max(c["s"], r["s"]), 1))
                ev.append((min(c["e"], r["e"]), -1))
        ev.sort()
        out, cur, t = defaultdict(float), 0, r["s"]
        for tt, d in ev:
            out[cur] += (tt - t) / 1e3
            cur, t = cur + d, tt
        out[cur] += (r["e"] - t) / 1e3
        return out

    print("# kernel, launches, mean / min / max span us, workgroups (mean), VGPRs, LDS; share of the span beside n conv launches of other queues")
    names = sorted({r["k"] for r in win}, key=lambda k: -sum(r["e"] - r["s"] for r in win if r["k"] == k))
    for k in names:
        rs = [r for r in win if r["k"] == k]
        sp = [(r["e"] - r["s"]) / 1e3 for r in rs]
        beside = defaultdict(float)
        for r in rs:
            for n, us in conv_beside(r).items():
                beside[n] += us
        tot = sum(beside.values()) or 1.0
        print("%-28s %5d  %8.1f %8.1f %8.1f  wgs %7.0f  vgpr %s lds %s   beside %s"
              % (k, len(rs), sum(sp) / len(sp), min(sp), max(sp), sum(r["wgs"] for r in rs) / len(rs), rs[0]["vgpr"], rs[0]["lds"],
                 " ".join("%d:%.0f%%" % (n, 100 * beside[n] / tot) for n in sorted(beside))))
    # conv launches in flight over the wall time
    ev = []
    for c in conv:
        ev.append((c["s"], 1))
        ev.append((c["e"], -1))
    ev.sort()
    hist, cur, t = defaultdict(float), 0, t0
    for tt, d in ev:
        hist[cur] += (tt - t) / 1e3
        cur, t = cur + d, tt
    hist[cur] += (t1 - t) / 1e3
    print("# conv-stack launches in flight, share of the window: " + "  ".join("%d: %.1f%%" % (n, 100 * hist[n] / wall) for n in sorted(hist)))
    print("# per queue: busy us (share of the window), mean gap between consecutive kernels of the chain, gap in front of k_step / conv / heads")
    for q in qs:
        rs = [r for r in win if r["q"] == q]
        busy = sum(r["e"] - r["s"] for r in rs) / 1e3
        gaps = defaultdict(list)
        for x, y in zip(rs, rs[1:]):
            gaps[y["k"]].append((y["s"] - x["e"]) / 1e3)
        allg = [g for v in gaps.values() for g in v]
        print("#   queue %s: busy %.0f us (%.0f%%), mean gap %.1f us; %s"
              % (q, busy, 100 * busy / wall, sum(allg) / max(1, len(allg)),
                 "  ".join("%s %.1f" % (k, sum(v) / len(v)) for k, v in sorted(gaps.items()))))
    # a few steps as a table
    first_q = qs[0]
    ks = [r for r in win if r["q"] == first_q and r["k"] == "k_step"]
    if len(ks) > a.print_steps:
        tA, tB = ks[0]["s"], ks[a.print_steps]["s"]
        names_q = {q: "q%d" % i for i, q in enumerate(qs)}
        print("# %d steps: start us, span us, queue, kernel, workgroups" % a.print_steps)
        for r in win:
            if tA <= r["s"] < tB:
                print("%9.1f %8.1f %s %-26s %6d" % ((r["s"] - tA) / 1e3, (r["e"] - r["s"]) / 1e3, names_q.get(r["q"], r["q"]), r["k"], r["wgs"]))


if __name__ == "__main__":
    main()
