#!/usr/bin/env python3
"""One training step as a timeline, from a rocprofv3 --kernel-trace directory of tools/train_bench.py:
    python tools/train_timeline.py gpurun_out/prof_dir > profiles/rNN_train_step_timeline.txt
Takes the last complete step of the hand-written path (from one k_im2col launch to the next) and prints start (us from the
step's first kernel), duration, queue and kernel name, plus the step's span and the busy time per queue."""
import csv, glob, os, re, sys


def main(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z0-9_]+(?:<[^>]*>)?)", r["Kernel_Name"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), m.group(1) if m else r["Kernel_Name"][:40]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[3].startswith("k_im2col")]
    if len(starts) < 3:
        raise SystemExit("no complete training step found")
    a, b = starts[-3], starts[-2]
    step = rows[a:b]
    t0 = step[0][0]
    queues = {q: "q%d" % i for i, q in enumerate(sorted({r[2] for r in step}))}
    print("# one graph-replayed training step under rocprofv3 --kernel-trace: start us, duration us, queue, kernel")
    busy = {}
    for s, e, q, n in step:
        print("%8.1f %6.1f %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, queues[q], n))
        busy[queues[q]] = busy.get(queues[q], 0.0) + (e - s) / 1e3
    print("# span %.1f us (to the next step's first kernel: %.1f us); %d launches; busy per queue: %s"
          % ((max(r[1] for r in step) - t0) / 1e3, (rows[b][0] - t0) / 1e3, len(step), {k: round(v, 1) for k, v in busy.items()}))


if __name__ == "__main__":
    main(sys.argv[1])
