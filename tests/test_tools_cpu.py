"""CPU: the measurement tools that turn profiler output into the committed summaries compute what they say (synthetic inputs)."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, rows):
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "t_kernel_trace.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "VGPR_Count",
                    "LDS_Block_Size", "Workgroup_Size_X", "Grid_Size_X"])
        for q, st, name, s, e, wgs in rows:
            w.writerow(["KERNEL_DISPATCH", 1, q, st, name, s, e, 64, 1024, 256, 256 * wgs])


def test_step_timeline_reads_queues_spans_and_overlap(tmp_path):
    """tools/step_timeline.py on a hand-made trace: two part-batches on two hardware queues, chains k_step -> conv -> heads of known
    length, the second part's conv overlapping the first part's tree kernel by a known amount."""
    rows, t = [], 0
    for step in range(12):                                   # queue 2: k_step 100 us, conv 400 us, heads 20 us, back to back
        base = step * 520_000
        rows += [(2, 1, "void ckr::k_step<float>(args)", base, base + 100_000, 342),
                 (2, 1, "ckrx::k_conv_stack_x3(ckrx::Args)", base + 100_000, base + 500_000, 683),
                 (2, 1, "void ckrp::k_policy_head<1>(args)", base + 500_000, base + 520_000, 86)]
        b2 = base + 260_000                                  # queue 3: the same chain half a step later
        rows += [(3, 2, "void ckr::k_step<float>(args)", b2, b2 + 100_000, 342),
                 (3, 2, "ckrx::k_conv_stack_x3(ckrx::Args)", b2 + 100_000, b2 + 500_000, 683),
                 (3, 2, "void ckrp::k_policy_head<1>(args)", b2 + 500_000, b2 + 520_000, 86)]
    _trace(str(tmp_path / "trace"), rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_timeline.py"), str(tmp_path / "trace"), "--steps", "8", "--print-steps", "2"],
                         capture_output=True, text=True, check=True).stdout
    assert "queue 2 <- stream 1" in out and "queue 3 <- stream 2" in out
    conv = next(ln for ln in out.splitlines() if ln.startswith("k_conv_stack_x3"))
    kst = next(ln for ln in out.splitlines() if ln.startswith("k_step "))
    assert " 400.0 " in conv and "wgs     683" in conv and " 100.0 " in kst
    # every k_step lies inside the other queue's conv launch (the last one of the trace excepted): beside ONE conv for >= 90 % of its span
    share = dict(x.split(":") for x in kst.split("beside")[1].split())
    assert float(share["1"].rstrip("%")) >= 90.0 and "2" not in share
    assert "conv-stack launches in flight" in out


def test_kernel_stats_averages_per_kernel(tmp_path):
    rows = [(2, 1, "ckrx::k_conv_stack_x3(ckrx::Args)", 0, 300_000, 10), (2, 1, "ckrx::k_conv_stack_x3(ckrx::Args)", 400_000, 600_000, 10),
            (2, 1, "void ckr::k_step<float>(args)", 700_000, 760_000, 4)]
    _trace(str(tmp_path / "trace"), rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_stats.py"), str(tmp_path / "trace")], capture_output=True, text=True, check=True).stdout
    lines = {ln.split(",")[0]: ln.split(",") for ln in out.strip().splitlines()[1:]}
    assert lines["k_conv_stack_x3"][1] == "2" and abs(float(lines["k_conv_stack_x3"][3]) - 250.0) < 1e-6
    assert lines["k_step"][1] == "1" and abs(float(lines["k_step"][3]) - 60.0) < 1e-6
