#!/usr/bin/env python3
"""Socket power and shader clock while the float32-grade conv stack runs back to back, on random planes
(the self-play workload's operand statistics) and on all-zero planes (the same instruction stream with
zero B operands): evidence for what bounds the kernel.  Samples `rocm-smi` from a side thread.

    python tools/power_probe.py [seconds per arm]
"""
import json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from checkers_mcts_amd import net as N
from checkers_mcts_amd.fused import FusedEvaluator

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
S = 4096


def sample():
    """(watts, sclk MHz) from rocm-smi; None when a field is missing."""
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showpower", "--showclocks", "--json"],
                             capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(out).values()))
    except Exception:
        return None, None
    watts = sclk = None
    for k, v in card.items():
        if "power" in k.lower() and watts is None:
            m = re.search(r"[\d.]+", str(v))
            watts = float(m.group(0)) if m else None
        if "sclk" in k.lower() and sclk is None:
            m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
            sclk = float(m.group(1)) if m else None
    return watts, sclk


def arm(name, fe, x):
    stop, samples = threading.Event(), []

    def poll():
        while not stop.is_set():
            samples.append(sample())
            time.sleep(0.2)
    for _ in range(20):
        fe.conv_only(x)
    torch.cuda.synchronize()
    th = threading.Thread(target=poll)
    th.start()
    n, t0 = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(50):
            fe.conv_only(x)
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    w = [a for a, _ in samples[2:] if a]
    c = [b for _, b in samples[2:] if b]
    flops = 2 * (64 * 9 * 14 * 128 + 7 * 64 * 9 * 128 * 128) * S
    return dict(arm=name, us_per_launch=us, algorithmic_tflops=flops / us / 1e6, executed_tflops=3 * flops / us / 1e6,
                watts_mean=sum(w) / len(w) if w else None, watts_max=max(w) if w else None,
                sclk_mhz_mean=sum(c) / len(c) if c else None, sclk_mhz_min=min(c) if c else None, samples=len(samples))


def main():
    m = N.PolicyValueNet(128).keras_init(0).eval().cuda()
    fe = FusedEvaluator(m, S, mode="f16x3")
    idle = sample()
    rnd = (torch.rand(S, 8, 8, 14, device="cuda") < 0.2).float().contiguous()
    out = [dict(arm="idle", watts=idle[0], sclk_mhz=idle[1])]
    out.append(arm("random planes (density 0.2)", fe, rnd))
    out.append(arm("all-zero planes", fe, torch.zeros_like(rnd)))
    try:
        cap = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        out.append(dict(arm="max power", raw=json.loads(cap)))
    except Exception:
        pass
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
