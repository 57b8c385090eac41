#!/usr/bin/env python3
"""The bf16 leg's two speeds (0.37 / 0.56 ms per step from run to run until round 4), made on purpose: three part-batches on torch
POOL streams (CKR_TORCH_STREAMS=1, as rounds 2-4 ran them), after BURN pool streams have been drawn by something else in the process
(in bench.py: graph captures of earlier legs, a number that depends on the tail of the main leg).  The HIP runtime maps pool stream
i onto one of GPU_MAX_HW_QUEUES = 4 hardware queues; with the right BURN two parts share a queue and their step chains serialise.
Run under `rocprofv3 --kernel-trace` and feed the trace to tools/step_timeline.py to see the queue of every part.

    BURN=0 python tools/queue_collision_demo.py      # one JSON line: ms per step
    BURN=1 python tools/queue_collision_demo.py"""
import json, os, sys, time
os.environ["CKR_TORCH_STREAMS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

a = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
burn = int(os.environ.get("BURN", "0"))
burned = [torch.cuda.Stream(device=dev) for _ in range(burn)]
for st in burned:                     # a stream gets its hardware queue at its first USE (least-used queue at that moment)
    with torch.cuda.stream(st):
        torch.zeros(8, device=dev).add_(1)
torch.cuda.synchronize(dev)
mode = os.environ.get("MODE", "bf16")
leg = bench.Leg(a, dev, mode, 0, a.slots, 64, True)
leg.warmup(3)
leg.step(int(os.environ.get("PREROLL", "4000")))
dt, d = bench.timed_window(leg, dev, 300)
print(json.dumps({"mode": mode, "burned_pool_streams": burn, "ms_per_step": dt / 300 * 1e3, "expansions_per_s": d["expansions"] / dt,
                  "part_streams": [hex(s.cuda_stream) for _, _, s in leg.runner.parts]}))
leg.close()
