"""GPU: bench.py itself as the driver launches it -- rehearsed on the one GPU of the box.  (i) `python bench.py --gpus 8` with
CKR_DIST_BACKEND=gloo: eight ranks share the GPU, every rank plays its block of workers, the timed window is bracketed by barriers,
rank 0 gathers the tuples of all ranks (training_pipeline.py:323-332: Pool.map's result hand-back) and prints ONE JSON line with
per-rank fields; the leaf caches are sized for eight ranks on one device.  (ii) the same script under torch.distributed.run with one
rank and the `nccl` backend (= RCCL): the RCCL branch of bench.py's own gather executes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--slots", "256", "--budget", "20", "--games-per-slot", "2", "--preroll", "300", "--warmup", "10", "--steps", "50",
         "--cpu-seconds", "0", "--extra-steps", "0", "--profile-steps", "4"]


def last_json(out):
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and lines, (out.stdout[-2000:], out.stderr[-4000:])
    return json.loads(lines[-1])


def check_line(d, world):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == world and d["steps"] == 50 and d["scaling"] == "weak" and d["value"] > 0
    assert len(d["ms_per_step_by_rank"]) == len(d["expansions_by_rank"]) == world
    assert abs(sum(d["expansions_by_rank"]) - d["expansions"]) < 0.5 and min(d["expansions_by_rank"]) > 0
    assert abs(max(d["ms_per_step_by_rank"]) - d["ms_per_step"]) < 1e-6
    # the host side of every rank: issuing a step's graph replays takes well under the step's own time (a host-bound rank would
    # otherwise hide behind the GPU figures), and every rank reports where its thread was placed
    assert len(d["host_issue_ms_per_step_by_rank"]) == len(d["host_placement_by_rank"]) == world
    for issue, ms in zip(d["host_issue_ms_per_step_by_rank"], d["ms_per_step_by_rank"]):
        assert 0 < issue < 0.5 * ms, (d["host_issue_ms_per_step_by_rank"], d["ms_per_step_by_rank"])
    for pl in d["host_placement_by_rank"]:
        assert pl["cpus"] >= 1 and pl["first_cpu"] <= pl["last_cpu"]
    w = d["whole_run"]
    assert w["games"] == world * 256 * 2 and len(w["seconds_by_rank"]) == world and w["pool_overflows"] == 0
    g = w["gather"]
    assert len(g["bytes_by_rank"]) == world and abs(sum(g["bytes_by_rank"]) - g["bytes"]) < 0.5 and g["bytes"] == g["tuples"] * 288
    assert min(g["bytes_by_rank"]) > 0                                         # every rank contributed its games' tuples
    assert w["plies"] <= g["tuples"] <= w["plies"] + w["games"]                # one tuple per ply + the terminal tuple of every game not adjudicated
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["rows_per_launch_in_window"] > 0
    assert abs(r["achieved"] - r["flops_per_unit"] * d["nn_evals_by_rank"][0] / (d["ms_per_step_by_rank"][0] * 50 / 1e3) / 1e12) < 1e-6 * r["achieved"]


def test_bench_eight_ranks_gloo_on_one_gpu(tmp_path):
    env = dict(os.environ, CKR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + SMALL, env=env, capture_output=True, text=True,
                         timeout=1500, cwd=str(tmp_path))
    d = last_json(out)
    check_line(d, 8)
    assert d["config"]["ranks_per_device"] == 8 and "gloo" in d["whole_run"]["gather"]["collective"]
    assert d["config"]["leaf_cache_log2"] <= 24                                  # sized for eight ranks on one device (256 slots: 2^24 at most)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_gloo_8ranks_one_gpu_small.json"), "w") as f:
        f.write(json.dumps(d) + "\n")


def test_bench_under_torchrun_with_rccl_world_of_one(tmp_path):
    env = dict(os.environ, CKR_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "CKR_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           "29631", os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=str(tmp_path))
    d = last_json(out)
    check_line(d, 1)
    assert "RCCL" in d["whole_run"]["gather"]["collective"]


def test_arena_job_on_two_ranks_sharing_the_gpu(tmp_path):
    """BASELINE cfg5's job shape through tools/selfplay_run.py (the command behind profiles/r04_cfg5_full_*): two ranks under
    torch.distributed.run share the box's GPU, every rank plays its block of arena games to their natural end, the W / L / D
    bookkeeping is summed over the ranks (training_pipeline.py:505-560: the tournament's outcome counts)."""
    env = dict(os.environ, CKR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29633", os.path.join(ROOT, "tools", "selfplay_run.py"), "--tournament", "--slots", "64", "--budget", "20", "--nn-dtype", "fp32"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=str(tmp_path))
    d = last_json(out)
    a = d["all_ranks"]
    assert d["n_gpus"] == 2 and d["games"] == 128 and a["failed"] == 0
    assert a["new_net_wins"] + a["old_net_wins"] + a["draws"] == 128
    assert sum(d["rank0_outcomes"].values()) == 64 and d["rank0_adjudicated"] == 0          # natural ends only: no TERMINATE_CNT in the arena
    assert d["expansions"] > 0 and a["longest_game_plies"] >= d["rank0_game_length"]["max"] >= d["rank0_game_length"]["min"] > 0
