"""CPU: the committed bench line of the round (profiles/r06_bench_default.json, written by `python bench.py` on an
MI355X) carries every field of the driver's contract, with the hot path's own metric and roofline / cpu_baseline objects; its
roofline traffic agrees with the committed PMC table; README.md's numbers are the ones rendered from this line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r06_bench_default.json")


def load():
    return json.loads(open(LINE).read().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_fields():
    d = load()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1.0e6 and abs(d["ms_per_step"] * d["value"] / 1e3 - d["expansions"] / d["steps"]) < 1.0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert isinstance(base.get("metric", ""), str)
    assert d["plies"] > 0 and d["terminal_visits"] > 0 and d["games_per_hour"] is not None      # steady state, M2 present
    # the roofline is derived from the window (reproducible from the line), the cache-off figure is in the same line, and
    # the per-rank arrays are there
    launches = r["launches_in_window"]
    assert abs(r["achieved"] - r["flops_per_unit"] * d["nn_evals_by_rank"][0] / (d["ms_per_step_by_rank"][0] * d["steps"] / 1e3) / 1e12) < 1e-6 * r["achieved"]
    assert abs(r["rows_per_launch_in_window"] * launches - d["nn_evals_by_rank"][0]) < 1.0 and r["kernel_alone"]["ms_per_launch"] > 0
    assert 0.3 * d["value"] < d["extra"]["cache_off"]["value"] < d["value"] and d["extra"]["cache_off"]["dup_leaves"] == 0
    assert len(d["ms_per_step_by_rank"]) == len(d["expansions_by_rank"]) == d["n_gpus"] and d["stalled_steps_in_window"] == 0
    assert d["whole_run"]["efficiency_vs_steady_state"] > 0.85
    # round 5: the host side of the rank, and cfg 5 as the WHOLE per-GPU share beside its mid-game rate
    assert len(d["host_issue_ms_per_step_by_rank"]) == d["n_gpus"] and 0 < d["host_issue_ms_per_step_by_rank"][0] < 0.5 * d["ms_per_step"]
    assert d["host_placement_by_rank"][0]["cpus"] >= 1
    a = d["extra"]["arena_cfg5_shape"]
    w = a["whole_share"]
    assert w["games"] == 4096 and w["new_net_wins"] + w["old_net_wins"] + w["draws"] == 4096 and w["pool_overflows"] == 0
    assert abs(w["sims_per_s"] - w["sims"] / w["seconds"]) < 1e-6 * w["sims_per_s"] and w["sims"] == 800 * w["plies"]
    # the quoted figure is the whole tournament's (since round 6 all 4 096 games play from the first step, and the share's rate -- its tail
    # chains many network-free simulations per step -- is no longer below the mid-game window's)
    assert a["sims_per_s"] == w["sims_per_s"] and 0.5 * w["sims_per_s"] < a["mid_game_window"]["sims_per_s"] < 2.0 * w["sims_per_s"]


def test_roofline_traffic_agrees_with_the_committed_pmc_table():
    """roofline.traffic is read off profiles/r05_pmc_conv_by_launch_size.csv at the window's own launch size (no linear scaling from
    another size): within 10 % of the nearest measured row."""
    sys.path.insert(0, ROOT)
    import bench
    d = load()
    r = d["roofline"]
    tab = bench.pmc_table()
    assert len(tab) >= 6 and 880 in tab and 4096 in tab
    rows = r["rows_per_launch_in_window"]
    near = min(tab, key=lambda k: abs(k - rows))
    assert abs(near - rows) < 0.15 * rows and abs(r["traffic"] - tab[near]) < 0.10 * tab[near]
    assert abs(bench.pmc_traffic("fp32", int(round(rows)))[0] - r["traffic"]) < 0.02 * r["traffic"]
    # traffic follows workgroup rounds, not rows: one round (<= 1 024 boards) costs about the same whatever its rows
    assert abs(tab[512] - tab[1024]) < 0.05 * tab[1024] and tab[2048] > 1.6 * tab[1024]


def test_readme_numbers_are_the_committed_line():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import readme_numbers
    block = readme_numbers.render()
    text = open(os.path.join(ROOT, "README.md")).read()
    assert block in text, "README.md is stale: run `python tools/readme_numbers.py`"
