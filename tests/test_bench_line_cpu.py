"""CPU: the committed bench line of the round (profiles/r04_bench_default.json, written by `python bench.py` on an
MI355X) carries every field of the driver's contract, with the hot path's own metric and roofline / cpu_baseline objects."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    line = open(os.path.join(ROOT, "profiles", "r04_bench_default.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
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
    # round 4: the roofline is derived from the window (reproducible from the line), the cache-off figure is in the same line, and
    # the per-rank arrays are there
    launches = r["launches_in_window"]
    assert abs(r["achieved"] - r["flops_per_unit"] * d["nn_evals_by_rank"][0] / (d["ms_per_step_by_rank"][0] * d["steps"] / 1e3) / 1e12) < 1e-6 * r["achieved"]
    assert abs(r["rows_per_launch_in_window"] * launches - d["nn_evals_by_rank"][0]) < 1.0 and r["kernel_alone"]["ms_per_launch"] > 0
    assert 0.3 * d["value"] < d["extra"]["cache_off"]["value"] < d["value"] and d["extra"]["cache_off"]["dup_leaves"] == 0
    assert len(d["ms_per_step_by_rank"]) == len(d["expansions_by_rank"]) == d["n_gpus"] and d["stalled_steps_in_window"] == 0
    assert d["whole_run"]["efficiency_vs_steady_state"] > 0.85
