#!/usr/bin/env python3
"""README.md's measured numbers come from ONE committed bench line: this script renders the table between the
`<!-- numbers:begin -->` / `<!-- numbers:end -->` markers from profiles/r06_bench_default.json (the output of `python bench.py` on an
MI355X), and tests/test_bench_line_cpu.py checks that README.md holds exactly what it renders.

    python tools/readme_numbers.py            # rewrite the block in README.md
    python tools/readme_numbers.py --print    # print the block"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r06_bench_default.json")
BEGIN, END = "<!-- numbers:begin -->", "<!-- numbers:end -->"


def render(path=LINE):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    e, r, w, c = d["extra"], d["roofline"], d["whole_run"], d["cpu_baseline"]
    ar = e["arena_cfg5_shape"]
    rows = [
        ("source", "`profiles/%s`: `python bench.py` on one MI355X (boxes of the pool differ: the same code read 7.1-8.0 M over the round's sessions, with the conv stack's clock -- its lone launch takes 0.261-0.280 ms)" % os.path.basename(path)),
        ("BASELINE cfg 3 (100 sims/move, 4 096 concurrent games), float32-grade network, steady state",
         "**%.2f M node-expansions/s** (%.3f ms per step of 4 096 slots); %.2f M of them per second are network rows, %.2f M/s are positions the "
         "network had already evaluated (leaf cache)" % (d["value"] / 1e6, d["ms_per_step"], d["nn_evals_per_s"] / 1e6, d["cache_served_per_s"] / 1e6)),
        ("the same without the leaf cache (`extra.cache_off`)", "%.2f M node-expansions/s" % (e["cache_off"]["value"] / 1e6)),
        ("dominant kernel `k_conv_stack_x3` over the timed window",
         "%.0f TFLOP/s algorithmic = **%.3f of the 2.5 PFLOP/s 16-bit dense peak** (%.2f executed: three MFMAs per multiply-add; %.1f x the "
         "float32 matrix peak); alone on %d rows: %.3f ms" % (r["achieved"], r["frac"], r["executed_frac"], r["vs_fp32_matrix_peak"],
                                                             round(r["kernel_alone"]["rows_per_launch"]), r["kernel_alone"]["ms_per_launch"])),
        ("complete run: %d self-play games on 4 096 slots" % w["games"],
         "%.1f s = **%.2f M games/hour**, %.3f of the steady-state rate over the whole run" % (w["seconds"], d["games_per_hour"] / 1e6, w["efficiency_vs_steady_state"])),
        ("bf16 throughput mode (not a parity mode)", "%.1f M node-expansions/s" % (e["bf16_throughput_mode"]["value"] / 1e6)),
        ("BASELINE cfg 5's per-GPU share: %d arena games, 800 sims/move, two networks, natural end" % ar["whole_share"]["games"],
         "**%.1f s, %.2f M simulations/s over the whole share** (new / old / draws %d / %d / %d); %.1f M simulations/s in mid-game while every slot plays"
         % (ar["whole_share"]["seconds"], ar["whole_share"]["sims_per_s"] / 1e6, ar["whole_share"]["new_net_wins"], ar["whole_share"]["old_net_wins"],
            ar["whole_share"]["draws"], ar["mid_game_window"]["sims_per_s"] / 1e6)),
        ("small jobs through the drop-in classes, 200 sims/move (`extra.small_jobs`: the reference's own job sizes)",
         "tournament of 400 games **%.1f s**, self-play batch of 128 games %.1f s (whole calls, engine creation and calibration included)"
         % (e["small_jobs"]["tournament_400_games"]["seconds"], e["small_jobs"]["selfplay_128_games"]["seconds"])),
        ("one game, one search at a time (`MCTS.begin_tree_search`)", "%.0f k simulations/s, %.1f ms per 400-simulation search"
         % (e["single_game_search"]["sims_per_s"] / 1e3, e["single_game_search"]["search_api_ms_per_400_simulations"])),
        ("random-rollout MCTS (`NEURAL_NET=False`)", "%.1f M complete random playouts/s" % (e["random_rollout_mode"]["rollouts_per_s"] / 1e6)),
        ("rules kernel K1 on 2^24 boards", "%.0f G boards/s = %.2f TB/s algorithmic (%.0f %% of the 8 TB/s spec; above the ~6.3 TB/s a plain copy reaches: "
         "the 268-MB input is re-read by back-to-back launches and partly served by the 256-MB Infinity Cache)"
         % (e["movegen_k1"]["boards_per_s"] / 1e9, e["movegen_k1"]["achieved"] / 1e3, 100 * e["movegen_k1"]["frac"])),
        ("drop-in output path: `generate_Checkers_data(...).generate_data()` end to end, %d games (`extra.dropin_generate_data`)" % e["dropin_generate_data"]["games"],
         "%.1f s: self-play %.1f s + tuples -> the reference's float64 lists %.1f s + `pickle.dump` of %.1f GB %.1f s (host tail %.2f x the self-play; "
         "16 384 games: 22.8 + 5.9 + 17.2 s for 17.3 GB, `profiles/r06_dropin_generate_data_16384.json`)"
         % (e["dropin_generate_data"]["seconds"], e["dropin_generate_data"]["selfplay_s"], e["dropin_generate_data"]["to_memory_s"],
            e["dropin_generate_data"]["pickle_bytes"] / 1e9, e["dropin_generate_data"]["pickle_s"], e["dropin_generate_data"]["host_tail_over_selfplay"])),
        ("training step, 128 boards (`csrc/ckr_train.hip`)", "%.2f ms = %.0f k samples/s, %.1f x PyTorch + MIOpen"
         % (e["training_step"]["ms_per_step"], e["training_step"]["samples_per_s"] / 1e3, e["training_step"]["speedup_vs_torch_miopen"])),
        ("CPU baseline in the same run (C oracle search + PyTorch-CPU network, %d host threads)" % c["cores"], "%.1f k node-expansions/s" % (c["value"] / 1e3)),
    ]
    out = [BEGIN, "| | |", "|---|---|"] + ["| %s | %s |" % kv for kv in rows] + [END]
    return "\n".join(out)


def main():
    block = render()
    if "--print" in sys.argv:
        print(block)
        return
    p = os.path.join(ROOT, "README.md")
    s = open(p).read()
    a, b = s.index(BEGIN), s.index(END) + len(END)
    open(p, "w").write(s[:a] + block + s[b:])


if __name__ == "__main__":
    main()
