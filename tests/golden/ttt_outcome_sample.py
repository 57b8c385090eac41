"""Outcome distribution of the REFERENCE's own Tic-Tac-Toe self-play (MCTS vs MCTS, random rollouts with NumPy's global
RNG, play_TTT.py's loop and kwargs at the README's budget of 1 000 rollouts per move) -- BUILD CONTAINER ONLY.

The README shows one such game ending in a draw ("optimal play ... will always result in a draw", README:100-168); over many
games the reference's search at UCT_C = 4 draws about three games in four and loses the rest as player 2.  The engine's
distribution is tested against this sample (tests/test_tictactoe_gpu.py).  Writes ttt_sample_v1.json.

    python tests/golden/ttt_outcome_sample.py [games per seed] [seeds...]
"""
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def play(args):
    budget, games, seed = args
    import numpy as np
    import ref_tools as rt  # noqa: F401  (installs the shim, puts /root/reference on the path)
    from MCTS import MCTS, MCTS_Node
    from TicTacToe import TicTacToe
    from make_golden import mcts_kwargs
    np.random.seed(seed)
    outs = []
    for _ in range(games):
        env = TicTacToe()
        initial = env.state
        mk = mcts_kwargs(budget, training=False, env=env)
        mk["NEURAL_NET"] = False
        MCTS(**mk)
        root1 = MCTS_Node(initial, parent=None)
        best1 = best2 = root2 = None
        while not env.done:
            if env.current_player(env.state) == "player1":
                if env.move_count != 0:
                    root1 = MCTS.new_root_node(best1)
                root = root1
            else:
                root2 = MCTS_Node(env.state, parent=None, initial_state=initial) if env.move_count == 1 else MCTS.new_root_node(best2)
                root = root2
            MCTS.begin_tree_search(root)
            best = MCTS.best_child(root)
            if root is root1:
                best1 = best
            else:
                best2 = best
            env.step(best.state)
        outs.append(env.outcome)
    return outs


if __name__ == "__main__":
    games = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    seeds = [int(v) for v in sys.argv[2:]] or [1, 2, 3, 4, 5, 6, 7, 8]
    with mp.Pool(min(8, len(seeds))) as pool:
        res = pool.map(play, [(1000, games, s) for s in seeds])
    flat = [o for r in res for o in r]
    out = {"budget": 1000, "games": len(flat), "seeds": seeds, "draw": flat.count("draw"), "player1_wins": flat.count("player1_wins"),
           "player2_wins": flat.count("player2_wins")}
    json.dump(out, open(os.path.join(os.environ.get("CKR_GOLDEN_OUT", HERE), "ttt_sample_v1.json"), "w"), indent=1)
    print(out)
