"""What the NumPy promotion rules change in the reference's search (BUILD CONTAINER ONLY: imports /root/reference).

The reference pins NumPy 1.19 (requirements.txt:68): there `python_number op np.float32` gives float64, so MCTS_Node's
total reward (MCTS.py:419-430) accumulates in float64.  Under NumPy >= 2 (NEP 50, the version installed here) it stays
float32, and the engine follows that.  The committed fixtures cannot tell the two apart: HashNet's values are dyadic
rationals, every sum is exact in either precision (they regenerate bit-identically under NumPy 1.26.4, see
VALIDATION.md).  This script drives the same searches with a network whose values are NOT exactly summable
(HashNet's v times 0.3, priors through a float32 softmax-like reweighting) and dumps the root statistics in float64;
run it under both interpreters and compare:

    python tests/golden/numpy_promotion_study.py /tmp/np2.npz
    /opt/conda/bin/python3.9 tests/golden/numpy_promotion_study.py /tmp/np1.npz
    python tests/golden/numpy_promotion_study.py --compare /tmp/np2.npz /tmp/np1.npz
"""
import sys

import numpy as np


def run(out):
    import ref_tools as rt
    import ref_shim
    from MCTS import MCTS, MCTS_Node
    from make_golden import mcts_kwargs

    class InexactNet(ref_shim.HashNet):
        def predict(self, x):
            p, v = ref_shim.HashNet.predict(self, x)
            p = (p * np.float32(0.7) + np.float32(1.0 / 3.0)).astype(np.float32)
            return [p, (v * np.float32(0.3)).astype(np.float32)]

    res = {}
    for ci, (budget, salt, max_plies) in enumerate(((60, 3, 40), (200, 5, 16), (25, 7, 200))):
        env = rt.new_env()
        env.neural_net = InexactNet(salt)
        MCTS(**mcts_kwargs(budget, training=False, env=env))
        ns, ws, chosen, wtypes = [], [], [], set()
        initial = env.state
        root1 = MCTS_Node(initial, parent=None)
        best1 = best2 = root2 = None
        while not env.done and env.move_count < max_plies:
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
            for c in root.children:
                ns.append(c.n); ws.append(float(c.w)); wtypes.add(type(c.w).__name__)
            chosen.append((int(best.state[14, 0, 0]) - 6) * 64 + 8 * int(best.state[14, 0, 1]) + int(best.state[14, 0, 2]))
            env.step(best.state)
        res["c%d_n" % ci] = np.array(ns, np.int64)
        res["c%d_w" % ci] = np.array(ws, np.float64)
        res["c%d_chosen" % ci] = np.array(chosen, np.int64)
        print("case", ci, "plies", len(chosen), "type of MCTS_Node.w:", sorted(wtypes), "numpy", np.__version__)
    np.savez_compressed(out, **res)


def compare(a, b):
    x, y = np.load(a), np.load(b)
    for k in x.files:
        u, v = x[k], y[k]
        if u.shape != v.shape:
            print(k, "different lengths", u.shape, v.shape)
        elif k.endswith("_w"):
            d = np.abs(u - v)
            print(k, "%d of %d differ, max |diff| %.3g (max |w| %.3g)" % (int((u != v).sum()), u.size, float(d.max()), float(np.abs(u).max())))
        else:
            print(k, "identical" if np.array_equal(u, v) else "%d of %d differ" % (int((u != v).sum()), u.size))


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        compare(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1])
