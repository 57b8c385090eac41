"""CPU: the injected test noise (noise_mode 1) is the same function in the reference-side shim (tests/golden/ref_shim.py, which
feeds it to the imported reference when the fixtures are made) and in the C oracle; and the shim's restatement of
np.random.choice's inverse-CDF draw picks what the real RandomState.choice picks, given the uniform that one consumes."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_noise_function_shim_equals_oracle(oracle):
    import ref_shim
    rng = np.random.RandomState(3)
    for _ in range(300):
        seed = int(rng.randint(0, 2 ** 31)) << int(rng.randint(0, 33))
        worker, ctr, n = int(rng.randint(0, 1 << 20)), int(rng.randint(0, 1 << 24)), int(rng.randint(1, 49))
        d = ref_shim.noise_dirichlet(seed, worker, ctr, n)
        assert d.dtype == np.float64 and d.tobytes() == oracle.noise_dirichlet(seed, worker, ctr, n).tobytes()
        assert ref_shim.noise_uniform(seed, worker, ctr) == oracle.noise_uniform(seed, worker, ctr)
        assert abs(d.sum() - 1.0) < 1e-12 and (d > 0).all()


def test_choice_restatement_equals_numpy(oracle):
    import ref_shim
    rng = np.random.RandomState(11)
    for t in range(4000):
        n = int(rng.randint(1, 31))
        visits = rng.randint(0, 400, n).astype(np.int64)
        if visits.sum() == 0:
            visits[0] = 1
        tau = [1.0, 0.9, 0.5, 0.30000000000000004, 0.10000000000000003][t % 5]
        ev = [int(v) ** (1 / tau) for v in visits]                  # MCTS.py:241-243
        total = np.sum(ev)
        p = [e / total for e in ev]
        seed = int(rng.randint(0, 2 ** 31))
        u = np.random.RandomState(seed).random_sample()             # the uniform RandomState.choice will draw
        real = int(np.random.RandomState(seed).choice(n, p=p))
        assert ref_shim.choice_given_uniform(list(range(n)), p, u) == real
        assert oracle.choice_index(p, u) == real
