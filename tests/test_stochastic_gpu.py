"""GPU: the reference-default stochastic paths against NumPy, distributionally.

MCTS.select_child mixes Dirichlet(alpha * 1_b) noise into the priors on every call (MCTS.py:104-111,
np.random.dirichlet) and MCTS.best_child samples the move with p ~ N^(1/tau) while training
(MCTS.py:240-246, np.random.choice).  NumPy's MT19937 stream cannot be matched bit for bit by the
engine's Philox generator, so these tests drive the engine's OWN device functions through the
ckr_probe_* entry points and compare the output distributions with the exact ones and with NumPy
samples: moments, Kolmogorov-Smirnov on the marginals (Dirichlet marginals are Beta(alpha,
(b-1) alpha)), chi-square on the pick frequencies.  Seeds are fixed: the thresholds are not flaky."""
import ctypes as C

import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from checkers_mcts_amd import _lib
    return _lib.load()


def dirichlet(L, alpha, b, samples, seed):
    from checkers_mcts_amd import _lib
    out = np.zeros((samples, b), np.float64)
    _lib.check(L.ckr_probe_dirichlet(alpha, b, samples, seed, out.ctypes.data))
    return out


def picks(L, visits, tau, samples, seed):
    from checkers_mcts_amd import _lib
    v = np.ascontiguousarray(visits, np.int32)
    out = np.zeros(samples, np.int32)
    _lib.check(L.ckr_probe_temperature(v.ctypes.data, len(v), tau, samples, seed, out.ctypes.data))
    return out


@pytest.mark.parametrize("alpha", [0.3, 1.0, 2.5])
@pytest.mark.parametrize("b", [2, 7, 30])
def test_dirichlet_noise_matches_numpy_distribution(L, alpha, b):
    n = 40000
    d = dirichlet(L, alpha, b, n, seed=1234 + b)
    assert np.all(d >= 0) and np.allclose(d.sum(1), 1.0, atol=1e-12)
    # exact moments of Dirichlet(alpha 1_b): mean 1/b, var (b-1) / (b^2 (b alpha + 1)), cov -1 / (b^2 (b alpha + 1))
    var = (b - 1) / (b * b * (b * alpha + 1.0))
    assert np.allclose(d.mean(0), 1.0 / b, atol=5 * np.sqrt(var / n))
    assert np.allclose(d.var(0), var, rtol=0.06)
    cov01 = np.cov(d[:, 0], d[:, 1])[0, 1]
    assert abs(cov01 - (-1.0 / (b * b * (b * alpha + 1.0)))) < 0.1 * var + 1e-5
    # every marginal is Beta(alpha, (b - 1) alpha); components of different lanes must not be correlated copies
    for j in (0, b // 2, b - 1):
        assert stats.kstest(d[:, j], "beta", args=(alpha, (b - 1) * alpha)).pvalue > 1e-4
    # two-sample test against NumPy's own generator (what the reference calls)
    ref = np.random.default_rng(7).dirichlet([alpha] * b, size=n)
    assert stats.ks_2samp(d[:, 0], ref[:, 0]).pvalue > 1e-4
    assert stats.ks_2samp(d.max(1), ref.max(1)).pvalue > 1e-4            # a statistic of the joint law (what argmax sees)
    # different draws (counter) and different workers are independent streams
    assert abs(np.corrcoef(d[:-1, 0], d[1:, 0])[0, 1]) < 0.02
    assert abs(np.corrcoef(d[:-1024, 0], d[1024:, 0])[0, 1]) < 0.02


def test_dirichlet_streams_are_seeded(L):
    a = dirichlet(L, 1.0, 5, 256, seed=1)
    assert (a == dirichlet(L, 1.0, 5, 256, seed=1)).all()
    assert (a != dirichlet(L, 1.0, 5, 256, seed=2)).any()


@pytest.mark.parametrize("tau", [1.0, 0.5, 0.1])
def test_temperature_pick_frequencies(L, tau):
    """p_i = N_i^(1/tau) / sum over the root's children in tree order (MCTS.py:240-242,246)."""
    n = 60000
    for visits in ([50, 30, 15, 5], [1, 99], [0, 40, 0, 35, 25, 0], list(range(1, 19)), [7] * 30):
        v = np.array(visits, np.float64)
        p = v ** (1.0 / tau)
        p = p / p.sum()
        got = picks(L, visits, tau, n, seed=99)
        assert got.min() >= 0 and got.max() < len(visits)
        cnt = np.bincount(got, minlength=len(visits)).astype(np.float64)
        assert (cnt[p == 0] == 0).all()                                   # a child without visits is never chosen
        keep = p * n >= 5                                                 # chi-square needs expected counts >= 5
        if keep.sum() >= 2:
            rest_o, rest_e = cnt[~keep].sum(), p[~keep].sum() * n
            obs = np.append(cnt[keep], rest_o) if rest_e > 0 else cnt[keep]
            exp = np.append(p[keep] * n, rest_e) if rest_e > 0 else p[keep] * n
            assert stats.chisquare(obs, exp * obs.sum() / exp.sum()).pvalue > 1e-4
        else:
            assert cnt[np.argmax(p)] >= 0.999 * n
        # same test on NumPy's choice: the reference's sampler passes it too (sanity of the threshold)
        ref = np.bincount(np.random.default_rng(3).choice(len(visits), size=n, p=p), minlength=len(visits))
        assert np.abs(cnt / n - ref / n).max() < 0.01


def test_temperature_pick_survives_exponents_beyond_float32(L):
    """The reference evaluates N ** (1 / tau) in float64 (MCTS.py:240-246), finite up to 1e308: a finer decay than the
    recorded runs' (tau 0.04, 0.02) or visit counts in the thousands at tau 0.1 put N^(1/tau) beyond float32's 3.4e38.
    The device pick works on (N / Nmax)^(1/tau): same distribution, largest weight 1."""
    n = 20000
    for visits, tau in (([200, 190, 40, 3], 0.02), ([40, 39, 1], 0.04), ([7080, 7000, 6000, 10], 0.1), ([1600, 1599, 2], 0.01)):
        v = np.array(visits, np.float64)
        p = (v / v.max()) ** (1.0 / tau)
        p = p / p.sum()
        got = picks(L, visits, tau, n, seed=5)
        cnt = np.bincount(got, minlength=len(visits)).astype(np.float64)
        assert np.abs(cnt / n - p).max() < 0.012, (visits, tau, cnt / n, p)
        assert cnt[-1] < 0.01 * n                                         # never the fall-through to the last child


def test_temperature_schedule_matches_reference_arithmetic(L):
    """tau -= TEMPERATURE_DECAY once move_count > TEMP_DECAY_DELAY, snapped to 0 by np.isclose; the
    class attribute is never reset (MCTS.py:243-245)."""
    from checkers_mcts_amd import _lib
    for tau0, decay, delay in ((1.0, 0.1, 10), (1.0, 0.3, 0), (0.5, 0.05, 3), (1.0, 0.0, 5)):
        moves = 40
        out = np.zeros(moves, np.float64)
        _lib.check(L.ckr_probe_tau_schedule(tau0, decay, delay, moves, out.ctypes.data))
        tau, want = tau0, []
        for move_count in range(moves):
            want.append(tau)
            if tau > 0:                                                   # training: the sampling branch runs
                if move_count > delay:
                    tau -= decay
                    if np.isclose(tau, 0):
                        tau = 0
        assert (out == np.array(want)).all()
        if (tau0, decay) == (1.0, 0.1):
            assert out[-1] == 0.0 and (out >= 0).all()                    # 0.1 - 0.1 leaves 2.8e-17: snapped to exactly 0
