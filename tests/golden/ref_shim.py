"""Import harness for the Python reference (BUILD CONTAINER ONLY).

/root/reference does not exist on the GPU box; nothing under tests/ that runs
with `-m gpu`, smoke() or bench.py imports this module.  It exists so that
make_golden.py / validate_oracle.py can (a) import Checkers.py, MCTS.py and
training_pipeline.py unmodified (TensorFlow/Keras are absent, so a
`sys.modules` stand-in provides the few names imported at module scope) and
(b) drive them with a deterministic integer "hash net".
"""
import os
import sys
import types

import numpy as np

REFERENCE = os.environ.get("CKR_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE, "Checkers.py"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Make `import Checkers, MCTS, training_pipeline` work."""
    global _installed
    if _installed:
        return
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)

    class Sequence:  # noqa: D401 - stand-in
        pass

    class Callback:
        pass

    _mod("tensorflow")
    _mod("tensorflow.keras")
    _mod("tensorflow.keras.utils", Sequence=Sequence)
    _mod("tensorflow.keras.callbacks", Callback=Callback, __all__=["Callback"])
    _mod("tensorflow.keras.backend")
    _mod("tensorflow.keras.models", load_model=net_of)
    _mod("keras")
    _mod("keras.callbacks", Callback=Callback)
    _mod("keras.backend")
    import matplotlib
    matplotlib.use("Agg")
    _installed = True


def salt_of(fn):
    """File name -> hash-net salt ('...salt7...' -> 7; default 0)."""
    import re
    m = re.search(r"salt(\d+)", str(fn))
    return int(m.group(1)) if m else 0


def net_of(fn):
    """File name -> test network: '...inexact...' selects InexactNet, otherwise HashNet; salt as in salt_of."""
    return InexactNet(salt_of(fn)) if "inexact" in str(fn) else HashNet(salt_of(fn))


_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(h):
    h = np.uint64(h) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


_SQ_X = np.repeat(np.arange(8), 4)
_SQ_Y = 2 * np.tile(np.arange(4), 8) + (1 - (_SQ_X & 1))


class HashNet:
    """Deterministic integer network with the Keras `.predict` contract
    (Checkers.py:433): x[1,8,8,14] -> [p[1,512] float32, v[1,1] float32].
    Same arithmetic as ckro_hashnet (oracle/ckr_oracle.c) and the engine's
    built-in test evaluator; every output is an exact float32."""

    def __init__(self, salt=0):
        self.salt = int(salt)
        self.calls = 0

    def predict(self, x):
        self.calls += 1
        x = np.asarray(x).reshape(8, 8, 14)
        words = []
        for p in range(4):
            bits = (x[_SQ_X, _SQ_Y, p] != 0).astype(np.uint64)
            words.append(int((bits << np.arange(32, dtype=np.uint64)).sum()))
        side = 1 if x[0, 0, 4] != 0 else 0
        k = int(np.rint(np.float32(x[0, 0, 5]) * np.float32(80.0)))
        h = np.uint64(0x9E3779B9 ^ (self.salt & 0xFFFFFFFF))
        for wd in words + [side, k]:
            h = (_fmix32(h ^ np.uint64(wd)) + np.uint64(0x7F4A7C15)) & _M32
        i = np.arange(512, dtype=np.uint64)
        hv = _fmix32((h + i * np.uint64(0x9E3779B1)) & _M32)
        p = ((hv >> np.uint64(16)) + np.uint64(1)).astype(np.float32) * np.float32(1.0 / 33554432.0)
        vv = int(_fmix32(h ^ np.uint64(0xDEADBEEF)) & np.uint64(0xFFFF)) - 32768
        v = np.float32(vv) * np.float32(1.0 / 65536.0)
        return [p.reshape(1, 512), np.array([[v]], np.float32)]


class InexactNet(HashNet):
    """HashNet whose outputs are NOT exactly summable (float32 p * 0.7 + 1/3, v * 0.3): unlike HashNet's dyadic
    rationals, sums of these values round differently in float32 and float64 and depend on the order of accumulation,
    so fixtures made with it pin the search's accumulation arithmetic (MCTS.py:149-186,389-394,419-430) in either NumPy
    promotion regime.  Same arithmetic as ckro_hashnet(..., inexact = 1) and ckr_hashnet_batch(..., inexact = 1)."""

    def predict(self, x):
        p, v = HashNet.predict(self, x)
        p = (p * np.float32(0.7) + np.float32(1.0 / 3.0)).astype(np.float32)
        return [p, (v * np.float32(0.3)).astype(np.float32)]


# ---------------------------------------------------------------------------------------------------------------------
# Injected noise: the stochastic search on IDENTICAL inputs (noise_mode 1 of oracle/ckr_oracle.h and include/ckr.h).
# The noise the reference consumes -- the vector np.random.dirichlet returns (MCTS.py:107-108) and the uniform behind
# np.random.choice (MCTS.py:246) -- is an input of the search.  NoiseInjector replaces those two NumPy entry points, for the
# duration of a `with` block, by a pure function of (seed, worker, draw counter, component) that the C oracle and the HIP
# engine evaluate as well; everything the reference does WITH the noise (the float32 / float64 mixing of the prior, PUCT,
# argmax, the temperature weights, their sum and normalisation) stays the reference's own code.
def noise_hash(seed, worker, ctr, lane):
    """ckro_noise_hash / the engine's noise_hash: five rounds of murmur3's fmix32."""
    def fm(h):
        h &= 0xFFFFFFFF
        h ^= h >> 16; h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13; h = (h * 0xC2B2AE35) & 0xFFFFFFFF
        return h ^ (h >> 16)
    k = 0x7F4A7C15
    h = (fm((seed & 0xFFFFFFFF) ^ 0x9E3779B9) + k) & 0xFFFFFFFF
    h = (fm(h ^ ((seed >> 32) & 0xFFFFFFFF)) + k) & 0xFFFFFFFF
    h = (fm(h ^ (worker & 0xFFFFFFFF)) + k) & 0xFFFFFFFF
    h = (fm(h ^ (ctr & 0xFFFFFFFF)) + k) & 0xFFFFFFFF
    return fm(h ^ (lane & 0xFFFFFFFF))


def noise_dirichlet(seed, worker, ctr, n):
    g = np.array([(noise_hash(seed, worker, ctr, i) >> 8) + 1 for i in range(n)], np.float64)
    return g / np.float64(sum(int(x) for x in g))


def noise_uniform(seed, worker, ctr):
    return noise_hash(seed, worker, ctr, 0xFFFFFFFF) / 4294967296.0


def choice_given_uniform(a, p, u):
    """RandomState.choice(a, p=p) with the uniform it would draw handed in (numpy/random/mtrand.pyx: cdf = p.cumsum();
    cdf /= cdf[-1]; idx = cdf.searchsorted(uniform, side='right')) -- NumPy's own array operations on NumPy's own types.
    tests/test_noise_cpu.py checks it against the real np.random.RandomState.choice draw by draw."""
    p = np.array(p, dtype=np.float64)
    cdf = p.cumsum()
    cdf /= cdf[-1]
    idx = int(cdf.searchsorted(u, side="right"))
    return a[idx]


class NoiseInjector:
    """with NoiseInjector(seed, worker): ... -- np.random.dirichlet and np.random.choice read the injected noise.
    The draw counter starts at 0 and advances with every Dirichlet draw that enters a score (MCTS.epsilon != 0: with
    epsilon == 0 the reference still calls np.random.dirichlet and multiplies the result by 0) and with every pick."""

    def __init__(self, seed, worker=0):
        self.seed, self.worker, self.ctr = int(seed), int(worker), 0
        self.n_dirichlet = self.n_choice = 0

    def _dirichlet(self, alpha, size=None):
        assert size is None
        n = len(alpha)
        eps = sys.modules["MCTS"].MCTS.epsilon
        if eps == 0:
            return np.full(n, 1.0 / n)
        self.n_dirichlet += 1
        d = noise_dirichlet(self.seed, self.worker, self.ctr, n)
        self.ctr += 1
        return d

    def _choice(self, a, size=None, replace=True, p=None):
        assert size is None and p is not None
        self.n_choice += 1
        u = noise_uniform(self.seed, self.worker, self.ctr)
        self.ctr += 1
        return choice_given_uniform(a, p, u)

    def __enter__(self):
        self._real = (np.random.dirichlet, np.random.choice)
        np.random.dirichlet, np.random.choice = self._dirichlet, self._choice
        return self

    def __exit__(self, *exc):
        np.random.dirichlet, np.random.choice = self._real
        return False
