"""Helpers that run the imported Python reference and express its results in
the build's record format (BUILD CONTAINER ONLY; see ref_shim.py)."""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
for p in (_ROOT, os.path.join(_ROOT, "oracle"), _HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import ref_shim  # noqa: E402

ref_shim.install()
import Checkers as ref_checkers  # noqa: E402

import checkers_mcts_amd.codec as codec  # noqa: E402

OUTCOME_CODE = {None: 0, "player1_wins": 1, "player2_wins": 2, "draw": 3}


def r_of(history):
    """Plies since the last man move / capture, by definition from the
    reference's draw scan (Checkers.py:335-343): number of immediately
    preceding states with the same piece count and men planes."""
    cur = history[-1]
    cnt = cur[0:4].sum()
    r = 0
    for prev in reversed(history[:-1]):
        if prev[0:4].sum() != cnt or not ((prev[0] == cur[0]).all() and (prev[2] == cur[2]).all()):
            break
        r += 1
        if r >= 127:
            break
    return r


def record_of(history, mover=None):
    """Board record of history[-1] (uint32[4])."""
    st = history[-1]
    if mover is None:
        mover = int(history[-2][4, 0, 0]) if len(history) > 1 else 1 - int(st[4, 0, 0])
    return codec.planes_to_boards(st, r=r_of(history), hist=len(history), mover=mover)[0]


def pack_planes(planes8):
    """state[6:14] -> uint32[8] mask words."""
    flat = np.asarray(planes8).reshape(8, 64)
    bits = (flat[:, codec.SQ_FLAT] != 0).astype(np.uint64)
    return (bits << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)


def ref_analyse(env, history):
    """Run the reference on history[-1]: returns (mask words, status word,
    children records in the reference's list order)."""
    state = history[-1]
    kids = env._check_moves(history)
    done, outcome = env.determine_outcome(history, legal_moves=kids)
    mask = pack_planes(state[6:14])
    k = int(round(float(state[5, 0, 0]) * 80))
    jump = int(any(int(c[14, 0, 0]) >= 10 for c in kids))
    status = OUTCOME_CODE[outcome] | (jump << 2) | (len(kids) << 8) | (k << 16)
    side = int(state[4, 0, 0])
    recs = np.zeros((len(kids), 4), np.uint32)
    for i, c in enumerate(kids):
        recs[i] = record_of(history + [c], mover=side)
    return mask, np.uint32(status), recs


def new_env(state=None):
    env = ref_checkers.Checkers()
    if state is not None:
        env.state = state
        env.history = [state]
        env.legal_next_states = env.get_legal_next_states(env.history)
        env.move_count = 0
        env.done = False
        env.outcome = None
    return env


def synthetic_state(rng, max_pieces=12, king_frac=None):
    """Random legal-looking position (SURVEY.md cfg2 generator): per side
    0..max_pieces pieces on random dark squares, men never on their own far
    row, random side to move."""
    s = np.zeros((15, 8, 8))
    squares = list(rng.permutation(32))
    kf = rng.rand() if king_frac is None else king_frac
    for side in (0, 1):
        n = rng.randint(0, max_pieces + 1)
        for _ in range(n):
            if not squares:
                break
            sq = squares.pop()
            x, y = int(codec.SQ_X[sq]), int(codec.SQ_Y[sq])
            king = rng.rand() < kf
            if not king and ((side == 0 and x == 7) or (side == 1 and x == 0)):
                king = True
            s[side * 2 + (1 if king else 0), x, y] = 1
    s[4] = rng.randint(0, 2)
    return s
