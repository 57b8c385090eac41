"""Data-format codec between the reference's (15, 8, 8) float64 state planes
(Checkers.py:37-49) and the engine's 16-byte bitboard records (include/ckr.h).

Pure layout conversion (bit <-> plane cell); no game logic lives here: legal
masks and the draw plane are produced by the HIP kernels and only *unpacked*
by this module.
"""
import numpy as np

# square index s = 4*x + (y >> 1) over the playable squares (x % 2 != y % 2)
SQ_X = np.repeat(np.arange(8), 4)
SQ_Y = 2 * np.tile(np.arange(4), 8) + (1 - (SQ_X & 1))
SQ_FLAT = SQ_X * 8 + SQ_Y                      # index into a flattened 8x8 plane

HIST_MAX = 0x1FFF


def make_meta(side, mover, action=0, has_action=0, r=0, hist=1):
    side, mover, action = np.asarray(side, np.uint32), np.asarray(mover, np.uint32), np.asarray(action, np.uint32)
    has_action, r = np.asarray(has_action, np.uint32), np.asarray(r, np.uint32)
    hist = np.minimum(np.asarray(hist, np.uint32), HIST_MAX)
    return ((side & 1) | ((mover & 1) << 1) | ((action & 0x1FF) << 2) | ((has_action & 1) << 11)
            | ((r & 0x7F) << 12) | ((hist & 0x1FFF) << 19)).astype(np.uint32)


def meta_side(m):   return np.asarray(m, np.uint32) & 1
def meta_mover(m):  return (np.asarray(m, np.uint32) >> 1) & 1
def meta_action(m): return (np.asarray(m, np.uint32) >> 2) & 0x1FF
def meta_hasact(m): return (np.asarray(m, np.uint32) >> 11) & 1
def meta_r(m):      return (np.asarray(m, np.uint32) >> 12) & 0x7F
def meta_hist(m):   return (np.asarray(m, np.uint32) >> 19) & 0x1FFF


def status_outcome(s): return np.asarray(s, np.uint32) & 3
def status_jump(s):    return (np.asarray(s, np.uint32) >> 2) & 1
def status_nlegal(s):  return (np.asarray(s, np.uint32) >> 8) & 0xFF
def status_drawk(s):   return (np.asarray(s, np.uint32) >> 16) & 0xFF


OUTCOME_NAMES = {0: None, 1: "player1_wins", 2: "player2_wins", 3: "draw"}


def _pack_bits(plane_flat):
    """[N, 64] 0/1 -> uint32[N] over the 32 playable squares."""
    bits = (plane_flat[:, SQ_FLAT] != 0).astype(np.uint64)
    return (bits << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)


def planes_to_boards(states, r=0, hist=1, mover=None):
    """states [N,15,8,8] (or [15,8,8]) -> uint32 [N,4] board records.

    r / hist / mover are not recoverable from the planes (the reference keeps
    them implicitly in its python history list); callers that track them pass
    them in.  mover defaults to the opponent of the side to move.
    """
    st = np.asarray(states)
    if st.ndim == 3:
        st = st[None]
    n = st.shape[0]
    flat = st.reshape(n, 15, 64)
    men1, k1, men2, k2 = (_pack_bits(flat[:, i]) for i in range(4))
    side = (flat[:, 4, 0] != 0).astype(np.uint32)
    plane = flat[:, 14, 0].astype(np.int64)
    x, y = flat[:, 14, 1].astype(np.int64), flat[:, 14, 2].astype(np.int64)
    has = (plane >= 6).astype(np.uint32)
    action = np.where(has == 1, (plane - 6) * 64 + 8 * x + y, 0).astype(np.uint32)
    if mover is None:
        mover = 1 - side
    out = np.empty((n, 4), np.uint32)
    out[:, 0] = men1 | k1
    out[:, 1] = men2 | k2
    out[:, 2] = k1 | k2
    out[:, 3] = make_meta(side, mover, action, has, r, hist)
    return out


def _unpack_bits(words):
    """uint32[N] -> float64 [N, 64] plane cells (playable squares only)."""
    w = np.asarray(words, np.uint32)
    out = np.zeros((w.shape[0], 64), np.float64)
    out[:, SQ_FLAT] = (w[:, None] >> np.arange(32, dtype=np.uint32)) & 1
    return out


def records_to_planes(boards, masks, status):
    """Board records + legal-mask words + status -> float64 [N,15,8,8] exactly
    as the reference leaves a state after _check_moves / determine_outcome
    (planes 5-13 populated in place, Checkers.py:137-198,338-360)."""
    b = np.asarray(boards, np.uint32).reshape(-1, 4)
    m = np.asarray(masks, np.uint32).reshape(-1, 8)
    s = np.asarray(status, np.uint32).reshape(-1)
    n = b.shape[0]
    out = np.zeros((n, 15, 64), np.float64)
    # the twelve bit planes (men / kings of both sides, the eight legal-move masks) in one scatter: bit s of a word = square s
    words = np.empty((n, 12), np.uint32)
    words[:, 0] = b[:, 0] & ~b[:, 2]
    words[:, 1] = b[:, 0] & b[:, 2]
    words[:, 2] = b[:, 1] & ~b[:, 2]
    words[:, 3] = b[:, 1] & b[:, 2]
    words[:, 4:] = m
    bits = np.unpackbits(words.view(np.uint8).reshape(n, 12, 4), axis=2, bitorder="little")            # [n, 12, 32] 0 / 1
    out[:, np.array((0, 1, 2, 3, 6, 7, 8, 9, 10, 11, 12, 13))[:, None], np.asarray(SQ_FLAT)[None, :]] = bits
    out[:, 4] = meta_side(b[:, 3])[:, None]
    out[:, 5] = (status_drawk(s).astype(np.float64) / 80)[:, None]
    a = meta_action(b[:, 3]).astype(np.int64)
    has = meta_hasact(b[:, 3]) == 1
    out[:, 14, 0] = np.where(has, (a >> 6) + 6, 0)
    out[:, 14, 1] = np.where(has, (a >> 3) & 7, 0)
    out[:, 14, 2] = np.where(has, a & 7, 0)
    return out.reshape(n, 15, 8, 8)


def pi_planes(actions, visits):
    """training_pipeline._create_prob_planes (:421-437): pi[layer,x,y] = N/sum N
    in float64, from the engine's (action code, visit count) pairs."""
    pi = np.zeros(512, np.float64)
    if len(actions):
        pi[np.asarray(actions, np.int64)] = np.asarray(visits, np.float64)
        pi /= np.sum(pi)
    return pi.reshape(8, 8, 8)
