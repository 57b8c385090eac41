"""GPU: the HIP rules kernels (through the C-ABI) against the oracle and the
committed golden vectors.  Bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import torch
    assert torch.cuda.is_available()
    from checkers_mcts_amd import rules
    return rules


def _np(t):
    return t.cpu().numpy().view(np.uint32)


def random_boards(n, seed):
    """Synthetic positions (SURVEY cfg2 generator) as records, numpy-vectorised."""
    rng = np.random.RandomState(seed)
    out = np.zeros((n, 4), np.uint32)
    for i in range(n):
        sq = rng.permutation(32)
        n1, n2 = rng.randint(0, 13), rng.randint(0, 13)
        kf = rng.rand()
        p1 = p2 = k = 0
        for j, s in enumerate(sq[:n1 + n2]):
            side = 0 if j < n1 else 1
            king = rng.rand() < kf or (side == 0 and s >= 28) or (side == 1 and s < 4)
            if side == 0:
                p1 |= 1 << int(s)
            else:
                p2 |= 1 << int(s)
            if king:
                k |= 1 << int(s)
        side = rng.randint(0, 2)
        hist = int(rng.choice([1, 5, 79, 80, 81, 200]))
        r = int(rng.randint(0, min(hist, 80)))
        out[i] = (p1, p2, k, side | ((1 - side) << 1) | (r << 12) | (hist << 19))
    return out


def test_golden_rules(R, oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "rules_v1.npz"))
    b = R.boards_to_device(g["boards"])
    mask, status = R.movegen(b)
    assert (_np(mask) == g["masks"]).all()
    assert (_np(status) == g["status"]).all()
    kids, cnt = R.children(b)
    kids, cnt = _np(kids), cnt.cpu().numpy()
    off = g["child_off"]
    assert (cnt == np.diff(off)).all()
    for i in range(len(off) - 1):
        assert (kids[i, :cnt[i]] == g["children"][off[i]:off[i + 1]]).all()


def test_random_boards_vs_oracle(R, oracle):
    boards = random_boards(20000, 11)
    b = R.boards_to_device(boards)
    mask, status = R.movegen(b)
    omask, ostatus = oracle.movegen(boards)
    assert (_np(mask) == omask).all() and (_np(status) == ostatus).all()
    kids, cnt = R.children(b)
    kids, cnt = _np(kids), cnt.cpu().numpy()
    for i in range(0, len(boards), 7):
        ok = oracle.children(boards[i])
        assert cnt[i] == len(ok) and (kids[i, :cnt[i]] == ok).all()


def test_packed_successor_lists_equal_the_slot_lists(R, golden_dir):
    """ckr_children_packed (dense output: lists back to back, offset + count per position) against ckr_children_batch on the 16 919 golden
    positions (whose slot lists are pinned to the reference above) and on 20 000 synthetic ones: every list record for record, offsets =
    the exclusive running sum of the counts; a buffer too small is reported, not overrun."""
    import torch
    g = np.load(os.path.join(golden_dir, "rules_v1.npz"))
    for boards in (g["boards"], random_boards(20000, 12), g["boards"][:70], g["boards"][:1]):
        b = R.boards_to_device(boards)
        kids, cnt = R.children(b)
        packed, off, pc = R.children_packed(b)
        kids, cnt, packed, off, pc = _np(kids), cnt.cpu().numpy(), _np(packed), off.cpu().numpy(), pc.cpu().numpy()
        assert (pc == cnt).all() and len(packed) == cnt.sum()
        assert (off == np.concatenate([[0], np.cumsum(cnt.astype(np.int64))[:-1]])).all()        # a CSR: position order, no gaps
        flat = np.concatenate([kids[i, :cnt[i]] for i in range(len(boards))])
        gather = np.concatenate([np.arange(off[i], off[i] + cnt[i]) for i in range(len(boards))])
        assert (packed[gather] == flat).all()
    b = R.boards_to_device(g["boards"][:4096])
    total_needed = int(R.children(b)[1].sum().item())
    small, off2, pc2 = R.children_packed(b, capacity=total_needed // 3)                        # first call too small: repeated with the exact size
    assert len(small) == total_needed and (_np(small)[off2.cpu().numpy()[0]:][:1] == _np(R.children(b)[0])[0, :1]).all()
    empty, off0, pc0 = R.children_packed(torch.zeros((0, 4), dtype=torch.int32, device="cuda"))
    assert len(empty) == 0 and len(off0) == 0


def playout_positions(oracle, n, seed):
    """Every position of seeded uniform-random playouts from the initial position, until n are
    collected; the oracle supplies rules and successors (it is the checker: pinned to the reference on 10^6 positions)."""
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        b = oracle.initial_board()
        while True:
            out.append(b)
            _, status = oracle.movegen(b[None])
            if status[0] & 3:
                break
            kids = oracle.children(b)
            b = kids[rng.randint(len(kids))]
    return np.array(out[:n], np.uint32)


def test_cfg2_65536_boards(R, oracle):
    """BASELINE config 2 with SURVEY 8(d)'s composition: 49 152 positions from uniform-random playouts + 16 384
    synthetic boards (0-12 pieces per side, random king fraction, random side / draw counters), seed 20260929;
    legal-move masks + status bit-exact."""
    play = playout_positions(oracle, 49152, 20260929)
    assert len(play) == 49152 and (play[0] == oracle.initial_board()).all()
    boards = np.concatenate([play, random_boards(16384, 20260929)])
    assert boards.shape == (65536, 4)
    mask, status = R.movegen(R.boards_to_device(boards))
    omask, ostatus = oracle.movegen(boards)
    assert (_np(mask) == omask).all() and (_np(status) == ostatus).all()
    # size-independent property: popcount(mask) == legal count in the status word
    pop = np.unpackbits(_np(mask).view(np.uint8), axis=1).sum(1)
    assert (pop == ((_np(status) >> 8) & 0xFF)).all()


def test_empty_and_ragged(R):
    import torch
    e = torch.empty((0, 4), dtype=torch.int32, device="cuda")
    m, s = R.movegen(e)
    assert m.shape == (0, 8) and s.shape == (0,)
    for n in (1, 63, 65, 257):
        boards = random_boards(n, n)
        m, s = R.movegen(R.boards_to_device(boards))
        assert m.shape == (n, 8)
    with pytest.raises(ValueError):
        R.movegen(torch.zeros((3, 5), dtype=torch.int32, device="cuda"))


def test_features_hashnet_renorm(R, oracle, golden_dir):
    import torch
    g = np.load(os.path.join(golden_dir, "predict_v1.npz"))
    b = R.boards_to_device(g["boards"])
    out = R.mask_renorm(b, torch.from_numpy(g["raw_p"]).cuda())
    assert (out.cpu().numpy().view(np.uint32) == g["planes"].view(np.uint32)).all()
    x = R.features(b).cpu().numpy()
    for i in range(0, len(g["boards"]), 5):
        assert (x[i] == oracle.features(g["boards"][i])).all()
    h = np.load(os.path.join(golden_dir, "hashnet_v1.npz"))
    for salt in np.unique(h["salt"]):
        sel = h["salt"] == salt
        p, v = R.hashnet(torch.from_numpy(h["x"][sel]).cuda().reshape(-1, 8, 8, 14), int(salt))
        assert (p.cpu().numpy() == h["p"][sel]).all() and (v.cpu().numpy() == h["v"][sel]).all()


def test_large_batch_properties(R):
    """2^22 boards (tiled): determinism and agreement between K1 and K2 counts."""
    import torch
    base = R.boards_to_device(random_boards(4096, 5))
    big = base.repeat(1024, 1).contiguous()
    m, s = R.movegen(big)
    m0, s0 = R.movegen(base)
    assert torch.equal(m.view(1024, 4096, 8)[777], m0) and torch.equal(s.view(1024, 4096)[1023], s0)
    kids, cnt = R.children(base)
    assert torch.equal(cnt, (s0 >> 8) & 0xFF)
