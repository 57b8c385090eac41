"""Batched Checkers rules on the GPU: torch-tensor wrappers around the K1/K2/K8
kernels of libckr.so.  All tensors live in HBM on a `cuda` (HIP) device; torch
is used for memory and streams only.

The reference counterparts are Checkers._check_moves / determine_outcome /
predict (Checkers.py:94-200, 306-364, 425-438).
"""
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _boards(t):
    if not (t.is_cuda and t.dtype == torch.int32 and t.dim() == 2 and t.shape[1] == 4 and t.is_contiguous()):
        raise ValueError("boards must be a contiguous cuda int32 tensor of shape [N, 4]")
    return t


def boards_to_device(boards_u32, device="cuda"):
    """numpy uint32 [N,4] -> cuda int32 [N,4] (bit pattern preserved)."""
    import numpy as np
    return torch.from_numpy(np.ascontiguousarray(boards_u32, np.uint32).view(np.int32).reshape(-1, 4)).to(device)


def movegen(boards):
    """[N,4] boards -> (mask int32 [N,8], status int32 [N])."""
    L = _lib.load()
    b = _boards(boards)
    n = b.shape[0]
    mask = torch.empty((n, 8), dtype=torch.int32, device=b.device)
    status = torch.empty((n,), dtype=torch.int32, device=b.device)
    _lib.check(L.ckr_movegen_batch(b.data_ptr(), n, mask.data_ptr(), status.data_ptr(), _stream()))
    return mask, status


def children(boards):
    """[N,4] -> (children int32 [N,48,4] in the reference's list order, count int32 [N])."""
    L = _lib.load()
    b = _boards(boards)
    n = b.shape[0]
    kids = torch.zeros((n, _lib.MAX_CHILDREN, 4), dtype=torch.int32, device=b.device)
    cnt = torch.empty((n,), dtype=torch.int32, device=b.device)
    _lib.check(L.ckr_children_batch(b.data_ptr(), n, kids.data_ptr(), cnt.data_ptr(), _stream()))
    return kids, cnt


def children_packed(boards, capacity=None):
    """[N,4] -> (packed int32 [total,4], offset int64 [N], count int32 [N]): the successor list of position i is
    packed[offset[i] : offset[i] + count[i]], in the reference's list order (include/ckr.h, ckr_children_packed: dense output,
    the lists back to back in position order -- offset is the exclusive running sum of count).  capacity: records to provide at first (default 8 per position);
    the call is repeated with the exact size when the positions have more."""
    L = _lib.load()
    b = _boards(boards)
    n = b.shape[0]
    offset = torch.empty((n,), dtype=torch.int64, device=b.device)
    cnt = torch.empty((n,), dtype=torch.int32, device=b.device)
    total = torch.zeros((1,), dtype=torch.int64, device=b.device)
    scratch = torch.empty(((n + 255) // 256 * 12 + 16,), dtype=torch.uint8, device=b.device)      # CKR_CHILDREN_PACKED_SCRATCH(n)
    cap = int(capacity if capacity is not None else 8 * n)
    while True:
        packed = torch.empty((max(cap, 1), 4), dtype=torch.int32, device=b.device)
        _lib.check(L.ckr_children_packed(b.data_ptr(), n, packed.data_ptr(), cap, offset.data_ptr(), cnt.data_ptr(), total.data_ptr(),
                                         scratch.data_ptr(), _stream()))
        t = int(total.item())
        if t <= cap:
            return packed[:t], offset, cnt
        cap = t


def features(boards):
    """[N,4] -> float32 [N,8,8,14] NHWC network input."""
    L = _lib.load()
    b = _boards(boards)
    n = b.shape[0]
    x = torch.empty((n, 8, 8, 14), dtype=torch.float32, device=b.device)
    _lib.check(L.ckr_features_batch(b.data_ptr(), n, x.data_ptr(), _stream()))
    return x


def mask_renorm(boards, p):
    """Checkers.predict post-processing on raw p [N,512] float32."""
    L = _lib.load()
    b = _boards(boards)
    n = b.shape[0]
    if not (p.is_cuda and p.dtype == torch.float32 and p.shape == (n, 512) and p.is_contiguous()):
        raise ValueError("p must be a contiguous cuda float32 tensor of shape [N, 512]")
    out = torch.empty_like(p)
    _lib.check(L.ckr_mask_renorm_batch(b.data_ptr(), n, p.data_ptr(), out.data_ptr(), _stream()))
    return out


def hashnet(x, salt=0, inexact=False):
    """Deterministic integer test network: x [N,8,8,14] float32 -> (p [N,512], v [N]).  inexact: the variant whose
    outputs do not sum exactly (tests/golden/ref_shim.InexactNet)."""
    L = _lib.load()
    n = x.shape[0]
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.numel() == n * 896):
        raise ValueError("x must be a contiguous cuda float32 tensor of shape [N, 8, 8, 14]")
    p = torch.empty((n, 512), dtype=torch.float32, device=x.device)
    v = torch.empty((n,), dtype=torch.float32, device=x.device)
    _lib.check(L.ckr_hashnet_batch(x.data_ptr(), n, int(salt) & 0xFFFFFFFF, int(bool(inexact)), p.data_ptr(), v.data_ptr(), _stream()))
    return p, v
