"""Host-side launch plumbing: torch tensors in, C-ABI calls on the current HIP stream, torch tensors out.

torch is used for device memory and stream ownership only; every number is produced by libgabo_hip.so.
Inputs living on the CPU are moved to the GPU, computed there and moved back (the reference's examples build CPU
tensors); without a GPU every call raises - there is no CPU implementation in this package.
"""
import torch

from . import _lib

_check_errors = True


def set_error_checking(flag):
    """Device-side data errors (non-SPD input) are read back after each call when enabled (costs a stream sync).
    The reference raises from torch.cholesky in that case; disable inside latency-critical loops."""
    global _check_errors
    _check_errors = bool(flag)


def _device_for(*tensors):
    for t in tensors:
        if t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("gabotorch_amd needs an MI355X: no HIP device is visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def _flatten_batch(x, tail_dims):
    """(..., tail) -> (tensor2d_or_3d contiguous, batch, batch_stride_in_elements).  A batch produced by
    `.expand()` (all batch strides 0) is passed as ONE shared set with batch stride 0."""
    bshape = x.shape[:-tail_dims]
    nb = 1
    for s in bshape:
        nb *= s
    if len(bshape) > 0 and nb > 1 and all(st == 0 for st, sz in zip(x.stride()[:len(bshape)], bshape) if sz > 1):
        base = x[(0,) * len(bshape)].contiguous()
        return base, nb, 0
    xc = x.contiguous()
    tail = 1
    for s in x.shape[-tail_dims:]:
        tail *= s
    return xc, nb, tail


def _prep(x, device):
    if x.dtype != torch.float64:
        x = x.double()
    return x.to(device)


def spd_ai_pairwise(x1, x2, beta=1.0, mode=_lib.GABO_OUT_GAUSSIAN, symmetric=False):
    """x1 (..., N1, d_vec), x2 (..., N2, d_vec) Mandel vectors -> (..., N1, N2) on x1's device."""
    lib = _lib.load()
    if x1.shape[:-2] != x2.shape[:-2] or x1.shape[-1] != x2.shape[-1]:
        raise RuntimeError(f"batch/feature shapes differ: {tuple(x1.shape)} vs {tuple(x2.shape)} (no broadcasting, as in the reference)")
    out_device = x1.device
    dev = _device_for(x1, x2)
    a, b = _prep(x1, dev), _prep(x2, dev)
    dv = a.shape[-1]
    d = int((-1.0 + (1.0 + 8.0 * dv) ** 0.5) / 2.0)
    if d * (d + 1) // 2 != dv:
        raise RuntimeError(f"last dimension {dv} is not d(d+1)/2")
    n1, n2 = a.shape[-2], b.shape[-2]
    bshape = a.shape[:-2]
    a2, nb, s1 = _flatten_batch(a, 2)
    b2, _, s2 = _flatten_batch(b, 2)
    out = torch.empty(bshape + (n1, n2), dtype=torch.float64, device=dev)
    if out.numel() == 0:
        return out.to(out_device)
    wsb = lib.gabo_spd_ai_workspace_bytes(nb, n1, n2, d)
    ws = torch.empty(max(wsb // 8, 1), dtype=torch.float64, device=dev)
    status = torch.zeros(2, dtype=torch.int32, device=dev)
    flags = int(mode) | (_lib.GABO_SYMMETRIC if symmetric else 0)
    with torch.cuda.device(dev):
        rc = lib.gabo_spd_ai_pairwise(a2.data_ptr(), b2.data_ptr(), out.data_ptr(), nb, n1, n2, d, s1, s2, float(beta), flags,
                                      ws.data_ptr(), wsb, status.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "gabo_spd_ai_pairwise")
    if _check_errors:
        st = status.tolist()
        if st[0] != 0:
            raise RuntimeError(f"gabo_spd_ai_pairwise: input matrix #{st[1]} is not positive definite "
                               "(Cholesky pivot <= 0)")
    return out.to(out_device)


def sphere_pairwise(x1, x2, beta=1.0, mode=_lib.GABO_OUT_GAUSSIAN, diag=False):
    """x1 (..., N1, dim), x2 (..., N2, dim) -> (..., N1, N2)   [diag: (..., N, 1)]."""
    lib = _lib.load()
    if x1.shape[:-2] != x2.shape[:-2] or x1.shape[-1] != x2.shape[-1]:
        raise RuntimeError(f"batch/feature shapes differ: {tuple(x1.shape)} vs {tuple(x2.shape)}")
    out_device = x1.device
    dev = _device_for(x1, x2)
    a, b = _prep(x1, dev), _prep(x2, dev)
    dim = a.shape[-1]
    n1, n2 = a.shape[-2], b.shape[-2]
    if diag and n1 != n2:
        raise RuntimeError("diag=True needs x1 and x2 of the same length")
    bshape = a.shape[:-2]
    a2, nb, s1 = _flatten_batch(a, 2)
    b2, _, s2 = _flatten_batch(b, 2)
    out = torch.empty(bshape + ((n1, 1) if diag else (n1, n2)), dtype=torch.float64, device=dev)
    if out.numel() == 0:
        return out.to(out_device)
    with torch.cuda.device(dev):
        rc = lib.gabo_sphere_pairwise(a2.data_ptr(), b2.data_ptr(), out.data_ptr(), nb, n1, n2, dim, s1, s2, float(beta),
                                      int(mode), 1 if diag else 0, _stream_ptr(dev))
    _lib.check(rc, "gabo_sphere_pairwise")
    return out.to(out_device)


def mandel_to_matrix(vec):
    """(..., d_vec) -> (..., d, d)."""
    lib = _lib.load()
    out_device = vec.device
    dev = _device_for(vec)
    v = _prep(vec, dev).contiguous()
    dv = v.shape[-1]
    d = int((-1.0 + (1.0 + 8.0 * dv) ** 0.5) / 2.0)
    if d * (d + 1) // 2 != dv:
        raise RuntimeError(f"last dimension {dv} is not d(d+1)/2")
    n = v.numel() // dv if dv else 0
    out = torch.empty(v.shape[:-1] + (d, d), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gabo_mandel_to_matrix(v.data_ptr(), out.data_ptr(), n, d, _stream_ptr(dev)), "gabo_mandel_to_matrix")
    return out.to(out_device)


def matrix_to_mandel(mat):
    """(..., d, d) -> (..., d_vec), averaging the two triangles."""
    lib = _lib.load()
    out_device = mat.device
    dev = _device_for(mat)
    m = _prep(mat, dev).contiguous()
    d = m.shape[-1]
    if m.shape[-2] != d:
        raise RuntimeError("last two dimensions must be square")
    n = m.numel() // (d * d)
    out = torch.empty(m.shape[:-2] + (d * (d + 1) // 2,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gabo_matrix_to_mandel(m.data_ptr(), out.data_ptr(), n, d, _stream_ptr(dev)), "gabo_matrix_to_mandel")
    return out.to(out_device)
