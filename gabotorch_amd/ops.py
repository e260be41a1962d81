"""Host-side launch plumbing: torch tensors in, C-ABI calls on the current HIP stream, torch tensors out.

torch is used for device memory and stream ownership only; every number is produced by libgabo_hip.so.
Inputs living on the CPU are moved to the GPU, computed there and moved back (the reference's examples build CPU
tensors); without a GPU every call raises - there is no CPU implementation in this package.
"""
import torch

from . import _lib

_check_errors = "deferred"


def set_error_checking(flag):
    """How device-side data errors (a non-SPD input matrix, a NaN entry) reach the caller.  The reference raises from torch.cholesky at the call
    (spd_utils_torch.py:87); a launch is asynchronous, so raising AT the call costs a stream synchronisation per call.
      "deferred" (default): every launch gets a status word of its own; the words are read back - ONE copy for all pending launches of a stream -
                 at the next synchronisation point: check_deferred(), which the acquisition sweep, the GP posterior and the solvers call where
                 they wait for the device anyway, or when 63 launches are pending on one stream.  The RuntimeError names the launch that failed.
      True / "sync": read back after every call (one stream synchronisation per call): the reference's behaviour to the statement.
      False:     never read.
    Returns the previous setting."""
    global _check_errors
    prev = _check_errors
    if flag == "sync":
        flag = True
    if flag not in (True, False, "deferred"):
        raise ValueError("set_error_checking: True / 'sync', 'deferred' or False")
    if prev is False and flag is not False:
        for ring in _status_rings.values():       # (nobody looked at the words while checking was off: start from clean ones)
            ring.reset()
    _check_errors = flag
    return prev


def _device_for(*tensors):
    for t in tensors:
        if t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("gabotorch_amd needs an MI355X: no HIP device is visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream_ptr(device):
    """raw hipStream_t of torch's current stream on `device` (torch.cuda.current_stream builds a Stream object around the same call: 4 us a time)"""
    idx = device.index if isinstance(device, torch.device) else torch.device(device).index
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


class _NoCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_CTX = _NoCtx()


def _on(device):
    """`with _on(dev):` = `with _on(dev):` when dev is not the current device, nothing otherwise (the guard object costs ~8 us)"""
    idx = device.index if isinstance(device, torch.device) else torch.device(device).index
    return _NO_CTX if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(idx)


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


def _require(dev, **tensors):
    """The pointer-taking wrappers below hand raw device addresses to the library: anything that is not a contiguous tensor on
    `dev` of the dtype the C ABI reads there would be reinterpreted silently.  fp64 unless the name says otherwise."""
    dev = torch.device(dev)
    for name, t in tensors.items():
        if t is None:
            continue
        want = (torch.uint8, torch.bool) if name in ("active", "invalid") else ((torch.int64,) if name == "iters" else (torch.float64,))
        on_dev = torch.is_tensor(t) and t.device.type == dev.type and (dev.index is None or t.device.index == dev.index)
        if not on_dev or t.dtype not in want or not t.is_contiguous():
            raise TypeError(f"{name}: expected a contiguous {want[0]} tensor on {dev}, got "
                            f"{getattr(t, 'dtype', type(t))} on {getattr(t, 'device', None)}"
                            f"{'' if not torch.is_tensor(t) or t.is_contiguous() else ' (not contiguous)'}")


class _StatusRing:
    """The status words (two int32: error code, index of the offender) of the launches on ONE stream of one device: a ring of `N` words in
    device memory, zeroed once - the kernels only ever write a word on an error, so no fill launch in front of a call (4 us of GPU time next to a
    6 us projection, profiles/r03_config5_kernel_stats.csv) - each launch taking the next word.  `pending` holds (slot, message, on_fail) of the
    launches nobody has checked yet; check() waits for the stream, reads the whole ring in one copy, forgets the launches that succeeded and raises
    for the first that did not (clearing its word, calling its on_fail first)."""
    N = 64

    def __init__(self, dev, stream):
        self.buf = torch.zeros(self.N, 2, dtype=torch.int32, device=dev)
        self.stream = stream
        self.next = 0
        self.pending = []

    def take(self):
        k = self.next = self.next % (self.N - 1) + 1          # slots 1 ... N - 1 (slot 0: the word of launches nobody checks)
        w = self.buf[k]
        w._gabo_slot = (self, k)
        return w

    def reset(self):
        with torch.cuda.stream(self.stream):
            self.buf.zero_()
        self.pending.clear()

    def prefetch(self):
        """Enqueue, behind everything on the ring's stream so far, a copy of the ring into page-locked host memory; -> how many pending launches
        it covers.  For a caller that is about to wait for the stream anyway: evaluate() afterwards costs no device access."""
        if getattr(self, "host", None) is None:
            self.host = torch.zeros(self.N, 2, dtype=torch.int32).pin_memory()
        if _stream_ptr(self.buf.device) == self.stream.cuda_stream:
            self.host.copy_(self.buf, non_blocking=True)
        else:
            with torch.cuda.stream(self.stream):
                self.host.copy_(self.buf, non_blocking=True)
        return len(self.pending)

    def check(self):
        if not self.pending:
            return
        with torch.cuda.stream(self.stream):
            vals = self.buf.cpu()              # (waits for everything enqueued on the ring's stream)
        self.evaluate(vals.tolist(), len(self.pending))

    def evaluate(self, vals, count):
        """the first `count` pending launches against a host copy of the ring taken after they had completed"""
        for _ in range(min(count, len(self.pending))):
            k, message, on_fail = self.pending.pop(0)
            st = vals[k]
            if st[0] != 0:
                st = [int(st[0]), int(st[1])]
                with torch.cuda.stream(self.stream):
                    self.buf[k].zero_()
                if on_fail is not None:
                    on_fail()
                raise RuntimeError(message(st) if callable(message) else message)


_status_rings = {}


def _status_word(dev, force=False):
    """A status word for the launch about to be enqueued on the current stream of `dev` (a view of that stream's ring: _StatusRing).
    force: a word that is registered and read even while error checking is off (the GP's Cholesky: see gp_factor)."""
    key = (dev.index, _stream_ptr(dev))
    ring = _status_rings.get(key)
    if ring is None:
        ring = _status_rings[key] = _StatusRing(dev, torch.cuda.current_stream(dev))
    if _check_errors is False and not force:
        return ring.buf[0]          # (never read while checking is off; set_error_checking clears the ring when it comes back on)
    if len(ring.pending) >= ring.N - 2:
        ring.check()
    return ring.take()


def _out(t, out_device):
    """The result on the caller's device.  A caller that handed CPU tensors gets a CPU tensor back - a device -> host copy that waits for the
    stream anyway: the natural point to look at the status words of the launches behind it (deferred error checking: one more small copy)."""
    if t.device == out_device:
        return t
    r = t.to(out_device)
    if _check_errors == "deferred" and out_device.type == "cpu":
        check_deferred()
    return r


def _mandel_dim(dv):
    d = int((-1.0 + (1.0 + 8.0 * dv) ** 0.5) / 2.0)
    if d * (d + 1) // 2 != dv:
        raise RuntimeError(f"last dimension {dv} is not d(d+1)/2")
    return d


def _not_spd_message(what):
    return lambda st: f"{what}: input matrix #{st[1]} is not positive definite (Cholesky pivot <= 0)"


def _raise_if_not_spd(status, what, message=None, on_fail=None, force=False):
    """Called after the launch that was handed `status`: registers the launch for the next check (deferred), or reads the word now (sync).
    message: the RuntimeError's text, or a callable of the two status ints; on_fail: called before raising (e.g. to drop a cache the failed launch
    was meant to fill)."""
    if _check_errors is False and not force:
        return
    message = message or _not_spd_message(what)
    slot = getattr(status, "_gabo_slot", None)
    if slot is not None:
        ring, k = slot
        ring.pending.append((k, message, on_fail))
        if _check_errors is True:
            ring.check()
        return
    # a word of the caller's own (fixed address: captured in a hipGraph, or owned by a solver object)
    if _check_errors is True:
        st = status.tolist()
        if st[0] != 0:
            status.zero_()
            if on_fail is not None:
                on_fail()
            raise RuntimeError(message(st) if callable(message) else message)
    else:
        _deferred.append((status, message, on_fail))
        if len(_deferred) > 32:
            check_deferred()


def _check_launch(rc, status, what, on_fail=None):
    """return code first, then the device status word.  A launch refused by its return code never ran: its word is not registered."""
    try:
        _lib.check(rc, what)
    except Exception:
        if on_fail is not None:
            on_fail()
        raise
    _raise_if_not_spd(status, what, on_fail=on_fail)


_deferred = []


def prefetch_deferred():
    """For a caller that is about to wait for its stream anyway (the acquisition sweep before its scoring call returns): enqueue the read-back of
    every status ring that has unchecked launches and return a token for check_prefetched - which then needs no device access at all."""
    return [(ring, ring.prefetch()) for ring in _status_rings.values() if ring.pending]


def check_prefetched(token):
    """check_deferred for the launches covered by prefetch_deferred's token; ONLY after the streams those copies were enqueued on have been
    waited for (the data is read from host memory as it is)."""
    for ring, count in token:
        ring.evaluate(ring.host.numpy(), count)


def check_deferred():
    """Reads the status words of every launch that has not been checked yet - one copy per stream that has any - and raises for the first failure,
    naming the launch.  Called at the natural host synchronisation points of a sweep (after the raw samples' scores have reached the host, after a
    solve, before a posterior); free when nothing is pending.  The launches that were still queued behind a failed one stay queued."""
    for ring in list(_status_rings.values()):
        ring.check()
    while _deferred:
        status, message, on_fail = _deferred.pop(0)
        st = status.tolist()
        if st[0] != 0:
            status.zero_()
            if on_fail is not None:
                on_fail()
            raise RuntimeError(message(st) if callable(message) else message)


def spd_ai_pairwise(x1, x2, beta=1.0, mode=_lib.GABO_OUT_GAUSSIAN, symmetric=False, return_dist=False):
    """x1 (..., N1, d_vec), x2 (..., N2, d_vec) Mandel vectors -> (..., N1, N2) on x1's device.
    return_dist=True also returns the distance matrix written by the same launch."""
    lib = _lib.load()
    if x1.shape[:-2] != x2.shape[:-2] or x1.shape[-1] != x2.shape[-1]:
        raise RuntimeError(f"batch/feature shapes differ: {tuple(x1.shape)} vs {tuple(x2.shape)} (no broadcasting, as in the reference)")
    out_device = x1.device
    dev = _device_for(x1, x2)
    a, b = _prep(x1, dev), _prep(x2, dev)
    d = _mandel_dim(a.shape[-1])
    n1, n2 = a.shape[-2], b.shape[-2]
    bshape = a.shape[:-2]
    a2, nb, s1 = _flatten_batch(a, 2)
    b2, _, s2 = _flatten_batch(b, 2)
    out = torch.empty(bshape + (n1, n2), dtype=torch.float64, device=dev)
    dist = torch.empty_like(out) if return_dist else None
    if out.numel() == 0:
        return (out.to(out_device), dist.to(out_device)) if return_dist else out.to(out_device)
    wsb = lib.gabo_spd_ai_workspace_bytes(nb, n1, n2, d)
    ws = torch.empty(max(wsb // 8, 1), dtype=torch.float64, device=dev)
    status = _status_word(dev)
    flags = int(mode) | (_lib.GABO_SYMMETRIC if symmetric else 0)
    with _on(dev):
        rc = lib.gabo_spd_ai_pairwise(a2.data_ptr(), b2.data_ptr(), out.data_ptr(), dist.data_ptr() if return_dist else None,
                                      nb, n1, n2, d, s1, s2, float(beta), flags, ws.data_ptr(), wsb, status.data_ptr(),
                                      _stream_ptr(dev))
    _check_launch(rc, status, "gabo_spd_ai_pairwise")
    if return_dist:
        return _out(out, out_device), dist.to(out_device)
    return _out(out, out_device)


def spd_ai_backward(x1, x2, grad_out, beta=1.0, mode=_lib.GABO_OUT_GAUSSIAN, wrt=1):
    """Gradient of sum(grad_out * spd_ai_pairwise(x1, x2)) with respect to x1 (wrt=1) or x2 (wrt=2), Mandel layout.
    grad_out (..., N1, N2).  Returns a tensor shaped like the differentiated argument."""
    lib = _lib.load()
    out_device = (x1 if wrt == 1 else x2).device
    dev = _device_for(x1, x2, grad_out)
    a, b, g = _prep(x1, dev), _prep(x2, dev), _prep(grad_out, dev).contiguous()
    d = _mandel_dim(a.shape[-1])
    n1, n2 = a.shape[-2], b.shape[-2]
    bshape = a.shape[:-2]
    a2, nb, s1 = _flatten_batch(a, 2)
    b2, _, s2 = _flatten_batch(b, 2)
    if wrt == 1:
        first, second, m1, m2, sf, ss = a2, b2, n1, n2, s1, s2
        go_si, go_sj = n2, 1
    else:       # exchange the roles of the two sets and read grad_out transposed
        first, second, m1, m2, sf, ss = b2, a2, n2, n1, s2, s1
        go_si, go_sj = 1, n2
    gx = torch.zeros(bshape + (m1, a.shape[-1]), dtype=torch.float64, device=dev)
    if gx.numel() == 0 or m2 == 0:
        return _out(gx, out_device)
    wsb = lib.gabo_spd_ai_workspace_bytes(nb, m1, m2, d)
    ws = torch.empty(max(wsb // 8, 1), dtype=torch.float64, device=dev)
    status = _status_word(dev)
    with _on(dev):
        rc = lib.gabo_spd_ai_backward(first.data_ptr(), second.data_ptr(), g.data_ptr(), gx.data_ptr(), nb, m1, m2, d, sf, ss,
                                      n1 * n2, go_si, go_sj, float(beta), int(mode), ws.data_ptr(), wsb, status.data_ptr(),
                                      _stream_ptr(dev))
    _lib.check(rc, "gabo_spd_ai_backward")
    _raise_if_not_spd(status, "gabo_spd_ai_backward")
    # (a set handed over as ONE expand()ed set, batch stride 0: the result is still the gradient with respect to the expanded tensor, one slice
    # per batch entry - autograd's ExpandBackward sums them for the base tensor; summing here as well would count the batch twice)
    return _out(gx, out_device)


def spd_ai_backward2(x1, x2, grad_out, u, beta=1.0, mode=_lib.GABO_OUT_GAUSSIAN, want_dgrad_out=True, want_mixed=False):
    """Second-order terms of the kernel (gabo_spd_ai_backward2): with grad_out fixed and u a direction shaped like x1, returns (hv, dg, mixed):
    hv = d/dt spd_ai_backward(x1 + t u, x2, grad_out)|_0 (shaped like x1), dg[..., i, j] = <u_i, dK_ij/dx1_i> (or None) and
    mixed = d <spd_ai_backward(x1, x2, grad_out), u> / d x2 (shaped like x2, or None)."""
    lib = _lib.load()
    out_device = x1.device
    dev = _device_for(x1, x2, grad_out, u)
    a, b = _prep(x1, dev), _prep(x2, dev)
    g, uu = _prep(grad_out, dev).contiguous(), _prep(u, dev).contiguous()
    if uu.shape != a.shape:
        raise RuntimeError(f"direction {tuple(uu.shape)} is not shaped like x1 {tuple(a.shape)}")
    d = _mandel_dim(a.shape[-1])
    n1, n2 = a.shape[-2], b.shape[-2]
    bshape = a.shape[:-2]
    a2, nb, s1 = _flatten_batch(a, 2)
    b2, _, s2 = _flatten_batch(b, 2)
    if s1 == 0 and nb > 1:                       # an expand()ed x1 with per-batch directions: give every batch its own copy
        a2, s1 = a.contiguous().reshape(nb, n1, -1), n1 * a.shape[-1]
    hv = torch.zeros(bshape + (n1, a.shape[-1]), dtype=torch.float64, device=dev)
    dg = torch.zeros(bshape + (n1, n2), dtype=torch.float64, device=dev) if want_dgrad_out else None
    mx = torch.zeros(bshape + (n2, a.shape[-1]), dtype=torch.float64, device=dev) if want_mixed else None
    if hv.numel() == 0 or n2 == 0:
        return _out(hv, out_device), (None if dg is None else dg.to(out_device)), (None if mx is None else mx.to(x2.device))
    wsb = lib.gabo_spd_ai_backward2_workspace_bytes(nb, n1, n2, d)
    ws = torch.empty(max(wsb // 8, 1), dtype=torch.float64, device=dev)
    status = _status_word(dev)
    with _on(dev):
        rc = lib.gabo_spd_ai_backward2(a2.data_ptr(), b2.data_ptr(), g.data_ptr(), uu.data_ptr(), hv.data_ptr(), None if dg is None else dg.data_ptr(),
                                       None if mx is None else mx.data_ptr(), nb, n1, n2, d, s1, s2, n1 * n2, n2, 1, float(beta), int(mode),
                                       ws.data_ptr(), wsb, status.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "gabo_spd_ai_backward2")
    _raise_if_not_spd(status, "gabo_spd_ai_backward2")
    # (x2 as one expand()ed set: mx stays per batch entry, like spd_ai_backward's result - ExpandBackward does the sum)
    return _out(hv, out_device), (None if dg is None else dg.to(out_device)), (None if mx is None else mx.to(x2.device))


class _SpdAiGradFunction(torch.autograd.Function):
    """The first-order gradient of sum(grad_out * K(x1, x2)) with respect to x1 (wrt = 1) or x2 (wrt = 2) as a differentiable function of
    (x1, x2, grad_out, beta): the node a second autograd pass runs through (create_graph=True; pymanopt_addons/tools/autodiff/_pytorch.py:103-116
    builds exact Hessian-vector products that way).  Its own backward is gabo_spd_ai_backward2: the diagonal block of the Hessian, the mixed
    block and the directional derivative of K for grad_out; for wrt = 2 the same call with the two sets exchanged (d(A, B) = d(B, A)).
    The mixed derivative with beta comes from the same launch: dK/dx = -beta K d(d^2)/dx (Gaussian), so d/dbeta (dK/dx) = (1/beta - d^2) dK/dx and
    d <g, u> / dbeta = sum_ij grad_out_ij (1/beta - d_ij^2) <u_i, dK_ij/dx_i>  (Laplace: d instead of d^2; distance output: 0)."""

    @staticmethod
    def forward(ctx, x1, x2, grad_out, beta, mode, wrt):
        bval = float(beta)
        ctx.save_for_backward(x1, x2, grad_out)
        ctx.bval, ctx.mode, ctx.wrt = bval, mode, wrt
        ctx.beta_shape, ctx.beta_device = beta.shape, beta.device
        return spd_ai_backward(x1, x2, grad_out, bval, mode, wrt=wrt).to((x1 if wrt == 1 else x2).dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, u):
        x1, x2, grad_out = ctx.saved_tensors
        need1, need2, needg, needb = ctx.needs_input_grad[:4]
        needb = needb and ctx.mode != _lib.GABO_OUT_DISTANCE
        if ctx.wrt == 1:
            hv, dg, mx = spd_ai_backward2(x1, x2, grad_out, u, ctx.bval, ctx.mode, want_dgrad_out=needg or needb, want_mixed=need2)
            d1, d2 = hv, mx
        else:
            hv, dg, mx = spd_ai_backward2(x2, x1, grad_out.transpose(-1, -2), u, ctx.bval, ctx.mode, want_dgrad_out=needg or needb, want_mixed=need1)
            d1, d2 = mx, hv
            dg = None if dg is None else dg.transpose(-1, -2)
        gb = None
        if ctx.needs_input_grad[3]:
            if needb:
                dist = spd_ai_pairwise(x1, x2, 1.0, _lib.GABO_OUT_DISTANCE).to(dg.device)
                w = dist * dist if ctx.mode == _lib.GABO_OUT_GAUSSIAN else dist
                gb = (grad_out.to(dg.device) * dg * (1.0 / ctx.bval - w)).sum()
            else:
                gb = torch.zeros((), dtype=torch.float64, device=grad_out.device)
            gb = gb.reshape(ctx.beta_shape).to(ctx.beta_device)
        return (d1.to(x1.dtype) if need1 else None), (d2.to(x2.dtype) if need2 else None), (dg.to(grad_out.dtype) if needg else None), gb, None, None


class _SpdAiKernelFunction(torch.autograd.Function):
    """K(x1, x2; beta) with the HIP forward and the HIP closed-form backward.  The gradient with respect to x1 is itself differentiable
    and so is the one with respect to x2 (in x1, x2 and the upstream gradient: _SpdAiGradFunction), which is what exact Hessian-vector
    products of a cost built on the kernel need (approx_hessian=False; the reference's SPD examples run with approx_hessian=True,
    manifold_optimize.py:198-202); so is the gradient with respect to beta (second order in x1, x2, beta and the upstream gradient)."""

    @staticmethod
    def forward(ctx, x1, x2, beta, mode):
        bval = float(beta)
        need = x1.requires_grad or x2.requires_grad or (torch.is_tensor(beta) and beta.requires_grad)
        if need:
            out, dist = spd_ai_pairwise(x1, x2, bval, mode, return_dist=True)
            ctx.save_for_backward(x1, x2, out, dist, beta)
        else:
            same = x1.data_ptr() == x2.data_ptr() and x1.shape == x2.shape and x1.stride() == x2.stride() and x1.dim() == 2
            out = spd_ai_pairwise(x1, x2, bval, mode, symmetric=same)
        ctx.bval, ctx.mode = bval, mode
        ctx.beta_shape = beta.shape if torch.is_tensor(beta) else None
        ctx.beta_device = beta.device if torch.is_tensor(beta) else None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x1, x2, out, dist, beta = ctx.saved_tensors
        g1 = g2 = gb = None
        second = torch.is_grad_enabled()              # create_graph=True: this pass is being recorded
        second = second and (x1.requires_grad or x2.requires_grad or grad_out.requires_grad or beta.requires_grad)
        if ctx.needs_input_grad[0]:
            if second:
                g1 = _SpdAiGradFunction.apply(x1, x2, grad_out, beta, ctx.mode, 1)
            else:
                g1 = spd_ai_backward(x1, x2, grad_out, ctx.bval, ctx.mode, wrt=1).to(x1.dtype)
        if ctx.needs_input_grad[1]:
            if second:
                g2 = _SpdAiGradFunction.apply(x1, x2, grad_out, beta, ctx.mode, 2)
            else:
                g2 = spd_ai_backward(x1, x2, grad_out, ctx.bval, ctx.mode, wrt=2).to(x2.dtype)
        if ctx.needs_input_grad[2]:
            if ctx.mode == _lib.GABO_OUT_DISTANCE:
                gb = torch.zeros((), dtype=grad_out.dtype, device=grad_out.device)
            elif second:
                # A recorded pass: dK/dbeta = -d^2 K (Gaussian) / -d K (Laplace) must itself be a function of (x1, x2, beta, grad_out), so it
                # is rebuilt from two differentiable evaluations (kernel value and distance) instead of the saved, constant matrices - the
                # marginal likelihood's Hessian in beta and the x-beta blocks then come out complete (third derivatives are not provided:
                # the nodes created here are first / second order like every other one).
                kk = _SpdAiKernelFunction.apply(x1, x2, beta, ctx.mode)
                dd = _SpdAiKernelFunction.apply(x1, x2, beta, _lib.GABO_OUT_DISTANCE)
                gb = -(grad_out * kk * (dd * dd if ctx.mode == _lib.GABO_OUT_GAUSSIAN else dd)).sum()
            else:
                go = grad_out.detach()
                if ctx.mode == _lib.GABO_OUT_GAUSSIAN:
                    gb = -(go * out * dist * dist).sum()      # dK/dbeta = -d^2 K
                else:
                    gb = -(go * out * dist).sum()             # dK/dbeta = -d K
            gb = gb.reshape(ctx.beta_shape).to(ctx.beta_device)
        return g1, g2, gb, None


def spd_ai_kernel(x1, x2, beta, mode=_lib.GABO_OUT_GAUSSIAN):
    """Differentiable entry point used by the kernel classes.  beta: python float or a one-element tensor."""
    if torch.is_tensor(beta):
        if beta.numel() != 1:
            raise RuntimeError("gabotorch_amd SPD kernels take a single beta (batch_shape == ()), as every reference example does")
        beta = beta.double()
    else:
        beta = torch.tensor(float(beta), dtype=torch.float64)
    return _SpdAiKernelFunction.apply(x1, x2, beta, int(mode))


def sphere_pairwise(x1, x2, beta=1.0, mode=_lib.GABO_OUT_GAUSSIAN, diag=False, symmetric=False):
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
        return _out(out, out_device)
    flags = int(mode) | (_lib.GABO_SYMMETRIC if symmetric and not diag else 0)
    with _on(dev):
        stream = _stream_ptr(dev)
        ktable = None
        # (not while a hipGraph is being captured: a build recorded into the graph would not have run for eager launches that find it cached)
        if lib.gabo_sphere_pairwise_uses_ktable(nb, n1, n2, dim, float(beta), flags, 1 if diag else 0) and not torch.cuda.is_current_stream_capturing():
            ktable = _sphere_ktable(lib, dev, stream, float(beta))
        rc = lib.gabo_sphere_pairwise_cached(a2.data_ptr(), b2.data_ptr(), out.data_ptr(), nb, n1, n2, dim, s1, s2, float(beta), flags,
                                             1 if diag else 0, None if ktable is None else ktable.data_ptr(), stream)
    _lib.check(rc, "gabo_sphere_pairwise")
    return _out(out, out_device)


_sphere_ktables = {}


def _sphere_ktable(lib, dev, stream, beta):
    """The kernel-value table of the large sphere Gram launches for this beta, built once per (device, stream, beta) by gabo_sphere_ktable_build
    and handed to gabo_sphere_pairwise_cached: a BO iteration evaluates every kernel matrix with the one fitted beta, so the 3-us table build
    leaves the prologue of each launch.  Keyed by the stream the build was enqueued on (stream order makes it visible to the launches behind it)."""
    key = (dev.type, dev.index, stream if isinstance(stream, int) else getattr(stream, "value", stream))
    ent = _sphere_ktables.get(key)
    if ent is None or ent[0] != beta:
        table = ent[1] if ent is not None else torch.empty(int(lib.gabo_sphere_ktable_doubles()), dtype=torch.float64, device=dev)
        if len(_sphere_ktables) > 16 and ent is None:
            _sphere_ktables.clear()
        _lib.check(lib.gabo_sphere_ktable_build(beta, table.data_ptr(), stream), "gabo_sphere_ktable_build")
        _sphere_ktables[key] = (beta, table)
        return table
    return ent[1]


def sphere_from_inner(inner, beta, mode, order):
    """Element-wise f^(order)(c) on an inner-product tensor (any shape)."""
    lib = _lib.load()
    out_device = inner.device
    dev = _device_for(inner)
    c = _prep(inner, dev).contiguous()
    out = torch.empty_like(c)
    with _on(dev):
        _lib.check(lib.gabo_sphere_from_inner(c.data_ptr(), out.data_ptr(), c.numel(), float(beta), int(mode), int(order),
                                              _stream_ptr(dev)), "gabo_sphere_from_inner")
    return _out(out, out_device)


class _SphereFromInner(torch.autograd.Function):
    """f^(order)(c), differentiable in c up to order 2 (value -> gradient -> Hessian-vector product) and, at order 0, in beta."""

    @staticmethod
    def forward(ctx, c, beta, mode, order):
        bval = float(beta)
        out = sphere_from_inner(c, bval, mode, order)
        ctx.save_for_backward(c, out)
        ctx.bval, ctx.mode, ctx.order = bval, mode, order
        ctx.beta_shape = beta.shape if torch.is_tensor(beta) else None
        ctx.beta_device = beta.device if torch.is_tensor(beta) else None
        return out

    @staticmethod
    def backward(ctx, g):
        c, out = ctx.saved_tensors
        gc = gb = None
        if ctx.needs_input_grad[0]:
            if ctx.order >= 2:
                raise RuntimeError("gabotorch_amd sphere kernels are differentiable twice in x, not three times")
            gc = g * _SphereFromInner.apply(c, torch.tensor(ctx.bval, dtype=torch.float64), ctx.mode, ctx.order + 1)
        if ctx.needs_input_grad[1]:
            if ctx.order != 0:
                raise RuntimeError("mixed x/beta second derivatives of the sphere kernels are not provided")
            with torch.no_grad():
                th = sphere_from_inner(c, 1.0, _lib.GABO_OUT_DISTANCE, 0)
                if ctx.mode == _lib.GABO_OUT_GAUSSIAN:
                    gb = -(g * out * th * th).sum()
                elif ctx.mode == _lib.GABO_OUT_LAPLACE:
                    gb = -(g * out * th).sum()
                else:
                    gb = torch.zeros((), dtype=g.dtype, device=g.device)
                gb = gb.reshape(ctx.beta_shape).to(ctx.beta_device)
        return gc, gb, None, None


def sphere_kernel(x1, x2, beta, mode=_lib.GABO_OUT_GAUSSIAN, diag=False):
    """Differentiable entry point of the sphere kernels.  Without autograd: one fused pairwise launch.  With autograd: the
    inner products are a plain batched GEMM (torch.matmul -> rocBLAS) followed by the element-wise HIP kernel, which keeps the
    expression differentiable twice in x1/x2."""
    if torch.is_tensor(beta):
        if beta.numel() != 1:
            raise RuntimeError("gabotorch_amd sphere kernels take a single beta (batch_shape == ())")
        beta = beta.double()
    else:
        beta = torch.tensor(float(beta), dtype=torch.float64)
    need = torch.is_grad_enabled() and (x1.requires_grad or x2.requires_grad or beta.requires_grad)
    if not need:
        # (no automatic GABO_SYMMETRIC here: the sphere launch is so short that the mirror pass costs more than it saves)
        return sphere_pairwise(x1, x2, float(beta), mode, diag=diag)
    dev = _device_for(x1, x2)
    a, b = x1.double().to(dev), x2.double().to(dev)
    if diag:
        c = (a * b).sum(-1, keepdim=True)
    else:
        c = a @ b.transpose(-1, -2)
    return _SphereFromInner.apply(c, beta, int(mode), 0).to(x1.device)


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
    with _on(dev):
        _lib.check(lib.gabo_mandel_to_matrix(v.data_ptr(), out.data_ptr(), n, d, _stream_ptr(dev)), "gabo_mandel_to_matrix")
    return _out(out, out_device)


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
    with _on(dev):
        _lib.check(lib.gabo_matrix_to_mandel(m.data_ptr(), out.data_ptr(), n, d, _stream_ptr(dev)), "gabo_matrix_to_mandel")
    return _out(out, out_device)


# ------------------------------------------------------------------------------------------ batched manifold operations
def _mats(dev, *xs):
    """Bring (…, d, d) inputs to contiguous fp64 device tensors of one common batch shape."""
    ts = [None if x is None else _prep(torch.as_tensor(x), dev) for x in xs]
    shape = None
    for t_ in ts:
        if t_ is not None:
            shape = t_.shape if shape is None else torch.broadcast_shapes(shape, t_.shape)
    return [None if t_ is None else t_.expand(shape).contiguous() for t_ in ts], shape


def spd_manifold_op(op, a, b=None, c=None, e=None, want_grad=False):
    """Batched SPD-manifold operation `op` (one of _lib.GABO_SPD_*) on (..., d, d) tensors; see include/gabo_hip.h."""
    lib = _lib.load()
    # the maximiser's own calls: contiguous fp64 tensors of one shape on one HIP device - nothing to convert, broadcast or move (the general
    # route below spends ~0.1 ms of host time on that per call, two calls in front of every 4-ms sweep's solve launch)
    if (torch.is_tensor(a) and a.is_cuda and a.dtype == torch.float64 and a.is_contiguous() and a.dim() >= 2
            and all(x is None or (torch.is_tensor(x) and x.dtype == torch.float64 and x.device == a.device and x.shape == a.shape
                                  and x.is_contiguous()) for x in (b, c, e))):
        out_device = dev = a.device
        (A, B, C, E), shape = (a, b, c, e), a.shape
    else:
        first = torch.as_tensor(a)
        out_device = first.device
        dev = _device_for(*[torch.as_tensor(x) for x in (a, b, c, e) if x is not None])
        (A, B, C, E), shape = _mats(dev, a, b, c, e)
    d = shape[-1]
    n = A.numel() // (d * d) if d else 0
    scalar = op in (_lib.GABO_SPD_INNER, _lib.GABO_SPD_NORM, _lib.GABO_SPD_DIST, _lib.GABO_SPD_EIGMAX, _lib.GABO_SPD_EIGMIN)
    out = torch.empty(shape[:-2] if scalar else shape, dtype=torch.float64, device=dev)
    matfun = op in (_lib.GABO_SPD_LOGM, _lib.GABO_SPD_EXPM, _lib.GABO_SPD_SQRTM)
    out2 = None
    if want_grad and scalar:
        out2 = torch.empty(shape, dtype=torch.float64, device=dev)               # v v^T of the extreme eigenvalue
    elif want_grad and matfun:
        out2 = torch.empty(shape[:-2] + (d * d + d,), dtype=torch.float64, device=dev)      # V and the eigenvalues, for the backward
    status = _status_word(dev)
    ptr = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
    with _on(dev):
        rc = lib.gabo_spd_manifold_op(int(op), ptr(A), ptr(B), ptr(C), ptr(E), out.data_ptr(), ptr(out2), n, d, status.data_ptr(),
                                      _stream_ptr(dev))
    _check_launch(rc, status, "gabo_spd_manifold_op")
    if out2 is not None:
        return _out(out, out_device), (out2 if matfun else out2.to(out_device))
    return _out(out, out_device)


class _SpdMatFun(torch.autograd.Function):
    """logm / expm / sqrtm of symmetric (SPD) matrices with the divided-difference adjoint as backward (first order)."""

    @staticmethod
    def forward(ctx, a, op):
        # the forward launch hands its eigen-decomposition (device-resident) to the backward: no second eigen-solve
        out, eig = spd_manifold_op(int(op), a, want_grad=True)
        ctx.save_for_backward(eig)
        ctx.op, ctx.shape, ctx.device, ctx.dtype = int(op), tuple(a.shape), a.device, a.dtype
        return out.to(a.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (eig,) = ctx.saved_tensors
        lib = _lib.load()
        dev = eig.device
        d = ctx.shape[-1]
        G = _prep(g, dev).expand(eig.shape[:-1] + (d, d)).contiguous()
        out = torch.empty_like(G)
        with _on(dev):
            _lib.check(lib.gabo_spd_matfun_backward_eig(ctx.op, eig.data_ptr(), G.data_ptr(), out.data_ptr(), G.numel() // (d * d), d,
                                                        _stream_ptr(dev)), "gabo_spd_matfun_backward_eig")
        return out.reshape(ctx.shape).to(ctx.device, ctx.dtype), None


def spd_matrix_function(a, op):
    """Differentiable GABO_SPD_LOGM / GABO_SPD_EXPM / GABO_SPD_SQRTM on (..., d, d) matrices."""
    return _SpdMatFun.apply(a, int(op)) if a.requires_grad else spd_manifold_op(int(op), a)


class NestedSpdReconstruction:
    """Reconstruction cost of the nested-SPD mapping for a FIXED data set, value and gradient in one launch
    (gabo_nested_spd_reconstruction; nested_spd_optimization.py:23-92).  metric: _lib.GABO_RECON_LOG_EUCLIDEAN / _AFFINE_INVARIANT.
    Built once per optimisation: logm X_n (or chol(X_n)^-1), Y_n^1/2 and the staging buffers stay on the device."""

    def __init__(self, x_data, x_data_projected, projection_matrix, metric, sqrt_low=None):
        lib = _lib.load()
        self.metric = int(metric)
        dev = self.device = _device_for(x_data, x_data_projected, projection_matrix)
        X = _prep(x_data, dev).contiguous()
        self.y = _prep(x_data_projected, dev).contiguous()
        self.w = _prep(projection_matrix, dev).contiguous()
        self.N, self.D, self.d = int(X.shape[0]), int(X.shape[-1]), int(self.w.shape[1])
        self.m = self.D - self.d
        if X.dim() != 3 or X.shape[1] != self.D or tuple(self.y.shape) != (self.N, self.d, self.d) or self.w.shape[0] != self.D:
            raise RuntimeError(f"shapes: x_data {tuple(X.shape)}, x_data_projected {tuple(self.y.shape)}, projection_matrix {tuple(self.w.shape)}")
        self.sqrt_y = (spd_manifold_op(_lib.GABO_SPD_SQRTM, self.y) if sqrt_low is None else _prep(sqrt_low, dev)).contiguous()
        self.data = torch.empty_like(X)
        status = _status_word(dev)
        with _on(dev):
            _lib.check(lib.gabo_nested_spd_reconstruction_prepare(X.data_ptr(), self.data.data_ptr(), self.N, self.D, self.metric,
                                                                  status.data_ptr(), _stream_ptr(dev)), "gabo_nested_spd_reconstruction_prepare")
        _raise_if_not_spd(status, "gabo_nested_spd_reconstruction_prepare")
        self.sizes = (self.D * self.m, self.m * self.m, self.d * self.m)
        self._buffers = {}

    def _staging(self, P):
        ent = self._buffers.get(P)
        if ent is None:
            lib = _lib.load()
            nV, nC, nK = self.sizes
            npar = nV + nC + nK
            ws = max(int(lib.gabo_nested_spd_reconstruction_workspace_bytes(P, max(self.N, 1), self.D, self.d)), 16)
            host_in, host_out = torch.empty(P * npar, dtype=torch.float64).pin_memory(), torch.empty(P * (1 + npar), dtype=torch.float64).pin_memory()
            dev_in = torch.empty(P * npar, dtype=torch.float64, device=self.device)
            dev_out = torch.empty(P * (1 + npar), dtype=torch.float64, device=self.device)
            o1, o2, o3 = P, P + P * nV, P + P * (nV + nC)
            ent = self._buffers[P] = dict(
                host_in=host_in, dev_in=dev_in, host_out=host_out, dev_out=dev_out, ws=torch.empty(ws, dtype=torch.uint8, device=self.device),
                # views made once: an evaluation is host-bound (one launch of ~0.14 ms), every Python-level slice on its path counts
                hin=host_in.numpy(), hout=host_out.numpy(), v=dev_in[:P * nV], c=dev_in[P * nV:P * (nV + nC)], k=dev_in[P * (nV + nC):],
                cost=dev_out[:P], gv=dev_out[o1:o2], gc=dev_out[o2:o3], gk=dev_out[o3:], ho_cost=host_out[:P], do_cost=dev_out[:P],
                stream=torch.cuda.current_stream(self.device))
        return ent

    def launch(self, v, c, k, cost, gv, gc, gk, P, ws):
        """Raw launch on contiguous fp64 device tensors (gv = gc = gk = None: values only)."""
        lib = _lib.load()
        ptr = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(lib.gabo_nested_spd_reconstruction(self.data.data_ptr(), self.y.data_ptr(), self.sqrt_y.data_ptr(), self.w.data_ptr(),
                                                          v.data_ptr(), c.data_ptr(), k.data_ptr(), cost.data_ptr(), ptr(gv), ptr(gc), ptr(gk), None,
                                                          None, P, self.N, self.D, self.d, self.metric, ws.data_ptr(), ws.numel(),
                                                          _stream_ptr(self.device)),
                       "gabo_nested_spd_reconstruction")

    def evaluate_host(self, V, C, K, grad=True):
        """numpy in, numpy out: V (P, D, m) or (D, m), C, K likewise -> cost (P,) [, gV, gC, gK].  One pinned host-to-device copy, one
        launch, one copy back: what an evaluation of the augmented-Lagrangian line search costs."""
        import numpy as np
        V, C, K = (np.asarray(a, dtype=np.float64) for a in (V, C, K))
        single = V.ndim == 2
        P = 1 if single else V.shape[0]
        ent = self._staging(P)
        nV, nC, nK = self.sizes
        hin = ent["hin"]
        hin[:P * nV] = V.ravel()
        hin[P * nV:P * (nV + nC)] = C.ravel()
        hin[P * (nV + nC):] = K.ravel()
        ent["dev_in"].copy_(ent["host_in"], non_blocking=True)
        if grad:
            self.launch(ent["v"], ent["c"], ent["k"], ent["cost"], ent["gv"], ent["gc"], ent["gk"], P, ent["ws"])
            ent["host_out"].copy_(ent["dev_out"], non_blocking=True)
        else:
            self.launch(ent["v"], ent["c"], ent["k"], ent["cost"], None, None, None, P, ent["ws"])
            ent["ho_cost"].copy_(ent["do_cost"], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        hout = ent["hout"]
        costs = hout[:P].copy()
        if not grad:
            return costs[0] if single else costs
        o1, o2, o3 = P, P + P * nV, P + P * (nV + nC)
        gV = hout[o1:o2].reshape(V.shape).copy()
        gC = hout[o2:o3].reshape(C.shape).copy()
        gK = hout[o3:].reshape(K.shape).copy()
        return (costs[0] if single else costs), gV, gC, gK

    def __call__(self, V, C, K):
        """Differentiable (first order) torch form: V (D, m), C (m, m), K (d, m) tensors -> 0-dim cost."""
        return _NestedSpdReconstructionFn.apply(self, V, C, K)

    def solve_host(self, V, C, unit, raw, options):
        """The augmented-Lagrangian / conjugate-gradient run from the start point (V, C, unit, raw) as ONE call of the native host loop
        (gabo_nested_spd_reconstruction_solve): numpy in, numpy out -> (V, C, unit, raw, log dict).  options: _lib.ReconSolveOptions."""
        import ctypes

        import numpy as np
        lib = _lib.load()
        ent = self._buffers.get("solve")
        if ent is None:
            dev_bytes, pin_doubles = ctypes.c_size_t(0), ctypes.c_size_t(0)
            lib.gabo_nested_spd_reconstruction_solve_workspace_bytes(self.N, self.D, self.d, ctypes.byref(dev_bytes), ctypes.byref(pin_doubles))
            ent = self._buffers["solve"] = dict(ws=torch.empty(max(dev_bytes.value, 16), dtype=torch.uint8, device=self.device),
                                                pinned=torch.empty(pin_doubles.value, dtype=torch.float64).pin_memory(),
                                                w_host=np.ascontiguousarray(self.w.cpu().numpy(), dtype=np.float64))
        v, c, u = (np.array(a, dtype=np.float64, order="C") for a in (V, C, unit))
        r = np.array([float(np.asarray(raw).reshape(-1)[0])], dtype=np.float64)
        if v.shape != (self.D, self.m) or c.shape != (self.m, self.m) or u.size != self.d * self.m:
            raise RuntimeError(f"start point shapes: V {v.shape}, C {c.shape}, unit {u.shape}")
        log = _lib.ReconSolveLog()
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
        with torch.cuda.device(self.device):
            code = lib.gabo_nested_spd_reconstruction_solve(
                self.data.data_ptr(), self.y.data_ptr(), self.sqrt_y.data_ptr(), self.w.data_ptr(), ptr(ent["w_host"]), ptr(v), ptr(c), ptr(u),
                ptr(r), self.N, self.D, self.d, self.metric, ent["ws"].data_ptr(), ent["ws"].numel(), ent["pinned"].data_ptr(),
                ent["pinned"].numel(), ctypes.byref(options), ctypes.byref(log), _stream_ptr(self.device))
        _lib.check(code, "gabo_nested_spd_reconstruction_solve")
        return v, c, u, r, _recon_log(log)


def _recon_log(log):
    return {"iterations": int(log.outer_iterations), "inner_iterations": int(log.inner_iterations), "evaluations": int(log.evaluations),
            "launches": int(log.launches), "stop_reason": _lib.GABO_RECON_STOP[log.stop_reason], "violation": float(log.violation),
            "rho": float(log.rho), "gammas": [float(log.gamma)], "final_cost": float(log.final_cost), "time": float(log.seconds),
            "time_in_evaluator": float(log.seconds_evaluator), "host_threads": int(log.host_threads)}


def nested_spd_reconstruction_solve_with(evaluate, w_host, V, C, unit, raw, options):
    """The same native loop around a host callable  evaluate(V (P, D, m), C (P, m, m), K (P, d, m)) -> (cost (P,), gV, gC, gK)  in numpy
    (gabo_nested_spd_reconstruction_solve_with): a user-supplied cost function, or a test's.  -> (V, C, unit, raw, log dict)."""
    import ctypes

    import numpy as np
    lib = _lib.load()
    w = np.ascontiguousarray(w_host, dtype=np.float64)
    D, d = w.shape
    m = D - d
    v, c, u = (np.array(a, dtype=np.float64, order="C") for a in (V, C, unit))
    r = np.array([float(np.asarray(raw).reshape(-1)[0])], dtype=np.float64)
    npar = D * m + m * m + d * m
    staging = np.empty(_lib.GABO_RECON_MAX_LOOKAHEAD * (2 * npar + 1 + m + m * m))
    failure = []

    def trampoline(_ctx, P, pv, pc, pk, pcost, pgv, pgc, pgk):
        try:
            as_np = lambda q, *shape: np.ctypeslib.as_array(q, shape=(int(np.prod(shape)),)).reshape(shape)   # noqa: E731
            cost, gv, gc, gk = evaluate(as_np(pv, P, D, m).copy(), as_np(pc, P, m, m).copy(), as_np(pk, P, d, m).copy())
            as_np(pcost, P)[:] = np.asarray(cost, dtype=np.float64).reshape(P)
            as_np(pgv, P, D, m)[:] = gv
            as_np(pgc, P, m, m)[:] = gc
            as_np(pgk, P, d, m)[:] = gk
            return _lib.GABO_OK
        except Exception as exc:          # nothing may propagate through the C frames: the loop stops on the code, the error is raised below
            failure.append(exc)
            return _lib.GABO_ERR_ARG

    log = _lib.ReconSolveLog()
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    code = lib.gabo_nested_spd_reconstruction_solve_with(_lib.ReconEvalFn(trampoline), None, ptr(w), ptr(v), ptr(c), ptr(u), ptr(r), D, d,
                                                         ptr(staging), staging.size, ctypes.byref(options), ctypes.byref(log))
    if failure:
        raise failure[0]
    _lib.check(code, "gabo_nested_spd_reconstruction_solve_with")
    return v, c, u, r, _recon_log(log)


class _NestedSpdReconstructionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rec, V, C, K):
        dev = rec.device
        v, c, k = (_prep(t_.detach(), dev).contiguous() for t_ in (V, C, K))
        need = any(t_.requires_grad for t_ in (V, C, K))
        cost = torch.empty(1, dtype=torch.float64, device=dev)
        gv, gc, gk = (torch.empty_like(t_) for t_ in (v, c, k)) if need else (None, None, None)
        rec.launch(v, c, k, cost, gv, gc, gk, 1, rec._staging(1)["ws"])
        if need:
            ctx.save_for_backward(gv, gc, gk)
        ctx.meta = [(t_.device, t_.dtype) for t_ in (V, C, K)]
        return cost[0].to(V.device, V.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        out = [None]
        for grad, (dev, dt), need in zip(ctx.saved_tensors, ctx.meta, ctx.needs_input_grad[1:]):
            out.append((grad * g.to(grad.device)).to(dev, dt) if need else None)
        return tuple(out)


def nested_spd_lift_prepare(w, v, c, k):
    """(x0, p) of gabo_nested_spd_lift_prepare for the mapping (W, V, C, K) on w's device (detached fp64)."""
    lib = _lib.load()
    dev = _device_for(w, v, c, k)
    W, V, C, K = (_prep(t_.detach(), dev).contiguous() for t_ in (w, v, c, k))
    D, d = int(W.shape[0]), int(W.shape[1])
    if tuple(V.shape) != (D, D - d) or tuple(C.shape) != (D - d, D - d) or tuple(K.shape) != (d, D - d):
        raise RuntimeError(f"nested SPD mapping shapes: W {tuple(W.shape)}, V {tuple(V.shape)}, C {tuple(C.shape)}, K {tuple(K.shape)}")
    x0 = torch.empty(D, D, dtype=torch.float64, device=dev)
    p = torch.empty(D, d, dtype=torch.float64, device=dev)
    with _on(dev):
        _lib.check(lib.gabo_nested_spd_lift_prepare(V.data_ptr(), C.data_ptr(), K.data_ptr(), x0.data_ptr(), p.data_ptr(), D, d, _stream_ptr(dev)),
                   "gabo_nested_spd_lift_prepare")
    return W, x0, p


def nested_spd_extreme_eigenvalues(y, w, p, x0, want_grad=False):
    """y (..., d, d) latent points -> lam (..., 2) = (lambda_max, lambda_min) of the lifted points [, grad (..., 2, d, d)]."""
    lib = _lib.load()
    dev = w.device
    Y = _prep(y, dev).contiguous()
    D, d = int(w.shape[0]), int(w.shape[1])
    if Y.shape[-2:] != (d, d):
        raise RuntimeError(f"latent points must be (..., {d}, {d}), got {tuple(Y.shape)}")
    R = Y.numel() // (d * d)
    lam = torch.empty(Y.shape[:-2] + (2,), dtype=torch.float64, device=dev)
    grad = torch.empty(Y.shape[:-2] + (2, d, d), dtype=torch.float64, device=dev) if want_grad else None
    with _on(dev):
        _lib.check(lib.gabo_nested_spd_extreme_eigenvalues(Y.data_ptr(), w.data_ptr(), p.data_ptr(), x0.data_ptr(), lam.data_ptr(),
                                                           None if grad is None else grad.data_ptr(), R, D, d, _stream_ptr(dev)),
                   "gabo_nested_spd_extreme_eigenvalues")
    return (lam, grad) if want_grad else lam


class _NestedSpdExtremes(torch.autograd.Function):
    """(lambda_max, lambda_min) of the lifted latent point, differentiable (first order) in the latent point."""

    @staticmethod
    def forward(ctx, y, w, p, x0):
        if not y.requires_grad:
            return nested_spd_extreme_eigenvalues(y, w, p, x0).to(y.device, y.dtype)
        lam, grad = nested_spd_extreme_eigenvalues(y, w, p, x0, want_grad=True)
        ctx.save_for_backward(grad)
        return lam.to(y.device, y.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        gy = (g.to(grad.device)[..., None, None] * grad).sum(-3)
        return gy.to(g.device, g.dtype), None, None, None


def nested_spd_extremes(y, w, p, x0):
    return _NestedSpdExtremes.apply(y, w, p, x0)


def spd_project(x_mandel, w):
    """(..., D_vec) Mandel, w (D, dl) -> (..., dl_vec) Mandel of W^T X W."""
    lib = _lib.load()
    out_device = x_mandel.device
    dev = _device_for(x_mandel, w)
    x = _prep(x_mandel, dev).contiguous()
    W = _prep(w, dev).contiguous()
    D = _mandel_dim(x.shape[-1])
    if W.dim() != 2 or W.shape[0] != D:
        raise RuntimeError(f"projection matrix must be ({D}, d_latent), got {tuple(W.shape)}")
    dl = W.shape[1]
    n = x.numel() // x.shape[-1]
    out = torch.empty(x.shape[:-1] + (dl * (dl + 1) // 2,), dtype=torch.float64, device=dev)
    with _on(dev):
        _lib.check(lib.gabo_spd_project(x.data_ptr(), W.data_ptr(), out.data_ptr(), n, D, dl, _stream_ptr(dev)), "gabo_spd_project")
    return _out(out, out_device)


class _SpdProject(torch.autograd.Function):
    """Mandel(W^T X W), differentiable in X (adjoint = the same kernel with W^T) and in W (2 sum_n X_n W G_n)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return spd_project(x, w)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = spd_project(g, w.t().contiguous()).to(x.dtype)
        if ctx.needs_input_grad[1]:
            dev = _device_for(x, w, g)
            X = mandel_to_matrix(_prep(x, dev)).reshape(-1, w.shape[0], w.shape[0])
            G = mandel_to_matrix(_prep(g, dev)).reshape(-1, w.shape[1], w.shape[1])
            gw = 2.0 * torch.einsum("nab,bc,ncd->ad", X, _prep(w, dev), G).to(w.device, w.dtype)
        return gx, gw


def spd_project_diff(x_mandel, w):
    """Differentiable spd_project."""
    if x_mandel.requires_grad or w.requires_grad:
        return _SpdProject.apply(x_mandel, w)
    return spd_project(x_mandel, w)


def nested_spd_gram(x1, x2, w, beta, metric=_lib.GABO_METRIC_AFFINE_INVARIANT, mode=_lib.GABO_OUT_GAUSSIAN):
    """Gram matrix of the nested SPD kernels in two launches, no gradient (gabo_nested_spd_gram): x1 (..., N1, D_vec), x2 (..., N2, D_vec) Mandel
    vectors of the ORIGINAL space, w (D, d_latent) with 2 <= d_latent <= 4 -> (..., N1, N2).  metric: _lib.GABO_METRIC_AFFINE_INVARIANT
    (exp(-beta d_AI^2) of the projected points) or _lib.GABO_METRIC_LOG_EUCLIDEAN (exp(-beta ||logm - logm + 1e-15||_F^2))."""
    lib = _lib.load()
    if x1.shape[:-2] != x2.shape[:-2] or x1.shape[-1] != x2.shape[-1]:
        raise RuntimeError(f"batch/feature shapes differ: {tuple(x1.shape)} vs {tuple(x2.shape)} (no broadcasting, as in the reference)")
    out_device = x1.device
    dev = _device_for(x1, x2, w)
    same = x2 is x1
    a = _prep(x1, dev).contiguous()
    b = a if same else _prep(x2, dev).contiguous()
    W = _prep(w, dev).contiguous()
    D = _mandel_dim(a.shape[-1])
    if W.dim() != 2 or W.shape[0] != D:
        raise RuntimeError(f"projection matrix must be ({D}, d_latent), got {tuple(W.shape)}")
    dl = int(W.shape[1])
    n1, n2 = a.shape[-2], b.shape[-2]
    bshape = a.shape[:-2]
    nb = 1
    for k in bshape:
        nb *= int(k)
    out = torch.empty(bshape + (n1, n2), dtype=torch.float64, device=dev)
    if out.numel() == 0:
        return _out(out, out_device)
    wsb = lib.gabo_nested_spd_gram_workspace_bytes(nb, n1, n2, dl)
    ws = torch.empty(max(wsb // 8, 1), dtype=torch.float64, device=dev)
    status = _status_word(dev)
    with _on(dev):
        rc = lib.gabo_nested_spd_gram(a.data_ptr(), b.data_ptr(), W.data_ptr(), out.data_ptr(), nb, n1, n2, D, dl, int(metric), float(beta), int(mode),
                                      ws.data_ptr(), wsb, status.data_ptr(), _stream_ptr(dev))
    _check_launch(rc, status, "gabo_nested_spd_gram")
    return _out(out, out_device)


def nested_spd_gram_applicable(x1, x2, w, *params):
    """the fused two-launch Gram serves evaluations that nobody differentiates, latent dimensions 2 ... 4 and dense, equally batched inputs"""
    if torch.is_grad_enabled() and any(torch.is_tensor(t_) and t_.requires_grad for t_ in (x1, x2, w) + params):
        return False
    if w.dim() != 2 or not (2 <= w.shape[1] <= 4) or w.shape[0] > _lib.GABO_SPD_MAX_DIM or x1.dim() < 2 or x1.shape[:-2] != x2.shape[:-2]:
        return False
    D = w.shape[0]
    if x1.shape[-1] != D * (D + 1) // 2 or x2.shape[-1] != x1.shape[-1]:
        return False
    dv = w.shape[1] * (w.shape[1] + 1) // 2
    return (D * w.shape[1] + dv * x1.shape[-1]) * 8 <= 48 * 1024 and (x1.is_cuda or torch.cuda.is_available())


def spd_logm_mandel(x_mandel):
    lib = _lib.load()
    out_device = x_mandel.device
    dev = _device_for(x_mandel)
    x = _prep(x_mandel, dev).contiguous()
    d = _mandel_dim(x.shape[-1])
    out = torch.empty_like(x)
    with _on(dev):
        _lib.check(lib.gabo_spd_logm_mandel(x.data_ptr(), out.data_ptr(), x.numel() // x.shape[-1], d, _stream_ptr(dev)),
                   "gabo_spd_logm_mandel")
    return _out(out, out_device)


def frobenius_pairwise(x1, x2, beta=1.0, mode=_lib.GABO_OUT_GAUSSIAN):
    """Mandel vectors x1 (..., N1, d_vec), x2 (..., N2, d_vec) -> (..., N1, N2) Frobenius distance / kernel."""
    lib = _lib.load()
    if x1.shape[:-2] != x2.shape[:-2] or x1.shape[-1] != x2.shape[-1]:
        raise RuntimeError(f"batch/feature shapes differ: {tuple(x1.shape)} vs {tuple(x2.shape)}")
    out_device = x1.device
    dev = _device_for(x1, x2)
    a, b = _prep(x1, dev), _prep(x2, dev)
    d = _mandel_dim(a.shape[-1])
    n1, n2 = a.shape[-2], b.shape[-2]
    a2, nb, s1 = _flatten_batch(a, 2)
    b2, _, s2 = _flatten_batch(b, 2)
    out = torch.empty(a.shape[:-2] + (n1, n2), dtype=torch.float64, device=dev)
    with _on(dev):
        _lib.check(lib.gabo_frobenius_pairwise(a2.data_ptr(), b2.data_ptr(), out.data_ptr(), nb, n1, n2, d, s1, s2, float(beta),
                                               int(mode), _stream_ptr(dev)), "gabo_frobenius_pairwise")
    return _out(out, out_device)


def spd_logm_mandel_backward(x_mandel, grad_y):
    """Gradient w.r.t. x (Mandel) of a loss whose gradient w.r.t. spd_logm_mandel(x) is grad_y."""
    lib = _lib.load()
    out_device = x_mandel.device
    dev = _device_for(x_mandel, grad_y)
    x = _prep(x_mandel, dev).contiguous()
    g = _prep(grad_y, dev).expand(x.shape).contiguous()
    d = _mandel_dim(x.shape[-1])
    out = torch.empty_like(x)
    with _on(dev):
        _lib.check(lib.gabo_spd_logm_mandel_backward(x.data_ptr(), g.data_ptr(), out.data_ptr(), x.numel() // x.shape[-1], d,
                                                     _stream_ptr(dev)), "gabo_spd_logm_mandel_backward")
    return _out(out, out_device)


class _SpdLogmMandel(torch.autograd.Function):
    """Mandel(logm(X)) with the Daleckii-Krein adjoint as backward (first order)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return spd_logm_mandel(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return spd_logm_mandel_backward(x, g).to(x.dtype)


def spd_logm_mandel_diff(x_mandel):
    """Differentiable spd_logm_mandel."""
    return _SpdLogmMandel.apply(x_mandel) if x_mandel.requires_grad else spd_logm_mandel(x_mandel)


def frobenius_backward(x1, x2, grad_out, beta=1.0, mode=_lib.GABO_OUT_GAUSSIAN, wrt=1):
    """Gradient of sum(grad_out * frobenius_pairwise(x1, x2)) with respect to x1 (wrt=1) or x2 (wrt=2)."""
    lib = _lib.load()
    out_device = (x1 if wrt == 1 else x2).device
    dev = _device_for(x1, x2, grad_out)
    a, b, g = _prep(x1, dev), _prep(x2, dev), _prep(grad_out, dev).contiguous()
    d = _mandel_dim(a.shape[-1])
    n1, n2 = a.shape[-2], b.shape[-2]
    bshape = a.shape[:-2]
    a2, nb, s1 = _flatten_batch(a, 2)
    b2, _, s2 = _flatten_batch(b, 2)
    if wrt == 1:
        first, second, m1, m2, sf, ss, go_si, go_sj, sgn = a2, b2, n1, n2, s1, s2, n2, 1, 1.0
    else:
        first, second, m1, m2, sf, ss, go_si, go_sj, sgn = b2, a2, n2, n1, s2, s1, 1, n2, -1.0
    gx = torch.zeros(bshape + (m1, a.shape[-1]), dtype=torch.float64, device=dev)
    if gx.numel() == 0 or m2 == 0:
        return _out(gx, out_device)
    with _on(dev):
        _lib.check(lib.gabo_frobenius_backward(first.data_ptr(), second.data_ptr(), g.data_ptr(), gx.data_ptr(), nb, m1, m2, d, sf,
                                               ss, n1 * n2, go_si, go_sj, float(beta), int(mode), sgn, _stream_ptr(dev)),
                   "gabo_frobenius_backward")
    if sf == 0 and nb > 1:
        gx = gx.reshape(nb, m1, -1).sum(0).expand(bshape + (m1, a.shape[-1]))
    return _out(gx, out_device)


class _FrobeniusKernelFunction(torch.autograd.Function):
    """frobenius_pairwise(x1, x2; beta) with HIP forward and backward (first order), differentiable in x1, x2 and beta."""

    @staticmethod
    def forward(ctx, x1, x2, beta, mode):
        bval = float(beta)
        out = frobenius_pairwise(x1, x2, bval, mode)
        ctx.save_for_backward(x1, x2, out)
        ctx.bval, ctx.mode = bval, mode
        ctx.beta_shape = beta.shape if torch.is_tensor(beta) else None
        ctx.beta_device = beta.device if torch.is_tensor(beta) else None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        x1, x2, out = ctx.saved_tensors
        g1 = g2 = gb = None
        if ctx.needs_input_grad[0]:
            g1 = frobenius_backward(x1, x2, grad_out, ctx.bval, ctx.mode, wrt=1).to(x1.dtype)
        if ctx.needs_input_grad[1]:
            g2 = frobenius_backward(x1, x2, grad_out, ctx.bval, ctx.mode, wrt=2).to(x2.dtype)
        if ctx.needs_input_grad[2]:
            if ctx.mode == _lib.GABO_OUT_DISTANCE:
                gb = torch.zeros((), dtype=grad_out.dtype, device=grad_out.device)
            else:       # K = exp(-beta t) with t = d^2 (Gaussian) or d (Laplace): dK/dbeta = -t K
                dist = frobenius_pairwise(x1, x2, 1.0, _lib.GABO_OUT_DISTANCE).to(out.device)
                t = dist * dist if ctx.mode == _lib.GABO_OUT_GAUSSIAN else dist
                gb = -(grad_out * out * t).sum()
            gb = gb.reshape(ctx.beta_shape).to(ctx.beta_device)
        return g1, g2, gb, None


def frobenius_kernel(x1, x2, beta, mode=_lib.GABO_OUT_GAUSSIAN):
    """Differentiable entry point used by SpdFrobeniusGaussianKernel / SpdLogEuclideanGaussianKernel."""
    if torch.is_tensor(beta):
        if beta.numel() != 1:
            raise RuntimeError("gabotorch_amd SPD kernels take a single lengthscale (batch_shape == ())")
        beta = beta.double()
    else:
        beta = torch.tensor(float(beta), dtype=torch.float64)
    return _FrobeniusKernelFunction.apply(x1, x2, beta, int(mode))


def gp_acquisition(kstar, alpha, linv, linv_t, mean, outputscale, kxx, best_f, kind, maximize, out_sign=1.0, need_grad=True):
    """Fused GP posterior + acquisition on the strip kstar (R x n, base kernel values): -> (value R, d value / d kstar R x n or None).
    All tensors fp64 on one HIP device."""
    lib = _lib.load()
    dev = kstar.device
    ks = kstar.contiguous()
    r, n = ks.shape
    value = torch.empty(r, dtype=torch.float64, device=dev)
    grad = torch.empty_like(ks) if need_grad else None
    ptr = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
    with _on(dev):
        _lib.check(lib.gabo_gp_acquisition(ks.data_ptr(), alpha.data_ptr(), ptr(linv), ptr(linv_t), value.data_ptr(), ptr(grad), r, n,
                                           float(mean), float(outputscale), float(kxx), float(best_f), int(kind), 1 if maximize else 0,
                                           float(out_sign), _stream_ptr(dev)), "gabo_gp_acquisition")
    return value, grad


_GP_FACTOR_MESSAGE = "gabo_gp_factor: the training covariance outputscale * K + noise * I is not positive definite (Cholesky pivot <= 0)"


def gp_factor(kbase, y, outputscale, noise, mean, defer_check=False, on_fail=None, want_kinv=False):
    """Prediction cache of the exact GP in one launch (gabo_gp_factor): kbase n x n BASE kernel matrix of the training set, y its targets ->
    (L^-1, L^-T, alpha) with L = chol(outputscale kbase + noise I), alpha = (outputscale kbase + noise I)^-1 (y - mean); n <= GABO_GP_FACTOR_MAX_N.
    Raises like torch.linalg.cholesky when the matrix is not positive definite - at once, or (defer_check=True) at the next check_deferred();
    on_fail: called before that error is raised (the caller drops whatever it built on the factor).  All tensors fp64 on one HIP device."""
    lib = _lib.load()
    dev = kbase.device
    kb, yy = kbase.contiguous(), y.to(dev, torch.float64).contiguous()
    n = kb.shape[-1]
    _require(dev, kbase=kb, y=yy)
    linv = torch.empty(n, n, dtype=torch.float64, device=dev)
    linv_t = torch.empty(n, n, dtype=torch.float64, device=dev)
    alpha = torch.empty(n, dtype=torch.float64, device=dev)
    kinv = torch.empty(n, n, dtype=torch.float64, device=dev) if want_kinv else None
    status = _status_word(dev, force=True)
    with _on(dev):
        rc = lib.gabo_gp_factor(kb.data_ptr(), yy.data_ptr(), n, float(outputscale), float(noise), float(mean), linv.data_ptr(), linv_t.data_ptr(),
                                alpha.data_ptr(), None if kinv is None else kinv.data_ptr(), status.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "gabo_gp_factor")
    # The status is read back whatever set_error_checking says (torch.linalg.cholesky would raise here too, and a silent garbage factor would
    # poison every acquisition value) - but not HERE: the read-back would park the host behind the launch (~0.1 ms of a 4-ms sweep) while it
    # has the rest of the sweep's launches to issue.  It is registered and checked at the caller's next natural synchronisation point
    # (check_deferred: after the raw samples' scores have been copied to the host, after a solve, or whenever error checking reads a status).
    _raise_if_not_spd(status, "gabo_gp_factor", message=_GP_FACTOR_MESSAGE, on_fail=on_fail, force=True)
    if defer_check is False:
        check_deferred()
    if want_kinv:          # (+ the symmetric inverse L^-T L^-1: what the fused SPD acquisition kernels take in place of the two factors)
        return linv, linv_t, alpha, kinv
    return linv, linv_t, alpha


_gp_prepare_ws = {}


def spd_gp_prepare(train_mandel, y, beta, mode, outputscale, noise, mean, want_factors=True, on_fail=None):
    """Everything an acquisition sweep needs from a fitted exact GP with an affine-invariant kernel, from ONE host call (gabo_spd_gp_prepare):
    the training Gram matrix, (L^-1, L^-T, alpha) of gp_factor, the entry-major training factors of spd_acq_prepare_train (d_vec x n, or None) and
    the symmetric inverse A = L^-T L^-1 -> (linv, linv_t, alpha, factors, kinv).
    The Cholesky status is checked at the next check_deferred() (as gp_factor(defer_check=True)); on_fail as there."""
    lib = _lib.load()
    dev = train_mandel.device
    x = train_mandel if train_mandel.is_contiguous() else train_mandel.contiguous()
    yy = y if (y.device == dev and y.dtype == torch.float64 and y.is_contiguous()) else y.to(dev, torch.float64).contiguous()
    if not (x.is_cuda and x.dtype == torch.float64 and x.dim() == 2 and yy.numel() == x.shape[0]):
        raise TypeError("spd_gp_prepare: train_mandel must be an n x d_vec fp64 tensor on a HIP device and y its n targets")
    n, dv = x.shape
    d = _mandel_dim(dv)
    # one allocation for the outputs: [L^-1 | L^-T | alpha | A = L^-T L^-1 | factors]
    nn = n * n
    flat = torch.empty(3 * nn + n + (dv * n if want_factors else 0), dtype=torch.float64, device=dev)
    base = flat.data_ptr()
    stream = _stream_ptr(dev)
    key = (dev.index, stream, n, d)
    ent = _gp_prepare_ws.get(key)
    if ent is None:
        if len(_gp_prepare_ws) > 16:
            _gp_prepare_ws.clear()
        wsb = int(lib.gabo_spd_gp_prepare_workspace_bytes(n, d))
        ent = _gp_prepare_ws[key] = (torch.empty(wsb, dtype=torch.uint8, device=dev), wsb)
    ws, wsb = ent
    status, fstatus = _status_word(dev), _status_word(dev, force=True)
    with _on(dev):
        rc = lib.gabo_spd_gp_prepare(x.data_ptr(), yy.data_ptr(), n, d, beta, mode, outputscale, noise, mean, base, base + 8 * nn, base + 16 * nn,
                                     base + 8 * (2 * nn + n), base + 8 * (3 * nn + n) if want_factors else None, ws.data_ptr(), wsb,
                                     status.data_ptr(), fstatus.data_ptr(), stream)
    _check_launch(rc, status, "gabo_spd_gp_prepare", on_fail=on_fail)
    _raise_if_not_spd(fstatus, "gabo_gp_factor", message=_GP_FACTOR_MESSAGE, on_fail=on_fail, force=True)
    linv, linv_t, alpha = flat[:nn].view(n, n), flat[nn:2 * nn].view(n, n), flat[2 * nn:2 * nn + n]
    kinv = flat[2 * nn + n:3 * nn + n].view(n, n)
    factors = flat[3 * nn + n:].view(dv, n) if want_factors else None
    return linv, linv_t, alpha, factors, kinv


_mll_large_ws = {}


def _gp_mll_large(e, y, theta, outputscale, noise, mean, gram, want_w):
    """gabo_gp_mll_large on contiguous device tensors -> (out (6,) device tensor, W or None); the workspace is kept per (device, n)."""
    lib = _lib.load()
    dev, n = e.device, e.shape[-1]
    key = (dev, n)
    ws = _mll_large_ws.get(key)
    if ws is None:
        if len(_mll_large_ws) > 4:
            _mll_large_ws.clear()
        ws = _mll_large_ws[key] = torch.empty(int(lib.gabo_gp_mll_large_workspace_bytes(n)) // 8 + 1, dtype=torch.float64, device=dev)
    out = torch.empty(6, dtype=torch.float64, device=dev)
    w = torch.empty(n, n, dtype=torch.float64, device=dev) if want_w else None
    with _on(dev):
        _lib.check(lib.gabo_gp_mll_large(e.data_ptr(), y.data_ptr(), n, float(theta), float(outputscale), float(noise), float(mean),
                                         1 if gram else 0, out.data_ptr(), None if w is None else w.data_ptr(), ws.data_ptr(),
                                         ws.numel() * 8, _stream_ptr(dev)), "gabo_gp_mll_large")
    return out, w


def gp_mll(e, y, theta, outputscale, noise, mean):
    """One evaluation of the exact-GP marginal log likelihood with K = outputscale * exp(-theta * e) + noise * I (gabo_gp_mll; the tiled
    gabo_gp_mll_large beyond GABO_GP_MLL_MAX_N points): -> [ll, dll/dtheta, dll/doutputscale, dll/dnoise, dll/dmean,
    not_positive_definite] as Python floats (one device->host copy).  e: n x n fp64 on a HIP device (n <= GABO_GP_MLL_LARGE_MAX_N), y: n."""
    lib = _lib.load()
    dev = e.device
    n = e.shape[-1]
    if e.dim() != 2 or e.shape[0] != n or y.numel() != n or not e.is_contiguous() or not y.is_contiguous():
        raise RuntimeError("gp_mll: e must be a contiguous n x n matrix and y a contiguous vector of n targets")
    if n > _lib.GABO_GP_MLL_MAX_N:
        return _gp_mll_large(e, y, theta, outputscale, noise, mean, False, False)[0].tolist()
    # the six results go straight into page-locked host memory the kernel addresses itself (one buffer per device): an L-BFGS evaluation is the launch
    # and a stream wait - no device buffer, no copy back (a surrogate fit is ~35 of these; worth 4 % of it: 1.57 -> 1.51 ms on 5 ... 20
    # observations - the rest is scipy's L-BFGS-B and the Python chain rule around the launch)
    ent = _mll_host_out.get(dev.index)
    if ent is None:
        host = torch.zeros(8, dtype=torch.float64).pin_memory()
        ent = _mll_host_out[dev.index] = (host, host.numpy())
    host, host_np = ent
    with _on(dev):
        stream = _stream_ptr(dev)
        _lib.check(lib.gabo_gp_mll(e.data_ptr(), y.data_ptr(), n, float(theta), float(outputscale), float(noise), float(mean),
                                   host.data_ptr(), stream), "gabo_gp_mll")
        torch.cuda.current_stream(dev).synchronize()
    return host_np[:6].tolist()


_mll_host_out = {}


def gp_mll_gram(k, y, outputscale, noise, mean, want_w=True):
    """gabo_gp_mll_gram: exact-GP marginal log likelihood for Ky = outputscale * k + noise * I with k an arbitrary base Gram matrix
    (n x n fp64 on a HIP device, n <= GABO_GP_MLL_LARGE_MAX_N).  -> (out (6,) device tensor: ll, 0, dll/doutputscale, dll/dnoise,
    dll/dmean, not-positive-definite flag;  W = alpha alpha^T - Ky^-1 (n x n) or None).  No host synchronisation."""
    lib = _lib.load()
    dev = k.device
    n = k.shape[-1]
    if k.dim() != 2 or k.shape[0] != n or y.numel() != n:
        raise RuntimeError("gp_mll_gram: k must be an n x n matrix and y a vector of n targets")
    k, y = k.contiguous(), y.contiguous()
    if n > _lib.GABO_GP_MLL_MAX_N:
        return _gp_mll_large(k, y, 0.0, outputscale, noise, mean, True, want_w)
    out = torch.empty(6, dtype=torch.float64, device=dev)
    w = torch.empty(n, n, dtype=torch.float64, device=dev) if want_w else None
    with _on(dev):
        _lib.check(lib.gabo_gp_mll_gram(k.data_ptr(), y.data_ptr(), n, float(outputscale), float(noise), float(mean), out.data_ptr(),
                                        None if w is None else w.data_ptr(), _stream_ptr(dev)), "gabo_gp_mll_gram")
    return out, w


def spd_acq_prepare_train(train_mandel):
    """Entry-major Cholesky factors of the training matrices for spd_acq_eval (d_vec x n)."""
    lib = _lib.load()
    x = train_mandel.contiguous()
    n, dv = x.shape
    d = _mandel_dim(dv)
    out = torch.empty(dv, n, dtype=torch.float64, device=x.device)
    status = _status_word(x.device)
    with _on(x.device):
        _lib.check(lib.gabo_spd_acq_prepare_train(x.data_ptr(), out.data_ptr(), n, d, status.data_ptr(), _stream_ptr(x.device)),
                   "gabo_spd_acq_prepare_train")
    _raise_if_not_spd(status, "gabo_spd_acq_prepare_train")
    return out


def spd_acq_eval(x_mandel, train_factors, alpha, linv, linv_t, beta, mode, mean, outputscale, kxx, best_f, kind, maximize,
                 out_sign=1.0, need_grad=True, active_ptr=None, out=None):
    """Single-launch acquisition value (R) and Euclidean gradient (R x d_vec, Mandel) at SPD candidates."""
    lib = _lib.load()
    dev = _device_for(train_factors, x_mandel)
    x = _prep(x_mandel, dev).contiguous()
    _require(dev, train_factors=train_factors, alpha=alpha, linv=linv, linv_t=linv_t)
    r, dv = x.shape
    n = train_factors.shape[1]
    if out is not None:           # (value, grad) buffers to update in place: masked candidates keep their previous entries
        value, grad = out
    else:
        value = torch.empty(r, dtype=torch.float64, device=dev)
        grad = torch.empty_like(x) if need_grad else None
    affine = (int(mode) & 24) == 0           # only the affine-invariant metric spills logm(M_j) to scratch
    scratch = torch.empty(r * dv * n, dtype=torch.float64, device=dev) if (need_grad and affine) else None
    status = _status_word(dev)
    ptr = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
    with _on(dev):
        _lib.check(lib.gabo_spd_acq_eval(x.data_ptr(), train_factors.data_ptr(), alpha.data_ptr(), ptr(linv), ptr(linv_t),
                                         value.data_ptr(), ptr(grad), ptr(scratch), r, n, _mandel_dim(dv), float(beta), int(mode),
                                         float(mean), float(outputscale), float(kxx), float(best_f), int(kind), 1 if maximize else 0,
                                         float(out_sign), active_ptr, status.data_ptr(), _stream_ptr(dev)), "gabo_spd_acq_eval")
    _raise_if_not_spd(status, "gabo_spd_acq_eval")
    return value, grad


def spd_sample(n, d, min_eig, max_eig, seed, device, mandel=False, first=0):
    """n random SPD matrices (n, d, d) - or Mandel vectors (n, d_vec) - drawn on the device with spd_sample's distribution: samples
    first ... first + n - 1 of the stream keyed by `seed` (a rank's shard of the raw samples when first > 0)."""
    lib = _lib.load()
    dev = torch.device(device)
    out = torch.empty((n, d * (d + 1) // 2) if mandel else (n, d, d), dtype=torch.float64, device=dev)
    with _on(dev):
        _lib.check(lib.gabo_spd_sample_range(out.data_ptr(), int(first), n, d, float(min_eig), float(max_eig), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                             1 if mandel else 0, _stream_ptr(dev)), "gabo_spd_sample_range")
    return out


def nested_sphere_epilogue(rotated, dist_to_axis, mode=0):
    """Per-point part of the nested-sphere projection on already rotated points (..., d) -> (..., d-1) [mode 0] / (..., d) [mode 1]."""
    lib = _lib.load()
    out_device = rotated.device
    dev = _device_for(rotated)
    u = _prep(rotated, dev).contiguous()
    d = u.shape[-1]
    n = u.numel() // d
    out = torch.empty(u.shape[:-1] + ((d - 1) if mode == 0 else d,), dtype=torch.float64, device=dev)
    with _on(dev):
        _lib.check(lib.gabo_nested_sphere_epilogue(u.data_ptr(), out.data_ptr(), n, d, float(dist_to_axis), int(mode), _stream_ptr(dev)),
                   "gabo_nested_sphere_epilogue")
    return _out(out, out_device)


def nested_sphere_epilogue_backward(rotated, grad_out, dist_to_axis):
    lib = _lib.load()
    out_device = rotated.device
    dev = _device_for(rotated, grad_out)
    u = _prep(rotated, dev).contiguous()
    g = _prep(grad_out, dev).contiguous()
    d = u.shape[-1]
    gu = torch.empty_like(u)
    with _on(dev):
        _lib.check(lib.gabo_nested_sphere_epilogue_backward(u.data_ptr(), g.data_ptr(), gu.data_ptr(), u.numel() // d, d,
                                                            float(dist_to_axis), _stream_ptr(dev)), "gabo_nested_sphere_epilogue_backward")
    return _out(gu, out_device)


class _NestedSphereEpilogue(torch.autograd.Function):
    """mode-0 epilogue, differentiable (first order) in the rotated points."""

    @staticmethod
    def forward(ctx, u, dist):
        ctx.save_for_backward(u)
        ctx.dist = float(dist)
        return nested_sphere_epilogue(u, ctx.dist, 0).to(u.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        return nested_sphere_epilogue_backward(u, g, ctx.dist).to(u.dtype), None


def pack_nested_sphere_axes(sphere_axes, dim):
    """[axis of S^(dim-1) (dim entries), axis of the next subsphere (dim - 1 entries), ...] -> one float64 numpy vector, level after level
    (the layout of gabo_nested_sphere_frames / _reconstruction / _fit_evaluate)."""
    import numpy as np
    parts = []
    for k, a in enumerate(sphere_axes):
        v = (a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)).reshape(-1)
        if v.size != dim - k:
            raise RuntimeError(f"nested-sphere axis of level {k} must have {dim - k} entries, got {v.size}")
        parts.append(v)
    return np.ascontiguousarray(np.concatenate(parts))


def _nested_sphere_frames(sphere_axes, sphere_distances, dim, dev):
    """(frames, distances) device tensors of the fused nested-sphere launches from the lists the reference's functions take."""
    lib = _lib.load()
    levels = len(sphere_axes)
    axes = torch.cat([a.detach().reshape(-1).to(dev, torch.float64) for a in sphere_axes])
    if axes.numel() != levels * dim - levels * (levels - 1) // 2:
        raise RuntimeError("nested-sphere axes: level k must have dim - k entries")
    dists = torch.cat([(d.detach().reshape(-1)[:1].to(dev, torch.float64) if torch.is_tensor(d)
                        else torch.tensor([float(d)], dtype=torch.float64, device=dev)) for d in sphere_distances])
    frames = torch.empty(axes.numel() + 2 * levels, dtype=torch.float64, device=dev)
    with _on(dev):
        _lib.check(lib.gabo_nested_sphere_frames(axes.data_ptr(), frames.data_ptr(), dim, levels, _stream_ptr(dev)), "gabo_nested_sphere_frames")
    return frames, dists


def _level_offsets(dim, levels):
    return [k * dim - k * (k - 1) // 2 for k in range(levels + 1)]


def nested_sphere_project_all(x, sphere_axes, sphere_distances):
    """projection_from_sphere_to_subsphere (nested_spheres_utils.py:117-146) in one launch, no autograd: x (N, D) -> the list
    [x, x_{D-1}, ..., x_{D-levels}] of the points on every nested subsphere."""
    lib = _lib.load()
    dev = _device_for(x)
    X = _prep(x, dev).contiguous()
    n, dim = int(X.shape[0]), int(X.shape[1])
    levels = len(sphere_axes)
    frames, dists = _nested_sphere_frames(sphere_axes, sphere_distances, dim, dev)
    off = _level_offsets(dim, levels)
    z = torch.empty(n, dim - levels, dtype=torch.float64, device=dev)
    store = torch.empty(n, off[-1], dtype=torch.float64, device=dev)
    with _on(dev):
        _lib.check(lib.gabo_nested_sphere_project(X.data_ptr(), frames.data_ptr(), dists.data_ptr(), z.data_ptr(), store.data_ptr(), n, dim, levels,
                                                  _stream_ptr(dev)), "gabo_nested_sphere_project")
    return [store[:, off[k]:off[k + 1]] for k in range(levels)] + [z]


def nested_sphere_lift_all(x_subsphere, sphere_axes, sphere_distances):
    """projection_from_subsphere_to_sphere (nested_spheres_utils.py:182-218) in one launch, no autograd: x_subsphere (N, D - levels) -> the
    list [x_subsphere, x_{D-levels+1}, ..., x_D] (the axes are used in reverse order, as in the reference)."""
    lib = _lib.load()
    dev = _device_for(x_subsphere)
    Z = _prep(x_subsphere, dev).contiguous()
    levels = len(sphere_axes)
    n, dim = int(Z.shape[0]), int(Z.shape[1]) + levels
    frames, dists = _nested_sphere_frames(sphere_axes, sphere_distances, dim, dev)
    off = _level_offsets(dim, levels)
    store = torch.empty(n, off[-1], dtype=torch.float64, device=dev)
    with _on(dev):
        _lib.check(lib.gabo_nested_sphere_lift(Z.data_ptr(), frames.data_ptr(), dists.data_ptr(), None, store.data_ptr(), n, dim, levels,
                                               _stream_ptr(dev)), "gabo_nested_sphere_lift")
    return [Z] + [store[:, off[k]:off[k + 1]] for k in range(levels - 1, -1, -1)]


class NestedSphereReconstruction:
    """min_error_reconstruction_cost (nested_spheres_optimization.py:20-38) for FIXED data, subsphere points and axes, as a function of the
    distances to the axes: value and gradient of P parameter sets in one launch (gabo_nested_sphere_reconstruction)."""

    def __init__(self, x_data, x_subsphere, sphere_axes):
        lib = _lib.load()
        dev = self.device = _device_for(x_data, x_subsphere)
        self.x = _prep(x_data, dev).contiguous()
        self.z = _prep(x_subsphere, dev).contiguous()
        self.N, self.D = int(self.x.shape[0]), int(self.x.shape[1])
        self.L = self.D - int(self.z.shape[1])
        if self.x.dim() != 2 or self.z.dim() != 2 or self.z.shape[0] != self.N or self.L < 1 or len(sphere_axes) != self.L:
            raise RuntimeError(f"shapes: x_data {tuple(self.x.shape)}, x_subsphere {tuple(self.z.shape)}, {len(sphere_axes)} axes")
        axes = torch.as_tensor(pack_nested_sphere_axes(sphere_axes, self.D), device=dev)
        self.frames = torch.empty(axes.numel() + 2 * self.L, dtype=torch.float64, device=dev)
        with _on(dev):
            _lib.check(lib.gabo_nested_sphere_frames(axes.data_ptr(), self.frames.data_ptr(), self.D, self.L, _stream_ptr(dev)), "gabo_nested_sphere_frames")
        self._buffers = {}

    def evaluate(self, distances, grad=True):
        """distances: (P, L) or (L,) numpy -> cost (P,) [, d cost / d distances (P, L)] as numpy (one copy in, one launch, one copy out)."""
        import numpy as np
        lib = _lib.load()
        r = np.asarray(distances, dtype=np.float64)
        single = r.ndim == 1
        r = np.ascontiguousarray(r.reshape(-1, self.L))
        P = r.shape[0]
        ent = self._buffers.get(P)
        if ent is None:
            ws = max(int(lib.gabo_nested_sphere_reconstruction_workspace_bytes(P, max(self.N, 1), self.D, self.L)), 16)
            ent = self._buffers[P] = dict(hin=torch.empty(P * self.L, dtype=torch.float64).pin_memory(),
                                          hout=torch.empty(P * (1 + self.L), dtype=torch.float64).pin_memory(),
                                          din=torch.empty(P * self.L, dtype=torch.float64, device=self.device),
                                          dout=torch.empty(P * (1 + self.L), dtype=torch.float64, device=self.device),
                                          ws=torch.empty(ws, dtype=torch.uint8, device=self.device))
        ent["hin"].numpy()[:] = r.ravel()
        ent["din"].copy_(ent["hin"], non_blocking=True)
        dout = ent["dout"]
        with torch.cuda.device(self.device):
            _lib.check(lib.gabo_nested_sphere_reconstruction(self.x.data_ptr(), self.z.data_ptr(), self.frames.data_ptr(), ent["din"].data_ptr(),
                                                             dout.data_ptr(), dout[P:].data_ptr() if grad else None, P, self.N, self.D, self.L,
                                                             ent["ws"].data_ptr(), ent["ws"].numel(), _stream_ptr(self.device)),
                       "gabo_nested_sphere_reconstruction")
        ent["hout"].copy_(dout, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        out = ent["hout"].numpy()
        cost = out[:P].copy()
        if not grad:
            return cost[0] if single else cost
        g = out[P:].reshape(P, self.L).copy()
        return (cost[0], g[0]) if single else (cost, g)


def nested_sphere_next(rotated, dist_to_axis):
    """Differentiable mode-0 epilogue."""
    if rotated.requires_grad:
        return _NestedSphereEpilogue.apply(rotated, float(dist_to_axis))
    return nested_sphere_epilogue(rotated, dist_to_axis, 0)


class SpdTcg:
    """Device-resident truncated CG of the SPD trust regions (gabo_spd_tcg_*): thin handle around the workspace."""

    def __init__(self, r, d, n_constraints, device):
        self.lib = _lib.load()
        self.r, self.d, self.c, self.dev = int(r), int(d), int(n_constraints), device
        self.wsb = self.lib.gabo_spd_tcg_workspace_bytes(self.r, self.d, self.c)
        self.ws = torch.zeros(self.wsb // 8 + 1, dtype=torch.float64, device=device)
        self.x_fd = torch.empty(self.r, d * (d + 1) // 2, dtype=torch.float64, device=device)
        self.any_running = torch.zeros(1, dtype=torch.int32, device=device)
        self.running_ptr = self.ws.data_ptr() + self.lib.gabo_spd_tcg_running_offset(self.r, self.d, self.c)
        self.status = torch.zeros(2, dtype=torch.int32, device=device)

    def begin(self, x, g, gc, fc, active, Delta):
        ptr = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
        self._keep = (x.contiguous(), g.contiguous(), None if gc is None else gc.contiguous(), None if fc is None else fc.contiguous(),
                      active.to(torch.uint8).contiguous(), Delta.contiguous())
        xx, gg, gcc, fcc, act, dl = self._keep
        with _on(self.dev):
            _lib.check(self.lib.gabo_spd_tcg_begin(xx.data_ptr(), gg.data_ptr(), ptr(gcc), ptr(fcc), act.data_ptr(), dl.data_ptr(),
                                                   self.ws.data_ptr(), self.wsb, self.r, self.d, self.c, self.status.data_ptr(),
                                                   _stream_ptr(self.dev)), "gabo_spd_tcg_begin")

    def begin_rand(self, eta0, heta0):
        """use_rand (robust_trust_regions.py:173-181, 407-452), after begin(): start from eta0 with heta0 = hess(x, eta0), no preconditioner."""
        self._keep_rand = (eta0.contiguous(), heta0.contiguous())
        _require(self.dev, eta0=self._keep_rand[0], heta0=self._keep_rand[1])
        with _on(self.dev):
            _lib.check(self.lib.gabo_spd_tcg_begin_rand(self.ws.data_ptr(), self._keep_rand[0].data_ptr(), self._keep_rand[1].data_ptr(),
                                                        self.r, self.d, self.c, _stream_ptr(self.dev)), "gabo_spd_tcg_begin_rand")

    def fd_point(self):
        with _on(self.dev):
            _lib.check(self.lib.gabo_spd_tcg_fd_point(self.ws.data_ptr(), self.x_fd.data_ptr(), self.r, self.d, self.c,
                                                      _stream_ptr(self.dev)), "gabo_spd_tcg_fd_point")
        return self.x_fd

    def step(self, egrad_mandel, neq, delta_cons, theta, kappa, mininner):
        eg = egrad_mandel.contiguous()
        with _on(self.dev):
            _lib.check(self.lib.gabo_spd_tcg_step(self.ws.data_ptr(), eg.data_ptr(), self.any_running.data_ptr(), self.r, self.d,
                                                  self.c, int(neq), float(delta_cons), float(theta), float(kappa), int(mininner),
                                                  _stream_ptr(self.dev)), "gabo_spd_tcg_step")

    def end(self):
        eta = torch.empty(self.r, self.d, self.d, dtype=torch.float64, device=self.dev)
        heta = torch.empty_like(eta)
        stop = torch.empty(self.r, dtype=torch.int32, device=self.dev)
        with _on(self.dev):
            _lib.check(self.lib.gabo_spd_tcg_end(self.ws.data_ptr(), eta.data_ptr(), heta.data_ptr(), stop.data_ptr(), self.r,
                                                 self.d, self.c, _stream_ptr(self.dev)), "gabo_spd_tcg_end")
        return eta, heta, stop.long()


class SpdTr:
    """Device-resident trust-region iteration (gabo_spd_tr_propose / gabo_spd_tr_update) on persistent state tensors."""

    def __init__(self, r, d, n_constraints, acq_params, n_train, device):
        import ctypes
        self.lib = _lib.load()
        self.r, self.d, self.c, self.n, self.dev = int(r), int(d), int(n_constraints), int(n_train), device
        self.acq = acq_params                      # _lib.AcqParams; the tensors it points to are owned by the caller
        self.acq_ref = ctypes.byref(self.acq)
        self.wsb = self.lib.gabo_spd_tr_workspace_bytes(self.r, self.d, self.c, self.n)
        self.ws = torch.zeros(self.wsb // 8 + 1, dtype=torch.float64, device=device)
        self.x_prop = torch.zeros(self.r, d, d, dtype=torch.float64, device=device)
        self.any_active = torch.ones(1, dtype=torch.int32, device=device)
        self.status = torch.zeros(2, dtype=torch.int32, device=device)

    def propose(self, x, g, Delta, active, gc, fc, neq, delta_cons, theta, kappa, mininner, maxinner):
        _require(self.dev, x=x, g=g, Delta=Delta, active=active, gc=gc, fc=fc)
        ptr = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
        with _on(self.dev):
            _lib.check(self.lib.gabo_spd_tr_propose(x.data_ptr(), g.data_ptr(), Delta.data_ptr(), active.data_ptr(), ptr(gc), ptr(fc),
                                                    self.acq_ref, self.ws.data_ptr(), self.wsb, self.x_prop.data_ptr(), self.r, self.d,
                                                    self.c, int(neq), float(delta_cons), float(theta), float(kappa), int(mininner),
                                                    int(maxinner), self.any_active.data_ptr(), self.status.data_ptr(),
                                                    _stream_ptr(self.dev)), "gabo_spd_tr_propose")
        return self.x_prop

    def solve(self, x, fx, g, ng, Delta, active, iters, kinds, bounds, strict, delta_cons, theta, kappa, mininner, maxinner, delta_bar,
              rho_prime, rho_regularization, mingradnorm, maxiter, lift=None, record=None):
        """The whole solve in one launch (built-in eigenvalue constraints `kinds`/`bounds`, or none).  lift: the (w, x0, p) of
        nested_spd_lift_prepare when some kinds are the nested ones (bounds stated in the original space of a nested SPD mapping).
        record: (K, r, d*d + 2) tensor pre-filled with NaN -> per-iteration record of the launch (gabo_tr_solve_record)."""
        _require(self.dev, x=x, fx=fx, g=g, ng=ng, Delta=Delta, active=active, iters=iters, record=record)
        import ctypes
        nc = len(kinds)
        ck = (ctypes.c_int * max(nc, 1))(*kinds)
        cb = (ctypes.c_double * max(nc, 1))(*bounds)
        if lift is not None:
            lw, lx0, lp = lift                         # (the order nested_spd_lift_prepare returns them in)
            _require(self.dev, lift_w=lw, lift_p=lp, lift_x0=lx0)
            self._lift_ref = lift                      # (keeps the buffers alive until the next solve)
            lift_args = (lw.data_ptr(), lp.data_ptr(), lx0.data_ptr(), int(lw.shape[0]))
        else:
            lift_args = (None, None, None, 0)
        with _on(self.dev):
            if record is not None:
                assert tuple(record.shape[1:]) == (self.r, self.d * self.d + 2)
                _lib.check(self.lib.gabo_tr_solve_record(record.data_ptr(), int(record.shape[0])), "gabo_tr_solve_record")
            _lib.check(self.lib.gabo_spd_tr_solve(x.data_ptr(), fx.data_ptr(), g.data_ptr(), ng.data_ptr(), Delta.data_ptr(),
                                                  active.data_ptr(), iters.data_ptr(), self.acq_ref, nc, ck, cb, 1 if strict else 0,
                                                  self.ws.data_ptr(), self.wsb, self.r, self.d, float(delta_cons), float(theta),
                                                  float(kappa), int(mininner), int(maxinner), float(delta_bar), float(rho_prime),
                                                  float(rho_regularization), float(mingradnorm), int(maxiter), *lift_args,
                                                  self.status.data_ptr(), _stream_ptr(self.dev)), "gabo_spd_tr_solve")

    def update(self, x, fx, g, ng, Delta, active, iters, invalid, delta_bar, rho_prime, rho_regularization, mingradnorm, maxiter):
        _require(self.dev, x=x, fx=fx, g=g, ng=ng, Delta=Delta, active=active, iters=iters, invalid=invalid)
        with _on(self.dev):
            _lib.check(self.lib.gabo_spd_tr_update(x.data_ptr(), fx.data_ptr(), g.data_ptr(), ng.data_ptr(), Delta.data_ptr(),
                                                   active.data_ptr(), iters.data_ptr(), None if invalid is None else invalid.data_ptr(),
                                                   self.x_prop.data_ptr(), self.ws.data_ptr(), self.r, self.d, self.c, self.n,
                                                   float(delta_bar), float(rho_prime), float(rho_regularization), float(mingradnorm),
                                                   int(maxiter), self.any_active.data_ptr(), _stream_ptr(self.dev)), "gabo_spd_tr_update")


def sphere_acq_eval(x, acq_params, need_grad=True):
    """Single-launch acquisition value (R) and Euclidean gradient (R x dim) at points of the sphere; acq_params: _lib.SphereAcqParams."""
    import ctypes
    lib = _lib.load()
    dev = _device_for(x)
    xx = _prep(x, dev).contiguous()           # a float32 / host tensor would otherwise be read as fp64 device memory
    r = xx.shape[0]
    value = torch.empty(r, dtype=torch.float64, device=dev)
    grad = torch.empty_like(xx) if need_grad else None
    with _on(dev):
        _lib.check(lib.gabo_sphere_acq_eval(xx.data_ptr(), ctypes.byref(acq_params), value.data_ptr(), None if grad is None else grad.data_ptr(),
                                            r, _stream_ptr(dev)), "gabo_sphere_acq_eval")
    return value, grad


class SphereTr:
    """Device-resident trust-region iteration on the sphere (gabo_sphere_tr_*): same interface as SpdTr."""

    def __init__(self, r, dim, n_constraints, acq_params, device, exact_hessian=False):
        import ctypes
        self.lib = _lib.load()
        self.r, self.d, self.c, self.dev = int(r), int(dim), int(n_constraints), device
        self.exact = 1 if exact_hessian else 0
        self.acq = acq_params
        self.acq_ref = ctypes.byref(self.acq)
        self.wsb = self.lib.gabo_sphere_tr_workspace_bytes(self.r, self.d, self.c)
        self.ws = torch.zeros(self.wsb // 8 + 1, dtype=torch.float64, device=device)
        self.x_prop = torch.zeros(self.r, dim, dtype=torch.float64, device=device)
        self.any_active = torch.ones(1, dtype=torch.int32, device=device)

    def propose(self, x, g, Delta, active, gc, fc, neq, delta_cons, theta, kappa, mininner, maxinner):
        _require(self.dev, x=x, g=g, Delta=Delta, active=active, gc=gc, fc=fc)
        ptr = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
        with _on(self.dev):
            _lib.check(self.lib.gabo_sphere_tr_propose(x.data_ptr(), g.data_ptr(), Delta.data_ptr(), active.data_ptr(), ptr(gc), ptr(fc),
                                                       self.acq_ref, self.ws.data_ptr(), self.wsb, self.x_prop.data_ptr(), self.r, self.c,
                                                       int(neq), float(delta_cons), float(theta), float(kappa), int(mininner),
                                                       int(maxinner), self.exact, self.any_active.data_ptr(), _stream_ptr(self.dev)),
                       "gabo_sphere_tr_propose")
        return self.x_prop

    def update(self, x, fx, g, ng, Delta, active, iters, invalid, delta_bar, rho_prime, rho_regularization, mingradnorm, maxiter):
        _require(self.dev, x=x, fx=fx, g=g, ng=ng, Delta=Delta, active=active, iters=iters, invalid=invalid)
        with _on(self.dev):
            _lib.check(self.lib.gabo_sphere_tr_update(x.data_ptr(), fx.data_ptr(), g.data_ptr(), ng.data_ptr(), Delta.data_ptr(),
                                                      active.data_ptr(), iters.data_ptr(), None if invalid is None else invalid.data_ptr(),
                                                      self.ws.data_ptr(), self.r, self.d, self.c, float(delta_bar), float(rho_prime),
                                                      float(rho_regularization), float(mingradnorm), int(maxiter),
                                                      self.any_active.data_ptr(), _stream_ptr(self.dev)), "gabo_sphere_tr_update")

    def solve(self, x, fx, g, ng, Delta, active, iters, kinds, bounds, strict, delta_cons, theta, kappa, mininner, maxinner, delta_bar,
              rho_prime, rho_regularization, mingradnorm, maxiter, record=None):
        _require(self.dev, x=x, fx=fx, g=g, ng=ng, Delta=Delta, active=active, iters=iters, record=record)
        assert not kinds, "the sphere has no built-in constraints"
        with _on(self.dev):
            if record is not None:                     # (K, r, dim + 2), pre-filled with NaN: gabo_tr_solve_record
                assert tuple(record.shape[1:]) == (self.r, self.d + 2)
                _lib.check(self.lib.gabo_tr_solve_record(record.data_ptr(), int(record.shape[0])), "gabo_tr_solve_record")
            _lib.check(self.lib.gabo_sphere_tr_solve(x.data_ptr(), fx.data_ptr(), g.data_ptr(), ng.data_ptr(), Delta.data_ptr(),
                                                     active.data_ptr(), iters.data_ptr(), self.acq_ref, self.ws.data_ptr(), self.wsb, self.r,
                                                     float(theta), float(kappa), int(mininner), int(maxinner), self.exact, float(delta_bar),
                                                     float(rho_prime), float(rho_regularization), float(mingradnorm), int(maxiter),
                                                     _stream_ptr(self.dev)), "gabo_sphere_tr_solve")


def sphere_manifold_op(op, x, u, v=None, w=None):
    """Batched sphere-manifold operation (one of _lib.GABO_SPH_*) on (..., dim) tensors."""
    lib = _lib.load()
    first = torch.as_tensor(x)
    out_device = first.device
    dev = _device_for(*[torch.as_tensor(t_) for t_ in (x, u, v, w) if t_ is not None])
    ts = [None if t_ is None else _prep(torch.as_tensor(t_), dev) for t_ in (x, u, v, w)]
    shape = torch.broadcast_shapes(*[t_.shape for t_ in ts if t_ is not None])
    X, U, V, W = [None if t_ is None else t_.expand(shape).contiguous() for t_ in ts]
    dim = shape[-1]
    n = X.numel() // dim
    out = torch.empty(shape[:-1] if op == _lib.GABO_SPH_DIST else shape, dtype=torch.float64, device=dev)
    ptr = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
    with _on(dev):
        _lib.check(lib.gabo_sphere_manifold_op(int(op), ptr(X), ptr(U), ptr(V), ptr(W), out.data_ptr(), n, dim, _stream_ptr(dev)),
                   "gabo_sphere_manifold_op")
    return _out(out, out_device)
