"""ORACLE (test infrastructure only - never imported by the product): the restart selection of an acquisition sweep as the device kernel
`sweep_select_kernel` (gabotorch_amd/csrc/spd_sweep.hip) performs it, restated in numpy.

What is restated: [3P] botorch.optim.initializers.initialize_q_batch_nonneg - the heuristic gen_batch_initial_conditions_manifold applies to
non-negative acquisition functions (BoManifolds/manifold_optimization/manifold_optimize.py:296-317) - with its sampling without replacement written
as the exponential race torch.multinomial runs (keys w_i / E_i, E_i ~ Exp(1), the n largest keys), and the library's counter-based random stream
(Philox4x32-10, Salmon et al. SC'11: key = seed, counter = (sample index, draw 0, tag)).  PARITY UNPINNED for the heuristic itself (botorch is not
in /root/reference and not installed: SURVEY App. B); the Philox rounds are checked against the published known-answer vectors in
tests/test_oracle_selection.py.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
SELECT_TAG = 0x73656C65


def philox4x32_10(counter, key):
    """counter: (..., 4) uint32, key: (..., 2) uint32 -> (..., 4) uint32, ten rounds"""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k = np.array(np.broadcast_to(np.asarray(key, dtype=np.uint32), c.shape[:-1] + (2,)), dtype=np.uint32, copy=True)
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[..., 0].astype(np.uint64)
            p1 = M1 * c[..., 2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[..., 1] ^ k[..., 0]
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[..., 3] ^ k[..., 1]
            c = np.stack([n0, (p1 & mask).astype(np.uint32), n2, (p0 & mask).astype(np.uint32)], axis=-1)
            k = np.stack([k[..., 0] + W0, k[..., 1] + W1], axis=-1)
    return c


def selection_uniforms(seed, total, tag=SELECT_TAG):
    """u1 in (0, 1] of draw 0 of items 0 ... total - 1 (Philox.uniform2 of gabo_philox.hpp)"""
    idx = np.arange(total, dtype=np.uint64)
    ctr = np.stack([(idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                    np.zeros(total, dtype=np.uint32), np.full(total, tag, dtype=np.uint32)], axis=-1)
    out = philox4x32_10(ctr, [np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)])
    a = ((out[:, 0].astype(np.uint64) << np.uint64(32)) | out[:, 1].astype(np.uint64)) >> np.uint64(11)
    return (a.astype(np.float64) + 1.0) * 2.0 ** -53


def select_nonneg(y, n, seed, eta=1.0, alpha=1e-4):
    """-> (sample index of every restart (n,), keys (total,)) or (None, None) when the heuristic falls back (no positive value, fewer positive
    values than n, a NaN): exactly the cases in which the kernel raises its flag."""
    y = np.asarray(y, dtype=np.float64)
    total = y.shape[0]
    if np.isnan(y).any() or not (y.max() > 0) or int((y > 0).sum()) < n:
        return None, None
    max_idx = int(np.argmax(y))
    max_val = y[max_idx]
    thr = alpha * max_val
    while int((y >= thr).sum()) < n:
        alpha = 0.1 * alpha
        thr = alpha * max_val
    w = np.exp(eta * (y / max_val - 1.0))
    with np.errstate(divide="ignore"):
        keys = np.where(y >= thr, w / -np.log(selection_uniforms(seed, total)), -1.0)
    order = np.lexsort((np.arange(total), -keys))          # descending key, ties by the lower index
    picked = order[:n].copy()
    if max_idx not in picked:
        picked[-1] = max_idx
    return picked, keys


SPHERE_TAG = 0x73706872


def sphere_samples(seed, count, dim):
    """count points uniform on S^(dim-1) as `sphere_sample_kernel` (gabotorch_amd/csrc/spd_sweep.hip) draws them: for sample i, draw k of the Philox stream
    (seed, item i, tag "sphr") gives two uniforms (Philox.uniform2 of gabo_philox.hpp), Box-Muller two normals (coordinates 2k, 2k + 1); the row is then
    normalised - the distribution of [3P] pymanopt's Sphere.rand (randn / norm).  Agreement with the device is to rounding of log / sincospi / sqrt."""
    idx = np.arange(count, dtype=np.uint64)
    out = np.empty((count, dim), dtype=np.float64)
    key = [np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)]
    for k in range((dim + 1) // 2):
        ctr = np.stack([(idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                        np.full(count, k, dtype=np.uint32), np.full(count, SPHERE_TAG, dtype=np.uint32)], axis=-1)
        o = philox4x32_10(ctr, key)
        a = ((o[:, 0].astype(np.uint64) << np.uint64(32)) | o[:, 1].astype(np.uint64)) >> np.uint64(11)
        b = ((o[:, 2].astype(np.uint64) << np.uint64(32)) | o[:, 3].astype(np.uint64)) >> np.uint64(11)
        u1 = (a.astype(np.float64) + 1.0) * 2.0 ** -53
        u2 = b.astype(np.float64) * 2.0 ** -53
        r = np.sqrt(-2.0 * np.log(u1))
        out[:, 2 * k] = r * np.cos(2.0 * np.pi * u2)
        if 2 * k + 1 < dim:
            out[:, 2 * k + 1] = r * np.sin(2.0 * np.pi * u2)
    return out / np.sqrt((out * out).sum(1))[:, None]
