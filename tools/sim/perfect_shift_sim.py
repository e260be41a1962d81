"""Simulation for the SPD backward: implicit QL with eigenvectors where the shift of stage l is the eigenvalue a first,
eigenvalue-only pass delivered for position l ("perfect shift": one sweep deflates e_l in exact arithmetic), on the benchmark distribution
(N = 4096, d = 10: M = L^-1 B L^-T of random SPD pairs, eigenvalues U[0.05, 5]).  Counts, per WAVE of 64 problems (a wave leaves a stage with its
slowest lane), the sweep steps of (a) today's Wilkinson-shift QL with vectors and (b) values-only pass + perfect-shift vector pass with Wilkinson
sweeps as the fall-back when the deflation test fails, and reports the accuracy of V diag(log lam) V^T against numpy."""
import sys
import numpy as np

D = int(sys.argv[1]) if len(sys.argv) > 1 else 10
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 200       # waves
EPS2 = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-26
rng = np.random.default_rng(0)


def rand_spd(n):
    q = np.linalg.qr(rng.standard_normal((n, D, D)))[0]
    lam = rng.uniform(0.05, 5.0, (n, D))
    return np.einsum("nab,nb,ncb->nac", q, lam, q)


def tridiag(M):
    """Householder tridiagonalisation (numpy/scipy-free, batched via loop): returns dg, e, Q with Q^T M Q = T"""
    import scipy.linalg as sl
    dg, e, Qs = [], [], []
    for m in M:
        h, q = sl.hessenberg(m, calc_q=True)
        dg.append(np.diag(h).copy())
        e.append(np.r_[np.diag(h, -1), 0.0])
        Qs.append(q)
    return np.array(dg), np.array(e), np.array(Qs)


def ql_sweep(dg, e, z, l, shift, active):
    """one implicit QL sweep of stage l (rows active) with the given shift; the device recurrence (spd_eigvec.hpp) vectorised over the batch"""
    n = dg.shape[0]
    g = dg[:, D - 1] - shift
    s = np.ones(n); c = np.ones(n); p = np.zeros(n)
    el = e[:, l].copy()
    for i in range(D - 2, l - 1, -1):
        ei = el if i == l else e[:, i]
        f = s * ei
        b = c * ei
        g = np.copysign(np.maximum(np.abs(g), 1e-150), g)
        r = np.sqrt(f * f + g * g)
        e_new = r
        s = f / r
        c = g / r
        g2 = dg[:, i + 1] - p
        rr = (dg[:, i] - g2) * s + 2.0 * c * b
        p = s * rr
        dnew = g2 + p
        g = c * rr - b
        upd = active
        e[:, i + 1] = np.where(upd, e_new, e[:, i + 1])
        dg[:, i + 1] = np.where(upd, dnew, dg[:, i + 1])
        if z is not None:
            zi, zj = z[:, :, i].copy(), z[:, :, i + 1].copy()
            z[:, :, i + 1] = np.where(upd[:, None], s[:, None] * zi + c[:, None] * zj, zj)
            z[:, :, i] = np.where(upd[:, None], c[:, None] * zi - s[:, None] * zj, zi)
    dg[:, l] = np.where(active, dg[:, l] - p, dg[:, l])
    e[:, l] = np.where(active, g, e[:, l])
    e[:, D - 1] = 0.0


def wilkinson(dg, e, l):
    sa, sb, se = dg[:, l], dg[:, l + 1], e[:, l]
    delta = 0.5 * (sb - sa)
    root = np.sqrt(delta * delta + se * se)
    return sa - np.copysign(root - np.abs(delta), delta)


def done(dg, e, l):
    s01 = np.abs(dg[:, l]) + np.abs(dg[:, l + 1])
    return e[:, l] ** 2 <= EPS2 * s01 * s01


def run(dg0, e0, z0, known=None):
    """-> eigenvalues, z, per-stage sweep counts per problem.  known: eigenvalue per position from a previous pass -> first sweep of a stage uses it"""
    dg, e = dg0.copy(), e0.copy()
    z = None if z0 is None else z0.copy()
    n = dg.shape[0]
    sweeps = np.zeros((n, D - 1), dtype=int)
    for l in range(D - 1):
        for it in range(60):
            dn = done(dg, e, l)
            if dn.all():
                break
            act = ~dn
            if known is not None and it == 0:
                shift = known[:, l]
            else:
                shift = wilkinson(dg, e, l)
            ql_sweep(dg, e, z, l, shift, act)
            sweeps[:, l] += act
    return dg, z, sweeps


n = NW * 64
A, B = rand_spd(n), rand_spd(n)
L = np.linalg.cholesky(A)
Li = np.linalg.inv(L)
M = Li @ B @ Li.transpose(0, 2, 1)
M = 0.5 * (M + M.transpose(0, 2, 1))
dg, e, Q = tridiag(M)
lam_a, z_a, sw_a = run(dg, e, Q)
lam_v, _, sw_v = run(dg, e, None)
assert np.array_equal(lam_a, lam_v)
lam_b, z_b, sw_b = run(dg, e, Q, known=lam_v)


def wave_steps(sw):
    """steps = sum over stages of (D-1-l) rotations x sweeps of the slowest lane of each wave"""
    w = sw.reshape(NW, 64, D - 1).max(1)
    rot = np.array([D - 1 - l for l in range(D - 1)])
    return (w * rot).sum(1)


def logm_err(lam, z):
    F = np.einsum("nik,nk,njk->nij", z, np.log(lam), z)
    w, v = np.linalg.eigh(M)
    Fr = np.einsum("nik,nk,njk->nij", v, np.log(w), v)
    return np.abs(F - Fr).max(axis=(1, 2)) / np.abs(Fr).max(axis=(1, 2))


sa, sv, sb = wave_steps(sw_a), wave_steps(sw_v), wave_steps(sw_b)
print(f"d = {D}, {NW} waves, eps2 = {EPS2}")
print(f"today:  vector-QL sweep steps per wave   mean {sa.mean():6.1f}  (per lane alone {(sw_a * np.arange(D - 1, 0, -1)).sum(1).mean():6.1f})")
print(f"values-only pass steps per wave          mean {sv.mean():6.1f}")
print(f"perfect-shift vector pass steps per wave mean {sb.mean():6.1f}   (ideal {sum(range(1, D))})")
print(f"   stages needing > 1 sweep: per lane {(sw_b > 1).mean():.4f}, per wave {(sw_b.reshape(NW, 64, D - 1).max(1) > 1).mean():.4f}; max sweeps {sw_b.max()}")
VEC, VAL = 17 + 4 * D, 17
print(f"instructions per pair (QL part): today {sa.mean() * VEC:7.0f}; two-pass {sv.mean() * VAL + sb.mean() * VEC:7.0f}  ({100 * (1 - (sv.mean() * VAL + sb.mean() * VEC) / (sa.mean() * VEC)):.1f} % less)")
ea, eb = logm_err(lam_a, z_a), logm_err(lam_b, z_b)
print(f"logm error (max rel): today {ea.max():.2e} (median {np.median(ea):.2e}); perfect shift {eb.max():.2e} (median {np.median(eb):.2e})")
print(f"orthogonality |Z^T Z - I|max: today {np.abs(np.einsum('nki,nkj->nij', z_a, z_a) - np.eye(D)).max():.2e}; perfect {np.abs(np.einsum('nki,nkj->nij', z_b, z_b) - np.eye(D)).max():.2e}")
print(f"eigenvalue drift |lam_b - lam_v|/|lam| max {np.abs(np.sort(lam_b, 1) - np.sort(lam_v, 1)).max() / np.abs(lam_v).max():.2e}")
