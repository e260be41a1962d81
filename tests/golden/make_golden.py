#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by IMPORTING the reference.

Runs only in the development container (it needs /root/reference); the GPU box never sees the
reference, only the .npz files this script wrote.  Re-run with

    python -W ignore tests/golden/make_golden.py

The only shim is `torch.symeig` (removed from torch >= 1.13; the reference calls it at
BoManifolds/Riemannian_utils/spd_utils_torch.py:25,45,110): it is mapped onto
`torch.linalg.eigh(UPLO='U')`, which is what symeig(upper=True) computed.

Every array stored here is either an INPUT drawn from a seeded numpy Generator or the OUTPUT of a
reference function called on that input.  No reference source text is stored.
"""
import collections
import os
import sys

import numpy as np
import torch

REF = os.environ.get("GABO_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

_R = collections.namedtuple("symeig", ["eigenvalues", "eigenvectors"])
torch.symeig = lambda A, eigenvectors=False, upper=True: _R(*torch.linalg.eigh(A, UPLO="U" if upper else "L"))
sys.path.insert(0, REF)

from BoManifolds.Riemannian_utils import spd_utils, sphere_utils  # noqa: E402
from BoManifolds.Riemannian_utils import spd_utils_torch as sut  # noqa: E402
from BoManifolds.Riemannian_utils import sphere_utils_torch as sphut  # noqa: E402
from BoManifolds.Riemannian_utils import spd_constraints_utils_torch as scut  # noqa: E402
from BoManifolds.nested_mappings import nested_spd_utils as nsu  # noqa: E402
from BoManifolds.nested_mappings import nested_spheres_utils as nsph  # noqa: E402
from BoManifolds.BO_test_functions import test_functions_spd as tf_spd  # noqa: E402
from BoManifolds.BO_test_functions import test_functions_sphere as tf_sph  # noqa: E402
from BoManifolds.pymanopt_addons.tools import multi as ref_multi  # noqa: E402


def rand_spd(rng, n, d, lo=0.05, hi=5.0):
    """Random SPD batch (n,d,d): Q diag(U[lo,hi]) Q^T.  Same recipe as SURVEY 8(d)."""
    out = np.empty((n, d, d))
    for k in range(n):
        lam = rng.uniform(lo, hi, size=d)
        q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        m = (q * lam) @ q.T
        out[k] = 0.5 * (m + m.T)
    return out


def to_mandel_ref(mats):
    return np.stack([spd_utils.symmetric_matrix_to_vector_mandel(m) for m in mats])


def gen_spd_ai():
    """a1/a2/a3/a7: Mandel -> matrices -> AI distance -> K, and autograd grad wrt x1 (Mandel)."""
    out = {}
    case = 0
    for seed, d, n1, n2, batch in [
        (0, 2, 7, 5, ()), (1, 3, 7, 7, ()), (2, 5, 32, 9, ()), (3, 10, 12, 17, ()),
        (4, 10, 1, 1, ()), (5, 3, 4, 6, (3,)), (6, 5, 1, 7, (2,)), (7, 2, 33, 33, ()),
    ]:
        rng = np.random.default_rng(seed)
        nb = int(np.prod(batch)) if batch else 1
        x1 = np.stack([to_mandel_ref(rand_spd(rng, n1, d)) for _ in range(nb)]).reshape(batch + (n1, -1))
        x2 = np.stack([to_mandel_ref(rand_spd(rng, n2, d)) for _ in range(nb)]).reshape(batch + (n2, -1))
        gup = rng.standard_normal(batch + (n1, n2))
        beta = [0.2, 0.6, 1.2931472][case % 3]
        t1 = torch.tensor(x1, requires_grad=True)
        t2 = torch.tensor(x2, requires_grad=True)
        m1 = sut.vector_to_symmetric_matrix_mandel_torch(t1)
        m2 = sut.vector_to_symmetric_matrix_mandel_torch(t2)
        dist = sut.affine_invariant_distance_torch(m1, m2)
        k = torch.exp(-(dist * dist) * beta)
        (k * torch.tensor(gup)).sum().backward()
        p = f"c{case}_"
        out[p + "x1"], out[p + "x2"], out[p + "gup"] = x1, x2, gup
        out[p + "beta"] = np.float64(beta)
        out[p + "m1"] = m1.detach().numpy()
        out[p + "dist"] = dist.detach().numpy()
        out[p + "K"] = k.detach().numpy()
        out[p + "grad_x1"] = t1.grad.numpy()
        out[p + "grad_x2"] = t2.grad.numpy()
        # independent in-repo statement of the same distance (spd_utils.py:180-197), pair by pair, batch 0 only
        if not batch:
            m2n = m2.detach().numpy()
            m1n = m1.detach().numpy()
            dn = np.array([[np.real(spd_utils.affine_invariant_distance(a, b)) for b in m2n] for a in m1n])
            out[p + "dist_np"] = dn
        case += 1
    out["ncases"] = np.int64(case)
    # diagonal_distance=True shape contract (spd_utils_torch.py:72-75)
    z = sut.affine_invariant_distance_torch(torch.zeros(3, 4, 2, 2, dtype=torch.float64),
                                            torch.zeros(3, 6, 2, 2, dtype=torch.float64), diagonal_distance=True)
    out["diag_shape"] = np.array(z.shape)
    # known answers quoted in SURVEY App. A
    a = torch.tensor([[[2., .5], [.5, 1.]]], dtype=torch.float64)
    b = torch.tensor([[[1., .2], [.2, 3.]]], dtype=torch.float64)
    out["kat_a"], out["kat_b"] = a.numpy(), b.numpy()
    out["kat_ab"] = sut.affine_invariant_distance_torch(a, b).numpy()
    out["kat_ab_np"] = np.float64(np.real(spd_utils.affine_invariant_distance(a[0].numpy(), b[0].numpy())))
    i10 = torch.eye(10, dtype=torch.float64)[None]
    out["kat_I_eI"] = sut.affine_invariant_distance_torch(i10, np.e * i10).numpy()
    np.savez_compressed(os.path.join(HERE, "spd_ai.npz"), **out)


def gen_mandel():
    """a3/a4 round trips, torch and numpy twins."""
    out = {}
    rng = np.random.default_rng(11)
    for d in (2, 3, 5, 10, 20):
        v = rng.standard_normal((2, 3, d * (d + 1) // 2))
        m = sut.vector_to_symmetric_matrix_mandel_torch(torch.tensor(v))
        out[f"d{d}_vec"] = v
        out[f"d{d}_mat"] = m.numpy()
        # a4 averages the two triangles: feed a NON-symmetric matrix to pin that
        ns = rng.standard_normal((4, d, d))
        out[f"d{d}_nonsym"] = ns
        out[f"d{d}_nonsym_vec"] = sut.symmetric_matrix_to_vector_mandel_torch(torch.tensor(ns)).numpy()
        out[f"d{d}_mat_np"] = spd_utils.vector_to_symmetric_matrix_mandel(v[0, 0])
    out["kat_mandel"] = spd_utils.symmetric_matrix_to_vector_mandel(np.array([[2., .5], [.5, 1.]]))
    np.savez_compressed(os.path.join(HERE, "mandel.npz"), **out)


def gen_sphere():
    """a5/a6/a7 sphere distance, kernel, grads, including the clamp edge cases."""
    out = {}
    case = 0
    for seed, dim, n1, n2, batch in [(0, 3, 9, 6, ()), (1, 10, 17, 33, ()), (2, 51, 5, 8, ()), (3, 4, 6, 3, (2,)),
                                     (4, 3, 1, 1, ())]:
        rng = np.random.default_rng(100 + seed)
        x1 = rng.standard_normal(batch + (n1, dim))
        x1 /= np.linalg.norm(x1, axis=-1, keepdims=True)
        x2 = rng.standard_normal(batch + (n2, dim))
        x2 /= np.linalg.norm(x2, axis=-1, keepdims=True)
        if not batch and n1 > 3 and n2 > 3:
            x2[0] = x1[0]                      # identical
            x2[1] = -x1[1]                     # antipodal
            x2[2] = x1[2] + 1e-9 * x2[2]       # near-identical
            x2[2] /= np.linalg.norm(x2[2])
        gup = rng.standard_normal(batch + (n1, n2))
        beta = [6.5, 0.6 + 0.6931472, 0.21][case % 3]
        t1 = torch.tensor(x1, requires_grad=True)
        t2 = torch.tensor(x2, requires_grad=True)
        dist = sphut.sphere_distance_torch(t1, t2)
        k = torch.exp(-(dist * dist) * beta)
        (k * torch.tensor(gup)).sum().backward()
        p = f"c{case}_"
        out[p + "x1"], out[p + "x2"], out[p + "gup"] = x1, x2, gup
        out[p + "beta"] = np.float64(beta)
        out[p + "dist"] = dist.detach().numpy()
        out[p + "K"] = k.detach().numpy()
        out[p + "grad_x1"] = t1.grad.numpy()
        out[p + "grad_x2"] = t2.grad.numpy()
        case += 1
    out["ncases"] = np.int64(case)
    # diag=True branch (sphere_utils_torch.py:45-49)
    rng = np.random.default_rng(7)
    xd = rng.standard_normal((5, 4))         # the reference's diag branch only accepts 2-D inputs (bmm)
    xd /= np.linalg.norm(xd, axis=-1, keepdims=True)
    yd = rng.standard_normal((5, 4))
    yd /= np.linalg.norm(yd, axis=-1, keepdims=True)
    out["diag_x"], out["diag_y"] = xd, yd
    out["diag_dist"] = sphut.sphere_distance_torch(torch.tensor(xd), torch.tensor(yd), diag=True).numpy()
    e = torch.eye(3, dtype=torch.float64)
    out["kat_e1e2"] = sphut.sphere_distance_torch(e[0:1], e[1:2]).numpy()
    out["kat_e1me1"] = sphut.sphere_distance_torch(e[0:1], -e[0:1]).numpy()
    out["kat_e1e1"] = sphut.sphere_distance_torch(e[0:1], e[0:1]).numpy()
    # numpy exp/log maps (sphere_utils.py:14-65), one pair per call
    xs = rng.standard_normal((6, 5))
    xs /= np.linalg.norm(xs, axis=-1, keepdims=True)
    base = rng.standard_normal((6, 5))
    base /= np.linalg.norm(base, axis=-1, keepdims=True)
    logs = np.stack([sphere_utils.logmap(xs[i], base[i])[:, 0] for i in range(6)])
    exps = np.stack([sphere_utils.expmap(logs[i], base[i])[:, 0] for i in range(6)])
    out["map_x"], out["map_base"], out["map_log"], out["map_exp"] = xs, base, logs, exps
    out["map_dist"] = np.array([sphere_utils.sphere_distance(xs[i], base[i]).item() for i in range(6)])
    # rotation_from_sphere_points_torch (sphere_utils_torch.py:58-93)
    out["rot"] = np.stack([sphut.rotation_from_sphere_points_torch(torch.tensor(xs[i]), torch.tensor(base[i])).numpy()
                           for i in range(6)])
    np.savez_compressed(os.path.join(HERE, "sphere.npz"), **out)


def gen_spd_maps():
    """a10/a13/a14 + pymanopt-equivalent helpers vendored in the reference (multilog/multiexp)."""
    out = {}
    rng = np.random.default_rng(21)
    for d in (2, 3, 5):
        S = rand_spd(rng, 6, d, 0.2, 4.0)
        X = rand_spd(rng, 6, d, 0.2, 4.0)
        U = np.stack([spd_utils.logmap(X[i], S[i]) for i in range(6)])
        U = np.real(U)
        Xb = np.real(np.stack([spd_utils.expmap(U[i], S[i]) for i in range(6)]))
        out[f"d{d}_S"], out[f"d{d}_X"], out[f"d{d}_log"], out[f"d{d}_explog"] = S, X, U, Xb
        sym = rng.standard_normal((6, d, d))
        sym = 0.5 * (sym + sym.transpose(0, 2, 1))
        out[f"d{d}_sym"] = sym
        out[f"d{d}_multiexp"] = ref_multi.multiexp(sym, sym=True)
        out[f"d{d}_multilog"] = ref_multi.multilog(X, pos_def=True)
        out[f"d{d}_logm_torch"] = np.stack([sut.logm_torch(torch.tensor(x)).numpy() for x in X])
        out[f"d{d}_sqrtm_torch"] = np.stack([sut.sqrtm_torch(torch.tensor(x)).numpy() for x in X])
        out[f"d{d}_maxeig"] = np.array([scut.max_eigenvalue_constraint_torch(torch.tensor(x), 5.0).item() for x in X])
        out[f"d{d}_mineig"] = np.array([scut.min_eigenvalue_constraint_torch(torch.tensor(x), 0.01).item() for x in X])
        # gradient of the constraints (autograd through symeig)
        g = []
        for x in X:
            t = torch.tensor(x, requires_grad=True)
            scut.max_eigenvalue_constraint_torch(t, 5.0).backward()
            g.append(t.grad.numpy())
        out[f"d{d}_maxeig_grad"] = np.stack(g)
        out[f"d{d}_frob"] = sut.frobenius_distance_torch(torch.tensor(S), torch.tensor(X)).numpy()

    # spd_sample (spd_utils.py:290-306): global numpy RNG, pinned by seeding it
    class _M:
        pass
    man = _M()
    man._n, man.min_eig, man.max_eig = 5, 0.001, 5.0
    np.random.seed(1234)
    out["sample_seed"] = np.int64(1234)
    out["sample_out"] = np.stack([spd_utils.spd_sample(man) for _ in range(3)])
    np.savez_compressed(os.path.join(HERE, "spd_maps.npz"), **out)


def gen_nested():
    """a15: W'XW projection, log-Euclidean distance (the op sequence of kernels_spd.py:267-313)."""
    out = {}
    rng = np.random.default_rng(31)
    D, d, n1, n2 = 20, 2, 6, 5
    X1 = rand_spd(rng, n1, D)
    X2 = rand_spd(rng, n2, D)
    W, _ = np.linalg.qr(rng.standard_normal((D, D)))
    W = W[:, :d]
    Y1 = nsu.projection_from_spd_to_nested_spd(torch.tensor(X1), torch.tensor(W))
    Y2 = nsu.projection_from_spd_to_nested_spd(torch.tensor(X2), torch.tensor(W))
    out["X1"], out["X2"], out["W"] = X1, X2, W
    out["x1_mandel"], out["x2_mandel"] = to_mandel_ref(X1), to_mandel_ref(X2)
    out["Y1"], out["Y2"] = Y1.numpy(), Y2.numpy()
    out["ai_dist"] = sut.affine_invariant_distance_torch(Y1, Y2).numpy()
    L1 = torch.stack([sut.logm_torch(y) for y in Y1])
    L2 = torch.stack([sut.logm_torch(y) for y in Y2])
    out["logY1"], out["logY2"] = L1.numpy(), L2.numpy()
    out["le_dist"] = sut.frobenius_distance_torch(L1, L2).numpy()
    # D=5 -> d=2 as shipped in examples/hd_gabo_spd.py:96-98
    X5 = rand_spd(rng, 4, 5)
    W5, _ = np.linalg.qr(rng.standard_normal((5, 5)))
    W5 = W5[:, :2]
    out["X5"], out["W5"] = X5, W5
    out["Y5"] = nsu.projection_from_spd_to_nested_spd(torch.tensor(X5), torch.tensor(W5)).numpy()
    # log-Euclidean Gaussian kernel (op sequence of kernels_spd.py:289-311) with autograd gradients w.r.t. both Mandel inputs
    for dd_, n1_, n2_ in ((2, 5, 4), (3, 4, 6)):
        a_ = to_mandel_ref(rand_spd(rng, n1_, dd_, 0.3, 3.0))
        b_ = to_mandel_ref(rand_spd(rng, n2_, dd_, 0.3, 3.0))
        ta, tb = torch.tensor(a_, requires_grad=True), torch.tensor(b_, requires_grad=True)
        ma, mb = sut.vector_to_symmetric_matrix_mandel_torch(ta), sut.vector_to_symmetric_matrix_mandel_torch(tb)
        la = torch.stack([sut.logm_torch(m) for m in ma])
        lb = torch.stack([sut.logm_torch(m) for m in mb])
        dist_ = sut.frobenius_distance_torch(la, lb)
        ls_ = 0.9
        k_ = torch.exp(-(dist_ * dist_) / (ls_ * ls_))
        gup_ = rng.standard_normal((n1_, n2_))
        (k_ * torch.tensor(gup_)).sum().backward()
        out[f"le{dd_}_x1"], out[f"le{dd_}_x2"], out[f"le{dd_}_gup"], out[f"le{dd_}_ls"] = a_, b_, gup_, np.float64(ls_)
        out[f"le{dd_}_K"], out[f"le{dd_}_g1"], out[f"le{dd_}_g2"] = k_.detach().numpy(), ta.grad.numpy(), tb.grad.numpy()
    # approximate right inverse of the projection (nested_spd_utils.py:51-118)
    Vc = np.linalg.qr(rng.standard_normal((5, 5)))[0]
    W5b, V5b = Vc[:, :2], Vc[:, 2:]
    bottom = rand_spd(rng, 1, 3, 0.5, 2.0)[0]
    contraction = rng.standard_normal((2, 3))
    contraction /= 2.0 * np.linalg.norm(contraction, 2)
    ylow = rand_spd(rng, 4, 2, 0.3, 3.0)
    back = nsu.projection_from_nested_spd_to_spd(torch.tensor(ylow), torch.tensor(W5b), torch.tensor(V5b), torch.tensor(bottom),
                                                 torch.tensor(contraction))
    out["back_W"], out["back_V"], out["back_bottom"], out["back_K"], out["back_ylow"], out["back_X"] = W5b, V5b, bottom, contraction, ylow, back.numpy()
    np.savez_compressed(os.path.join(HERE, "nested_spd.npz"), **out)


def gen_letters():
    """(7) the 2x2 SPD set examples/kernels/spd/spd_kernels.py:98-107 derives from data/2Dletters/C.mat."""
    from scipy.io import loadmat
    demos = loadmat(os.path.join(REF, "data/2Dletters/C.mat"))["demos"][0]
    pos = demos[0]["pos"][0][0]                      # 2 x 200, first demonstration
    pts = pos[:, :40:2]                              # 20 points
    mats = np.stack([np.real(spd_utils.expmap(0.01 * np.outer(pts[:, n], pts[:, n]), np.eye(2)))
                     for n in range(pts.shape[1])])
    x = to_mandel_ref(mats)
    t = torch.tensor(x)
    m = sut.vector_to_symmetric_matrix_mandel_torch(t)
    dist = sut.affine_invariant_distance_torch(m, m)
    out = {"x_mandel": x, "dist": dist.numpy(), "beta": np.float64(0.6 + 0.6931472)}
    out["K"] = torch.exp(-(dist * dist) * out["beta"]).numpy()
    np.savez_compressed(os.path.join(HERE, "letters_spd2.npz"), **out)


def gen_objectives():
    """Objective values for configs 1/4/5 with manifold.log := the reference's own numpy log maps."""
    out = {}

    class _SpdMan:
        def __init__(self, n):
            self._n = n

        def log(self, base, x):      # pymanopt order: Log at `base` of `x`
            return np.real(spd_utils.logmap(x, base))

        def exp(self, base, u):
            return np.real(spd_utils.expmap(u, base))

    class _SphMan:
        def __init__(self, n):
            self._shape = (n,)
            self._n = n

        def log(self, base, x):      # called with (1,n) arrays (test_functions_sphere.py:47-49); returns (1,n)
            return sphere_utils.logmap(np.ravel(x), np.ravel(base))[:, 0][None]

        def exp(self, base, u):
            return sphere_utils.expmap(np.ravel(u), np.ravel(base))[:, 0][None]

    rng = np.random.default_rng(41)
    for d in (2, 5):
        X = rand_spd(rng, 5, d, 0.1, 4.0)
        xm = to_mandel_ref(X)
        man = _SpdMan(d)
        out[f"spd{d}_x"] = xm
        out[f"spd{d}_ackley"] = np.array([np.asarray(tf_spd.ackley_function_spd(torch.tensor(v), man)).item()
                                          for v in xm])
        out[f"spd{d}_rosenbrock"] = np.array(
            [np.asarray(tf_spd.rosenbrock_function_spd(torch.tensor(v), man)).item() for v in xm])
    for n in (3, 5):
        x = rng.standard_normal((5, n))
        x /= np.linalg.norm(x, axis=-1, keepdims=True)
        man = _SphMan(n)
        out[f"sph{n}_x"] = x
        out[f"sph{n}_ackley"] = np.array(
            [np.asarray(tf_sph.ackley_function_sphere(torch.tensor(v), man)).item() for v in x])
    np.savez_compressed(os.path.join(HERE, "objectives.npz"), **out)


def gen_nested_sphere():
    """f3: nested-sphere projections (nested_spheres_utils.py:13-218) and the op sequence of NestedSphereGaussianKernel.forward
    (kernels_nested_sphere.py:125-152), with autograd gradients with respect to the inputs and the axes."""
    out = {}
    rng = np.random.default_rng(41)
    for tag, dim, latent, dist in (("a", 5, 3, np.pi / 2), ("b", 6, 2, np.pi / 4), ("c", 4, 3, 1.1)):
        n1, n2 = 7, 5
        x1 = rng.standard_normal((n1, dim)); x1 /= np.linalg.norm(x1, axis=1, keepdims=True)
        x2 = rng.standard_normal((n2, dim)); x2 /= np.linalg.norm(x2, axis=1, keepdims=True)
        axes = []
        for d in range(dim, latent, -1):
            a = rng.standard_normal((1, d)); a /= np.linalg.norm(a)
            axes.append(a)
        dists = [torch.tensor([[dist]], dtype=torch.float64) for _ in axes]
        beta = 0.9
        gup = rng.standard_normal((n1, n2))
        t1 = torch.tensor(x1, requires_grad=True)
        t2 = torch.tensor(x2, requires_grad=True)
        tax = [torch.tensor(a, requires_grad=True) for a in axes]
        levels1 = nsph.projection_from_sphere_to_subsphere(t1, tax, dists)
        levels2 = nsph.projection_from_sphere_to_subsphere(t2, tax, dists)
        dd = sphut.sphere_distance_torch(levels1[-1], levels2[-1])
        K = torch.exp(-dd * dd * beta)
        (K * torch.tensor(gup)).sum().backward()
        out.update({f"{tag}_x1": x1, f"{tag}_x2": x2, f"{tag}_dist": np.array(dist), f"{tag}_beta": np.array(beta), f"{tag}_gup": gup,
                    f"{tag}_K": K.detach().numpy(), f"{tag}_g1": t1.grad.numpy(), f"{tag}_g2": t2.grad.numpy(),
                    f"{tag}_nlevels": np.array(len(axes))})
        for k, a in enumerate(axes):
            out[f"{tag}_axis{k}"] = a
            out[f"{tag}_gaxis{k}"] = tax[k].grad.numpy()
        for k, lv in enumerate(levels1):
            out[f"{tag}_level{k}"] = lv.detach().numpy()
        # first level: the nested-sphere point in the ambient sphere, and the rotation it uses
        nested = nsph.projection_from_sphere_to_nested_sphere(torch.tensor(x1), torch.tensor(axes[0]), dists[0])
        north = torch.zeros(1, dim, dtype=torch.float64); north[:, -1] = 1.0
        out[f"{tag}_nested0"] = nested.numpy()
        out[f"{tag}_rot0"] = sphut.rotation_from_sphere_points_torch(torch.tensor(axes[0]), north).numpy()
        # back projection from the latent sphere to the ambient one
        back = nsph.projection_from_subsphere_to_sphere(levels1[-1].detach(), [torch.tensor(a) for a in axes], dists)
        for k, b in enumerate(back):
            out[f"{tag}_back{k}"] = b.numpy()
    np.savez_compressed(os.path.join(HERE, "nested_sphere.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    gen_spd_ai()
    gen_mandel()
    gen_sphere()
    gen_spd_maps()
    gen_nested()
    gen_nested_sphere()
    gen_letters()
    gen_objectives()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
