"""Pins the CPU oracle (oracle/*.py) to vectors produced by importing the reference (tests/golden/make_golden.py).

Tolerances: the reference stores SPD eigenvalues through an fp32 buffer (spd_utils_torch.py:108), so anything that
went through `affine_invariant_distance_torch` agrees to ~2e-7 relative only; everything else is fp64-tight.
"""
import numpy as np
import pytest

from oracle import spd as ospd
from oracle import sphere as osph

F32 = 5e-7      # fp32-sink floor of the reference distance


def test_mandel(golden):
    g = golden("mandel.npz")
    for d in (2, 3, 5, 10, 20):
        m = ospd.vector_to_symmetric_matrix_mandel(g[f"d{d}_vec"])
        np.testing.assert_allclose(m, g[f"d{d}_mat"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(m[0, 0], g[f"d{d}_mat_np"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(ospd.symmetric_matrix_to_vector_mandel(m), g[f"d{d}_vec"], rtol=1e-15, atol=1e-15)
        np.testing.assert_allclose(ospd.symmetric_matrix_to_vector_mandel(g[f"d{d}_nonsym"]), g[f"d{d}_nonsym_vec"],
                                   rtol=1e-15, atol=1e-15)
    np.testing.assert_allclose(ospd.symmetric_matrix_to_vector_mandel(np.array([[2., .5], [.5, 1.]])),
                               g["kat_mandel"], rtol=1e-15)
    np.testing.assert_allclose(g["kat_mandel"], [2.0, 1.0, 0.5 * 2 ** 0.5], rtol=1e-15)


def test_spd_ai_distance_kernel_and_grads(golden):
    g = golden("spd_ai.npz")
    for c in range(int(g["ncases"])):
        p = f"c{c}_"
        x1, x2, beta = g[p + "x1"], g[p + "x2"], float(g[p + "beta"])
        m1 = ospd.vector_to_symmetric_matrix_mandel(x1)
        m2 = ospd.vector_to_symmetric_matrix_mandel(x2)
        np.testing.assert_allclose(m1, g[p + "m1"], rtol=0, atol=1e-15)
        dist = ospd.affine_invariant_distance(m1, m2)
        np.testing.assert_allclose(dist, g[p + "dist"], rtol=F32, atol=F32)
        if p + "dist_np" in g:   # independent fp64 statement inside the reference (spd_utils.py:180-197)
            np.testing.assert_allclose(dist, np.sqrt(g[p + "dist_np"] ** 2 + 1e-15), rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(ospd.affine_invariant_distance_faithful(m1, m2), dist, rtol=1e-12, atol=1e-13)
        k = ospd.spd_ai_gaussian_kernel(x1, x2, beta)
        np.testing.assert_allclose(k, g[p + "K"], rtol=1e-5, atol=1e-7)
        g1, g2 = ospd.spd_ai_gaussian_kernel_grads(x1, x2, beta, g[p + "gup"])
        scale = max(1.0, np.abs(g[p + "grad_x1"]).max())
        np.testing.assert_allclose(g1, g[p + "grad_x1"], rtol=1e-5, atol=2e-6 * scale)
        scale = max(1.0, np.abs(g[p + "grad_x2"]).max())
        np.testing.assert_allclose(g2, g[p + "grad_x2"], rtol=1e-5, atol=2e-6 * scale)
    assert tuple(g["diag_shape"]) == ospd.affine_invariant_distance(np.zeros((3, 4, 2, 2)), np.zeros((3, 6, 2, 2)),
                                                                    diagonal_distance=True).shape
    np.testing.assert_allclose(ospd.affine_invariant_distance(g["kat_a"], g["kat_b"]), g["kat_ab"], rtol=F32)
    np.testing.assert_allclose(ospd.affine_invariant_distance(g["kat_a"], g["kat_b"])[0, 0], float(g["kat_ab_np"]),
                               rtol=1e-13)
    np.testing.assert_allclose(float(g["kat_ab_np"]), 1.4033966394735078, rtol=1e-14)
    np.testing.assert_allclose(ospd.affine_invariant_distance(np.eye(10)[None], np.e * np.eye(10)[None]),
                               g["kat_I_eI"], rtol=F32)
    np.testing.assert_allclose(ospd.affine_invariant_distance(np.eye(10)[None], np.e * np.eye(10)[None])[0, 0],
                               10 ** 0.5, rtol=1e-14)


def test_sphere_distance_kernel_and_grads(golden):
    g = golden("sphere.npz")
    for c in range(int(g["ncases"])):
        p = f"c{c}_"
        x1, x2, beta = g[p + "x1"], g[p + "x2"], float(g[p + "beta"])
        np.testing.assert_allclose(osph.sphere_distance(x1, x2), g[p + "dist"], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(osph.sphere_gaussian_kernel(x1, x2, beta), g[p + "K"], rtol=1e-12, atol=1e-15)
        g1, g2 = osph.sphere_gaussian_kernel_grads(x1, x2, beta, g[p + "gup"])
        # near the clamp the gradient magnitude is ~1e7 x noise in <x,y>; compare relative to the row scale
        for mine, ref in ((g1, g[p + "grad_x1"]), (g2, g[p + "grad_x2"])):
            np.testing.assert_allclose(mine, ref, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(osph.sphere_distance(g["diag_x"], g["diag_y"], diag=True), g["diag_dist"], rtol=1e-13)
    e = np.eye(3)
    assert osph.sphere_distance(e[0:1], e[1:2])[0, 0] == g["kat_e1e2"][0, 0] == np.pi / 2
    np.testing.assert_allclose(osph.sphere_distance(e[0:1], -e[0:1]), g["kat_e1me1"], rtol=1e-15)
    np.testing.assert_allclose(osph.sphere_distance(e[0:1], e[0:1]), g["kat_e1e1"], rtol=1e-12)
    np.testing.assert_allclose(g["kat_e1e1"], 4.4703483581542975e-08, rtol=1e-12)
    np.testing.assert_allclose(osph.logmap(g["map_x"], g["map_base"]), g["map_log"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(osph.expmap(g["map_log"], g["map_base"]), g["map_exp"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(osph.expmap(g["map_log"], g["map_base"]), g["map_x"], rtol=1e-10, atol=1e-12)
    for i in range(6):
        np.testing.assert_allclose(osph.rotation_from_sphere_points(g["map_x"][i], g["map_base"][i]), g["rot"][i],
                                   rtol=1e-12, atol=1e-13)


def test_spd_maps(golden):
    g = golden("spd_maps.npz")
    for d in (2, 3, 5):
        S, X, U = g[f"d{d}_S"], g[f"d{d}_X"], g[f"d{d}_log"]
        np.testing.assert_allclose(ospd.logmap(X, S), U, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(ospd.expmap(U, S), g[f"d{d}_explog"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(ospd.expmap(ospd.logmap(X, S), S), X, rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(ospd.spd_log(S, X), U, rtol=1e-9, atol=1e-10)          # pymanopt arg order
        np.testing.assert_allclose(ospd.expm_sym(g[f"d{d}_sym"]), g[f"d{d}_multiexp"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(ospd.logm(X), g[f"d{d}_multilog"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(ospd.logm(X), g[f"d{d}_logm_torch"], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(ospd.sqrtm(X), g[f"d{d}_sqrtm_torch"], rtol=1e-10, atol=1e-11)
        c, gc = ospd.max_eigenvalue_constraint(X, 5.0)
        np.testing.assert_allclose(c, g[f"d{d}_maxeig"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(gc, g[f"d{d}_maxeig_grad"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(ospd.min_eigenvalue_constraint(X, 0.01)[0], g[f"d{d}_mineig"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(ospd.frobenius_distance(S, X), g[f"d{d}_frob"], rtol=1e-13)
        # AI distance == ||Log_S(X)||_S  (ties a10 to a2)
        dist = ospd.affine_invariant_distance(S, X)
        np.testing.assert_allclose(np.sqrt(ospd.spd_inner(S, U, U) + 1e-15), np.diagonal(dist), rtol=1e-9)
    np.random.seed(int(g["sample_seed"]))
    mine = np.stack([ospd.spd_sample(5, 0.001, 5.0) for _ in range(3)])
    np.testing.assert_allclose(mine, g["sample_out"], rtol=1e-13, atol=1e-14)


def test_nested_projection_and_log_euclid(golden):
    g = golden("nested_spd.npz")
    y1 = ospd.projection_from_spd_to_nested_spd(g["X1"], g["W"])
    y2 = ospd.projection_from_spd_to_nested_spd(g["X2"], g["W"])
    np.testing.assert_allclose(y1, g["Y1"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(ospd.vector_to_symmetric_matrix_mandel(g["x1_mandel"]), g["X1"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(ospd.affine_invariant_distance(y1, y2), g["ai_dist"], rtol=F32, atol=F32)
    np.testing.assert_allclose(ospd.logm(y1), g["logY1"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(ospd.log_euclidean_distance(y1, y2), g["le_dist"], rtol=1e-10)
    np.testing.assert_allclose(ospd.projection_from_spd_to_nested_spd(g["X5"], g["W5"]), g["Y5"], rtol=1e-12, atol=1e-13)


def test_letters_fixture(golden):
    g = golden("letters_spd2.npz")
    k = ospd.spd_ai_gaussian_kernel(g["x_mandel"], g["x_mandel"], float(g["beta"]))
    np.testing.assert_allclose(k, g["K"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(k, k.T, rtol=1e-12)
    np.testing.assert_allclose(np.diagonal(k), 1.0, atol=1e-12)


def test_log_euclidean_kernel_and_gradients(golden):
    g = golden("nested_spd.npz")
    for d in (2, 3):
        x1, x2, ls = g[f"le{d}_x1"], g[f"le{d}_x2"], float(g[f"le{d}_ls"])
        np.testing.assert_allclose(ospd.log_euclidean_gaussian_kernel(x1, x2, ls), g[f"le{d}_K"], rtol=1e-10)
        g1, g2 = ospd.log_euclidean_gaussian_kernel_grads(x1, x2, ls, g[f"le{d}_gup"])
        np.testing.assert_allclose(g1, g[f"le{d}_g1"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(g2, g[f"le{d}_g2"], rtol=1e-8, atol=1e-10)


def test_nested_sphere_projections_and_kernel(golden):
    g = golden("nested_sphere.npz")
    for tag in "abc":
        nl = int(g[f"{tag}_nlevels"])
        axes = [g[f"{tag}_axis{k}"] for k in range(nl)]
        dists = [float(g[f"{tag}_dist"])] * nl
        np.testing.assert_allclose(osph.rotation_from_sphere_points(axes[0], np.eye(axes[0].shape[1])[-1:]), g[f"{tag}_rot0"], atol=1e-14)
        np.testing.assert_allclose(osph.projection_from_sphere_to_nested_sphere(g[f"{tag}_x1"], axes[0], dists[0]), g[f"{tag}_nested0"],
                                   atol=1e-13)
        levels = osph.projection_from_sphere_to_subsphere(g[f"{tag}_x1"], axes, dists)
        for k, lv in enumerate(levels):
            np.testing.assert_allclose(lv, g[f"{tag}_level{k}"], atol=1e-13)
        back = osph.projection_from_subsphere_to_sphere(levels[-1], axes, dists)
        for k, b in enumerate(back):
            np.testing.assert_allclose(b, g[f"{tag}_back{k}"], atol=1e-13)
        np.testing.assert_allclose(osph.nested_sphere_gaussian_kernel(g[f"{tag}_x1"], g[f"{tag}_x2"], axes, dists, float(g[f"{tag}_beta"])),
                                   g[f"{tag}_K"], rtol=1e-11)


def test_reconstruction_costs_and_matrix_function_adjoints(golden):
    g = golden("reconstruction.npz")
    for tag in "ab":
        args = [g[f"{tag}_{k}"] for k in ("X", "Y", "W", "V", "C", "K")]
        # the reference accumulates the per-matrix distances in float32 (torch.zeros(n_data)): 1e-6 is its own resolution
        np.testing.assert_allclose(ospd.reconstruction_cost(*args, metric="ai"), g[f"{tag}_ai_cost"], rtol=2e-6)
        np.testing.assert_allclose(ospd.reconstruction_cost(*args, metric="le"), g[f"{tag}_le_cost"], rtol=2e-6)
    np.testing.assert_allclose(ospd.logm(g["mf_A"]), g["logm_val"], atol=1e-12)
    np.testing.assert_allclose(ospd.sqrtm(g["mf_A"]), g["sqrtm_val"], atol=1e-12)
    sym = lambda a: 0.5 * (a + a.transpose(0, 2, 1))     # noqa: E731  autograd's gradient w.r.t. a general matrix; we compare its symmetric part
    np.testing.assert_allclose(ospd.matfun_adjoint(g["mf_A"], g["mf_G"], "log"), sym(g["logm_grad"]), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(ospd.matfun_adjoint(g["mf_A"], g["mf_G"], "sqrt"), sym(g["sqrtm_grad"]), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(ospd.frobenius_distance(g["frob_x1"], g["frob_x2"]), g["frob_d"], rtol=1e-12)


def test_nested_sphere_reconstruction_cost(golden):
    g = golden("reconstruction.npz")
    axes = [g["ns_axis0"], g["ns_axis1"]]
    cost = lambda r: osph.nested_sphere_reconstruction_cost(g["ns_x"], g["ns_sub"], axes, list(r))   # noqa: E731
    np.testing.assert_allclose(cost(g["ns_r"]), g["ns_cost"], rtol=1e-10)
    h = 1e-6
    num = [(cost(g["ns_r"] + h * np.eye(2)[k]) - cost(g["ns_r"] - h * np.eye(2)[k])) / (2 * h) for k in range(2)]
    np.testing.assert_allclose(num, g["ns_grad"], rtol=1e-6)


def test_gp_oracle_against_the_ei_fixture(golden):
    """oracle/gp.py (GP posterior + expected improvement) on oracle/spd.py's kernel against -EI as make_golden_ei_optimum.py states it in torch
    on the reference's `affine_invariant_distance_torch`: at the 32 starts and at the reference solver's 32 end points."""
    from oracle import gp as ogp
    g = golden("ei_optimum.npz")
    beta = float(g["beta"])
    np.testing.assert_allclose(g["mandel_check"], ospd.symmetric_matrix_to_vector_mandel(g["X"][:1])[0], rtol=0, atol=1e-15)
    kxx = ospd.spd_ai_gaussian_kernel(g["Xv"], g["Xv"], beta)
    for pts, want in ((g["x0"], g["f0"]), (g["x"], g["f"])):
        v = ospd.symmetric_matrix_to_vector_mandel(pts)
        ks = ospd.spd_ai_gaussian_kernel(v, g["Xv"], beta)
        ei = ogp.expected_improvement(*ogp.gp_posterior(kxx, ks, np.exp(-beta * 1e-15) * np.ones(len(v)), g["y"], float(g["mean"]),
                                                        float(g["outputscale"]), float(g["noise"])), best_f=float(g["best_f"]), maximize=False)
        np.testing.assert_allclose(-ei, want, rtol=2e-7, atol=1e-12)
    assert int(g["best"]) == int(np.argmin(g["f"])) and (g["nit"] == 100).sum() == 1


def test_second_order_statements_against_the_reference_backend(golden):
    """oracle.spd.spd_ai_kernel_hvp / oracle.sphere.sphere_gaussian_kernel_hvp against `ehess` of the reference's own PytorchBackend
    (pymanopt_addons/tools/autodiff/_pytorch.py:103-116) on costs built from its distance functions (tests/golden/make_golden_hvp.py)"""
    z = golden("hvp.npz")

    def mandel(m):
        return ospd.symmetric_matrix_to_vector_mandel(0.5 * (m + np.swapaxes(m, -1, -2)))
    assert not bool(z["exact_repeat_is_finite"])       # (the reference itself returns NaN at an exactly repeated eigenvalue of M)
    for d in z["spd_dims"]:
        x1, x2, G, U, beta = (z[f"spd{d}_{k}"] for k in ("x1", "x2", "G", "U", "beta"))
        for mode in ("gaussian", "laplace", "distance"):
            want = mandel(z[f"spd{d}_{mode}_ehess"])       # chain rule through the (isometric) Mandel map: P^T vec(ehess)
            got = ospd.spd_ai_kernel_hvp(mandel(x1), mandel(x2), float(beta), G, mandel(U), mode)
            np.testing.assert_allclose(got, want, rtol=0, atol=5e-12 * np.abs(want).max())
        ga, _ = ospd.spd_ai_gaussian_kernel_grads(mandel(x1), mandel(x2), float(beta), G)
        np.testing.assert_allclose(ga, mandel(z[f"spd{d}_gaussian_egrad"]), rtol=0, atol=5e-12 * np.abs(ga).max())
    for dim in z["sphere_dims"]:
        x1, x2, G, U, beta = (z[f"sph{dim}_{k}"] for k in ("x1", "x2", "G", "U", "beta"))
        got = osph.sphere_gaussian_kernel_hvp(x1, x2, float(beta), G, U)
        np.testing.assert_allclose(got, z[f"sph{dim}_ehess"], rtol=0, atol=1e-12 * np.abs(got).max())
