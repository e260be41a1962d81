"""The nested SPD Gram in two launches (gabo_nested_spd_gram: projection + factorisation / logm fused, then the Gram launch) against the
separate-launch chain it replaces and against the oracle (kernel_utils/kernels_nested_spd.py:104-136, 191-246; nested_spd_utils.py:13-48)."""
import numpy as np
import pytest
import torch

from gabotorch_amd import _lib, ops
from gabotorch_amd.kernel_utils.kernels_spd import NestedSpdAffineInvariantGaussianKernel, NestedSpdLogEuclideanGaussianKernel
from oracle import spd as ospd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _data(rng, n, D, dl):
    q = np.linalg.qr(rng.standard_normal((n, D, D)))[0]
    m = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.1, 4.0, (n, D)), q)
    x = ospd.symmetric_matrix_to_vector_mandel(0.5 * (m + m.transpose(0, 2, 1)))
    w = np.ascontiguousarray(np.linalg.qr(rng.standard_normal((D, D)))[0][:, :dl])
    return x, w


@pytest.mark.parametrize("D,dl", [(20, 2), (12, 3), (9, 4), (5, 2)])
@pytest.mark.parametrize("metric", ["ai", "le"])
def test_fused_nested_gram_matches_the_chain_and_the_oracle(D, dl, metric):
    rng = np.random.default_rng(D * 10 + dl)
    x1, w = _data(rng, 70, D, dl)
    x2, _ = _data(rng, 130, D, dl)
    X1, X2, W = (torch.tensor(a, device=DEV) for a in (x1, x2, w))
    beta = 0.7
    m = _lib.GABO_METRIC_AFFINE_INVARIANT if metric == "ai" else _lib.GABO_METRIC_LOG_EUCLIDEAN
    got = ops.nested_spd_gram(X1, X2, W, beta, m)
    y1, y2 = ops.spd_project(X1, W), ops.spd_project(X2, W)
    if metric == "ai":
        chain = ops.spd_ai_pairwise(y1, y2, beta=beta)
        yo1 = ospd.symmetric_matrix_to_vector_mandel(ospd.projection_from_spd_to_nested_spd(ospd.vector_to_symmetric_matrix_mandel(x1), w))
        yo2 = ospd.symmetric_matrix_to_vector_mandel(ospd.projection_from_spd_to_nested_spd(ospd.vector_to_symmetric_matrix_mandel(x2), w))
        want = ospd.spd_ai_gaussian_kernel(yo1, yo2, beta)
    else:
        chain = ops.frobenius_pairwise(ops.spd_logm_mandel(y1), ops.spd_logm_mandel(y2), beta=beta)
        yo1 = ospd.symmetric_matrix_to_vector_mandel(ospd.projection_from_spd_to_nested_spd(ospd.vector_to_symmetric_matrix_mandel(x1), w))
        yo2 = ospd.symmetric_matrix_to_vector_mandel(ospd.projection_from_spd_to_nested_spd(ospd.vector_to_symmetric_matrix_mandel(x2), w))
        want = ospd.log_euclidean_gaussian_kernel(yo1, yo2, 1.0 / np.sqrt(beta))
    np.testing.assert_allclose(got.cpu().numpy(), chain.cpu().numpy(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-9, atol=1e-12)
    # x1 is x2: one projection serves both roles
    same = ops.nested_spd_gram(X1, X1, W, beta, m).cpu().numpy()
    np.testing.assert_allclose(same, ops.nested_spd_gram(X1, X1.clone(), W, beta, m).cpu().numpy(), rtol=1e-13, atol=1e-15)
    # a batch of point sets
    xb = torch.stack([X1[:30], X1[30:60]])
    xc = torch.stack([X2[:40], X2[40:80]])
    gb = ops.nested_spd_gram(xb, xc, W, beta, m).cpu().numpy()
    np.testing.assert_allclose(gb[0], got[:30, :40].cpu().numpy(), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(gb[1], got[30:60, 40:80].cpu().numpy(), rtol=1e-13, atol=1e-15)


def test_kernel_classes_take_the_fused_path_only_without_gradients():
    rng = np.random.default_rng(1)
    x, w = _data(rng, 40, 20, 2)
    X = torch.tensor(x, device=DEV)
    for cls, kw in ((NestedSpdAffineInvariantGaussianKernel, dict(beta_min=0.6)), (NestedSpdLogEuclideanGaussianKernel, {})):
        k = cls(20, 2, **kw).to(DEV)
        k.projection_matrix = torch.tensor(w, device=DEV)
        with torch.no_grad():
            fused = k.forward(X, X)
        graph = k.forward(X, X)                            # parameters require a gradient: the differentiable chain
        assert graph.requires_grad and not fused.requires_grad
        np.testing.assert_allclose(fused.cpu().numpy(), graph.detach().cpu().numpy(), rtol=1e-12, atol=1e-14)


def test_fused_nested_gram_error_semantics(raising):
    rng = np.random.default_rng(2)
    x, w = _data(rng, 20, 8, 2)
    bad = x.copy()
    bad[3] = -bad[3]                                       # negative definite: its projection too
    X, B, W = torch.tensor(x, device=DEV), torch.tensor(bad, device=DEV), torch.tensor(w, device=DEV)
    with raising("not positive definite"):
        ops.nested_spd_gram(B, X, W, 0.5)                  # x1 is factored: raises (spd_utils_torch.py:87)
    k = ops.nested_spd_gram(X, B, W, 0.5).cpu().numpy()    # x2: a NaN column (spd_utils_torch.py:109-120)
    assert np.isnan(k[:, 3]).all() and not np.isnan(np.delete(k, 3, axis=1)).any()
    ops.nested_spd_gram(X, X, W, 0.5)                      # the status word is clean again after the raise
    ops.check_deferred()
