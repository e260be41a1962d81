"""Error behaviour of the host layer and the C ABI on a GPU box (argument validation, shape contracts, dtype/device handling)."""
import numpy as np
import pytest
import torch

from gabotorch_amd import _lib, ops
from gabotorch_amd.kernel_utils.kernels_spd import SpdAffineInvariantGaussianKernel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_shape_contracts():
    a = torch.rand(4, 6, dtype=torch.float64, device=DEV) + torch.tensor([3., 3, 3, 0, 0, 0], dtype=torch.float64, device=DEV)
    with pytest.raises(RuntimeError, match="not d\\(d\\+1\\)/2"):
        ops.spd_ai_pairwise(torch.ones(3, 7, device=DEV), torch.ones(3, 7, device=DEV))
    with pytest.raises(RuntimeError, match="shapes differ"):
        ops.spd_ai_pairwise(a[None].expand(2, 4, 6), a)                      # batch shapes must match (no broadcasting)
    with pytest.raises(RuntimeError, match="shapes differ"):
        ops.sphere_pairwise(torch.ones(3, 4, device=DEV), torch.ones(3, 5, device=DEV))
    with pytest.raises(RuntimeError, match="same length"):
        ops.sphere_pairwise(torch.ones(3, 4, device=DEV), torch.ones(2, 4, device=DEV), diag=True)
    # fp32 / CPU inputs are accepted, the result is fp64 on the caller's device
    k = ops.spd_ai_pairwise(a.float().cpu(), a.float().cpu(), beta=0.3)
    assert k.dtype == torch.float64 and k.device.type == "cpu" and k.shape == (4, 4)
    # non-contiguous views
    big = torch.zeros(4, 12, dtype=torch.float64, device=DEV)
    big[:, ::2] = a
    np.testing.assert_allclose(ops.spd_ai_pairwise(big[:, ::2], a, beta=0.3).cpu().numpy(), ops.spd_ai_pairwise(a, a, beta=0.3).cpu().numpy())


def test_c_abi_argument_errors_on_device():
    lib = _lib.load()
    x = torch.ones(2, 3, dtype=torch.float64, device=DEV)
    out = torch.empty(2, 2, dtype=torch.float64, device=DEV)
    st = torch.zeros(2, dtype=torch.int32, device=DEV)
    ws = torch.empty(64, dtype=torch.float64, device=DEV)
    # workspace too small / symmetric with n1 != n2 / negative sizes
    assert lib.gabo_spd_ai_pairwise(x.data_ptr(), x.data_ptr(), out.data_ptr(), None, 1, 2, 2, 2, 6, 6, 1.0, 0, ws.data_ptr(), 8, st.data_ptr(), None) == _lib.GABO_ERR_ARG
    assert lib.gabo_spd_ai_pairwise(x.data_ptr(), x.data_ptr(), out.data_ptr(), None, 1, 2, 1, 2, 6, 6, 1.0, _lib.GABO_SYMMETRIC, ws.data_ptr(), 512, st.data_ptr(), None) == _lib.GABO_ERR_ARG
    assert lib.gabo_spd_ai_pairwise(x.data_ptr(), x.data_ptr(), out.data_ptr(), None, -1, 2, 2, 2, 6, 6, 1.0, 0, ws.data_ptr(), 512, st.data_ptr(), None) == _lib.GABO_ERR_ARG
    assert lib.gabo_spd_manifold_op(99, x.data_ptr(), None, None, None, out.data_ptr(), None, 1, 2, None, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_spd_project(x.data_ptr(), x.data_ptr(), out.data_ptr(), 1, 2, 65, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_sphere_from_inner(x.data_ptr(), out.data_ptr(), 4, 1.0, 0, 3, None) == _lib.GABO_ERR_ARG


def test_device_solve_reports_a_non_spd_start():
    """The single-launch solve writes the device status word like every other entry point: a non-SPD starting matrix raises."""
    from gabotorch_amd import manifolds, models
    from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
    from gabotorch_amd.manifold_optimization.manifold_optimize import gen_candidates_manifold
    from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch, vector_to_symmetric_matrix_mandel_torch
    rng = np.random.default_rng(0)
    q = np.linalg.qr(rng.standard_normal((8, 3, 3)))[0]
    Xm = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.5, 2.0, (8, 3)), q)
    X = torch.tensor(Xm, device=DEV)
    xv = symmetric_matrix_to_vector_mandel_torch(X)
    gp = models.ExactGP(xv, torch.tensor(rng.standard_normal(8), device=DEV), SpdAffineInvariantGaussianKernel(beta_min=0.5), outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=0.0, maximize=False)
    bad = xv[:3].clone()
    bad[1, :3] = torch.tensor([1.0, -2.0, 1.0], dtype=torch.float64, device=DEV)        # a negative diagonal entry: not SPD
    with pytest.raises(RuntimeError, match="not positive definite"):
        gen_candidates_manifold(bad[:, None], acq, manifolds.PositiveDefinite(3), BatchedTrustRegions(maxiter=3),
                                vector_to_symmetric_matrix_mandel_torch, symmetric_matrix_to_vector_mandel_torch, approx_hessian=True)


def test_a_refused_solve_consumes_its_record_buffer():
    """gabo_tr_solve_record hands ONE pending buffer to the next single-launch solve.  A solve that is refused for its arguments must still
    consume it - otherwise the pointer would be written by a later, unrelated solve (possibly after the tensor is gone)."""
    from gabotorch_amd import manifolds, models
    from gabotorch_amd.manifold_optimization.batched_trust_regions import BatchedTrustRegions
    from gabotorch_amd.manifold_optimization.manifold_optimize import gen_candidates_manifold
    from gabotorch_amd.Riemannian_utils.spd_utils_torch import symmetric_matrix_to_vector_mandel_torch, vector_to_symmetric_matrix_mandel_torch
    lib = _lib.load()
    buf = torch.full((4, 3, 11), float("nan"), dtype=torch.float64, device=DEV)
    assert lib.gabo_tr_solve_record(buf.data_ptr(), 4) == _lib.GABO_OK
    # refused: d = 1
    assert lib.gabo_spd_tr_solve(None, None, None, None, None, None, None, None, 0, None, None, 0, None, 0, 3, 1, 1e-6, 1.0, 0.1, 1, 3, 1.0, 0.1,
                                 1e3, 1e-6, 10, None, None, None, 0, None, None) == _lib.GABO_ERR_DIM
    rng = np.random.default_rng(0)
    q = np.linalg.qr(rng.standard_normal((8, 3, 3)))[0]
    Xm = np.einsum("nab,nb,ncb->nac", q, rng.uniform(0.5, 2.0, (8, 3)), q)
    xv = symmetric_matrix_to_vector_mandel_torch(torch.tensor(Xm, device=DEV))
    gp = models.ExactGP(xv, torch.tensor(rng.standard_normal(8), device=DEV), SpdAffineInvariantGaussianKernel(beta_min=0.5), outputscale=1.0, noise=1e-2)
    acq = models.ExpectedImprovement(gp, best_f=0.0, maximize=False)
    solver = BatchedTrustRegions(maxiter=3)
    gen_candidates_manifold(xv[:3, None], acq, manifolds.PositiveDefinite(3), solver, vector_to_symmetric_matrix_mandel_torch,
                            symmetric_matrix_to_vector_mandel_torch, approx_hessian=True)
    torch.cuda.synchronize()
    assert "one_launch_solve" in solver.log and bool(torch.isnan(buf).all())
    # and withdrawing a pending buffer works
    assert lib.gabo_tr_solve_record(buf.data_ptr(), 4) == _lib.GABO_OK and lib.gabo_tr_solve_record(None, 0) == _lib.GABO_OK
    gen_candidates_manifold(xv[:3, None], acq, manifolds.PositiveDefinite(3), solver, vector_to_symmetric_matrix_mandel_torch,
                            symmetric_matrix_to_vector_mandel_torch, approx_hessian=True)
    torch.cuda.synchronize()
    assert bool(torch.isnan(buf).all())
