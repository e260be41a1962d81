/*
 * gabo_hip.h - C ABI of libgabo_hip.so, the MI355X (gfx950) implementation of GaBOtorch's manifold-kernel hot path.
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream; no torch types cross this boundary.
 * All matrices/vectors are fp64, row-major, contiguous in their last dimension(s).  The library allocates nothing
 * persistent: scratch is passed in by the caller (`workspace`, sized by the matching *_workspace_bytes call).
 * Calls are stream-ordered, re-entrant, and keep no global state.  Return value: 0 on success, a negative
 * GABO_ERR_* code for argument errors detected on the host.  Data errors detected on the device (a non-positive
 * Cholesky pivot = input not SPD) are reported through `status`: a caller-owned device int[2], zeroed by the
 * caller, that receives {GABO_ERR_NOT_SPD, index of the first offending matrix}.  Nothing throws across the ABI.
 *
 * Each function cites the reference code it replaces (paths relative to the GaBOtorch tree, BoManifolds/...).
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 */
#ifndef GABO_HIP_H
#define GABO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gabo_stream_t; /* hipStream_t */

#define GABO_OK 0
#define GABO_ERR_DIM (-1)       /* unsupported matrix / vector dimension */
#define GABO_ERR_ARG (-2)       /* null pointer, negative size, workspace too small */
#define GABO_ERR_NOT_SPD (-3)   /* device-side: Cholesky pivot <= 0 (reference: torch.cholesky RuntimeError) */
#define GABO_ERR_LAUNCH (-4)    /* hipGetLastError() != hipSuccess after a launch */

/* `flags` of the pairwise kernels */
#define GABO_OUT_GAUSSIAN 0     /* out = exp(-beta * d^2)            kernels_spd.py:96-98, kernels_sphere.py:91-93 */
#define GABO_OUT_DISTANCE 1     /* out = d                           spd_utils_torch.py:120, sphere_utils_torch.py:55 */
#define GABO_OUT_LAPLACE 2      /* out = exp(-beta * d)              kernels_spd.py:185, kernels_sphere.py:133 */
#define GABO_OUT_MASK 3
#define GABO_SYMMETRIC 4        /* x1 and x2 are the same set (n1 == n2): evaluate i <= j only and mirror */
/* distance used by the fused acquisition kernels (gabo_spd_acq_params.flags, OR-ed with GABO_OUT_GAUSSIAN / GABO_OUT_LAPLACE) */
#define GABO_METRIC_AFFINE_INVARIANT 0
#define GABO_METRIC_LOG_EUCLIDEAN 8   /* kernels_spd.py:244-313; Gaussian only, d <= 8 */
#define GABO_METRIC_FROBENIUS 16      /* kernels_spd.py:190-241; Gaussian only, d <= 8 in the fused acquisition / trust-region kernels (round 5; larger d
                                         and the Laplace form: the separate-launch chain gabo_frobenius_pairwise -> gabo_gp_acquisition ->
                                         gabo_frobenius_backward); train_factors = the Mandel vectors of the training points, entry-major */
#define GABO_METRIC_MASK 24

/* SPD pairwise kernels: 2 <= d <= GABO_SPD_MAX_DIM.  The forward runs the register-resident lane-per-pair kernels (the fast path,
 * the metric) up to GABO_SPD_FWD_REG_MAX_DIM; the closed-form backward up to GABO_SPD_BWD_REG_MAX_DIM, the fused acquisition kernels
 * up to GABO_SPD_REG_MAX_DIM; larger d falls back to one wave per pair with LDS tiles. */
#define GABO_SPD_REG_MAX_DIM 12
#define GABO_SPD_BWD_REG_MAX_DIM 16   /* the closed-form backward stays register-resident up to here (two lanes per pair from d = 12) */
#define GABO_SPD_FWD_REG_MAX_DIM 20   /* the forward kernels stay register-resident up to here (one wave per SIMD above 12) */
#define GABO_SPD_MAX_DIM 32

int gabo_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * SPD affine-invariant pairwise kernel matrix.
 * Replaces  SpdAffineInvariantGaussianKernel.forward        kernel_utils/kernels_spd.py:72-100
 *           SpdAffineInvariantLaplaceKernel.forward         kernel_utils/kernels_spd.py:157-187
 *           affine_invariant_distance_torch                 Riemannian_utils/spd_utils_torch.py:53-121
 *           vector_to_symmetric_matrix_mandel_torch (fused) Riemannian_utils/spd_utils_torch.py:159-194
 *
 * x1: batch x n1 x d_vec Mandel vectors, x2: batch x n2 x d_vec, d_vec = d(d+1)/2, out: batch x n1 x n2.
 * dist_out: NULL, or a second batch x n1 x n2 buffer that receives the distances d_ij alongside `out` (saved for the
 * beta-gradient by the autograd wrapper).
 * x{1,2}_batch_stride: distance in doubles between consecutive batches (0 = one set shared by every batch, which is
 * how gpytorch hands over the expanded training inputs).  2 <= d <= GABO_SPD_MAX_DIM.
 * d_ij = sqrt(sum_k log^2 lambda_k(L_i^-1 X2_j L_i^-T) + 1e-15), L_i = chol(X1_i).
 */
size_t gabo_spd_ai_workspace_bytes(int64_t batch, int64_t n1, int64_t n2, int d);
int gabo_spd_ai_pairwise(const double* x1, const double* x2, double* out, double* dist_out, int64_t batch, int64_t n1,
                         int64_t n2, int d, int64_t x1_batch_stride, int64_t x2_batch_stride, double beta, int flags,
                         void* workspace, size_t workspace_bytes, int* status, gabo_stream_t stream);

/* Gradient of the above with respect to x1 (Mandel), given grad_out = dLoss/d(out) for the SAME `flags` output mode.
 * Replaces the reference's autograd pass through cholesky/inverse/bmm/symeig(eigenvectors=True)
 * (Riemannian_utils/spd_utils_torch.py:87-120, "Eigenvalue True necessary for derivation" :110); closed form in SURVEY App. C.
 * grad_out is read as grad_out[b*go_batch_stride + i*go_row_stride + j*go_col_stride] (element strides), so the gradient
 * with respect to x2 is this same call with the two sets exchanged and the row/column strides swapped.
 * grad_x1: batch x n1 x d_vec, dense (when x1 is shared across the batch, stride 0, the caller sums over the batch).
 * Workspace: gabo_spd_ai_workspace_bytes(batch, n1, n2, d).  GABO_SYMMETRIC is not accepted. */
int gabo_spd_ai_backward(const double* x1, const double* x2, const double* grad_out, double* grad_x1, int64_t batch, int64_t n1,
                         int64_t n2, int d, int64_t x1_batch_stride, int64_t x2_batch_stride, int64_t go_batch_stride,
                         int64_t go_row_stride, int64_t go_col_stride, double beta, int flags, void* workspace,
                         size_t workspace_bytes, int* status, gabo_stream_t stream);

/* Second-order term of the same kernels with respect to x1: what a SECOND autograd pass through the reference's chain returns
 * (pymanopt_addons/tools/autodiff/_pytorch.py:103-116 asks torch.autograd.grad of <gradient, vector> for exact Hessian-vector products; the chain
 * is Riemannian_utils/spd_utils_torch.py:87-120 with symeig(eigenvectors=True) at :110).  With grad_out held fixed and u (batch x n1 x d_vec,
 * Mandel) a direction for x1:
 *   hv_x1[b,i,:]       = d/dt grad_x1(x1 + t u)[b,i,:] at t = 0     (the Hessian of sum(grad_out * out) is block diagonal in i)
 *   d_grad_out[b,i,j]  = <u[b,i,:], d out[b,i,j] / d x1[b,i,:]>     (derivative of <grad_x1, u> with respect to grad_out; NULL to skip; dense)
 *   mixed_x2[b,j,:]    = d <grad_x1[b], u[b]> / d x2[b,j,:]          (the mixed block; NULL to skip; batch x n2 x d_vec dense - with x2 shared across
 *                        the batch, stride 0, the caller sums over the batch)
 * Closed form through second divided differences of t log t on the eigen-decomposition of each pair (csrc/spd_backward2.hip); every
 * 2 <= d <= GABO_SPD_MAX_DIM, one wave per pair.  grad_out strides as in gabo_spd_ai_backward; x1 is per batch (x1_batch_stride >= n1 * d_vec, or 0
 * with batch = 1).  The second derivative with respect to x2 alone is this call with the two sets exchanged (the distance is symmetric in its
 * arguments); mixed derivatives with beta are not provided. */
size_t gabo_spd_ai_backward2_workspace_bytes(int64_t batch, int64_t n1, int64_t n2, int d);
int gabo_spd_ai_backward2(const double* x1, const double* x2, const double* grad_out, const double* u, double* hv_x1, double* d_grad_out,
                          double* mixed_x2, int64_t batch, int64_t n1, int64_t n2, int d, int64_t x1_batch_stride, int64_t x2_batch_stride,
                          int64_t go_batch_stride, int64_t go_row_stride, int64_t go_col_stride, double beta, int flags, void* workspace,
                          size_t workspace_bytes, int* status, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Sphere pairwise kernel matrix.
 * Replaces  SphereGaussianKernel.forward / SphereLaplaceKernel.forward   kernel_utils/kernels_sphere.py:71-94,118-134
 *           sphere_distance_torch                                        Riemannian_utils/sphere_utils_torch.py:12-55
 * x1: batch x n1 x dim, x2: batch x n2 x dim, out: batch x n1 x n2;  d_ij = acos(clamp(<x1_i,x2_j>, -1+1e-15, 1-1e-15)).
 * diag != 0 pairs row k of x1 with row k of x2 (n1 == n2) and writes batch x n1 x 1 (sphere_utils_torch.py:45-49).
 */
int gabo_sphere_pairwise(const double* x1, const double* x2, double* out, int64_t batch, int64_t n1, int64_t n2,
                         int dim, int64_t x1_batch_stride, int64_t x2_batch_stride, double beta, int flags, int diag,
                         gabo_stream_t stream);
/* The same launch for a caller that evaluates many Gram matrices with ONE beta (every kernel matrix of a BO iteration after the fit:
 * kernels_sphere.py:90-94 is called with the fitted beta throughout).  Large Gaussian Gram matrices read the kernel VALUE from a table that
 * depends on beta only (csrc/sphere_pairwise.hip); built inside every launch it is a prologue without stores in flight.  The caller may
 * build it once - gabo_sphere_ktable_build into a buffer of gabo_sphere_ktable_doubles() doubles, on the stream of the launches that use
 * it - and hand it over as `ktable` (NULL: built in the launch, exactly gabo_sphere_pairwise).  The table must belong to the `beta`
 * passed here; it is ignored when the launch does not take the table path (gabo_sphere_pairwise_uses_ktable == 0).  Results are
 * bit-identical with and without it. */
int64_t gabo_sphere_ktable_doubles(void);
int gabo_sphere_pairwise_uses_ktable(int64_t batch, int64_t n1, int64_t n2, int dim, double beta, int flags, int diag);
int gabo_sphere_ktable_build(double beta, double* table, gabo_stream_t stream);
int gabo_sphere_pairwise_cached(const double* x1, const double* x2, double* out, int64_t batch, int64_t n1, int64_t n2,
                                int dim, int64_t x1_batch_stride, int64_t x2_batch_stride, double beta, int flags, int diag,
                                const double* ktable, gabo_stream_t stream);

/* Element-wise value (order 0), first (1) or second (2) derivative, with respect to the inner product c, of
 * f(c) = g(acos(clamp(c, -1+1e-15, 1-1e-15))) with g chosen by `flags` as above; derivatives are 0 where the clamp is active.
 * `inner` and `out`: n doubles.  This is the differentiable route (c = x1 x2^T is a plain GEMM): the reference differentiates
 * sphere_distance_torch / SphereGaussianKernel.forward by autograd, and twice for the exact Hessian-vector products of the
 * sphere trust region (pymanopt_addons/tools/autodiff/_pytorch.py:103-116). */
int gabo_sphere_from_inner(const double* inner, double* out, int64_t n, double beta, int flags, int order, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Mandel vector <-> symmetric matrix.
 * Replaces  vector_to_symmetric_matrix_mandel_torch   Riemannian_utils/spd_utils_torch.py:159-194
 *           symmetric_matrix_to_vector_mandel_torch   Riemannian_utils/spd_utils_torch.py:197-226 (averages both triangles, :219)
 * vec: n x d_vec, mat: n x d x d.   1 <= d <= 64.
 */
int gabo_mandel_to_matrix(const double* vec, double* mat, int64_t n, int d, gabo_stream_t stream);
int gabo_matrix_to_mandel(const double* mat, double* vec, int64_t n, int d, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Batched Riemannian operations on the SPD manifold (one wave per matrix, LDS tiles, 1 <= d <= 32).
 * All matrices are full n x d x d row-major batches (what pymanopt's PositiveDefinite and the reference's numpy maps use).
 * Replaces, per restart of the acquisition maximiser (BoManifolds/manifold_optimization/...):
 *   GABO_SPD_EXP          out = X expm(X^-1 U)             a=X b=U        spd_utils.py:104-120 ; [3P] PositiveDefinite.exp = retr
 *   GABO_SPD_LOG          out = Log_X(Y)                   a=X b=Y        spd_utils.py:123-139 ; tools/multi.py:55-64
 *   GABO_SPD_INNER        out[n] = tr(X^-1 U X^-1 V)       a=X b=U c=V    [3P] inner  (robust_trust_regions.py:148)
 *   GABO_SPD_NORM         out[n] = sqrt(inner(U,U))        a=X b=U        [3P] norm
 *   GABO_SPD_DIST         out[n] = ||logm(L^-1 Y L^-T)||_F a=X b=Y        [3P] dist   (examples/gabo_spd.py:289)
 *   GABO_SPD_EGRAD2RGRAD  out = X sym(G) X                 a=X b=G        [3P] (pymanopt_addons/problem.py:135)
 *   GABO_SPD_EHESS2RHESS  out = X sym(H) X + sym(U sym(G) X)  a=X b=G c=H e=U   [3P] (problem.py:156)
 *   GABO_SPD_LOGM/EXPM/SQRTM  out = f(A)                   a=A            spd_utils_torch.py:13-50 ; tools/multi.py:55-75
 *                             out2 (NULL to skip) = the eigen-decomposition, n x (d*d + d): V row-major, then the d eigenvalues;
 *                             gabo_spd_matfun_backward_eig takes it instead of solving the eigen-problem again
 *   GABO_SPD_EIGMAX/EIGMIN    out[n] = extreme eigenvalue, out2 = v v^T (NULL to skip)   spd_constraints_utils_torch.py:17-50
 * status: device int[2] as above (non-SPD base point) or NULL.
 */
#define GABO_SPD_EXP 0
#define GABO_SPD_LOG 1
#define GABO_SPD_INNER 2
#define GABO_SPD_NORM 3
#define GABO_SPD_DIST 4
#define GABO_SPD_EGRAD2RGRAD 5
#define GABO_SPD_EHESS2RHESS 6
#define GABO_SPD_LOGM 7
#define GABO_SPD_EXPM 8
#define GABO_SPD_SQRTM 9
#define GABO_SPD_EIGMAX 10
#define GABO_SPD_EIGMIN 11
int gabo_spd_manifold_op(int op, const double* a, const double* b, const double* c, const double* e, double* out, double* out2,
                         int64_t n, int d, int* status, gabo_stream_t stream);

/* Y = W^T X W, Mandel in (n x D(D+1)/2) -> Mandel out (n x dl(dl+1)/2); w: D x dl row-major.  1 <= D, dl <= 64
 * (dl > D is the adjoint map G -> W G W^T of the projection, i.e. its gradient with respect to X).
 * Replaces projection_from_spd_to_nested_spd (nested_mappings/nested_spd_utils.py:13-48) fused with both Mandel maps. */
int gabo_spd_project(const double* x_mandel, const double* w, double* y_mandel, int64_t n, int D, int dl, gabo_stream_t stream);

/* logm of n SPD matrices, Mandel in -> Mandel out: the per-point half of SpdLogEuclideanGaussianKernel.forward
 * (kernel_utils/kernels_spd.py:289-305, logm_torch spd_utils_torch.py:13-30). */
int gabo_spd_logm_mandel(const double* x_mandel, double* y_mandel, int64_t n, int d, gabo_stream_t stream);

/* Pairwise Frobenius distance of symmetric matrices given as Mandel vectors, with the reference's +1e-15 on every matrix
 * element of the difference (frobenius_distance_torch, spd_utils_torch.py:124-156); `flags`/`beta` as for the other pairwise
 * kernels (SpdFrobeniusGaussianKernel / SpdLogEuclideanGaussianKernel use exp(-d^2 / lengthscale^2): beta = 1/lengthscale^2,
 * kernels_spd.py:238-240,309-311).  out: batch x n1 x n2. */
int gabo_frobenius_pairwise(const double* x1, const double* x2, double* out, int64_t batch, int64_t n1, int64_t n2, int d,
                            int64_t x1_batch_stride, int64_t x2_batch_stride, double beta, int flags, gabo_stream_t stream);

/* Gram matrix of the nested SPD kernels in two launches (no gradient): project both point sets (Y = W^T X W, nested_spd_utils.py:13-48) and finish
 * every latent matrix in the same launch - Cholesky factor / inverse for the affine-invariant kernel, Mandel vector of logm for the log-Euclidean
 * one - then the Gram launch.  Replaces NestedSpdAffineInvariantGaussianKernel.forward / NestedSpdLogEuclideanGaussianKernel.forward
 * (kernel_utils/kernels_nested_spd.py:104-136, 191-246) when no gradient is requested.
 *   x1: batch x n1 x D_vec, x2: batch x n2 x D_vec (Mandel, dense), w: D x dl (row-major), out: batch x n1 x n2; 2 <= dl <= 4, dl <= D <= 32.
 *   metric: GABO_METRIC_AFFINE_INVARIANT (out = exp(-beta d_AI^2), Laplace / distance with `flags` as in gabo_spd_ai_pairwise, GABO_SYMMETRIC when
 *           x1 is x2) or GABO_METRIC_LOG_EUCLIDEAN (out = exp(-beta ||logm Y1 - logm Y2 + 1e-15||_F^2), beta = 1 / lengthscale^2).
 *   x1 == x2 (same pointer, n1 == n2): the set is projected once.  status as in gabo_spd_ai_pairwise (a latent matrix of x1 that is not SPD is reported). */
size_t gabo_nested_spd_gram_workspace_bytes(int64_t batch, int64_t n1, int64_t n2, int dl);
int gabo_nested_spd_gram(const double* x1, const double* x2, const double* w, double* out, int64_t batch, int64_t n1, int64_t n2, int D, int dl,
                         int metric, double beta, int flags, void* workspace, size_t workspace_bytes, int* status, gabo_stream_t stream);

/* Gradients of the two calls above (the reference differentiates them by autograd; kernels_spd.py:238-240,305-311).
 * gabo_spd_logm_mandel_backward: grad_x = Mandel( V ((V^T G V) o F) V^T ), the adjoint Frechet derivative of logm at X applied
 *   to G = grad_y (Mandel), F the divided differences of log at the eigenvalues of X.
 * gabo_frobenius_backward: grad_x1[b,i,:] = sum_j grad_out[b,i,j] dOut_ij/dx1_i for the same `flags`/`beta`; grad_out strides as
 *   in gabo_spd_ai_backward.  The gradient with respect to x2 is the same call with the sets exchanged, the row/column strides
 *   swapped and eps_sign = -1 (the reference's +1e-15 is added to x1 - x2, so it changes sign with the roles); eps_sign = +1
 *   otherwise. */
/* Gradient of GABO_SPD_LOGM / GABO_SPD_EXPM / GABO_SPD_SQRTM (gabo_spd_manifold_op) with respect to the input matrices:
 * grad_a = V ((V^T sym(grad_out) V) o F) V^T, F = divided differences of log / exp / sqrt at the eigenvalues of a (n x d x d each).
 * Replaces autograd through logm_torch / sqrtm_torch (spd_utils_torch.py:13-50), e.g. in the reconstruction costs of
 * nested_mappings/nested_spd_optimization.py:23-92. */
int gabo_spd_matfun_backward(int op, const double* a, const double* grad_out, double* grad_a, int64_t n, int d, gabo_stream_t stream);
/* The same from the eigen-decomposition the forward call wrote to out2 (`eig`: n x (d*d + d)): no eigen-solve, four small products. */
int gabo_spd_matfun_backward_eig(int op, const double* eig, const double* grad_out, double* grad_a, int64_t n, int d,
                                 gabo_stream_t stream);
int gabo_spd_logm_mandel_backward(const double* x_mandel, const double* grad_y, double* grad_x, int64_t n, int d,
                                  gabo_stream_t stream);
int gabo_frobenius_backward(const double* x1, const double* x2, const double* grad_out, double* grad_x1, int64_t batch, int64_t n1,
                            int64_t n2, int d, int64_t x1_batch_stride, int64_t x2_batch_stride, int64_t go_batch_stride,
                            int64_t go_row_stride, int64_t go_col_stride, double beta, int flags, double eps_sign,
                            gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused exact-GP posterior + analytic acquisition on a cross-covariance strip: the consumer of the kernel strip
 * K(X*, X_train) inside the acquisition maximiser (manifold_optimize.py:182-184; botorch ExpectedImprovement / PosteriorMean
 * over a gpytorch ExactGP, [3P] semantics in SURVEY App. B).  One launch replaces the posterior algebra and its autograd.
 *   kstar: r x n BASE kernel values k(x*_r, x_train_j) (before the outputscale); alpha = (K + noise I)^-1 (y - mean), n;
 *   linv = L^-1 and linv_t = L^-T (both n x n row-major DENSE, L = chol(K + noise I), with exact zeros outside their triangle: the
 *   SPD kernels sum over whole rows / columns); kxx = base kernel k(x*, x*).
 *   value[r]      = out_sign * acquisition(x*_r)
 *   grad_kstar    = NULL or r x n: out_sign * d acquisition / d kstar   (chain it into gabo_spd_ai_backward / the sphere backward)
 *   kind GABO_ACQ_EXPECTED_IMPROVEMENT: sigma = sqrt(max(var, 1e-9)), u = +-(mean - best_f)/sigma, EI = sigma (phi(u) + u Phi(u))
 *        GABO_ACQ_POSTERIOR_MEAN:       +-mean          (maximize != 0 selects '+'). */
#define GABO_ACQ_EXPECTED_IMPROVEMENT 0
#define GABO_ACQ_POSTERIOR_MEAN 1
int gabo_gp_acquisition(const double* kstar, const double* alpha, const double* linv, const double* linv_t, double* value,
                        double* grad_kstar, int64_t r, int64_t n, double mean, double outputscale, double kxx, double best_f,
                        int kind, int maximize, double out_sign, gabo_stream_t stream);

/* The prediction cache those two entry points read, in one launch: L = chol(outputscale * k + noise * I) (k: n x n BASE kernel matrix of the
 * training set, row-major, lower triangle read), linv = L^-1 and linv_t = L^-T (n x n row-major, exact zeros outside their triangle) and
 * alpha = (outputscale * k + noise * I)^-1 (y - mean).  Replaces the Cholesky / cholesky_solve / triangular solve a fitted [3P] gpytorch
 * ExactGP runs when it is first asked for a posterior (behind manifold_optimize.py:182-184).  One workgroup, both factors in LDS:
 * n <= GABO_GP_FACTOR_MAX_N (GABO_ERR_DIM beyond: factor with a library call instead).  status = {GABO_ERR_NOT_SPD, 0} when a pivot is
 * not positive (outputs then unspecified).
 * kinv (may be NULL): the symmetric inverse (outputscale * k + noise * I)^-1 = L^-T L^-1, n x n.  The fused SPD acquisition kernels
 * (gabo_spd_acq_eval and everything built on it) accept it IN PLACE OF the two factors - pass the same pointer as linv and linv_t: the posterior
 * variance k** - ks^T A ks and its gradient -2 A ks then cost one n-term product per training point instead of two triangular ones.
 * gabo_gp_acquisition takes the factors. */
#define GABO_GP_FACTOR_MAX_N 96
int gabo_gp_factor(const double* k, const double* y, int64_t n, double outputscale, double noise, double mean, double* linv,
                   double* linv_t, double* alpha, double* kinv, int* status, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Exact-GP marginal log likelihood and its analytic gradient, one launch per evaluation of the surrogate fit
 * (fit_gpytorch_model(mll) at examples/bo_spd/benchmark_examples/gabo_spd.py:194 and its siblings; [3P] gpytorch
 * ExactMarginalLogLikelihood semantics).  Every kernel of the path is exp(-theta * E) with E = d^2 (Gaussian) or
 * d (Laplace), fixed during a fit: evaluate the distances once with GABO_OUT_DISTANCE, then call this per L-BFGS evaluation.
 *   e: n x n (row-major, symmetric; the lower triangle is read), y: n targets,
 *   Ky = outputscale * exp(-theta * e) + noise * I.
 *   out[0] = log N(y | mean, Ky)  (no priors, not divided by n)
 *   out[1..4] = d out[0] / d theta, d outputscale, d noise, d mean
 *   out[5] = 1.0 when Ky is not numerically positive definite (out[0..4] are then 0), else 0.0.
 * One workgroup, the bordered matrix [[Ky, r], [r^T, 0]] held in registers and inverted in place by n symmetric sweeps (one barrier
 * each): n <= GABO_GP_MLL_MAX_N (GABO_ERR_DIM beyond). */
#define GABO_GP_MLL_MAX_N 160
int gabo_gp_mll(const double* e, const double* y, int64_t n, double theta, double outputscale, double noise, double mean,
                double* out, gabo_stream_t stream);
/* The same for a Gram matrix that is not of the exp(-theta E) form (kernels with parameters inside the distance: the nested kernels
 * fitted by fit_gpytorch_manifold, manifold_gp_fit.py:54-222): k = BASE kernel matrix (n x n, lower triangle read),
 * Ky = outputscale * k + noise * I.  out as above with out[1] = 0; w (NULL to skip) receives W = alpha alpha^T - Ky^-1 (n x n), so that
 * d out[0] / d k = outputscale * W / 2 chains into the kernel's own backward - no Cholesky / solve / log-det autograd. */
int gabo_gp_mll_gram(const double* k, const double* y, int64_t n, double outputscale, double noise, double mean, double* out, double* w,
                     gabo_stream_t stream);
/* Both of the above for larger training sets, n <= GABO_GP_MLL_LARGE_MAX_N: the same sweep operator on 32 x 32 tiles of the bordered
 * matrix in the caller's workspace, one launch per block of 32 pivots (csrc/gp_mll_large.hip).  gram = 0: e as in gabo_gp_mll;
 * gram != 0: e is the base Gram matrix as in gabo_gp_mll_gram (theta unused, out[1] = 0).  w: NULL or n x n as above.  Both triangles
 * of e are read. */
#define GABO_GP_MLL_LARGE_MAX_N 2048
size_t gabo_gp_mll_large_workspace_bytes(int64_t n);
int gabo_gp_mll_large(const double* e, const double* y, int64_t n, double theta, double outputscale, double noise, double mean, int gram,
                      double* out, double* w, void* workspace, size_t workspace_bytes, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * One evaluation of the surrogate-fit objective of HD-GaBO with its gradient, as one HOST call (it returns with the numbers):
 * the marginal log likelihood of ScaleKernel(NestedSpdLogEuclideanGaussianKernel) at a projection matrix W and scalar hyper-parameters.
 * Replaces the closure fit_gpytorch_manifold differentiates by autograd   manifold_optimization/manifold_gp_fit.py:54-222
 *   kernel: NestedSpdLogEuclideanGaussianKernel.forward                    kernel_utils/kernels_nested_spd.py:139-250
 *           = projection_from_spd_to_nested_spd (nested_spd_utils.py:20-48) -> logm_torch (spd_utils_torch.py:13-30) -> exp(-theta ||.||_F^2)
 *   likelihood: [3P] gpytorch ExactMarginalLogLikelihood.
 * It issues gabo_spd_project, gabo_spd_logm_mandel, gabo_frobenius_pairwise, gabo_gp_mll_gram (gabo_gp_mll_large above GABO_GP_MLL_MAX_N) and
 * their backward launches back to back on `stream`, and waits.
 * x_mandel: n x D(D+1)/2 training points, x_matrices: the same as n x D x D, y: n targets (device); w_host: D x d (host).
 * out_host (host, 7 + D d doubles): [ll, 0, d ll/d outputscale, d ll/d noise, d ll/d mean, not-positive-definite flag, d ll/d theta,
 *   d ll / d W (D x d, Euclidean)]; want_grad = 0: only the first six are computed (the rest zero).  theta = 1 / lengthscale^2.
 * workspace: device, gabo_nested_spd_fit_workspace_bytes(n, D, d); pinned: >= 2 D d + 7 doubles of page-locked host memory.
 * 1 <= n <= GABO_GP_MLL_LARGE_MAX_N, 1 <= d < D <= GABO_SPD_MAX_DIM.
 */
size_t gabo_nested_spd_fit_workspace_bytes(int64_t n, int D, int d);
int gabo_nested_spd_fit_evaluate(const double* x_mandel, const double* x_matrices, const double* y, const double* w_host, int64_t n, int D,
                                 int d, double theta, double outputscale, double noise, double mean, int want_grad, double* out_host,
                                 void* workspace, size_t workspace_bytes, double* pinned, size_t pinned_doubles, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Nested-sphere mappings over all their levels in one launch (HD-GaBO on the sphere, examples/hd_bo_sphere/benchmark_examples/hd_gabo_sphere.py).
 * Level k = 0 .. levels-1 has the axis a_k on S^(D-k-1) (dimension D - k) and the distance r_k; `axes` are packed level after level
 * (sum_k (D - k) doubles).  frames: the rotation frames of the axes, sum_k (D - k) + 2 levels doubles, written by gabo_nested_sphere_frames
 *   (rotation_from_sphere_points_torch(axis, north pole), Riemannian_utils/sphere_utils_torch.py:58-93, as a rank-2 update).
 * gabo_nested_sphere_reconstruction replaces  min_error_reconstruction_cost   nested_mappings/nested_spheres_optimization.py:20-38
 *   = projection_from_subsphere_to_sphere (nested_spheres_utils.py:149-218) + sphere_distance_torch(diag) (sphere_utils_torch.py:12-55) + autograd:
 *   x_data: N x D, x_subsphere: N x (D - levels), distances: P x levels (P parameter sets in one launch: the optimiser's start candidates);
 *   cost[p] = sum_n acos(clamp(<x_n, reconstruction_p(z_n)>))^2, grad: P x levels = d cost / d distances, or NULL.
 * gabo_nested_sphere_fit_evaluate: one evaluation of the surrogate-fit objective of the same example with its gradient, one HOST call
 *   (fit_gpytorch_manifold's closure, manifold_optimization/manifold_gp_fit.py:54-222, for ScaleKernel(NestedSphereGaussianKernel),
 *   kernel_utils/kernels_nested_sphere.py:19-152 = projection_from_sphere_to_subsphere (nested_spheres_utils.py:68-146) -> Gaussian sphere kernel;
 *   [3P] gpytorch ExactMarginalLogLikelihood): x: n x D training points, y: n targets (device); axes_host (packed), distances_host (levels): host.
 *   out_host (7 + sum_k (D - k) doubles): [ll, 0, d/d outputscale, d/d noise, d/d mean, not-positive-definite flag, d/d beta, d ll / d axes
 *   (packed, Euclidean)]; want_grad = 0: the first six only.  workspace: device, ..._fit_workspace_bytes; pinned: >= 2 sum_k (D - k) + levels + 7
 *   doubles of page-locked host memory.  D - levels <= 64, n <= GABO_GP_MLL_LARGE_MAX_N.
 */
int gabo_nested_sphere_frames(const double* axes, double* frames, int D, int levels, gabo_stream_t stream);
/* gabo_nested_sphere_project: x (n x D) through every level -> z (n x (D - levels));  levels_in (or NULL): n x sum_k (D - k), the input of
 *   every level - the list projection_from_sphere_to_subsphere returns, without its last entry z (nested_spheres_utils.py:117-146).
 * gabo_nested_sphere_lift: x_subsphere (n x (D - levels)) back up -> x (n x D, or NULL); levels_out (or NULL): n x sum_k (D - k), the output of
 *   every level, level 0 (the final point) first - the list of projection_from_subsphere_to_sphere reversed (:182-218).  distances: `levels` doubles on the device. */
int gabo_nested_sphere_project(const double* x, const double* frames, const double* distances, double* z, double* levels_in, int64_t n, int D,
                               int levels, gabo_stream_t stream);
int gabo_nested_sphere_lift(const double* x_subsphere, const double* frames, const double* distances, double* x, double* levels_out, int64_t n,
                            int D, int levels, gabo_stream_t stream);
size_t gabo_nested_sphere_reconstruction_workspace_bytes(int64_t P, int64_t N, int D, int levels);
int gabo_nested_sphere_reconstruction(const double* x_data, const double* x_subsphere, const double* frames, const double* distances,
                                      double* cost, double* grad, int64_t P, int64_t N, int D, int levels, void* workspace,
                                      size_t workspace_bytes, gabo_stream_t stream);
size_t gabo_nested_sphere_fit_workspace_bytes(int64_t n, int D, int levels);
int gabo_nested_sphere_fit_evaluate(const double* x, const double* y, const double* axes_host, const double* distances_host, int64_t n, int D,
                                    int levels, double beta, double outputscale, double noise, double mean, int want_grad, double* out_host,
                                    void* workspace, size_t workspace_bytes, double* pinned, size_t pinned_doubles, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * n random SPD matrices with the distribution of spd_sample (Riemannian_utils/spd_utils.py:290-306; the raw samples of
 * gen_batch_initial_conditions_manifold, manifold_optimize.py:288): eigenvalues U[min_eig, max_eig], eigenvectors = orthogonal
 * factor of a Gaussian matrix.  out: n x d x d (mandel == 0) or n x d_vec Mandel vectors.  Counter-based stream (Philox4x32-10,
 * key = seed, counter = matrix index): reproducible for a seed, independent of the launch geometry, NOT numpy's stream. d <= 16. */
int gabo_spd_sample(double* out, int64_t n, int d, double min_eig, double max_eig, uint64_t seed, int mandel, gabo_stream_t stream);
/* Samples first ... first + n - 1 of the same stream (out holds n of them): the raw samples of one acquisition sweep sharded by sample
 * index over the GPUs of a node - the union over the ranks is bit-identical to one gabo_spd_sample call of the total. */
int gabo_spd_sample_range(double* out, int64_t first, int64_t n, int d, double min_eig, double max_eig, uint64_t seed, int mandel,
                          gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Nested-sphere projection S^(d-1) c R^d -> next subsphere, per-point part.
 * Replaces projection_from_sphere_to_nested_sphere / projection_from_sphere_to_next_subsphere
 * (nested_mappings/nested_spheres_utils.py:13-114), the body of NestedSphereGaussianKernel.forward
 * (kernel_utils/kernels_nested_sphere.py:125-152), and autograd through them.
 * rotated: n x d, the points already rotated so that the nested sphere's axis is the north pole (U = X R^T, a plain GEMM with
 * R = rotation_from_sphere_points_torch(axis, north), sphere_utils_torch.py:58-93).
 *   mode 0: out n x (d-1), the points of the next subsphere  (+1e-6 regularisations of :56,105,110 included)
 *   mode 1: out n x d, the nested-sphere points, still in the rotated frame (multiply by R to rotate back)
 * backward (mode 0): grad_rotated n x d from grad_out n x (d-1). */
int gabo_nested_sphere_epilogue(const double* rotated, double* out, int64_t n, int d, double dist_to_axis, int mode,
                                gabo_stream_t stream);
int gabo_nested_sphere_epilogue_backward(const double* rotated, const double* grad_out, double* grad_rotated, int64_t n, int d,
                                         double dist_to_axis, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Acquisition value and gradient at r candidate SPD points in one launch (one wave per candidate): the affine-invariant kernel
 * strip against the training set, gabo_gp_acquisition's posterior + EI / posterior mean, and the closed-form gradient back to
 * the candidate, fused.  Same reference code as gabo_spd_ai_pairwise + gabo_gp_acquisition + gabo_spd_ai_backward; exists because
 * the lock-step trust regions evaluate this once per inner iteration and are launch-count bound.  2 <= d <= GABO_SPD_REG_MAX_DIM.
 *   gabo_spd_acq_prepare_train: Cholesky factors of the n training matrices, entry-major (d_vec x n), once per surrogate.
 *   gabo_spd_acq_eval: x_mandel r x d_vec; value r; grad_mandel NULL or r x d_vec (then scratch: r * d_vec * n doubles);
 *     flags GABO_OUT_GAUSSIAN or GABO_OUT_LAPLACE; active: NULL, or r ints - candidates with active[i] == 0 are skipped and their
 *     outputs left untouched (the trust regions pass the running flags of gabo_spd_tcg_*); remaining arguments as in
 *     gabo_gp_acquisition. */
typedef struct {
    const double* train_factors; /* affine-invariant: gabo_spd_acq_prepare_train output; log-Euclidean / Frobenius: the Mandel vectors of
                                    logm(X_j) / X_j; always entry-major d_vec x n */
    const double* alpha;         /* n */
    const double* linv;          /* n x n dense, zeros above the diagonal */
    const double* linv_t;        /* n x n dense, zeros below the diagonal */
    int64_t n;
    double beta;
    int flags;                   /* GABO_OUT_GAUSSIAN / GABO_OUT_LAPLACE | GABO_METRIC_* */
    double mean, outputscale, kxx, best_f;
    int kind, maximize;          /* GABO_ACQ_*, maximize != 0 */
    double out_sign;
} gabo_spd_acq_params;           /* the surrogate, as one argument (same fields as gabo_spd_acq_eval's scalar list) */

/* largest training-set size the fused kernels take at dimension d (LDS budget; 2048 up to d = 11, 1755 at d = 12) */
int64_t gabo_spd_acq_max_train(int d);
int gabo_spd_acq_prepare_train(const double* x_train_mandel, double* train_factors, int64_t n, int d, int* status,
                               gabo_stream_t stream);
int gabo_spd_acq_eval(const double* x_mandel, const double* train_factors, const double* alpha, const double* linv,
                      const double* linv_t, double* value, double* grad_mandel, double* scratch, int64_t r, int64_t n, int d,
                      double beta, int flags, double mean, double outputscale, double kxx, double best_f, int kind, int maximize,
                      double out_sign, const int* active, int* status, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Truncated conjugate gradients of the Riemannian trust-region solvers on S^d_++, R restarts in lock step (one wave each).
 * Replaces the per-restart numpy loops of  TrustRegions._truncated_conjugate_gradient   robust_trust_regions.py:417-570
 *                                          ConstrainedTrustRegions (linearised constraints) constrained_trust_regions.py:530-732
 *                                          get_hessianfd                                    approximate_hessian.py:11-62
 * with the preconditioner of manifold_optimize.py:190-193.  One inner iteration = gabo_spd_tcg_fd_point, the caller's
 * acquisition gradient at the finite-difference points (Euclidean, Mandel), gabo_spd_tcg_step.
 *   begin:    x, grad (Riemannian gradient at x): r x d x d; cons_grads: n_constraints x r x d x d Riemannian gradients of the
 *             constraints (equalities first), cons_values: r x n_constraints; active: r bytes (0 = restart already converged);
 *             trust_radius: r.  Non-SPD x is reported through `status` like the other entry points.
 *   begin_rand (optional, after begin): `use_rand=True` of the reference (robust_trust_regions.py:173-181, 407-452) - the iteration starts
 *             from the caller's tiny random tangent vectors eta0 (r x d x d) with heta0 = hess(x, eta0) instead of zero, and the steps
 *             that follow run without the preconditioner (":411 and therefore, no preconditioner").  The comparison with the Cauchy
 *             point after `end` (:196-219) is the caller's.
 *   fd_point: writes the points x1 = retr(x, 2^-14 delta / ||delta||_x) as Mandel vectors (r x d_vec).
 *   step:     consumes the Euclidean gradient at those points (Mandel, r x d_vec) and advances every running restart;
 *             *any_running (device int) = 1 while at least one restart continues.  `mininner`: no residual test before.
 *   end:      eta, heta (r x d x d tangent vectors at x) and the stop reasons (0 negative curvature, 1 exceeded trust region,
 *             2/3 reached target linear/superlinear, 4 max inner iterations, 5 model increased, 6 reached constraints).
 * The state lives in `workspace` (gabo_spd_tcg_workspace_bytes); n_constraints <= 8. */
size_t gabo_spd_tcg_workspace_bytes(int64_t r, int d, int n_constraints);
/* byte offset, inside the workspace, of the r ints "restart still running" (readable mask for gabo_spd_acq_eval's `active`) */
size_t gabo_spd_tcg_running_offset(int64_t r, int d, int n_constraints);
int gabo_spd_tcg_begin(const double* x, const double* grad, const double* cons_grads, const double* cons_values,
                       const uint8_t* active, const double* trust_radius, void* workspace, size_t workspace_bytes, int64_t r, int d,
                       int n_constraints, int* status, gabo_stream_t stream);
int gabo_spd_tcg_begin_rand(void* workspace, const double* eta0, const double* heta0, int64_t r, int d, int n_constraints,
                            gabo_stream_t stream);
int gabo_spd_tcg_fd_point(void* workspace, double* x_fd_mandel, int64_t r, int d, int n_constraints, gabo_stream_t stream);
int gabo_spd_tcg_step(void* workspace, const double* egrad_fd_mandel, int* any_running, int64_t r, int d, int n_constraints,
                      int n_equalities, double delta_cons, double theta, double kappa, int mininner, gabo_stream_t stream);
int gabo_spd_tcg_end(void* workspace, double* eta, double* heta, int* stop_reason, int64_t r, int d, int n_constraints,
                     gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * One trust-region iteration of the acquisition maximiser on S^d_++ for r restarts (one wave each), 2 <= d <= 12.
 * Replaces one pass of the while-loop of TrustRegions.solve / ConstrainedTrustRegions.solve / StrictConstrainedTrustRegions.solve
 * (robust_trust_regions.py:180-415, constrained_trust_regions.py:190-528, 812-1160) together with everything it calls.
 *   propose: tCG from (x, grad = Riemannian gradient, trust_radius) under the linearised constraints, then the proposal
 *            x_prop = retr(x, eta) (r x d x d, written for the caller: the strict variant evaluates the constraints there) and the
 *            acquisition value / gradient at x_prop (kept in the workspace).  Restarts with active[i] == 0 are skipped.
 *   update:  rho test against `fx`, radius update (shrink / grow up to delta_bar), acceptance (rho > rho_prime), in-place update of
 *            x, fx, grad, grad_norm, trust_radius, iters; active[i] <- 0 when grad_norm < mingradnorm or iters >= maxiter;
 *            *any_active = 1 while a restart remains.  invalid: NULL or r bytes (strict variant: proposal violates a constraint).
 * The caller loops  [constraints at x] -> propose -> [constraints at x_prop] -> update  until *any_active == 0. */
size_t gabo_spd_tr_workspace_bytes(int64_t r, int d, int n_constraints, int64_t n_train);
int gabo_spd_tr_propose(const double* x, const double* grad, const double* trust_radius, const uint8_t* active,
                        const double* cons_grads, const double* cons_values, const gabo_spd_acq_params* acq, void* workspace,
                        size_t workspace_bytes, double* x_prop, int64_t r, int d, int n_constraints, int n_equalities, double delta_cons,
                        double theta, double kappa, int mininner, int maxinner, int* any_active, int* status, gabo_stream_t stream);
/* 1 when gabo_spd_tr_propose / gabo_spd_tr_update iterate a surrogate with these GABO_METRIC_* / GABO_OUT_* flags at dimension d:
 * affine-invariant 2 ... 12, log-Euclidean and Frobenius 2 ... 8.  Otherwise the caller composes the iteration from gabo_spd_tcg_* and gabo_spd_acq_eval
 * (or, where gabo_spd_tr_solve_supported, runs the single launch). */
int gabo_spd_tr_propose_supported(int flags, int d);
int gabo_spd_tr_update(double* x, double* fx, double* grad, double* grad_norm, double* trust_radius, uint8_t* active, int64_t* iters,
                       const uint8_t* invalid, const double* x_prop, void* workspace, int64_t r, int d, int n_constraints,
                       int64_t n_train, double delta_bar, double rho_prime, double rho_regularization, double mingradnorm,
                       int64_t maxiter, int* any_active, gabo_stream_t stream);

/* The whole solve in ONE launch: every wave iterates propose/update for its restart until its gradient norm or iteration limit is
 * reached.  Possible when the constraints need no host callable: none, or bounds on the extreme eigenvalues of the iterate
 * (max_eigenvalue_constraint_torch / min_eigenvalue_constraint_torch, spd_constraints_utils_torch.py:17-50) or of the iterate lifted
 * to the original space of a nested SPD mapping (max/min_eigenvalue_nested_spd_constraint, nested_spd_constraints_utils.py:14-73: the
 * constraints of HD-GaBO's latent sweep, examples/hd_bo_spd/benchmark_examples/hd_gabo_spd.py:244-257), all inequalities;
 * strict != 0 rejects infeasible proposals (StrictConstrainedTrustRegions).  2 <= d <= 8; affine-invariant and log-Euclidean
 * surrogates.  State arrays as in gabo_spd_tr_update, updated in place; active[i] is 0 for every restart on return.
 * lift_w, lift_p (lift_dim x d), lift_x0 (lift_dim x lift_dim): the mapping of the nested kinds (gabo_nested_spd_lift_prepare;
 * 5 <= lift_dim <= GABO_TR_NESTED_MAX_DIM), NULL / 0 without them. */
#define GABO_TR_NESTED_MAX_DIM 24
#define GABO_CONSTRAINT_MAX_EIGENVALUE 0          /* bound - lambda_max(x) >= 0 */
#define GABO_CONSTRAINT_MIN_EIGENVALUE 1          /* lambda_min(x) - bound >= 0 */
#define GABO_CONSTRAINT_MAX_EIGENVALUE_NESTED 2   /* bound - lambda_max(lift(x)) >= 0 */
#define GABO_CONSTRAINT_MIN_EIGENVALUE_NESTED 3   /* lambda_min(lift(x)) - bound >= 0 */
int gabo_spd_tr_solve(double* x, double* fx, double* grad, double* grad_norm, double* trust_radius, uint8_t* active, int64_t* iters,
                      const gabo_spd_acq_params* acq, int n_constraints, const int* constraint_kind, const double* constraint_bound,
                      int strict, void* workspace, size_t workspace_bytes, int64_t r, int d, double delta_cons, double theta, double kappa,
                      int mininner, int maxinner, double delta_bar, double rho_prime, double rho_regularization, double mingradnorm,
                      int64_t maxiter, const double* lift_w, const double* lift_p, const double* lift_x0, int lift_dim, int* status,
                      gabo_stream_t stream);
/* 1 when gabo_spd_tr_solve runs this problem in its single launch, 0 when the caller has to iterate through gabo_spd_tr_propose /
 * gabo_spd_tr_update instead (same results, two launches per iteration): d outside 2 ... 8, an unknown surrogate metric, more dynamic LDS than a block has - or an instantiation this build of the library leaves out
 * (-DGABO_LE_MAX_GENERIC_DIM, csrc/spd_tr_body.hpp).  lift_dim: as in gabo_spd_tr_solve, 0 without nested kinds. */
int gabo_spd_tr_solve_supported(const gabo_spd_acq_params* acq, int64_t r, int d, int n_constraints, int lift_dim);
/* The two-wave form of the single launch (csrc/spd_tr_duo_body.hpp): for the affine-invariant surrogate with the symmetric inverse of the Gram matrix
 * (acq->linv == acq->linv_t), d <= 6, at most 512 restarts, constraints of kinds 0 / 1 or none, gabo_spd_tr_solve gives every restart a second wave that
 * evaluates the proposal truncated CG is about to make while the first wave is still evaluating the finite-difference point (same results, bit for bit).
 * gabo_spd_tr_two_waves(0 / 1) turns the form off / on for this process and returns the previous setting (-1: only query; initially on unless GABO_TR_DUO=0
 * is in the environment); gabo_spd_tr_two_waves_counters reads (and with reset != 0 clears) how many trust-region iterations of such launches so far found
 * truncated CG leaving with the speculated step (hits) and how many did not (misses: the proposal was evaluated again, the one-wave schedule). */
int gabo_spd_tr_two_waves(int enable);
int gabo_spd_tr_two_waves_counters(long long* hits, long long* misses, int reset);

/* ------------------------------------------------------------------------------------------------------------
 * One multi-start acquisition sweep on S^d_++ through a native host driver: the device work of
 *   gen_batch_initial_conditions_manifold   manifold_optimize.py:232-321  (raw samples drawn on the device and scored, the restarts selected)
 *   gen_candidates_manifold + get_best_candidates   :124-228, :118-120   (initial value / gradient, the whole trust-region solve, argmax)
 * Conditions of gabo_spd_tr_solve (2 <= d <= 8, constraints = bounds on the extreme eigenvalues of the iterate or none); `acq` as for
 * gabo_spd_acq_eval.  The configuration of one sweep: */
#define GABO_SWEEP_MAX_CONSTRAINTS 8
typedef struct {
    gabo_spd_acq_params acq;
    int d;
    double min_eig, max_eig;                       /* the sampler: eigenvalues U[min_eig, max_eig] (spd_utils.py:290-306) */
    int n_constraints;
    int constraint_kind[GABO_SWEEP_MAX_CONSTRAINTS];        /* GABO_CONSTRAINT_MAX_EIGENVALUE / _MIN_EIGENVALUE */
    double constraint_bound[GABO_SWEEP_MAX_CONSTRAINTS];
    int strict;
    double delta_bar, delta0, delta_cons, theta, kappa;     /* robust_trust_regions.py:111-160 / constrained_trust_regions.py:120-190 */
    int mininner, maxinner;
    double rho_prime, rho_regularization, mingradnorm;
    int64_t maxiter;
} gabo_spd_sweep_config;

/* ------------------------------------------------------------------------------------------------------------
 * The sweep with its set-up, start and end INSIDE the launches, laid out for sharding over the GPUs of a node.
 *
 * gabo_spd_gp_prepare: everything the sweep needs from a fitted exact GP with an affine-invariant kernel, from ONE host call - the Gram matrix
 *   K(X, X) of the training set (SpdAffineInvariant{Gaussian,Laplace}Kernel.forward, kernels_spd.py:72-100,157-187; `flags` = GABO_OUT_GAUSSIAN /
 *   GABO_OUT_LAPLACE), the prediction cache gabo_gp_factor computes from it ([3P] gpytorch's prediction strategy behind manifold_optimize.py:182-184)
 *   (kinv != NULL: its symmetric inverse too) and, train_factors != NULL, gabo_spd_acq_prepare_train's factors (d_vec x n).  status: a training matrix that is not SPD; factor_status: a
 *   covariance that is not positive definite (both device int[2], zeroed by the caller).  n <= GABO_GP_FACTOR_MAX_N, d <= GABO_SPD_REG_MAX_DIM.
 *
 * The sweep keeps two tables in its workspace (gabo_spd_sweep_rows_tables returns their device addresses):
 *   raw rows     max_raw x (1 + d_vec):    [acquisition value, raw sample as a Mandel vector]
 *   result rows  restarts x (2 + d_vec):   [final cost = -acquisition, trust-region iterations, final iterate as a Mandel vector]
 * A multi-GPU sweep all_gathers exactly these two tables (raw rows by sample index after scoring, result rows after solving): the caller's
 * collective, on the tables, between / after the calls - the driver has none of its own.
 *   gabo_spd_sweep_score_rows: raw samples first_sample ... first_sample + count - 1 of the stream `seed` (gabo_spd_sample_range; raw_matrices_host !=
 *     NULL: the count x d x d matrices the caller's sampler drew) into rows first_row ... first_row + count - 1 of the table (< max_raw: a rank of a
 *     sharded sweep fills its own block), scored (gen_batch_initial_conditions_manifold, manifold_optimize.py:288-309).
 *     values_mapped (may be NULL): count doubles the KERNEL writes the values to as well - memory the device can address: page-locked host memory
 *     (hipHostMalloc, torch's pinned allocator) or device memory; pageable host memory is refused (GABO_ERR_ARG).
 *   gabo_spd_sweep_solve_rows: two launches for gen_candidates_manifold (manifold_optimize.py:124-228) on the restarts that start from rows
 *     picked_mapped[0 ... restarts - 1] (device-addressable like values_mapped): the start of every restart (the two Mandel maps, cost, gradient,
 *     [3P] egrad2rgrad / norm), then the whole trust-region solve (gabo_spd_tr_solve's kernel), which ends with the result row.  results_mapped (may be NULL): restarts x (2 + d_vec) doubles
 *     of device-addressable memory that receive the result rows as well.  get_best_candidates (:118-120) is an argmax over column 0 of the result rows:
 *     the caller's, after its all_gather if there is one.
 *   skip_flag (solve; may be NULL): the DEVICE int gabo_spd_sweep_select_rows wrote its flag to - when it is non-zero (nothing was picked) the
 *     launches do no work (result rows: cost NaN, 0 iterations) and the caller selects on the host and solves again.
 *   status: device int[2] as everywhere (zeroed by the caller); status_mapped (may be NULL): int[2] of device-addressable host memory, zeroed by
 *     the caller, that receives the same two ints when a launch of the call reports an error - the host reads it after the stream has drained,
 *     without a copy.
 *   synchronize != 0: the call returns after the stream has drained (the mapped copies are then readable by the host).
 * The numbers are those of this package's Python path (the same device statements in the same order), bit for bit. */
size_t gabo_spd_gp_prepare_workspace_bytes(int64_t n, int d);
int gabo_spd_gp_prepare(const double* train_mandel, const double* y, int64_t n, int d, double beta, int flags, double outputscale, double noise,
                        double mean, double* linv, double* linv_t, double* alpha, double* kinv, double* train_factors, void* workspace,
                        size_t workspace_bytes, int* status, int* factor_status, gabo_stream_t stream);
size_t gabo_spd_sweep_rows_workspace_bytes(int64_t n_train, int d, int64_t max_raw, int64_t restarts, int n_constraints);
int gabo_spd_sweep_rows_tables(void* workspace, int64_t n_train, int d, int64_t max_raw, int64_t restarts, int n_constraints, double** raw_rows,
                               double** result_rows);
int gabo_spd_sweep_score_rows(const gabo_spd_sweep_config* cfg, int64_t first_sample, int64_t first_row, int64_t count, int64_t max_raw,
                              int64_t restarts, uint64_t seed, const double* raw_matrices_host, double* values_mapped, void* workspace,
                              size_t workspace_bytes, int* status, int* status_mapped, int synchronize, gabo_stream_t stream);
/* The selection between the two, on the device (one launch): which `restarts` of the `total` scored raw samples become restarts - [3P] botorch's
 * initialize_q_batch_nonneg (the heuristic gen_batch_initial_conditions_manifold applies to non-negative acquisition functions,
 * manifold_optimize.py:296-317): samples with value >= alpha * max (alpha shrunk by tens until `restarts` qualify) drawn without replacement with
 * weights exp(eta (y / max - 1)) by the exponential race torch.multinomial runs (keys w_i / E_i, the largest win; E_i from the library's Philox
 * stream (seed, sample index) instead of the host generator), the arg-max forced into the last slot when the draw missed it.
 *   raw_rows: the raw-row table (gabo_spd_sweep_rows_tables) laid out in `world` blocks of per_rank + 1 rows - a header row, then the block's
 *     samples: sample s sits in row (s / per_rank) * (per_rank + 1) + 1 + s % per_rank (one rank: world = 1, per_rank = total); seed_in_header != 0:
 *     the seed is read from column 0 of row 0 (after an all_gather: rank 0's proposal, identical on every rank).
 *   picked_rows: device memory, ceil(restarts / world) int64 - the table rows of the restarts this rank owns (restart k belongs to rank k % world),
 *     ready for gabo_spd_sweep_solve_rows; picked_samples (may be NULL): `restarts` int64, the sample index of every restart.
 *   flag (device int) / flag_mapped (device-addressable host int, may be NULL): 0 = picked; 1 = the heuristic has to fall back on its random
 *     choices (no positive value, fewer positive values than restarts) or a value is NaN: nothing was picked, the caller selects on the host.
 * gabo_spd_sweep_select_supported: total <= 8192 and restarts < total (otherwise the caller selects on the host as well). */
int gabo_spd_sweep_select_supported(int64_t total, int64_t restarts);
int gabo_spd_sweep_select_rows(const double* raw_rows, int d, int64_t total, int64_t per_rank, int64_t restarts, double eta, double alpha, uint64_t seed,
                               int seed_in_header, int rank, int world, int64_t* picked_rows, int64_t* picked_samples, int* flag, int* flag_mapped,
                               gabo_stream_t stream);
int gabo_spd_sweep_solve_rows(const gabo_spd_sweep_config* cfg, const int64_t* picked_mapped, int64_t restarts, int64_t max_raw,
                              double* results_mapped, const int* skip_flag, void* workspace, size_t workspace_bytes, int* status,
                              int* status_mapped, int synchronize, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Acquisition maximisation on the sphere S^(dim-1) (the sphere twins of gabo_spd_acq_eval / gabo_spd_tr_*): kernel strip of
 * SphereGaussianKernel / SphereLaplaceKernel (kernels_sphere.py:71-94,118-134) + exact-GP posterior + EI / posterior mean + gradient
 * in one launch, and the trust-region iteration of robust_trust_regions.py / constrained_trust_regions.py with the finite-difference
 * Hessian (approximate_hessian.py:11-62) on pymanopt's Sphere geometry, one wave per restart.  State arrays: x, grad r x dim
 * (grad = Riemannian gradient), fx, grad_norm, trust_radius r; active r bytes; iters r int64; cons_grads n_constraints x r x dim
 * (Riemannian gradients, equalities first), cons_values r x n_constraints.  gabo_sphere_tr_solve runs the whole solve in one launch
 * when there are no constraints (constraints on the sphere are user callables).  exact_hessian != 0: the tCG uses the exact
 * Riemannian Hessian-vector product (closed form of the double backward the reference runs through its sphere kernel,
 * pymanopt_addons/tools/autodiff/_pytorch.py:103-116; the stock TrustRegions of examples/gabo_sphere.py:151) instead of
 * get_hessianfd. */
typedef struct {
    const double* train;     /* n x dim training points, row-major */
    const double* train_t;   /* dim x n, the same points transposed (coalesced strip evaluation) */
    const double* alpha;     /* n */
    const double* linv;      /* n x n */
    const double* linv_t;    /* n x n */
    int64_t n;
    int dim;
    double beta;
    int flags;               /* GABO_OUT_GAUSSIAN / GABO_OUT_LAPLACE */
    double mean, outputscale, kxx, best_f;
    int kind, maximize;
    double out_sign;
} gabo_sphere_acq_params;
int gabo_sphere_acq_eval(const double* x, const gabo_sphere_acq_params* acq, double* value, double* grad, int64_t r,
                         gabo_stream_t stream);
size_t gabo_sphere_tr_workspace_bytes(int64_t r, int dim, int n_constraints);
/* byte offset, inside the workspace, of the r ints "stop reason of the last truncated-CG run" (codes as gabo_spd_tcg_end): what a
 * per-iteration record of the solver reads between gabo_sphere_tr_propose and gabo_sphere_tr_update (robust_trust_regions.py:190 `srstr`) */
size_t gabo_sphere_tr_stop_offset(int64_t r, int dim, int n_constraints);
/* Per-iteration record of the single-launch solves - what the reference's solvers keep in their optlog (robust_trust_regions.py:300-340:
 * iterate, radius, `srstr`) and tests/golden/tr_traces.npz holds for them.  The NEXT call of gabo_spd_tr_solve or gabo_sphere_tr_solve of
 * this process (one pending buffer; not thread-safe: a parity / debugging facility) writes, for outer iteration k < max_iterations of restart
 * i, buffer[(k * r + i) * (L + 2) + ...] = the iterate (L = d * d doubles, or dim for the sphere), the trust radius and the stop reason of the
 * truncated-CG run that made the proposal judged in that iteration.  buffer: device memory of max_iterations * r * (L + 2) doubles that the
 * caller pre-fills (NaN: iterations a restart did not run stay NaN).  max_iterations = 0 withdraws a pending buffer. */
int gabo_tr_solve_record(double* buffer, int64_t max_iterations);
int gabo_sphere_tr_propose(const double* x, const double* grad, const double* trust_radius, const uint8_t* active,
                           const double* cons_grads, const double* cons_values, const gabo_sphere_acq_params* acq, void* workspace,
                           size_t workspace_bytes, double* x_prop, int64_t r, int n_constraints, int n_equalities, double delta_cons,
                           double theta, double kappa, int mininner, int maxinner, int exact_hessian, int* any_active, gabo_stream_t stream);
int gabo_sphere_tr_update(double* x, double* fx, double* grad, double* grad_norm, double* trust_radius, uint8_t* active, int64_t* iters,
                          const uint8_t* invalid, void* workspace, int64_t r, int dim, int n_constraints, double delta_bar,
                          double rho_prime, double rho_regularization, double mingradnorm, int64_t maxiter, int* any_active,
                          gabo_stream_t stream);
int gabo_sphere_tr_solve(double* x, double* fx, double* grad, double* grad_norm, double* trust_radius, uint8_t* active, int64_t* iters,
                         const gabo_sphere_acq_params* acq, void* workspace, size_t workspace_bytes, int64_t r, double theta, double kappa,
                         int mininner, int maxinner, int exact_hessian, double delta_bar, double rho_prime, double rho_regularization,
                         double mingradnorm, int64_t maxiter, gabo_stream_t stream);

/* The sphere counterpart, as two host calls around the caller's selection heuristic (score, then solve): one multi-start acquisition sweep of the reference's gabo_sphere examples
 * (examples/bo_sphere/benchmark_examples/gabo_sphere.py:151-175: stock TrustRegions, no constraints; manifold_optimize.py:36-321) as two host
 * calls around the caller's selection heuristic.  The raw samples are the caller's (count x dim points drawn by `manifold.rand` on the host).
 * Launches: gabo_sphere_acq_eval, gabo_sphere_manifold_op (proj), gabo_sphere_tr_solve.  candidates_dev: the final iterates, restarts x dim. */
typedef struct {
    gabo_sphere_acq_params acq;
    double delta_bar, delta0, theta, kappa;
    int mininner, maxinner, exact_hessian;
    double rho_prime, rho_regularization, mingradnorm;
    int64_t maxiter;
} gabo_sphere_sweep_config;
size_t gabo_sphere_sweep_workspace_bytes(int dim, int64_t max_raw, int64_t restarts);
int gabo_sphere_sweep_score(const gabo_sphere_sweep_config* cfg, int64_t count, int64_t max_raw, int64_t restarts, const double* raw_points_host,
                            double* values_host, void* workspace, size_t workspace_bytes, gabo_stream_t stream);
int gabo_sphere_sweep_solve(const gabo_sphere_sweep_config* cfg, const int64_t* picked_host, int64_t restarts, int64_t max_raw,
                            int64_t* best_index_host, double* best_value_host, int64_t* max_iterations_host, double** candidates_dev,
                            double** cost_dev, int64_t** iterations_dev, void* workspace, size_t workspace_bytes, gabo_stream_t stream);
/* The same sweep in ONE call with one host wait: raw samples (count x dim from the caller, or - raw_points_host NULL - drawn on the library's Philox stream
 * `sample_seed`: normal deviates, normalised, as [3P] Sphere.rand) -> values -> restart selection ON THE DEVICE (botorch's initialize_q_batch_nonneg as
 * gabo_spd_sweep_select_rows runs it, weights exp(eta (y / max - 1)), threshold alpha, stream `select_seed`; needs gabo_spd_sweep_select_supported(count,
 * restarts)) -> start and solve of every restart -> arg-max.  Outputs as gabo_sphere_sweep_solve; workspace of gabo_sphere_sweep_workspace_bytes(dim, count,
 * restarts); *picked_dev (may be NULL): the raw-sample index of every restart, restarts int64 in the workspace.  *fallback_host = 1: the selection needs the heuristic's random fall-backs (no positive value, fewer positive values than restarts, a NaN):
 * nothing else is valid and the caller takes gabo_sphere_sweep_score / its own selection / gabo_sphere_sweep_solve. */
int gabo_sphere_sweep_run(const gabo_sphere_sweep_config* cfg, int64_t count, int64_t restarts, const double* raw_points_host, uint64_t sample_seed,
                          double eta, double alpha, uint64_t select_seed, int64_t* best_index_host, double* best_value_host,
                          int64_t* max_iterations_host, double** candidates_dev, double** cost_dev, int64_t** iterations_dev, int64_t** picked_dev,
                                     int* fallback_host,
                          void* workspace, size_t workspace_bytes, gabo_stream_t stream);

/* Batched sphere-manifold operations, x/u/v/w/out: n x dim (GABO_SPH_DIST writes n scalars).
 *   GABO_SPH_PROJ   out = U - <X,U> X        [3P] Sphere.proj = egrad2rgrad; transp(X,Y,U) = proj(Y,U)
 *   GABO_SPH_RETR   out = (X+U)/|X+U|        [3P] Sphere.retr  (robust_trust_regions.py:228)
 *   GABO_SPH_EXP    out = Exp_X(U)           Riemannian_utils/sphere_utils.py:14-38
 *   GABO_SPH_LOG    out = Log_X(U=point)     sphere_utils.py:41-65
 *   GABO_SPH_DIST   out[n] = acos(clip(<X,U>))
 *   GABO_SPH_EHESS2RHESS  out = proj(X, v=ehess) - <X, u=egrad> w=tangent      [3P]
 */
#define GABO_SPH_PROJ 0
#define GABO_SPH_RETR 1
#define GABO_SPH_EXP 2
#define GABO_SPH_LOG 3
#define GABO_SPH_DIST 4
#define GABO_SPH_EHESS2RHESS 5
int gabo_sphere_manifold_op(int op, const double* x, const double* u, const double* v, const double* w, double* out, int64_t n,
                            int dim, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Reconstruction cost of the nested-SPD mapping, with its gradient, in one launch (HD-GaBO: the objective of
 * optimize_reconstruction_parameters_nested_spd, nested_mappings/nested_spd_optimization.py:95-186).
 * Replaces  min_affine_invariant_distance_reconstruction_cost   nested_spd_optimization.py:23-56   (metric 0)
 *           min_log_euclidean_distance_reconstruction_cost      nested_spd_optimization.py:59-92   (metric 1)
 *           projection_from_nested_spd_to_spd                    nested_mappings/nested_spd_utils.py:51-118
 *           and the reference's autograd pass through them (sqrtm_torch / logm_torch: spd_utils_torch.py:13-50; symeig).
 *
 * cost[p] = sum_n dist(X_n, R_p [[Y_n, B_pn], [B_pn^T, C_p]] R_p^T)^2,  R_p = [W, V_p],  B_pn = Y_n^1/2 K_p C_p^1/2, with
 *   metric 1: dist^2 = ||logm X_n - logm Xrec + 1e-15||_F^2 (the 1e-15 on every matrix entry, spd_utils_torch.py:156),
 *   metric 0: dist^2 = sum_k log^2 lambda_k(L_n^-1 Xrec L_n^-T) + 1e-15, L_n = chol(X_n)      (spd_utils_torch.py:120).
 * data: N x D x D, what the metric needs of the FIXED data, written by gabo_nested_spd_reconstruction_prepare from X (N x D x D):
 *   logm X_n (metric 1) or L_n^-1 (metric 0).  y, sqrt_y: N x d x d (Y_n = W^T X_n W and its square root, e.g. GABO_SPD_SQRTM);
 * w: D x d;  v: P x D x (D-d), c: P x (D-d) x (D-d), k: P x d x (D-d): P parameter sets evaluated by one launch (the optimiser's
 * initial candidates; P = 1 inside the line searches).  cost: P.  grad_v / grad_c / grad_k: the Euclidean partial derivatives, same
 * shapes as v / c / k, or all three NULL (values only).  2 <= D <= GABO_SPD_MAX_DIM, 1 <= d < D.  status: as above (prepare, metric 0).
 * c_eigenvalues: P x (D-d), c_eigenvectors: P x (D-d) x (D-d) with the vectors in the COLUMNS (row-major), or both NULL: the
 *   eigen-decomposition of sym(C_p), when the caller has it (the kernel then skips its own - a lone wave's ~50 us at D = 20).
 */
size_t gabo_nested_spd_reconstruction_workspace_bytes(int64_t P, int64_t N, int D, int d);
int gabo_nested_spd_reconstruction_prepare(const double* x, double* data, int64_t N, int D, int metric, int* status, gabo_stream_t stream);
int gabo_nested_spd_reconstruction(const double* data, const double* y, const double* sqrt_y, const double* w, const double* v,
                                   const double* c, const double* k, double* cost, double* grad_v, double* grad_c, double* grad_k,
                                   const double* c_eigenvalues, const double* c_eigenvectors, int64_t P, int64_t N, int D, int d, int metric,
                                   void* workspace, size_t workspace_bytes, gabo_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * The optimisation of the reconstruction parameters itself, as a native host loop around the launch above.
 * Replaces  optimize_reconstruction_parameters_nested_spd            nested_mappings/nested_spd_optimization.py:168-186 (the run from the
 *             chosen start point; the random candidates of :158-166 stay with the caller: one launch with P = nb_init_candidates)
 *           AugmentedLagrangeMethod.solve with one equality constraint  manifold_optimization/augmented_Lagrange_method.py:72-326
 *           [3P] pymanopt ConjugateGradient + LineSearchAdaptive         (the inner_solver of examples/hd_bo_spd/.../hd_gabo_spd.py)
 * on  V in G(D, D-d) x C in S^(D-d)_++ x unit vector of R^(d (D-d)) x raw in R,  K = sigmoid(raw) * unit reshaped d x (D-d)  (:139, 155),
 * under ||V^T W||_F = 0 (:142-147).  HOST function: it returns when the optimisation has ended (every evaluation waits for its launch).
 * v, c, unit, raw: HOST arrays, in: the start point, out: the optimum.  w_host: D x d host copy of w.
 * gabo_nested_spd_reconstruction_solve: data / y / sqrt_y / w as for gabo_nested_spd_reconstruction (device); workspace: device memory,
 *   pinned: page-locked host memory (hipHostMalloc / torch pin_memory), sizes from ..._solve_workspace_bytes (pinned counted in doubles).
 * gabo_nested_spd_reconstruction_solve_with: the same loop around ANY evaluator (a user-supplied cost_function, :98) - called with
 *   P <= GABO_RECON_MAX_LOOKAHEAD parameter sets, v: P x D x (D-d), c: P x (D-d)^2, k: P x d x (D-d), it fills cost[P] and the Euclidean partial derivatives
 *   grad_v / grad_c / grad_k (same shapes) and returns GABO_OK; staging: >= GABO_RECON_MAX_LOOKAHEAD (2 npar + 1 + (D-d) + (D-d)^2) doubles of host memory,
 *   npar = D (D-d) + (D-d)^2 + d (D-d).
 * Returns GABO_OK, an argument error, GABO_ERR_NOT_SPD when an iterate left the cone (never in exact arithmetic), or the evaluator's code.
 */
typedef struct {
    double bound, rho_init, thetarho, tau, starting_tolgradnorm, ending_tolgradnorm, gammas_fact, minstepsize, maxtime; /* augmented_Lagrange_method.py:36-70 */
    int64_t maxiter;
    double cg_minstepsize, cg_maxtime, cg_orth_value;   /* [3P] pymanopt ConjugateGradient(minstepsize, maxtime, orth_value) */
    int64_t cg_maxiter;
    int64_t lookahead;   /* step lengths alpha, alpha/2, ... evaluated by the first launch of a line search: 1..GABO_RECON_MAX_LOOKAHEAD,
                            0 = default (4 up to D - d = 16, else 2).
                            It changes which launch computes a value, never the values or the steps taken. */
    int64_t host_threads; /* threads that retract / factor the candidates of a line search side by side: 1..GABO_RECON_MAX_LOOKAHEAD,
                             0 = default (one per candidate from D - d = 8 when the process may run on >= 16 cores, else 1).  Same results either way. */
} gabo_recon_solve_options;
#define GABO_RECON_MAX_LOOKAHEAD 4
#define GABO_RECON_STOP_MAXITER 0
#define GABO_RECON_STOP_MAXTIME 1
#define GABO_RECON_STOP_MINSTEP 2
#define GABO_RECON_STOP_MINGRAD 3
typedef struct {
    int64_t outer_iterations, inner_iterations, evaluations, launches;
    int stop_reason;                                     /* GABO_RECON_STOP_* */
    double violation, rho, gamma, final_cost, seconds;
    double seconds_evaluator;                            /* of `seconds`, inside the evaluator (enqueue + wait); the rest is host arithmetic */
    int64_t host_threads;                                /* threads used (see the options) */
} gabo_recon_solve_log;
typedef int (*gabo_recon_eval_fn)(void* ctx, int64_t P, const double* v, const double* c, const double* k, double* cost, double* grad_v,
                                  double* grad_c, double* grad_k);
void gabo_nested_spd_reconstruction_solve_workspace_bytes(int64_t N, int D, int d, size_t* device_bytes, size_t* pinned_doubles);
int gabo_nested_spd_reconstruction_solve(const double* data, const double* y, const double* sqrt_y, const double* w, const double* w_host,
                                         double* v, double* c, double* unit, double* raw, int64_t N, int D, int d, int metric,
                                         void* workspace, size_t workspace_bytes, double* pinned, size_t pinned_doubles,
                                         const gabo_recon_solve_options* options, gabo_recon_solve_log* log, gabo_stream_t stream);
int gabo_nested_spd_reconstruction_solve_with(gabo_recon_eval_fn evaluate, void* ctx, const double* w_host, double* v, double* c,
                                              double* unit, double* raw, int D, int d, double* staging, size_t staging_doubles,
                                              const gabo_recon_solve_options* options, gabo_recon_solve_log* log);

/* ------------------------------------------------------------------------------------------------------------
 * Eigenvalue bounds of a latent (nested) SPD point stated in the original space (HD-GaBO's acquisition constraints).
 * Replaces  max_eigenvalue_nested_spd_constraint / min_eigenvalue_nested_spd_constraint   nested_mappings/nested_spd_constraints_utils.py:14-73
 *           (projection_from_nested_spd_to_spd, nested_spd_utils.py:51-118, followed by symeig(eigenvectors=True) and autograd).
 * gabo_nested_spd_lift_prepare, once per mapping (W, V, C, K):  x0 = V sym(C) V^T (D x D),  p = V (K C^1/2)^T (D x d), so that
 *   Xrec(Y) = x0 + W Y W^T + (W Y^1/2) p^T + p (W Y^1/2)^T.    v: D x (D-d), c: (D-d) x (D-d), k: d x (D-d).
 * gabo_nested_spd_extreme_eigenvalues: y: R x d x d latent points; lam: R x 2 = (lambda_max, lambda_min) of Xrec(Y_r);
 *   grad: R x 2 x d x d = their gradients with respect to the (symmetric) Y_r, or NULL.  One wave per point, both eigenpairs from one
 *   Householder reduction.  2 <= D <= GABO_SPD_MAX_DIM, 1 <= d < D.
 */
int gabo_nested_spd_lift_prepare(const double* v, const double* c, const double* k, double* x0, double* p, int D, int d, gabo_stream_t stream);
int gabo_nested_spd_extreme_eigenvalues(const double* y, const double* w, const double* p, const double* x0, double* lam, double* grad,
                                        int64_t R, int D, int d, gabo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GABO_HIP_H */
