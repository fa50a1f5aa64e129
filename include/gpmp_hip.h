/* gpmp_hip.h -- C ABI of libgpmp_hip.so, the MI355X (gfx950) exact-GP inner loop.
 *
 * Drop-in boundary for GPmp's numerical backend contract (gpmp/num/__init__.py:25-46): every entry
 * point below replaces the arithmetic one `gnp.*` call (or one fused run of them) performs on the
 * hot path.  The host side (Python: gpmp_amd/, ctypes) keeps GPmp's names and semantics.
 *
 * Conventions
 *  - All matrices are fp64, ROW-MAJOR, in device memory (HBM), described by (pointer, rows, cols, ld)
 *    with ld = leading dimension in elements (ld >= cols).  Fast paths need 16-byte aligned base
 *    pointers and even ld; anything else is still correct (checked/scalar loads).
 *  - Small parameter vectors (theta, mean parameters) are HOST pointers: they come from SciPy on the
 *    host (gpmp/kernel/parameter_selection.py:253-260) and are passed by value to the kernels.
 *  - Every function returns 0 on success, <0 for a bad argument (-k = k-th argument), or a LAPACK
 *    style info > 0 where stated.  No entry point synchronises: every function only enqueues work
 *    on `stream` (a hipStream_t passed as void*; NULL = the default stream); scalar results
 *    (info, log-det, gradient) land in caller-provided DEVICE words.
 *  - Callers (the torch allocator) own every matrix, vector and workspace; workspace sizes come from the
 *    gpmp_*_ws_* / gpmp_dinv_elems queries.  The library itself allocates only the 16.4 KB flag block of the one-launch
 *    triangular solve, once per (device, stream) that uses it (see gpmp_solve_status), plus host-side helper streams /
 *    events.  Lifetime of that block: from the stream's first single-vector solve until gpmp_stream_release(stream)
 *    (or gpmp_stream_destroy for the library's own streams) -- a caller that creates and destroys streams calls
 *    gpmp_stream_release before destroying one; otherwise the block stays allocated until the process ends (16.4 KB per
 *    stream ever used) and a later stream with the same handle value reuses it (harmless: it is zeroed before every solve).
 *  - DEVICES: every call works on the CURRENT device (hipSetDevice) and its pointers / stream must belong to it.  What the
 *    library keeps on the host -- the helper streams and events of the look-ahead factorisation, the flag blocks above,
 *    one-time kernel attributes -- is kept PER DEVICE ORDINAL (round 4; earlier versions refused a second device), so
 *    both launch models work with one copy of the library: one process per GPU (what bench.py and gpmp_amd/dist do), or
 *    one process with one host thread per GPU, each thread with its device current.  Threads on different devices do not
 *    wait for each other; two threads on the same device are serialised while they enqueue a factorisation.  One calling
 *    thread at a time per stream.  gpmp_device_release() returns what the library holds for the current device (safe against
 *    an enqueue section still running on another thread: the state is held by shared ownership).  Tested: one process per
 *    GPU; SEVERAL host threads on ONE device (eight thread-ranks of the distributed tests share a GPU); the per-device table
 *    under the host sanitizers (gpmp_debug_device_table_selftest).  NOT tested: more than one device ordinal driven from
 *    one process -- the GPU pool this was developed on has one-GPU boxes.
 *  - covparam layout (gpmp/kernel/matern.py:78-79,88-89): theta = [log sigma^2, log(1/rho_1..d)];
 *    with `noise` != 0 the layout is [log sigma^2, log sigma_noise^2, log(1/rho_1..d)]
 *    (examples/gpmp_example07_nd_regression.py:95-131).
 */
#ifndef GPMP_HIP_H
#define GPMP_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gpmp_stream_t;

#define GPMP_NB 128          /* diagonal block size of the blocked factorisation */
#define GPMP_MAX_DIM 64      /* largest input dimension d handled by the Gram kernels */
#define GPMP_MAX_P 16        /* largest Matern half-integer index p (nu = p + 1/2) */
#define GPMP_MAX_RANK 72     /* largest low-rank correction width in gpmp_matern_grad_trace */
#define GPMP_MAX_EXTENT (1 << 30) /* largest row / column / contraction count any entry point accepts: element OFFSETS are 64-bit
                                    (n = 131072 squared is in the tests), extents and launch arithmetic are 32-bit */
#define GPMP_BATCH_MAX_N 4096 /* largest (padded) problem size of the batched small-problem driver */
#define GPMP_BATCH_MAX_Q 16   /* largest number of mean-design columns of the batched small-problem driver */

int gpmp_hip_abi_version(void);
/* Last error text of the calling thread (HIP error string or argument message). */
const char* gpmp_last_error(void);

/* Opt-in kernel timing (HIP events recorded on the launch stream around every kernel launch of the
 * library).  gpmp_profile_end synchronises the recorded events and fills table_host[12][3] =
 * {launch count, total milliseconds, work} per kind: 0..3 GEMM (register-staged kernel) NT / NN / TN / TT
 * (work = executed flops), 4 diagonal-block kernel (work = blocks), 5 Gram kernel (work = bytes written),
 * 6 column reductions (work = bytes read), 7 gradient trace, 8..11 GEMM (LDS-direct kernel
 * gemm_f64_kernel_v2) NT / NN / TN / TT.  Used by bench.py for `roofline`. */
int gpmp_profile_begin(void);
/* Same, for the kinds whose bit is set only (bit k = kind k): a pair of events per launch costs about 1 us on the
 * launch stream, which adds up over the thousands of small launches of a factorisation (0.8 % of the benchmark step
 * with every kind recorded). */
int gpmp_profile_begin_kinds(unsigned kinds);
int gpmp_profile_end(double* table_host);

/* A stream whose kernels run on every CU except `reserve_cus` of them (taken one per XCD first).  The distributed
 * Cholesky (gpmp_amd/dist) runs its bulk trailing updates on such a stream so that the latency-bound panel chain on its
 * high-priority stream always finds a free CU instead of queueing behind 218-us MFMA workgroups.  The caller owns the
 * stream (gpmp_stream_destroy).  No counterpart in the reference (it has no device code). */
int gpmp_stream_create_reserving_cus(int reserve_cus, gpmp_stream_t* stream_out);
int gpmp_stream_destroy(gpmp_stream_t stream);
/* Hint for callers that enqueue latency-bound work (panel solves, small factorisations) on one stream while THEIR OWN
 * machine-filling GEMM runs on another (gpmp_amd/dist does): while the hint is on, NT products with K <= 512 take the
 * small-footprint kernel that starts beside that GEMM's resident workgroups (see gpmp_dgemm).  Returns the previous value. */
int gpmp_hint_machine_busy(int on);

/* ---- Matern kernels ------------------------------------------------------------------------ */

/* K[i,j] = sigma2 * Matern_p( || invrho * (x_i - y_j) || )  (+ diag_add on i == j when y == NULL).
 * Replaces gnp.scaled_distance + maternp_kernel + "+ nugget*eye" (gpmp/num/numpy_backend.py:432-436,
 * gpmp/kernel/matern.py:32-64,88-94,114-121) in one pass; the distance matrix is never stored.
 * y == NULL selects the ii/tt path (matern.py:139-141): nugget 10*sigma2*eps (or the noise variance)
 * is passed by the caller as diag_add.  lower_only != 0 (ii path only) writes tiles on/below the
 * diagonal only (enough for gpmp_potrf_lower_async). */
int gpmp_matern_gram(const double* x, const double* y, int n, int m, int d, int p,
                     const double* theta_host, int noise, double diag_add, int lower_only,
                     double* K, long ldk, gpmp_stream_t stream);

/* out[i] = sigma2 * Matern_p(|| invrho * (x_i - y_i) ||)  -- the pairwise=True path
 * (matern.py:114-121; gnp.scaled_distance_elementwise numpy_backend.py:438-446). */
int gpmp_matern_pairwise(const double* x, const double* y, int n, int d, int p,
                         const double* theta_host, int noise, double* out, gpmp_stream_t stream);

/* D[i,j] = || invrho * (x_i - y_j) ||  -- gnp.scaled_distance (numpy_backend.py:432-436). */
int gpmp_scaled_distance(const double* x, const double* y, int n, int m, int d,
                         const double* loginvrho_host, double* D, long ldd, gpmp_stream_t stream);

/* Matern_p(h) elementwise on a device vector -- maternp_kernel (matern.py:32-64). */
int gpmp_maternp_kernel(const double* h, long count, int p, double* out, gpmp_stream_t stream);

/* out (n x n) = dK/dtheta_jparam for the covariance of gpmp_matern_gram (ii path): dense derivative
 * matrices for the Fisher information I_ij = 1/2 tr(K^-1 dK_i K^-1 dK_j) (gpmp/core/fisher.py:18-78, which
 * differentiates the covariance by 5-point finite differences).  Diagnostic-sized n (<= 65535). */
int gpmp_matern_gram_deriv(const double* x, int n, int d, int p, const double* theta_host, int noise,
                           int jparam, double* out, long ld, gpmp_stream_t stream);

/* ---- Cholesky and triangular solves ---------------------------------------------------------- */

/* Number of doubles of the factorisation workspace `dinv` for an n x n matrix: the inverses of the
 * ceil(n/128) diagonal blocks, [block][128][128], followed (n > 1024) by one 1024 x 1024 scratch area that the
 * many-right-hand-side forward solve uses when the buffer is also passed as `scratch` (see gpmp_trsm_lower). */
size_t gpmp_dinv_elems(int n);

/* In-place lower Cholesky A = L L^T (replaces numpy.linalg.cholesky, numpy_backend.py:466).
 * Only the lower triangle of A is read; the strict upper triangle is left unspecified (use
 * gpmp_tril to zero it).  dinv receives inv(L_kk) for every GPMP_NB diagonal block (used by the
 * solves below).  *info_dev (a DEVICE int, written asynchronously): 0, or k > 0 if the leading
 * minor of order k is not positive definite (reference: numpy.linalg.LinAlgError, see
 * gpmp/num/numpy_backend.py:30-46,158-162); the factor is then unspecified.  Enqueue only. */
int gpmp_potrf_lower_async(double* A, int n, long lda, double* dinv, int* info_dev, gpmp_stream_t stream);

/* gpmp_potrf_lower_async followed by B <- L^-1 B (n x m, as gpmp_trsm_lower with trans = 0), with the solve of the leading
 * half of the rows enqueued on an internal stream as soon as those columns of L are final, so that it overlaps the
 * chain-bound trailing half of the factorisation (the path of one prediction: cholesky_solve on K(xi,xi) and K(xi,xt),
 * gpmp/core/kriging.py:59-62).  dinv: gpmp_dinv_elems(n) doubles (block inverses + scratch of the solve leaves).
 * Everything is joined into `stream` before return; *info_dev as gpmp_potrf_lower_async (B is then unspecified). */
int gpmp_potrf_trsm_lower_async(double* A, int n, long lda, double* dinv, int* info_dev, double* B, int m, long ldb,
                                gpmp_stream_t stream);

/* B <- op(L)^-1 B for an n x m row-major B; trans = 0: L, 1: L^T.  Replaces
 * scipy.linalg.solve_triangular (numpy_backend.py:467-468, gpmp/core/linalg.py:41).  If
 * dinv == NULL the diagonal-block inverses are recomputed into `scratch` (gpmp_dinv_elems(n)).
 * `scratch` (gpmp_dinv_elems(n) doubles, may be the same buffer as dinv, may be NULL when dinv is given): for
 * n > 1024 its tail behind the block inverses is overwritten by the fused solve leaves (trans = 0, m >= 512);
 * without it the solve runs launch-per-block leaves (same results, about 5 % slower at n = 32768). */
int gpmp_trsm_lower(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb,
                    int trans, double* scratch, gpmp_stream_t stream);

/* Solves with m <= 4 right-hand sides run as ONE launch whose workgroups hand the solved blocks over through flags in
 * device memory; every wait in it is bounded (no launch can hang the GPU).  If a wait ever ran out -- it does not in a
 * healthy run -- the rows from that block on are filled with NaN in EVERY right-hand side, and the event is counted per
 * stream.  gpmp_solve_status synchronises `stream`, stores that count in *status_host (may be NULL), resets it and
 * returns it (0 = every solve on this stream completed; < 0: HIP error).  The flag block (16.4 KB) is the one piece of
 * device memory the library allocates itself: one per (device, stream) that has run such a solve, on first use. */
int gpmp_solve_status(gpmp_stream_t stream, int* status_host);
/* Releases what the library holds for `stream` (today: that flag block): waits for the work queued on the stream -- the
 * one entry point besides gpmp_solve_status / gpmp_profile_end that synchronises -- frees the block and forgets the
 * stream.  0 when the stream holds nothing (no HIP call is made then).  Call it before destroying a stream that ran
 * single-vector solves; a later solve on the same handle simply allocates a fresh block. */
int gpmp_stream_release(gpmp_stream_t stream);
/* Per-device host state (helper streams / events of the look-ahead factorisation; created at a device's first large
 * factorisation).  gpmp_device_release: synchronises and destroys what the CURRENT device holds (0 if nothing; the next
 * factorisation on it creates fresh ones).  gpmp_device_state_count: devices that hold such state (host only, no HIP call).
 * gpmp_debug_device_table_selftest: host-only consistency check of that table under `threads` concurrent host threads
 * (made-up ordinals, no HIP call; 0 = consistent) -- what tests/test_asan_cpu.py runs under ASan + UBSan.
 * No counterpart in the reference (single-threaded Python, no device code). */
int gpmp_device_release(void);
int gpmp_device_state_count(void);
int gpmp_debug_device_table_selftest(int threads, int ordinals, int iters);

/* B <- B L^-T for an M x k row-major B and a k x k lower-triangular L (right-side solve: the panel step
 * A21 <- A21 inv(L11)^T of a blocked / distributed Cholesky).  dinv as produced by gpmp_potrf_lower_async
 * or gpmp_trtri_diag_blocks for L. */
int gpmp_trsm_right_lower(const double* L, int k, long ldl, const double* dinv, double* B, int M, long ldb,
                          gpmp_stream_t stream);

/* inv(L_kk) for every diagonal block of a given lower-triangular L (no factorisation). */
int gpmp_trtri_diag_blocks(const double* L, int n, long ldl, double* dinv, gpmp_stream_t stream);

/* T <- L^-1 (lower triangular, n x n, ldt); strict upper triangle of T is zeroed.
 * (diag_Kinv_from_chol forms this with solve_triangular(C, eye(n)), gpmp/core/linalg.py:39-41.) */
int gpmp_trtri_lower(const double* L, int n, long ldl, const double* dinv, double* T, long ldt,
                     gpmp_stream_t stream);

/* Kinv(lower) <- T^T T for lower-triangular T = L^-1 (n x n).  Only tiles on/below the diagonal of
 * Kinv are written. */
int gpmp_lauum_lower(const double* T, int n, long ldt, double* Kinv, long ldk, gpmp_stream_t stream);

/* Zero the strict upper triangle of an n x n matrix. */
int gpmp_tril(double* A, int n, long lda, gpmp_stream_t stream);
/* Mirror the lower triangle into the upper one. */
int gpmp_symmetrize_from_lower(double* A, int n, long lda, gpmp_stream_t stream);

/* C = alpha * op(A) op(B) + beta * C on the fp64 MFMA GEMM used by all the blocked routines
 * (exported for tests).  ta/tb: 0 = as stored, 1 = transposed; all row-major.
 * lower_only: bit 0 = compute tiles on / below the diagonal only; bit 1 = for NT products with K <= 512 (ta = 0, tb = 1),
 * take the small-footprint kernel (20 KB of LDS, 48 VGPRs) that the look-ahead Cholesky uses for its panel products
 * while a trailing update fills the machine (it starts beside that GEMM's resident workgroups); bit 2 = op(B) is upper
 * triangular (op(B)(l, j) = 0 for l > j, e.g. tb = 1 with B a lower-triangular T: C = A T^T): the k loop of a tile column
 * stops at its last non-zero row. */
int gpmp_dgemm(int ta, int tb, int M, int N, int K, double alpha, const double* A, long lda,
               const double* B, long ldb, double beta, double* C, long ldc, int lower_only,
               gpmp_stream_t stream);

/* ---- reductions ------------------------------------------------------------------------------ */

/* out[k*ldo + j] = sum_i V[i,j] * Y[i,k]  (k < r),   out[r*ldo + j] = sum_i V[i,j]^2.
 * V is n x m (ldv), Y is n x r (ldy, r <= GPMP_MAX_RANK).  Replaces the einsum("i..., i...") column
 * dots of gpmp/core/kriging.py:194 and gpmp/core/model.py:298-300, and the column sum-of-squares
 * of gpmp/core/linalg.py:44.  ws: m * gpmp_coldots_ws_rows(n) doubles. */
int gpmp_coldots(const double* V, int n, int m, long ldv, const double* Y, int r, long ldy,
                 double* out, long ldo, double* ws, gpmp_stream_t stream);
int gpmp_coldots_ws_rows(int n);

/* out[j] = sum_i A[i,j] * B[i,j]  (A, B n x m row-major, lda / ldb): the matrix x matrix form of einsum("i..., i...") --
 * the reference's posterior-variance reduction over lambda_t and Kit (gpmp/core/kriging.py:194), taken when the
 * kriging weights are requested (gpmp/core/model.py:305-306).  ws: m * gpmp_coldots_ws_rows(n) doubles (more than needed).
 * Enqueue only. */
int gpmp_coldots_pair(const double* A, long lda, const double* B, long ldb, int n, int m, double* out, double* ws,
                      gpmp_stream_t stream);

/* ONE sweep of the one-sided Jacobi SVD of a square matrix (what stands behind gnp.svd: gpmp/num/torch_backend.py:833-834,
 * scipy.linalg.svd in the NumPy backend; its caller on the path is the "svd" route of gpmp/core/sample_paths.py:54-58, the
 * symmetric square root of a positive SEMI-definite covariance).  G (n x n row-major, ldg) starts as A and W (n x n, ldw) as
 * the identity; a sweep rotates every pair of ROWS of G once (n - 1 launches of n / 2 independent pairs) and applies the same
 * rotations to W.  When the rows of G are mutually orthogonal, A = W^T diag(|g_i|) (g_i / |g_i|).  *conv_dev (device double) =
 * the largest |g_p . g_q| / (|g_p| |g_q|) met in this sweep; rows with |g| <= tiny_norm count as zero and are left alone.  The
 * caller repeats sweeps until *conv_dev is at rounding level.  Enqueue only. */
int gpmp_jacobi_sweep(double* G, long ldg, double* W, long ldw, int n, double tiny_norm, double* conv_dev, gpmp_stream_t stream);

/* *out_dev (device double) = 2 * sum_i log(L[i,i])  (gpmp/core/likelihood.py:50).  Enqueue only. */
int gpmp_logdet_chol(const double* L, int n, long ldl, double* out_dev, gpmp_stream_t stream);

/* ---- analytic gradient of the Matern covariance ---------------------------------------------- */

/* g_dev[j] (device vector) = sum_{i,k} M[i,k] * dK[i,k]/dtheta_j, j < 1 + noise + d, with
 *   M[i,k] = Kinv[i,k] - sum_{a<r} F[i,a] * G[k,a]     (Kinv: lower triangle used, symmetric)
 * and K the covariance of gpmp_matern_gram (ii path, diag_add = nugget or noise).  The reference has
 * no analytic form (torch autograd, gpmp/num/torch_backend.py:574-604); formulas in DESIGN.md.
 * ws: gpmp_grad_ws_elems(n, d) doubles.  Enqueue only. */
int gpmp_matern_grad_trace(const double* Kinv, long ldk, const double* x, int n, int d, int p,
                           const double* theta_host, int noise, const double* F, const double* G,
                           int r, long ldf, double* g_dev, double* ws, gpmp_stream_t stream);
size_t gpmp_grad_ws_elems(int n, int d);

/* ---- fused drivers (zero-mean GP, Matern covariance) ------------------------------------------- */

/* *nll_dev = 1/2 (n ln 2 pi + ln|K| + z^T K^-1 z) with K = gpmp_matern_gram(x) + nugget (or noise variance):
 * the whole of negative_log_likelihood_zero_mean (gpmp/core/likelihood.py:18-52) -- Gram build, Cholesky,
 * one triangular solve, two reductions -- enqueued by one call.  x: n x d row-major, z: n, both on the device.
 * *info_dev as gpmp_potrf_lower_async; when it is non-zero *nll_dev = +inf (the reference returns safe_inf(),
 * likelihood.py:47-48).  ws: gpmp_nll_ws_elems(n) doubles (holds K / L on return).  Enqueue only. */
size_t gpmp_nll_ws_elems(int n);
int gpmp_nll_zero_mean(const double* x, const double* z, int n, int d, int p, const double* theta_host,
                       int noise, double* ws, double* nll_dev, int* info_dev, gpmp_stream_t stream);

/* Posterior mean and variance at m points: kriging_predictor_with_zero_mean + _compute_posterior_variance
 * (gpmp/core/kriging.py:35-67,170-199) restated as ONE solve  V = L^-1 K(xi, xt) (gpmp_potrf_trsm_lower_async):
 *   zpm = V^T (L^-1 zi),   zpv = sigma^2 - colsumsq(V)   (clamped at 0 when zero_neg_variances != 0, as
 * Model.predict does, gpmp/core/model.py:290-296).  xi: n x d, zi: n, xt: m x d, zpm_dev / zpv_dev: m, all on the
 * device.  A failed factorisation (*info_dev != 0) fills both outputs with NaN.
 * ws: gpmp_predict_ws_elems(n, m) doubles.  Enqueue only. */
size_t gpmp_predict_ws_elems(int n, int m);
int gpmp_predict_zero_mean(const double* xi, const double* zi, const double* xt, int n, int m, int d, int p,
                           const double* theta_host, int noise, int zero_neg_variances, double* ws,
                           double* zpm_dev, double* zpv_dev, int* info_dev, gpmp_stream_t stream);

/* ---- fused drivers with a linear-predictor mean (REML, gradients, leave-one-out) ------------------------------- */

/* The same traces over a RECTANGULAR block: M is n x m (ldm), its rows belong to the points x (n x d), its columns to the
 * points y (m x d); every entry counts once.  g_dev[0] = sum M[i,k] sigma^2 Kc(x_i, y_k), g_dev[1 + j] = sum M[i,k]
 * dK(x_i, y_k)/dlog(1/rho_j), j < d, with M[i,k] = Minv[i,k] - sum_a F[i,a] G[k,a] (F: n x r, G: m x r, same ldf) -- no nugget /
 * noise term: those need tr(M), which only the caller of a blocked trace knows.  This is the building block of the gradient
 * on a DISTRIBUTED inverse (gpmp_amd/dist: the blocks (shard c, shard c') of K^-1 = T^T T meet the matching blocks of dK).
 * ws: gpmp_grad_ws_elems(n, d) doubles.  Enqueue only. */
int gpmp_matern_grad_trace_cross(const double* M, long ldm, const double* x, int n, const double* y, int m, int d, int p,
                                 const double* theta_host, int noise, const double* F, const double* G, int r, long ldf,
                                 double* g_dev, double* ws, gpmp_stream_t stream);

/* The three drivers below take the mean DESIGN matrix P = mean(xi, meanparam) (n x q row-major, leading dimension ldp, on
 * the device; q = 0 / P = NULL: zero-mean model) -- mean functions are user callables in the reference
 * (gpmp/core/model.py:30-52).  0 <= q <= GPMP_MAX_RANK - 1, q < n.  Each call only enqueues: Gram build, Cholesky, the
 * solves, and the q x q algebra (Cholesky of S = P^T K^-1 P and of P^T P, S^-1) in one small workgroup on the device.
 * *info_dev: 0; k in [1, n]: K not positive definite at leading minor k (as gpmp_potrf_lower_async); n + k: S or P^T P
 * not positive definite at pivot k (rank-deficient mean design).  ws: the matching gpmp_*_ws_elems doubles.
 *
 * gpmp_reml: *value_dev = 1/2 ((n - q) ln 2 pi + ln|W^T K W| + (W^T z)^T (W^T K W)^-1 (W^T z)), W an orthonormal basis
 * of Null(P^T) -- negative_log_restricted_likelihood (gpmp/core/likelihood.py:92-129; the n x n complete QR and the two
 * n^3 products of gpmp/core/linalg.py:49-88 are replaced by ln|K| + ln|S| - ln|P^T P| and z^T K^-1 z - b^T S^-1 b,
 * b = P^T K^-1 z).  q = 0: the zero-mean NLL (likelihood.py:18-52).  +inf when *info_dev != 0 (likelihood.py:123-124). */
size_t gpmp_reml_ws_elems(int n, int q);
int gpmp_reml(const double* x, const double* z, const double* P, long ldp, int n, int d, int q, int p,
              const double* theta_host, int noise, double* ws, double* value_dev, int* info_dev, gpmp_stream_t stream);

/* Value as gpmp_reml plus its gradient with respect to the covariance parameters, grad_dev[1 + noise + d] (device):
 *   d/dtheta_j = 1/2 tr((Qinv - beta beta^T) dK/dtheta_j),  Qinv = K^-1 - U S^-1 U^T, U = K^-1 P, beta = Qinv z
 * (q = 0: Qinv = K^-1, beta = K^-1 z: the ML gradient).  The reference has no analytic form: NumPy backend
 * gradient = None -> SciPy finite differences (gpmp/num/numpy_backend.py:333), torch backend autograd
 * (gpmp/num/torch_backend.py:574-604); this is what gpmp/kernel/parameter_selection.py:35-124 wraps for the optimiser.
 * K^-1 = T^T T with T = L^-1 (gpmp_trtri_lower, gpmp_lauum_lower), the trace in one fused pass
 * (gpmp_matern_grad_trace).  A failed factorisation gives value +inf and a zero gradient. */
size_t gpmp_nll_grad_ws_elems(int n, int d, int q);
int gpmp_nll_grad(const double* x, const double* z, const double* P, long ldp, int n, int d, int q, int p,
                  const double* theta_host, int noise, double* ws, double* value_dev, double* grad_dev, int* info_dev,
                  gpmp_stream_t stream);

/* Leave-one-out by virtual cross-validation: eloo_i = (Qinv z)_i / Qinv_ii, sigma2loo_i = 1 / Qinv_ii,
 * zloo_i = z_i - eloo_i (three device vectors of length n) -- _loo_with_zero_mean (gpmp/core/loo.py:65-83, q = 0) and
 * _loo_with_linear_predictor_mean_cpd (loo.py:103-130, q > 0); diag(K^-1) = column sums of squares of L^-1
 * (gpmp/core/linalg.py:17-46).  A failed factorisation fills the outputs with NaN. */
size_t gpmp_loo_ws_elems(int n, int q);
int gpmp_loo(const double* x, const double* z, const double* P, long ldp, int n, int d, int q, int p,
             const double* theta_host, int noise, double* ws, double* zloo_dev, double* sigma2loo_dev, double* eloo_dev,
             int* info_dev, gpmp_stream_t stream);

/* Posterior mean and variance at m points with a LINEAR mean whose parameters are unknown (universal kriging):
 * kriging_predictor (gpmp/core/kriging.py:69-167) + _compute_posterior_variance (kriging.py:170-199) + the clamp of
 * Model.predict (gpmp/core/model.py:290-296), restated over the Schur complement S = P^T K^-1 P so
 * that no (n + q) x (n + q) system is formed: one solve V = L^-1 K(xi, xt), W = L^-1 [zi, Pi], then per point
 *   r = V^T Wp - pt,   zpm = V^T w - (S^-1 Wp^T w) . r,   zpv = sigma^2 - (colsumsq(V) - r^T S^-1 r).
 * Pi: n x q (ldpi) and Pt: m x q (ldpt) are the mean design at the observation / prediction points, 1 <= q < GPMP_MAX_RANK
 * (q = 0: gpmp_predict_zero_mean).  *info_dev as gpmp_reml (k in [1, n]: K not positive definite; n + k: the mean design is
 * rank deficient to working precision); when it is non-zero both outputs are NaN.
 * ws: gpmp_predict_mean_ws_elems(n, m, q) doubles.  Enqueue only. */
size_t gpmp_predict_mean_ws_elems(int n, int m, int q);
int gpmp_predict_mean(const double* xi, const double* zi, const double* Pi, long ldpi, const double* xt, const double* Pt,
                      long ldpt, int n, int m, int d, int q, int p, const double* theta_host, int noise,
                      int zero_neg_variances, double* ws, double* zpm_dev, double* zpv_dev, int* info_dev,
                      gpmp_stream_t stream);

/* ---- many small problems at once (mini-batch criteria, posterior samplers) ------------------------------------- */

/* B independent criteria -- the zero-mean NLL (q = 0) or REML with a mean design of q <= GPMP_BATCH_MAX_Q columns -- and, when
 * grads_dev != NULL, their gradients with respect to the covariance parameters, every step ONE launch over all
 * problems (problem = blockIdx.y / .z of the diagonal-block, GEMM and solve kernels).  Callers: the weighted mean over
 * the batches of a loader (gpmp/num/torch_backend.py:607-718, gpmp/dataloader.py:484-513: B batches, one parameter
 * vector -> theta_stride = 0) and log_prob evaluations at many parameter vectors on one data set
 * (gpmp/mcmc/param_posterior.py:229-278: stride_x = stride_z = stride_p = 0, theta_stride = 1 + noise + d).
 *   x + b stride_x: n_b x d points (row-major);  z + b stride_z: n_b values;  P + b stride_p: n_b x q (ldp) mean design;
 *   n_host[b] in (q, nmax] (host array; NULL: every problem has nmax points);  nmax <= GPMP_BATCH_MAX_N;
 *   theta_host + b theta_stride: the parameters of problem b (host);
 *   values_dev[B], grads_dev[B x (1 + noise + d)] (may be NULL), info_dev[B] as gpmp_nll_grad (per problem): device.
 * Smaller problems occupy an identity-padded nmax x nmax slot (log-det and quadratic form unchanged).
 * ws: gpmp_batch_ws_elems(nmax, d, q, B, grads_dev != NULL) doubles.  Enqueue only. */
size_t gpmp_batch_ws_elems(int nmax, int d, int q, int B, int with_grad);
int gpmp_nll_grad_batch(const double* x, long stride_x, const double* z, long stride_z, const double* P, long ldp,
                        long stride_p, int q, const int* n_host, int nmax, int d, int B, int p,
                        const double* theta_host, int theta_stride, int noise, double* ws, double* values_dev,
                        double* grads_dev, int* info_dev, gpmp_stream_t stream);

/* ---- distributed (2-D block-cyclic) Cholesky: the LOCAL half of one block-column step --------------------------
 * SURVEY 8(b) `gpmp_dist_*`; the reference has no counterpart (README.md:39-40).  The library links no communication
 * library and takes no communicator: the HOST owns every collective (gpmp_amd/dist over torch.distributed = RCCL; a
 * C++ / RCCL host: examples/dist_potrf_rccl.cpp) and calls these between them.  Layout: global
 * block (I, J) of size nb (a multiple of 128, <= 1024) of the n x n matrix lives on rank (I mod Pr, J mod Pc) at local
 * block (I div Pr, J div Pc) of ONE dense row-major local matrix (gpmp_dist_local_shape); only the last block is short.
 * Step k:  owner of (k, k): gpmp_dist_diag_factor -> msg;  [broadcast msg down process column k mod Pc];
 *          ranks of that column: gpmp_dist_panel_solve -> panel (their block rows I > k);  [broadcast along process rows];
 *          per process row rp: gpmp_dist_exchange_pack on the holder, [broadcast inside the process column],
 *          gpmp_dist_exchange_unpack everywhere -> colop (block rows J > k of the owned block COLUMNS);
 *          every rank: gpmp_dist_trailing_update.  Everything only enqueues on `stream`. */
/* doubles of the diagonal-block message [L_kk (bk x ld16(bk)) | inverses of its 128-blocks | info as a double] */
size_t gpmp_dist_diag_msg_elems(int bk);
/* factor the bk x bk diagonal block D in place (numpy_backend.py:466 on one block) and fill msg; msg's last double is the
 * LAPACK-style info of the block (0, or the 1-based failing pivot inside it) */
int gpmp_dist_diag_factor(double* D, int bk, long ldd, double* msg, gpmp_stream_t stream);
size_t gpmp_dist_panel_ws_elems(int bk);
/* panel = P L_kk^-T for the `rows` local rows below the diagonal block (P: rows x bk view into the local matrix, also
 * overwritten with the result); ws: gpmp_dist_panel_ws_elems(bk) doubles (NULL or a ragged bk: substitution in place) */
int gpmp_dist_panel_solve(const double* msg, int bk, double* P, int rows, long ldp, double* panel, long ldo, double* ws,
                          gpmp_stream_t stream);
/* rows x cols of the local matrix of rank (r, c) */
int gpmp_dist_local_shape(int n, int nb, int pr, int pc, int r, int c, long* rows_out, long* cols_out);
/* step k on rank (r, c): rows of the panel buffer (owned block rows I > k), rows of the column operand (owned block
 * columns J > k), and the local row / column where they start */
int gpmp_dist_step_shape(int n, int nb, int pr, int pc, int r, int c, int k, long* panel_rows_out, long* colop_rows_out,
                         long* panel_row0_out, long* colop_col0_out);
/* rows of the piece that process row rp contributes to the column operand of process column c at step k
 * (the blocks J > k with J mod Pc == c and J mod Pr == rp); 0: nothing to exchange, < 0: bad arguments */
long gpmp_dist_exchange_rows(int n, int nb, int pr, int pc, int rp, int c, int k);
/* holder (r == rp): gather those blocks from its panel buffer into `piece` (consecutive rows) */
int gpmp_dist_exchange_pack(const double* panel, long ldp, double* piece, long ldq, int n, int nb, int pr, int pc, int r, int c,
                            int k, int bk, gpmp_stream_t stream);
/* every rank of the process column: scatter process row rp's piece into its column operand */
int gpmp_dist_exchange_unpack(const double* piece, long ldq, double* colop, long ldc, int n, int nb, int pr, int pc, int rp, int c,
                              int k, int bk, gpmp_stream_t stream);
/* A_IJ -= panel_I colop_J^T for the local blocks I >= J with local block-column index in [jlo, jhi) (jhi < 0: to the end)
 * and, when rows_after >= 0, global block row I > rows_after: a staircase of fp64 MFMA GEMMs over groups of 4 block rows */
int gpmp_dist_trailing_update(double* A, long lda, int n, int nb, int pr, int pc, int r, int c, int k, const double* panel,
                              long ldp, const double* colop, long ldc, int jlo, int jhi, int rows_after, gpmp_stream_t stream);

/* The blocks of T^T T2 on a block-cyclic INVERSE factor (what K^-1 = T^T T needs for the gradient of the ML / REML criteria;
 * gpmp/num/numpy_backend.py:458-463 forms the inverse; the reference differentiates by autograd / finite differences): T and T2
 * are this rank's local rows of T = L^-1 for the column sets c and c2 (c2 == c: T2 = T; otherwise the neighbour's part, received
 * by the host).  M (local columns of set c) x (local columns of set c2) <- T^T T2 with the contraction of block (I, J) starting at the
 * first local block row >= max(I, J) -- T is lower triangular, everything before is structurally zero -- in ONE launch; lower_only
 * != 0: only the blocks J <= I are computed (the rest of M is left untouched).  The sum over the process rows is the host's
 * (the trace against dK is linear: no matrix reduction is needed).  nb must be 1024.  Enqueue only. */
int gpmp_dist_inverse_gram(const double* T, long ldt, const double* T2, long ldt2, double* M, long ldm, int n, int nb, int pr, int pc,
                           int r, int c, int c2, int lower_only, gpmp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GPMP_HIP_H */
