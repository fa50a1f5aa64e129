// Fused drivers with a linear-predictor mean behind the C ABI: REML value, ML / REML value + analytic gradient,
// leave-one-out -- each ONE call that only enqueues, for hosts that are not Python:
//   negative_log_restricted_likelihood        gpmp/core/likelihood.py:92-129   (contrasts W = Q[:, q:], G = W^T K W)
//   its gradient w.r.t. the covariance parameters  (reference: torch autograd, gpmp/num/torch_backend.py:574-604;
//                                              criterion wrapper gpmp/kernel/parameter_selection.py:35-124)
//   _loo_with_zero_mean / _loo_with_linear_predictor_mean_cpd    gpmp/core/loo.py:65-83,103-130
// The n x n complete QR and the two n^3 products of the reference are replaced by the exact identities
//   W (W^T K W)^-1 W^T = K^-1 - U S^-1 U^T =: Qinv,   U = K^-1 P,  S = P^T K^-1 P
//   ln|W^T K W| = ln|K| + ln|S| - ln|P^T P|
// (DESIGN.md section 2); the q x q algebra (Cholesky of S and P^T P, S^-1, S^-1 b) runs in one small workgroup on the
// device, so no driver ever waits for the host.  The mean design P = mean(xi) (n x q, row-major) is passed in: mean
// functions are user callables in the reference (gpmp/core/model.py:30-52).
#include "common.h"
#include <cfloat>
#include <cmath>

namespace gpmp {
namespace {

constexpr int QMAX = GPMP_MAX_RANK - 1;   // mean-design columns: W = L^-1 [z, P] has 1 + q <= GPMP_MAX_RANK columns
constexpr int QLD = QMAX + 1;

inline long pad16(long v) { return (v + 15) / 16 * 16; }

// small[] layout (doubles): [0] ln|S|  [1] ln|P^T P|  [2] quad = z^T Qinv z  [3] z^T K^-1 z   [16 + a] c_a = (S^-1 b)_a
constexpr int SM_LDS = 0, SM_LDP = 1, SM_QUAD = 2, SM_ZKZ = 3, SM_C = 16, SM_TOTAL = 16 + QLD + 8;

// Y[i] = [z_i, P_i1 .. P_iq]
__global__ void pack_zp_kernel(const double* __restrict__ z, const double* __restrict__ P, long ldp, int n, int q,
                               double* __restrict__ Y, long ldy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Y[(long)i * ldy] = z[i];
  for (int a = 0; a < q; ++a) Y[(long)i * ldy + 1 + a] = P[(long)i * ldp + a];
}

// In-LDS Cholesky A = R R^T (lower, in place) of a q x q matrix by one workgroup; returns the first failing pivot (1-based) or 0.
// A pivot that is not above q * eps * (its original diagonal entry) counts as a failure: the matrix is singular to working
// precision (a rank-deficient mean design leaves a pivot of rounding-error size, of either sign).
__device__ int chol_lds(double (*A)[QLD + 1], int q, int* fail, double* d0) {
  const int t = threadIdx.x, nt = blockDim.x;
  for (int k = t; k < q; k += nt) d0[k] = A[k][k];
  for (int k = 0; k < q; ++k) {
    __syncthreads();
    if (t == 0) {
      const double dkk = A[k][k];
      if (!(dkk > (double)q * 2.220446049250313e-16 * d0[k])) { if (*fail == 0) *fail = k + 1; A[k][k] = 1.0; } else A[k][k] = sqrt(dkk);
    }
    __syncthreads();
    const double piv = A[k][k];
    for (int i = k + 1 + t; i < q; i += nt) A[i][k] /= piv;
    __syncthreads();
    const int rem = q - k - 1;
    for (int idx = t; idx < rem * rem; idx += nt) {
      const int i = k + 1 + idx / rem, j = k + 1 + idx % rem;
      if (j <= i) A[i][j] -= A[i][k] * A[j][k];
    }
  }
  __syncthreads();
  return *fail;
}

// One workgroup.  Gm: (1 + q) x (1 + q) Gram matrix of W = L^-1 [z, P] (row k, column j at Gm[k * ldg + j]);
// PtP: q x q.  Writes small[] (see above) and Sinv (q x q, ld = lds); a non-positive pivot of S or P^T P sets
// *info = n + pivot (LAPACK-style continuation of the potrf numbering) if *info was 0.
__global__ void __launch_bounds__(256) meanspace_kernel(const double* __restrict__ Gm, long ldg, const double* __restrict__ PtP,
                                                        long ldp, int q, int n, double* __restrict__ small,
                                                        double* __restrict__ Sinv, long lds, int* info) {
  extern __shared__ __attribute__((aligned(16))) double ms_lds[];   // 2 (q x q) images + b + c: 84 KB, above the static limit
  double (*A)[QLD + 1] = reinterpret_cast<double (*)[QLD + 1]>(ms_lds);
  double (*R)[QLD + 1] = reinterpret_cast<double (*)[QLD + 1]>(ms_lds + QLD * (QLD + 1));     // R^-1 (lower)
  double* b = ms_lds + 2 * QLD * (QLD + 1);
  double* c = b + QLD;
  double* d0 = c + QLD;
  __shared__ int fail;
  const int t = threadIdx.x, nt = blockDim.x;
  if (t == 0) fail = 0;
  // ---- ln |P^T P|
  for (int idx = t; idx < q * q; idx += nt) {
    const int i = idx / q, j = idx % q;
    A[i][j] = 0.5 * (PtP[(long)i * ldp + j] + PtP[(long)j * ldp + i]);
  }
  __syncthreads();
  chol_lds(A, q, &fail, d0);
  if (t == 0) {
    double s = 0.0;
    for (int k = 0; k < q; ++k) s += log(A[k][k]);
    small[SM_LDP] = 2.0 * s;
  }
  __syncthreads();
  // ---- S = sym(Gm[1:, 1:]), b = Gm[1:, 0]
  for (int idx = t; idx < q * q; idx += nt) {
    const int i = idx / q, j = idx % q;
    A[i][j] = 0.5 * (Gm[(long)(1 + i) * ldg + 1 + j] + Gm[(long)(1 + j) * ldg + 1 + i]);
  }
  for (int i = t; i < q; i += nt) b[i] = Gm[(long)(1 + i) * ldg];
  __syncthreads();
  chol_lds(A, q, &fail, d0);
  if (t == 0) {
    double s = 0.0;
    for (int k = 0; k < q; ++k) s += log(A[k][k]);
    small[SM_LDS] = 2.0 * s;
    small[SM_ZKZ] = Gm[0];
  }
  // ---- R^-1: column j by forward substitution (thread j)
  for (int j = t; j < q; j += nt) {
    for (int i = 0; i < q; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int l = j; l < i; ++l) s -= A[i][l] * R[l][j];
      R[i][j] = (i < j) ? 0.0 : s / A[i][i];
    }
  }
  __syncthreads();
  // ---- S^-1 = R^-T R^-1
  for (int idx = t; idx < q * q; idx += nt) {
    const int i = idx / q, j = idx % q;
    double s = 0.0;
    for (int l = (i > j ? i : j); l < q; ++l) s += R[l][i] * R[l][j];
    Sinv[(long)i * lds + j] = s;
    A[i][j] = s;
  }
  __syncthreads();
  for (int i = t; i < q; i += nt) {
    double s = 0.0;
    for (int j = 0; j < q; ++j) s += A[i][j] * b[j];
    c[i] = s;
    small[SM_C + i] = s;
  }
  __syncthreads();
  if (t == 0) {
    double s = 0.0;
    for (int i = 0; i < q; ++i) s += b[i] * c[i];
    small[SM_QUAD] = Gm[0] - s;
    if (fail != 0) atomicCAS(info, 0, n + fail);
  }
}

// value = 1/2 ((n - q) ln 2 pi + ln|K| + ln|S| - ln|P^T P| + quad); +inf when anything failed (likelihood.py:123-124)
__global__ void reml_finalize_kernel(const double* logdetK, const double* small, const double* ssq, int q, int n, const int* info,
                                     double* out) {
  double v;
  if (q > 0) v = 0.5 * ((double)(n - q) * 1.8378770664093454835606594728112 + *logdetK + small[SM_LDS] - small[SM_LDP] + small[SM_QUAD]);
  else v = 0.5 * ((double)n * 1.8378770664093454835606594728112 + *logdetK + *ssq);
  if (*info != 0 || !(v == v) || v > DBL_MAX || v < -DBL_MAX) v = __builtin_huge_val();
  *out = v;
}

// Row i: U_i = X[i, 1:], alpha_i = X[i, 0];  US_i = U_i S^-1;  beta_i = alpha_i - U_i c.
//   F[i] = [US_i, beta_i],  G[i] = [U_i, beta_i]     (low-rank part of Qinv - beta beta^T for gpmp_matern_grad_trace)
//   loo (dcol != NULL): Qd = dcol_i - US_i . U_i ;  eloo = beta_i / Qd ;  sigma2loo = 1 / Qd ;  zloo = z_i - eloo
__global__ void __launch_bounds__(256) rows_kernel(const double* __restrict__ X, long ldx, const double* __restrict__ Sinv, long lds,
                                                   const double* __restrict__ small, int q, int n, double* __restrict__ F,
                                                   double* __restrict__ G, long ldf, const double* __restrict__ dcol,
                                                   const double* __restrict__ z, const int* __restrict__ info,
                                                   double* __restrict__ zloo, double* __restrict__ s2loo, double* __restrict__ eloo) {
  __shared__ double S[QLD][QLD + 1];
  __shared__ double cs[QLD];
  for (int idx = threadIdx.x; idx < q * q; idx += blockDim.x) S[idx / q][idx % q] = Sinv[(long)(idx / q) * lds + idx % q];
  for (int a = threadIdx.x; a < q; a += blockDim.x) cs[a] = small[SM_C + a];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* xr = X + (long)i * ldx;
  double beta = xr[0], usu = 0.0;
  for (int a = 0; a < q; ++a) beta -= xr[1 + a] * cs[a];
  for (int a = 0; a < q; ++a) {
    double s = 0.0;
    for (int l = 0; l < q; ++l) s += xr[1 + l] * S[l][a];
    usu += s * xr[1 + a];
    if (F != nullptr) { F[(long)i * ldf + a] = s; G[(long)i * ldf + a] = xr[1 + a]; }
  }
  if (F != nullptr) { F[(long)i * ldf + q] = beta; G[(long)i * ldf + q] = beta; }
  if (dcol != nullptr) {
    const double nan = __builtin_nan("");
    const bool bad = *info != 0;
    const double qd = dcol[i] - usu;
    const double e = beta / qd;
    eloo[i] = bad ? nan : e;
    s2loo[i] = bad ? nan : 1.0 / qd;
    zloo[i] = bad ? nan : z[i] - e;
  }
}

// g[j] <- 1/2 g[j] (gpmp_matern_grad_trace returns the plain traces); zeros when the factorisation failed (the
// criterion is +inf there and the selection wrappers return a zero gradient, numpy_backend.py:344-350)
__global__ void grad_finalize2_kernel(double* g, int len, const int* info) {
  const int j = threadIdx.x;
  if (j < len) g[j] = (*info != 0) ? 0.0 : 0.5 * g[j];
}

struct MeanLayout {
  long ldn, ldq;
  size_t K, dinv, T, Y, X, F, G, Gm, PtP, Sinv, small, scal, dcol, cd, gws, total;
};

// want_T: trtri target (gradient, LOO); want_fg: F / G of the gradient; want_dcol: LOO
MeanLayout mean_layout(int n, int d, int q, bool want_T, bool want_fg, bool want_dcol) {
  MeanLayout l;
  l.ldn = pad16(n);
  l.ldq = pad16(1 + q);
  size_t o = 0;
  auto take = [&](size_t cnt) { size_t at = o; o += (size_t)pad16((long)cnt); return at; };
  l.K = take((size_t)n * l.ldn);
  l.dinv = take(gpmp_dinv_elems(n));
  l.T = want_T ? take((size_t)n * l.ldn) : 0;
  l.Y = take((size_t)n * l.ldq);
  l.X = take((size_t)n * l.ldq);
  l.F = want_fg ? take((size_t)n * l.ldq) : 0;
  l.G = want_fg ? take((size_t)n * l.ldq) : 0;
  l.Gm = take((size_t)(2 + q) * l.ldq);
  l.PtP = take((size_t)(1 + q) * l.ldq);
  l.Sinv = take((size_t)(q > 0 ? q : 1) * l.ldq);
  l.small = take(SM_TOTAL);
  l.scal = take(16);
  l.dcol = want_dcol ? take((size_t)n) : 0;
  const size_t cd_cols = want_dcol ? (size_t)n : (size_t)(1 + q);
  l.cd = take(cd_cols * (size_t)gpmp_coldots_ws_rows(n));
  l.gws = want_fg ? take(gpmp_grad_ws_elems(n, d)) : 0;
  l.total = o;
  return l;
}

int check_common(const double* x, const double* z, const double* P, long ldp, int n, int d, int q, const double* theta_host,
                 const double* ws, const int* info_dev) {
  GPMP_ARG(x != nullptr, 1, "x is NULL");
  GPMP_ARG(z != nullptr, 2, "z is NULL");
  GPMP_ARG(q >= 0 && q <= QMAX, 7, "q outside [0, GPMP_MAX_RANK - 1]");
  GPMP_ARG(q == 0 || (P != nullptr && ldp >= q), 3, "P is NULL or ldp < q");
  GPMP_ARG(n > q && n <= GPMP_MAX_EXTENT, 5, "n <= q or above GPMP_MAX_EXTENT");
  GPMP_ARG(d >= 1 && d <= GPMP_MAX_DIM, 6, "d outside [1, GPMP_MAX_DIM]");
  GPMP_ARG(theta_host != nullptr, 9, "theta is NULL");
  GPMP_ARG(ws != nullptr, 11, "ws is NULL");
  GPMP_ARG(info_dev != nullptr, 13, "info_dev is NULL");
  return 0;
}

// Gram (lower) -> Cholesky -> W = L^-1 [z, P] -> Gram of W -> q x q algebra.  Leaves L in ws + l.K, W in ws + l.Y.
int factor_and_meanspace(const MeanLayout& l, const double* x, const double* z, const double* P, long ldp, int n, int d, int q,
                         int p, const double* theta_host, int noise, double* ws, int* info_dev, gpmp_stream_t stream) {
  hipStream_t st = as_stream(stream);
  double* K = ws + l.K;
  double* dinv = ws + l.dinv;
  double* Y = ws + l.Y;
  const double sigma2 = std::exp(theta_host[0]);
  const double diag = noise ? std::exp(theta_host[1]) : 10.0 * sigma2 * DBL_EPSILON;   // matern.py:90
  int rc = gpmp_matern_gram(x, nullptr, n, n, d, p, theta_host, noise, diag, 1, K, l.ldn, stream);
  if (rc) return rc;
  rc = gpmp_potrf_lower_async(K, n, l.ldn, dinv, info_dev, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(pack_zp_kernel, dim3((n + 255) / 256), dim3(256), 0, st, z, P, ldp, n, q, Y, l.ldq);
  GPMP_HIP_TRY(hipGetLastError());
  rc = gpmp_trsm_lower(K, n, l.ldn, dinv, Y, 1 + q, l.ldq, 0, nullptr, stream);                 // W = L^-1 [z, P]
  if (rc) return rc;
  rc = gpmp_logdet_chol(K, n, l.ldn, ws + l.scal, stream);
  if (rc) return rc;
  rc = gpmp_coldots(Y, n, 1 + q, l.ldq, Y, 1 + q, l.ldq, ws + l.Gm, l.ldq, ws + l.cd, stream);   // rows k <= q: W^T W; row 1 + q: column sums of squares
  if (rc) return rc;
  if (q > 0) {
    rc = gpmp_coldots(P, n, q, ldp, P, q, ldp, ws + l.PtP, l.ldq, ws + l.cd, stream);
    if (rc) return rc;
    const size_t ms_bytes = sizeof(double) * (2 * QLD * (QLD + 1) + 3 * QLD);
    static DeviceOnce attr_once;
    if (const long long dev_bit = attr_once.need()) {
    if (dev_bit < 0) { set_error("hipGetDevice failed or device ordinal above 62"); return -1; }
      GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(meanspace_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)ms_bytes));
      attr_once.done(dev_bit);
    }
    hipLaunchKernelGGL(meanspace_kernel, dim3(1), dim3(256), ms_bytes, st, ws + l.Gm, l.ldq, ws + l.PtP, l.ldq, q, n, ws + l.small,
                       ws + l.Sinv, l.ldq, info_dev);
    GPMP_HIP_TRY(hipGetLastError());
  }
  return 0;
}

// ---- prediction with a linear mean (universal kriging, gpmp/core/kriging.py:69-167 restated over the Schur complement) -------
// Per prediction point j, with V = L^-1 K(xi, xt), W = L^-1 [z, P] = [w, Wp], S = Wp^T Wp, c = S^-1 Wp^T w, D = V^T W:
//   r_j = D[1:, j] - Pt[j, :]          (what the Lagrange multipliers mu_j = S^-1 r_j have to absorb)
//   mean_j = D[0, j] - c . r_j,        var_j = sigma^2 - (colsumsq(V)_j - r_j^T S^-1 r_j)
__global__ void __launch_bounds__(256) predict_mean_finalize_kernel(const double* __restrict__ D, long ldd, const double* __restrict__ Pt,
                                                                    long ldpt, int m, int q, const double* __restrict__ Sinv, long lds,
                                                                    const double* __restrict__ small, double sigma2, int clamp,
                                                                    const int* info, double* __restrict__ zpm, double* __restrict__ zpv) {
  extern __shared__ double pm_lds[];           // Sinv (q x q) | c (q)
  double* Si = pm_lds;
  double* cs = pm_lds + (size_t)q * q;
  for (int idx = threadIdx.x; idx < q * q; idx += blockDim.x) Si[idx] = Sinv[(long)(idx / q) * lds + idx % q];
  for (int a = threadIdx.x; a < q; a += blockDim.x) cs[a] = small[SM_C + a];
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  double mean = D[j], quad = 0.0;
  for (int a = 0; a < q; ++a) {
    const double ra = D[(long)(1 + a) * ldd + j] - Pt[(long)j * ldpt + a];
    mean -= cs[a] * ra;
    double ta = 0.0;
    for (int b = 0; b < q; ++b) ta += Si[a * q + b] * (D[(long)(1 + b) * ldd + j] - Pt[(long)j * ldpt + b]);
    quad += ra * ta;
  }
  double v = sigma2 - (D[(long)(1 + q) * ldd + j] - quad);
  if (clamp && v < 0.0) v = 0.0;
  const bool bad = *info != 0;
  const double nan = __builtin_nan("");
  zpm[j] = bad ? nan : mean;
  zpv[j] = bad ? nan : v;
}

struct PredictMeanLayout {
  long ldn, ldm, ldq;
  size_t K, dinv, Kit, Y, Gm, PtP, Sinv, small, D, cd, total;
};
PredictMeanLayout predict_mean_layout(int n, int m, int q) {
  PredictMeanLayout l;
  l.ldn = pad16(n);
  l.ldm = pad16(m + 2 + q);          // K(xi, xt) | z | P | (a zero column when that keeps the column count even)
  l.ldq = pad16(1 + q);
  size_t o = 0;
  auto take = [&](size_t cnt) { size_t at = o; o += (size_t)pad16((long)cnt); return at; };
  l.K = take((size_t)n * l.ldn);
  l.dinv = take(gpmp_dinv_elems(n));
  l.Kit = take((size_t)n * l.ldm);
  l.Y = take((size_t)n * l.ldq);
  l.Gm = take((size_t)(2 + q) * l.ldq);
  l.PtP = take((size_t)(1 + q) * l.ldq);
  l.Sinv = take((size_t)(q > 0 ? q : 1) * l.ldq);
  l.small = take(SM_TOTAL);
  l.D = take((size_t)(2 + q) * l.ldm);
  const size_t cols = (size_t)(m > 1 + q ? m : 1 + q);
  l.cd = take(cols * (size_t)gpmp_coldots_ws_rows(n));
  l.total = o;
  return l;
}

}  // namespace
}  // namespace gpmp

using namespace gpmp;

extern "C" size_t gpmp_reml_ws_elems(int n, int q) {
  return (n > 0 && q >= 0 && q <= QMAX) ? mean_layout(n, 1, q, false, false, false).total : 0;
}

extern "C" int gpmp_reml(const double* x, const double* z, const double* P, long ldp, int n, int d, int q, int p,
                         const double* theta_host, int noise, double* ws, double* value_dev, int* info_dev,
                         gpmp_stream_t stream) {
  int rc = check_common(x, z, P, ldp, n, d, q, theta_host, ws, info_dev);
  if (rc) return rc;
  GPMP_ARG(value_dev != nullptr, 12, "value_dev is NULL");
  const MeanLayout l = mean_layout(n, d, q, false, false, false);
  rc = factor_and_meanspace(l, x, z, P, ldp, n, d, q, p, theta_host, noise, ws, info_dev, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(reml_finalize_kernel, dim3(1), dim3(1), 0, as_stream(stream), ws + l.scal, ws + l.small,
                     ws + l.Gm + (size_t)(1 + q) * l.ldq, q, n, info_dev, value_dev);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" size_t gpmp_nll_grad_ws_elems(int n, int d, int q) {
  return (n > 0 && q >= 0 && q <= QMAX && d >= 1) ? mean_layout(n, d, q, true, true, false).total : 0;
}

extern "C" int gpmp_nll_grad(const double* x, const double* z, const double* P, long ldp, int n, int d, int q, int p,
                             const double* theta_host, int noise, double* ws, double* value_dev, double* grad_dev,
                             int* info_dev, gpmp_stream_t stream) {
  int rc = check_common(x, z, P, ldp, n, d, q, theta_host, ws, info_dev);
  if (rc) return rc;
  GPMP_ARG(value_dev != nullptr, 12, "value_dev is NULL");
  GPMP_ARG(grad_dev != nullptr, 13, "grad_dev is NULL");
  hipStream_t st = as_stream(stream);
  const MeanLayout l = mean_layout(n, d, q, true, true, false);
  rc = factor_and_meanspace(l, x, z, P, ldp, n, d, q, p, theta_host, noise, ws, info_dev, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(reml_finalize_kernel, dim3(1), dim3(1), 0, st, ws + l.scal, ws + l.small,
                     ws + l.Gm + (size_t)(1 + q) * l.ldq, q, n, info_dev, value_dev);
  GPMP_HIP_TRY(hipGetLastError());
  double* K = ws + l.K;
  double* dinv = ws + l.dinv;
  double* X = ws + l.X;
  // X = L^-T W = K^-1 [z, P]
  GPMP_HIP_TRY(hipMemcpyAsync(X, ws + l.Y, sizeof(double) * (size_t)n * l.ldq, hipMemcpyDeviceToDevice, st));
  rc = gpmp_trsm_lower(K, n, l.ldn, dinv, X, 1 + q, l.ldq, 1, nullptr, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, X, l.ldq, ws + l.Sinv, l.ldq, ws + l.small, q, n,
                     ws + l.F, ws + l.G, l.ldq, (const double*)nullptr, (const double*)nullptr, info_dev, (double*)nullptr,
                     (double*)nullptr, (double*)nullptr);
  GPMP_HIP_TRY(hipGetLastError());
  // K^-1 (lower) = T^T T over the factor's own buffer: L is dead once T and X exist
  double* T = ws + l.T;
  rc = gpmp_trtri_lower(K, n, l.ldn, dinv, T, l.ldn, stream);
  if (rc) return rc;
  rc = gpmp_lauum_lower(T, n, l.ldn, K, l.ldn, stream);
  if (rc) return rc;
  rc = gpmp_matern_grad_trace(K, l.ldn, x, n, d, p, theta_host, noise, ws + l.F, ws + l.G, q + 1, l.ldq, grad_dev, ws + l.gws, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(grad_finalize2_kernel, dim3(1), dim3(128), 0, st, grad_dev, 1 + (noise ? 1 : 0) + d, info_dev);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" size_t gpmp_loo_ws_elems(int n, int q) {
  return (n > 0 && q >= 0 && q <= QMAX) ? mean_layout(n, 1, q, true, false, true).total : 0;
}

extern "C" int gpmp_loo(const double* x, const double* z, const double* P, long ldp, int n, int d, int q, int p,
                        const double* theta_host, int noise, double* ws, double* zloo_dev, double* sigma2loo_dev,
                        double* eloo_dev, int* info_dev, gpmp_stream_t stream) {
  int rc = check_common(x, z, P, ldp, n, d, q, theta_host, ws, info_dev);
  if (rc) return rc;
  GPMP_ARG(zloo_dev != nullptr && sigma2loo_dev != nullptr && eloo_dev != nullptr, 12, "output is NULL");
  hipStream_t st = as_stream(stream);
  const MeanLayout l = mean_layout(n, d, q, true, false, true);
  rc = factor_and_meanspace(l, x, z, P, ldp, n, d, q, p, theta_host, noise, ws, info_dev, stream);
  if (rc) return rc;
  double* K = ws + l.K;
  double* dinv = ws + l.dinv;
  double* X = ws + l.X;
  GPMP_HIP_TRY(hipMemcpyAsync(X, ws + l.Y, sizeof(double) * (size_t)n * l.ldq, hipMemcpyDeviceToDevice, st));
  rc = gpmp_trsm_lower(K, n, l.ldn, dinv, X, 1 + q, l.ldq, 1, nullptr, stream);                    // K^-1 [z, P]
  if (rc) return rc;
  // diag(K^-1) = column sums of squares of T = L^-1 (gpmp/core/linalg.py:17-46)
  double* T = ws + l.T;
  rc = gpmp_trtri_lower(K, n, l.ldn, dinv, T, l.ldn, stream);
  if (rc) return rc;
  rc = gpmp_coldots(T, n, n, l.ldn, nullptr, 0, 1, ws + l.dcol, n, ws + l.cd, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, X, l.ldq, ws + l.Sinv, l.ldq, ws + l.small, q, n,
                     (double*)nullptr, (double*)nullptr, l.ldq, ws + l.dcol, z, info_dev, zloo_dev, sigma2loo_dev, eloo_dev);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" size_t gpmp_predict_mean_ws_elems(int n, int m, int q) {
  return (n > 0 && m > 0 && q >= 1 && q <= QMAX) ? predict_mean_layout(n, m, q).total : 0;
}

extern "C" int gpmp_predict_mean(const double* xi, const double* zi, const double* Pi, long ldpi, const double* xt, const double* Pt,
                                 long ldpt, int n, int m, int d, int q, int p, const double* theta_host, int noise,
                                 int zero_neg_variances, double* ws, double* zpm_dev, double* zpv_dev, int* info_dev,
                                 gpmp_stream_t stream) {
  GPMP_ARG(xi != nullptr, 1, "xi is NULL");
  GPMP_ARG(zi != nullptr, 2, "zi is NULL");
  GPMP_ARG(q >= 1 && q <= QMAX, 11, "q outside [1, GPMP_MAX_RANK - 1] (q = 0: gpmp_predict_zero_mean)");
  GPMP_ARG(Pi != nullptr && ldpi >= q, 3, "Pi is NULL or ldpi < q");
  GPMP_ARG(xt != nullptr, 5, "xt is NULL");
  GPMP_ARG(Pt != nullptr && ldpt >= q, 6, "Pt is NULL or ldpt < q");
  GPMP_ARG(n > q && n <= GPMP_MAX_EXTENT, 8, "n <= q or above GPMP_MAX_EXTENT");
  GPMP_ARG(m > 0 && m <= GPMP_MAX_EXTENT, 9, "m outside [1, GPMP_MAX_EXTENT]");
  GPMP_ARG(d >= 1 && d <= GPMP_MAX_DIM, 10, "d outside [1, GPMP_MAX_DIM]");
  GPMP_ARG(theta_host != nullptr, 13, "theta is NULL");
  GPMP_ARG(ws != nullptr, 16, "ws is NULL");
  GPMP_ARG(zpm_dev != nullptr && zpv_dev != nullptr, 17, "output is NULL");
  GPMP_ARG(info_dev != nullptr, 19, "info_dev is NULL");
  hipStream_t st = as_stream(stream);
  const PredictMeanLayout l = predict_mean_layout(n, m, q);
  double* K = ws + l.K;
  double* dinv = ws + l.dinv;
  double* Kit = ws + l.Kit;
  double* Y = ws + l.Y;
  const double sigma2 = std::exp(theta_host[0]);
  const double diag = noise ? std::exp(theta_host[1]) : 10.0 * sigma2 * DBL_EPSILON;   // matern.py:90
  int rc = gpmp_matern_gram(xi, nullptr, n, n, d, p, theta_host, noise, diag, 1, K, l.ldn, stream);
  if (rc) return rc;
  rc = gpmp_matern_gram(xi, xt, n, m, d, p, theta_host, noise, 0.0, 0, Kit, l.ldm, stream);
  if (rc) return rc;
  // [z, P] ride along as 1 + q more right-hand sides: [V | W] = L^-1 [K(xi, xt) | z | P] in ONE solve (gpmp_predict_zero_mean does
  // the same with z): the separate few-column sweep waited for the end of the factorisation; an even column count keeps the
  // LDS-direct GEMM's fast path
  (void)Y;
  double* W = Kit + m;
  const long ldw = l.ldm;
  const int mb = (m + 1 + q) + ((m + 1 + q) & 1);
  hipLaunchKernelGGL(pack_zp_kernel, dim3((n + 255) / 256), dim3(256), 0, st, zi, Pi, ldpi, n, q, W, ldw);
  GPMP_HIP_TRY(hipGetLastError());
  if (mb > m + 1 + q) GPMP_HIP_TRY(hipMemset2DAsync(Kit + m + 1 + q, (size_t)l.ldm * sizeof(double), 0, sizeof(double), n, st));
  rc = gpmp_potrf_trsm_lower_async(K, n, l.ldn, dinv, info_dev, Kit, mb, l.ldm, stream);         // in place
  if (rc) return rc;
  rc = gpmp_coldots(W, n, 1 + q, ldw, W, 1 + q, ldw, ws + l.Gm, l.ldq, ws + l.cd, stream);        // W^T W
  if (rc) return rc;
  rc = gpmp_coldots(Pi, n, q, ldpi, Pi, q, ldpi, ws + l.PtP, l.ldq, ws + l.cd, stream);
  if (rc) return rc;
  {
    const size_t ms_bytes = sizeof(double) * (2 * QLD * (QLD + 1) + 3 * QLD);
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(meanspace_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)ms_bytes));
    hipLaunchKernelGGL(meanspace_kernel, dim3(1), dim3(256), ms_bytes, st, ws + l.Gm, l.ldq, ws + l.PtP, l.ldq, q, n, ws + l.small,
                       ws + l.Sinv, l.ldq, info_dev);
    GPMP_HIP_TRY(hipGetLastError());
  }
  rc = gpmp_coldots(Kit, n, m, l.ldm, W, 1 + q, ldw, ws + l.D, l.ldm, ws + l.cd, stream);         // V^T [w, Wp] and colsumsq(V)
  if (rc) return rc;
  const size_t fin_bytes = sizeof(double) * ((size_t)q * q + q);
  GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(predict_mean_finalize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(sizeof(double) * ((size_t)QMAX * QMAX + QMAX))));
  hipLaunchKernelGGL(predict_mean_finalize_kernel, dim3((m + 255) / 256), dim3(256), fin_bytes, st, ws + l.D, l.ldm, Pt, ldpt, m, q,
                     ws + l.Sinv, l.ldq, ws + l.small, sigma2, zero_neg_variances, info_dev, zpm_dev, zpv_dev);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}
