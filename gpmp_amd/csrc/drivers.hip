// Fused drivers of the zero-mean hot path behind the C ABI: one call = the whole device-side sequence of
//   negative_log_likelihood_zero_mean  (gpmp/core/likelihood.py:18-52)
//   kriging_predictor_with_zero_mean + _compute_posterior_variance (gpmp/core/kriging.py:35-67,170-199)
// for the Matern covariance of gpmp_matern_gram, so that a host that is not Python (or the reference's own
// backend module) can bind two symbols instead of re-assembling the sequence.  They only enqueue the same
// kernels the Python layer (gpmp_amd/core) launches -- Gram build, blocked Cholesky, triangular solves, column
// reductions -- plus one tiny kernel that combines device scalars; results and info stay on the device.
#include "common.h"
#include <cfloat>
#include <cmath>

namespace gpmp {
namespace {

inline long pad16(long v) { return (v + 15) / 16 * 16; }

// nll = 1/2 (n ln 2 pi + ln|K| + z^T K^-1 z); +inf when the factorisation failed (the reference's safe_inf
// convention, likelihood.py:47-48) or the value is not finite.
__global__ void nll_finalize_kernel(const double* logdet, const double* quad, const int* info, int n, double* out) {
  double v = 0.5 * ((double)n * 1.8378770664093454835606594728112 + *logdet + *quad);
  if (*info != 0 || !(v == v) || v > DBL_MAX || v < -DBL_MAX) v = __builtin_huge_val();
  *out = v;
}

// zpm[j] = D[0][j];  zpv[j] = sigma2 - D[1][j]  (optionally clamped at 0: Model.predict, core/model.py:290-296)
__global__ void predict_finalize_kernel(const double* D, long ldo, int m, double sigma2, int clamp, const int* info,
                                        double* zpm, double* zpv) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const bool bad = *info != 0;
  const double nan = __builtin_nan("");
  double v = sigma2 - D[ldo + j];
  if (clamp && v < 0.0) v = 0.0;
  zpm[j] = bad ? nan : D[j];
  zpv[j] = bad ? nan : v;
}

struct NllLayout {
  long ldn;
  size_t K, dinv, w, scal, cd, total;
};
NllLayout nll_layout(int n) {
  NllLayout l;
  l.ldn = pad16(n);
  size_t o = 0;
  l.K = o; o += (size_t)n * l.ldn;
  l.dinv = o; o += pad16((long)gpmp_dinv_elems(n));
  l.w = o; o += (size_t)n * 16;                                  // n x 1 right-hand side, ld 16
  l.scal = o; o += 16;                                           // [0] log-det, [1] quadratic form
  l.cd = o; o += pad16((long)gpmp_coldots_ws_rows(n));           // reduction scratch for one column
  l.total = o;
  return l;
}

}  // namespace
}  // namespace gpmp

using namespace gpmp;

extern "C" size_t gpmp_nll_ws_elems(int n) { return n > 0 ? nll_layout(n).total : 0; }

extern "C" int gpmp_nll_zero_mean(const double* x, const double* z, int n, int d, int p, const double* theta_host,
                                  int noise, double* ws, double* nll_dev, int* info_dev, gpmp_stream_t stream) {
  GPMP_ARG(x != nullptr, 1, "x is NULL");
  GPMP_ARG(z != nullptr, 2, "z is NULL");
  GPMP_ARG(n > 0 && n <= GPMP_MAX_EXTENT, 3, "n outside [1, GPMP_MAX_EXTENT]");
  GPMP_ARG(theta_host != nullptr, 6, "theta is NULL");
  GPMP_ARG(ws != nullptr, 8, "ws is NULL");
  GPMP_ARG(nll_dev != nullptr, 9, "nll_dev is NULL");
  GPMP_ARG(info_dev != nullptr, 10, "info_dev is NULL");
  hipStream_t st = as_stream(stream);
  const NllLayout l = nll_layout(n);
  double* K = ws + l.K;
  double* dinv = ws + l.dinv;
  double* w = ws + l.w;
  double* scal = ws + l.scal;
  const double sigma2 = std::exp(theta_host[0]);
  const double diag = noise ? std::exp(theta_host[1]) : 10.0 * sigma2 * DBL_EPSILON;   // matern.py:90
  int rc = gpmp_matern_gram(x, nullptr, n, n, d, p, theta_host, noise, diag, 1, K, l.ldn, stream);
  if (rc) return rc;
  rc = gpmp_potrf_lower_async(K, n, l.ldn, dinv, info_dev, stream);
  if (rc) return rc;
  GPMP_HIP_TRY(hipMemcpy2DAsync(w, 16 * sizeof(double), z, sizeof(double), sizeof(double), n, hipMemcpyDeviceToDevice, st));
  rc = gpmp_trsm_lower(K, n, l.ldn, dinv, w, 1, 16, 0, nullptr, stream);
  if (rc) return rc;
  rc = gpmp_logdet_chol(K, n, l.ldn, scal, stream);
  if (rc) return rc;
  rc = gpmp_coldots(w, n, 1, 16, nullptr, 0, 1, scal + 1, 1, ws + l.cd, stream);   // sum of squares of the one column
  if (rc) return rc;
  hipLaunchKernelGGL(nll_finalize_kernel, dim3(1), dim3(1), 0, st, scal, scal + 1, info_dev, n, nll_dev);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

namespace gpmp {
namespace {
struct PredictLayout {
  long ldn, ldm;
  size_t K, dinv, w, Kit, D, cd, total;
};
PredictLayout predict_layout(int n, int m) {
  PredictLayout l;
  l.ldn = pad16(n);
  l.ldm = pad16(m + 2);             // K(xi, xt) | z | (a zero column when that keeps the column count even)
  size_t o = 0;
  l.K = o; o += (size_t)n * l.ldn;
  l.dinv = o; o += pad16((long)gpmp_dinv_elems(n));
  l.w = o; o += (size_t)n * 16;
  l.Kit = o; o += (size_t)n * l.ldm;
  l.D = o; o += 2 * (size_t)l.ldm;                               // rows: V^T w, colsumsq(V)
  l.cd = o; o += (size_t)m * gpmp_coldots_ws_rows(n);
  l.total = o;
  return l;
}
}  // namespace
}  // namespace gpmp

extern "C" size_t gpmp_predict_ws_elems(int n, int m) { return (n > 0 && m > 0) ? predict_layout(n, m).total : 0; }

extern "C" int gpmp_predict_zero_mean(const double* xi, const double* zi, const double* xt, int n, int m, int d, int p,
                                      const double* theta_host, int noise, int zero_neg_variances, double* ws,
                                      double* zpm_dev, double* zpv_dev, int* info_dev, gpmp_stream_t stream) {
  GPMP_ARG(xi != nullptr, 1, "xi is NULL");
  GPMP_ARG(zi != nullptr, 2, "zi is NULL");
  GPMP_ARG(xt != nullptr, 3, "xt is NULL");
  GPMP_ARG(n > 0 && n <= GPMP_MAX_EXTENT, 4, "n outside [1, GPMP_MAX_EXTENT]");
  GPMP_ARG(m > 0 && m <= GPMP_MAX_EXTENT, 5, "m outside [1, GPMP_MAX_EXTENT]");
  GPMP_ARG(theta_host != nullptr, 8, "theta is NULL");
  GPMP_ARG(ws != nullptr, 11, "ws is NULL");
  GPMP_ARG(zpm_dev != nullptr && zpv_dev != nullptr, 12, "output is NULL");
  GPMP_ARG(info_dev != nullptr, 14, "info_dev is NULL");
  hipStream_t st = as_stream(stream);
  const PredictLayout l = predict_layout(n, m);
  double* K = ws + l.K;
  double* dinv = ws + l.dinv;
  double* w = ws + l.w;
  double* Kit = ws + l.Kit;
  double* D = ws + l.D;
  const double sigma2 = std::exp(theta_host[0]);
  const double diag = noise ? std::exp(theta_host[1]) : 10.0 * sigma2 * DBL_EPSILON;
  int rc = gpmp_matern_gram(xi, nullptr, n, n, d, p, theta_host, noise, diag, 1, K, l.ldn, stream);
  if (rc) return rc;
  rc = gpmp_matern_gram(xi, xt, n, m, d, p, theta_host, noise, 0.0, 0, Kit, l.ldm, stream);
  if (rc) return rc;
  // z rides along as one more right-hand side: [V | w] = L^-1 [K(xi, xt) | z] in ONE solve (the separate single-vector sweep
  // for w = L^-1 z was 0.17 ms of a 5.3 ms prediction at n = 4096 and waited for the end of the factorisation; as a column of
  // the panel-by-panel solve it is hidden like the rest).  An even column count keeps the LDS-direct GEMM's fast path.
  (void)w;
  const int mb = (m + 1) + ((m + 1) & 1);
  GPMP_HIP_TRY(hipMemcpy2DAsync(Kit + m, (size_t)l.ldm * sizeof(double), zi, sizeof(double), sizeof(double), n, hipMemcpyDeviceToDevice, st));
  if (mb > m + 1) GPMP_HIP_TRY(hipMemset2DAsync(Kit + m + 1, (size_t)l.ldm * sizeof(double), 0, sizeof(double), n, st));
  // factorisation and the solve (in place) in one call: panel by panel behind the factorisation at chain-bound sizes
  rc = gpmp_potrf_trsm_lower_async(K, n, l.ldn, dinv, info_dev, Kit, mb, l.ldm, stream);
  if (rc) return rc;
  rc = gpmp_coldots(Kit, n, m, l.ldm, Kit + m, 1, l.ldm, D, l.ldm, ws + l.cd, stream);      // V^T w and colsumsq(V)
  if (rc) return rc;
  hipLaunchKernelGGL(predict_finalize_kernel, dim3((m + 255) / 256), dim3(256), 0, st, D, l.ldm, m, sigma2,
                     zero_neg_variances, info_dev, zpm_dev, zpv_dev);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}
