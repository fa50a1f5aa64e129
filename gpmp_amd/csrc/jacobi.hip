// One-sided Jacobi (Hestenes) singular value decomposition of a square matrix -- what stands behind `gnp.svd`
// (gpmp/num/numpy_backend.py: scipy.linalg.svd; torch_backend.py:833-834), whose one caller on the path is the "svd" route of
// gpmp/core/sample_paths.py:54-58: the symmetric square root U sqrt(s) U^T of a covariance matrix that is only positive
// SEMI-definite (observation and prediction points stacked, repeated points), where the Cholesky route has no factor.
//
// G (n x n, row-major) starts as A, W as the identity.  A sweep is n - 1 steps; step s rotates the n / 2 disjoint ROW pairs of a
// round-robin tournament -- one workgroup per pair: three dot products over the two rows of G (a = |g_p|^2, b = |g_q|^2,
// c = g_p . g_q), the rotation that zeroes c, applied to the two rows of G and of W.  After convergence the rows of G are mutually
// orthogonal: G = diag(s) Vt, W = U^T, A = U diag(s) Vt.  Rows are contiguous, so every access is a coalesced 8-byte stream:
// HBM-bound, 8 n doubles read + 4 n written per pair; a sweep moves 6 n^2 doubles.  The largest |c| / sqrt(a b) seen in the
// sweep lands in a device word (the caller decides when to stop); nothing synchronises.
#include "common.h"

namespace gpmp {
namespace {

__device__ __forceinline__ void tournament_pair(int n_even, int s, int k, int& p, int& q) {
  // players 0 .. n_even - 1, player n_even - 1 fixed; step s in [0, n_even - 1)
  const int m = n_even - 1;
  if (k == 0) { p = m; q = s; }
  else { p = (s + k) % m; q = (s - k + m) % m; }
  if (p > q) { const int t = p; p = q; q = t; }
}

__global__ void __launch_bounds__(256) jacobi_step_kernel(double* __restrict__ G, long ldg, double* __restrict__ W, long ldw, int n, int n_even,
                                                          int step, double tiny2, double* __restrict__ conv) {
  int p, q;
  tournament_pair(n_even, step, blockIdx.x, p, q);
  if (q >= n) return;                                  // the padding player of an odd n sits this pair out
  double* gp = G + (long)p * ldg;
  double* gq = G + (long)q * ldg;
  double a = 0.0, b = 0.0, c = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const double x = gp[j], y = gq[j];
    a = fma(x, x, a); b = fma(y, y, b); c = fma(x, y, c);
  }
  __shared__ double red[3][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; red[2][threadIdx.x >> 6] = c; }
  __syncthreads();
  a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  c = red[2][0] + red[2][1] + red[2][2] + red[2][3];
  const double ab = a * b;
  if (!(a > tiny2 && b > tiny2)) return;               // a numerically zero row (rank deficiency): its direction is noise, leave it
  const double off = fabs(c) / sqrt(ab);
  if (threadIdx.x == 0) {
    // non-negative doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long*>(conv), (unsigned long long)__double_as_longlong(off));
  }
  if (off <= 1e-15) return;
  const double zeta = (b - a) / (2.0 * c);
  const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
  double* wp = W + (long)p * ldw;
  double* wq = W + (long)q * ldw;
  for (int j = threadIdx.x; j < n; j += 256) {
    const double x = gp[j], y = gq[j];
    gp[j] = cs * x - sn * y;
    gq[j] = sn * x + cs * y;
    const double u = wp[j], v = wq[j];
    wp[j] = cs * u - sn * v;
    wq[j] = sn * u + cs * v;
  }
}

}  // namespace
}  // namespace gpmp

using namespace gpmp;

extern "C" int gpmp_jacobi_sweep(double* G, long ldg, double* W, long ldw, int n, double tiny_norm, double* conv_dev, gpmp_stream_t stream) {
  GPMP_ARG(n >= 0 && n <= GPMP_MAX_EXTENT, 5, "n outside [0, GPMP_MAX_EXTENT]");
  if (n <= 1) return 0;
  GPMP_ARG(G != nullptr && W != nullptr, 1, "G or W is NULL");
  GPMP_ARG(ldg >= n && ldw >= n, 2, "leading dimension < n");
  GPMP_ARG(tiny_norm >= 0.0, 6, "tiny_norm < 0");
  GPMP_ARG(conv_dev != nullptr, 7, "conv_dev is NULL");
  hipStream_t st = as_stream(stream);
  GPMP_HIP_TRY(hipMemsetAsync(conv_dev, 0, sizeof(double), st));
  const int n_even = n + (n & 1);
  for (int s = 0; s < n_even - 1; ++s) {
    hipLaunchKernelGGL(jacobi_step_kernel, dim3(n_even / 2), dim3(256), 0, st, G, ldg, W, ldw, n, n_even, s, tiny_norm * tiny_norm, conv_dev);
  }
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}
