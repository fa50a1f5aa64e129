// Triangular solves with a few right-hand sides (m <= 4 per pass): the single-vector solves of the
// likelihood (L^-1 z, gpmp/core/likelihood.py:46) and of the mean-space algebra (L^-1 [z, P]).
// HBM-bound (every element of L is read once, 4 n^2 bytes), so it runs as one small fused kernel
// per 128-row diagonal block instead of going through 128 x 128 MFMA tiles:
//   every workgroup recomputes x_k = inv(L_kk) b_k (128 x 128 mat-vec from L2) into LDS and applies it
//   to its 128 rows:  b_rows -= L[rows, k] x_k.  The residual block b_k itself is left untouched
//   during the sweep (the other workgroups of the same launch are reading it); one final launch turns
//   every b_k into x_k in place.
// Rows are walked one per wave with the 64 lanes along the contraction index (coalesced 1 KB row
// segments), partial sums reduced with DPP/shuffle adds.
#include "common.h"

namespace gpmp {
namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Forward step k: x_k = Dinv_k B[k0:k0+jb] (LDS only);  B[r] -= L[r, k0:k0+jb] x_k for r >= k0 + jb.
// grid.x = ceil((n - k0 - jb) / 128).   TRANS: backward step for L^T (rows above, L read by rows).
// FINAL: grid.x = number of diagonal blocks; block b stores x_b over b_b (no update).
template <int R, bool TRANS, bool FINAL>
__global__ void __launch_bounds__(256) trsv_step_kernel(const double* __restrict__ L, long ldl,
                                                        const double* __restrict__ dinv_k, double* __restrict__ B,
                                                        long ldb, int n, int k0, int jb, int m) {
  if (FINAL) {
    k0 = blockIdx.x * NB;
    jb = (n - k0) < NB ? (n - k0) : NB;
    dinv_k += (size_t)blockIdx.x * NB * NB;
  }
  __shared__ double xs[NB][R];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // ---- x_k = op(Dinv_k) * b_k   (rows i = wave*32 .. +31, lanes over l)
  double bl[2][R];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int c = 0; c < R; ++c) {
      const int l = 2 * lane + u;
      bl[u][c] = (l < jb && c < m) ? B[(long)(k0 + l) * ldb + c] : 0.0;
    }
  for (int ii = 0; ii < 32; ++ii) {
    const int i = wave * 32 + ii;
    double d0, d1;
    if (!TRANS) { d0 = dinv_k[i * NB + 2 * lane]; d1 = dinv_k[i * NB + 2 * lane + 1]; }
    else { d0 = dinv_k[(2 * lane) * NB + i]; d1 = dinv_k[(2 * lane + 1) * NB + i]; }
#pragma unroll
    for (int c = 0; c < R; ++c) {
      const double s = wave_sum(fma(d0, bl[0][c], d1 * bl[1][c]));
      if (lane == 0) xs[i][c] = s;
    }
  }
  __syncthreads();
  if (FINAL) {
    for (int idx = t; idx < jb * R; idx += 256) {
      const int i = idx / R, c = idx % R;
      if (c < m) B[(long)(k0 + i) * ldb + c] = xs[i][c];
    }
    return;
  }
  double xl[2][R];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int c = 0; c < R; ++c) xl[u][c] = xs[2 * lane + u][c];
  if (!TRANS) {
    const int r0 = k0 + jb + blockIdx.x * NB;
    for (int ii = 0; ii < 32; ++ii) {
      const int r = r0 + wave * 32 + ii;
      if (r >= n) break;
      const double* lp = L + (long)r * ldl + k0 + 2 * lane;
      const double l0 = (2 * lane < jb) ? lp[0] : 0.0, l1 = (2 * lane + 1 < jb) ? lp[1] : 0.0;
#pragma unroll
      for (int c = 0; c < R; ++c) {
        const double s = wave_sum(fma(l0, xl[0][c], l1 * xl[1][c]));
        if (lane == 0 && c < m) B[(long)r * ldb + c] -= s;
      }
    }
  } else {
    // rows above: B[j] -= sum_l L[k0 + l][j] x[l], j in this workgroup's 128 columns of L's block row
    const int j0 = blockIdx.x * NB;
    const int j = j0 + (t & 127), half = t >> 7;      // 2 threads per column split the 128 l's
    double acc[R];
#pragma unroll
    for (int c = 0; c < R; ++c) acc[c] = 0.0;
    if (j < k0) {
      for (int l = half * 64; l < half * 64 + 64 && l < jb; ++l) {
        const double v = L[(long)(k0 + l) * ldl + j];
#pragma unroll
        for (int c = 0; c < R; ++c) acc[c] = fma(v, xs[l][c], acc[c]);
      }
    }
    __shared__ double part[128][R];
    if (half == 1) {
#pragma unroll
      for (int c = 0; c < R; ++c) part[t & 127][c] = acc[c];
    }
    __syncthreads();
    if (half == 0 && j < k0) {
#pragma unroll
      for (int c = 0; c < R; ++c)
        if (c < m) B[(long)j * ldb + c] -= acc[c] + part[t][c];
    }
  }
}

template <int R>
int run(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, int trans, hipStream_t st) {
  const int nblk = (n + NB - 1) / NB;
  if (!trans) {
    for (int k = 0; k < nblk; ++k) {
      const int k0 = k * NB, jb = (n - k0) < NB ? (n - k0) : NB;
      const int rest = n - k0 - jb;
      if (rest <= 0) break;
      hipLaunchKernelGGL((trsv_step_kernel<R, false, false>), dim3((rest + NB - 1) / NB), dim3(256), 0, st, L, ldl,
                         dinv + (size_t)k * NB * NB, B, ldb, n, k0, jb, m);
    }
    hipLaunchKernelGGL((trsv_step_kernel<R, false, true>), dim3(nblk), dim3(256), 0, st, L, ldl, dinv, B, ldb, n, 0, 0, m);
  } else {
    for (int k = nblk - 1; k >= 1; --k) {
      const int k0 = k * NB, jb = (n - k0) < NB ? (n - k0) : NB;
      hipLaunchKernelGGL((trsv_step_kernel<R, true, false>), dim3((k0 + NB - 1) / NB), dim3(256), 0, st, L, ldl,
                         dinv + (size_t)k * NB * NB, B, ldb, n, k0, jb, m);
    }
    hipLaunchKernelGGL((trsv_step_kernel<R, true, true>), dim3(nblk), dim3(256), 0, st, L, ldl, dinv, B, ldb, n, 0, 0, m);
  }
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

// In-place op(L)^-1 B for an n x m B with m <= 4.
int trsv_few(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, int trans,
             hipStream_t st) {
  if (m <= 1) return run<1>(L, n, ldl, dinv, B, m, ldb, trans, st);
  if (m <= 2) return run<2>(L, n, ldl, dinv, B, m, ldb, trans, st);
  return run<4>(L, n, ldl, dinv, B, m, ldb, trans, st);
}

}  // namespace gpmp
