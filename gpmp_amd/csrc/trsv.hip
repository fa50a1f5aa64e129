// Triangular solves with a few right-hand sides (m <= 4 per pass): the single-vector solves of the
// likelihood (L^-1 z, gpmp/core/likelihood.py:46) and of the mean-space algebra (L^-1 [z, P]).
// HBM-bound (every element of L is read once: 4 n^2 bytes), so it runs as one small kernel per 128-row
// diagonal block instead of going through 128 x 128 MFMA tiles:
//   * x_k (the solution rows of block k) is already stored in B when step k starts;
//   * workgroup b applies it to its 128 rows, one row per thread pair:  b_rows -= L[rows, k] x_k ;
//   * workgroup 0 owns the NEXT diagonal block's rows: after its update it computes
//     x_{k+1} = inv(L_{k+1,k+1}) b_{k+1} (128 x 128 mat-vec) and stores it over b_{k+1} in place
//     (no other workgroup of this launch reads those rows).
// Every thread walks a contiguous piece of one row of L (16-byte loads), x_k is broadcast from LDS;
// no cross-lane reductions on the critical path.
#include "common.h"

namespace gpmp {
namespace {

// y[i] (i < jb) = sum_l M(i, l) v[l] for the 128 x 128 block Dinv (row-major, ld NB); TRANS uses Dinv^T.
// 256 threads: thread t handles row i = t & 127 and the half (t >> 7) of the l range; halves meet in LDS.
template <int R, bool TRANS>
__device__ __forceinline__ void block_matvec(const double* __restrict__ dinv, const double (*v)[R], double (*y)[R],
                                             double (*part)[R], int t) {
  const int i = t & 127, half = t >> 7;
  double acc[R];
#pragma unroll
  for (int c = 0; c < R; ++c) acc[c] = 0.0;
  if (!TRANS) {
    const double* row = dinv + i * NB + half * 64;
#pragma unroll 8
    for (int l = 0; l < 64; l += 2) {
      const d2 m2 = *reinterpret_cast<const d2*>(row + l);
#pragma unroll
      for (int c = 0; c < R; ++c) acc[c] = fma(m2[0], v[half * 64 + l][c], fma(m2[1], v[half * 64 + l + 1][c], acc[c]));
    }
  } else {
#pragma unroll 8
    for (int l = 0; l < 64; ++l) {
      const double m = dinv[(half * 64 + l) * NB + i];   // coalesced across i
#pragma unroll
      for (int c = 0; c < R; ++c) acc[c] = fma(m, v[half * 64 + l][c], acc[c]);
    }
  }
  if (half == 1) {
#pragma unroll
    for (int c = 0; c < R; ++c) part[i][c] = acc[c];
  }
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int c = 0; c < R; ++c) y[i][c] = acc[c] + part[i][c];
  }
  __syncthreads();
}

// INIT launch (grid 1): x_first = op(Dinv_first) b_first stored in place.
// STEP launch for block k: B[rows] -= op(L)[rows, k] x_k for the rows still to be solved; the workgroup that
// owns the next diagonal block then turns it into x_next in place.
template <int R, bool TRANS, bool INIT>
__global__ void __launch_bounds__(256) trsv_kernel(const double* __restrict__ L, long ldl, const double* __restrict__ dinv,
                                                   double* __restrict__ B, long ldb, int n, int k, int m) {
  __shared__ double xs[NB][R];
  __shared__ double ys[NB][R];
  __shared__ double part[NB][R];
  const int t = threadIdx.x;
  const int nblk = (n + NB - 1) / NB;
  const int k0 = k * NB;
  const int jb = (n - k0) < NB ? (n - k0) : NB;
  if (INIT) {
    for (int idx = t; idx < NB * R; idx += 256) {
      const int l = idx / R, c = idx % R;
      xs[l][c] = (l < jb && c < m) ? B[(long)(k0 + l) * ldb + c] : 0.0;
    }
    __syncthreads();
    block_matvec<R, TRANS>(dinv + (size_t)k * NB * NB, xs, ys, part, t);
    for (int idx = t; idx < jb * R; idx += 256) {
      const int l = idx / R, c = idx % R;
      if (c < m) B[(long)(k0 + l) * ldb + c] = ys[l][c];
    }
    return;
  }
  // x_k from B
  for (int idx = t; idx < NB * R; idx += 256) {
    const int l = idx / R, c = idx % R;
    xs[l][c] = (l < jb && c < m) ? B[(long)(k0 + l) * ldb + c] : 0.0;
  }
  __syncthreads();
  // target block of this workgroup: forward -> blocks k+1+b ; backward -> blocks k-1-b
  const int tb = TRANS ? (k - 1 - (int)blockIdx.x) : (k + 1 + (int)blockIdx.x);
  const int r0 = tb * NB;
  const int rb = (n - r0) < NB ? (n - r0) : NB;
  const int i = t & 127, half = t >> 7;
  double acc[R];
#pragma unroll
  for (int c = 0; c < R; ++c) acc[c] = 0.0;
  if (i < rb) {
    if (!TRANS) {
      // row r0+i of L, columns k0 + half*64 .. +64 (contiguous per thread)
      const double* row = L + (long)(r0 + i) * ldl + k0 + half * 64;
      const int lmax = (jb - half * 64) < 64 ? (jb - half * 64) : 64;
      if (lmax == 64 && ((ldl & 1) == 0) && ((reinterpret_cast<uintptr_t>(L) & 15) == 0)) {
#pragma unroll 8
        for (int l = 0; l < 64; l += 2) {
          const d2 a2 = *reinterpret_cast<const d2*>(row + l);
#pragma unroll
          for (int c = 0; c < R; ++c) acc[c] = fma(a2[0], xs[half * 64 + l][c], fma(a2[1], xs[half * 64 + l + 1][c], acc[c]));
        }
      } else {
        for (int l = 0; l < lmax; ++l) {
          const double a0 = row[l];
#pragma unroll
          for (int c = 0; c < R; ++c) acc[c] = fma(a0, xs[half * 64 + l][c], acc[c]);
        }
      }
    } else {
      // (L^T)[r0+i, k0+l] = L[k0+l][r0+i]: coalesced across i
      const int lmax = (jb - half * 64) < 64 ? (jb - half * 64) : 64;
      for (int l = 0; l < lmax; ++l) {
        const double a = L[(long)(k0 + half * 64 + l) * ldl + r0 + i];
#pragma unroll
        for (int c = 0; c < R; ++c) acc[c] = fma(a, xs[half * 64 + l][c], acc[c]);
      }
    }
  }
  if (half == 1) {
#pragma unroll
    for (int c = 0; c < R; ++c) part[i][c] = acc[c];
  }
  __syncthreads();
  const bool owner_of_next = (blockIdx.x == 0);
  if (half == 0 && i < rb) {
#pragma unroll
    for (int c = 0; c < R; ++c) {
      if (c < m) {
        const double v = B[(long)(r0 + i) * ldb + c] - (acc[c] + part[i][c]);
        if (owner_of_next) xs[i][c] = v; else B[(long)(r0 + i) * ldb + c] = v;
      }
    }
  }
  if (!owner_of_next) return;
  // this workgroup holds the fully updated residual of the next diagonal block in xs: solve it
  if (half == 0 && i >= rb) {
#pragma unroll
    for (int c = 0; c < R; ++c) xs[i][c] = 0.0;
  }
  if (half == 0) {
#pragma unroll
    for (int c = 0; c < R; ++c) if (c >= m) xs[i][c] = 0.0;
  }
  __syncthreads();
  block_matvec<R, TRANS>(dinv + (size_t)tb * NB * NB, xs, ys, part, t);
  for (int idx = t; idx < rb * R; idx += 256) {
    const int l = idx / R, c = idx % R;
    if (c < m) B[(long)(r0 + l) * ldb + c] = ys[l][c];
  }
  (void)nblk;
}

template <int R>
int run(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, int trans, hipStream_t st) {
  const int nblk = (n + NB - 1) / NB;
  if (!trans) {
    hipLaunchKernelGGL((trsv_kernel<R, false, true>), dim3(1), dim3(256), 0, st, L, ldl, dinv, B, ldb, n, 0, m);
    for (int k = 0; k + 1 < nblk; ++k)
      hipLaunchKernelGGL((trsv_kernel<R, false, false>), dim3(nblk - 1 - k), dim3(256), 0, st, L, ldl, dinv, B, ldb, n, k, m);
  } else {
    hipLaunchKernelGGL((trsv_kernel<R, true, true>), dim3(1), dim3(256), 0, st, L, ldl, dinv, B, ldb, n, nblk - 1, m);
    for (int k = nblk - 1; k >= 1; --k)
      hipLaunchKernelGGL((trsv_kernel<R, true, false>), dim3(k), dim3(256), 0, st, L, ldl, dinv, B, ldb, n, k, m);
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
