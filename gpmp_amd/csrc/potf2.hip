// Diagonal-block kernel of the blocked Cholesky: factor one NB x NB (128 x 128) block in LDS and
// produce the inverse of its triangular factor.  One workgroup, whole block resident in LDS
// (128 x 129 doubles = 132 KB of the CU's 160 KB).
//
//  phase 1  right-looking Cholesky, column by column (two barriers per column)
//  phase 2  T = L^-1 by forward substitution, one column of T per thread pair, no barriers:
//           T is kept in the (otherwise unused) upper triangle of the LDS image, transposed.
//
// The inverse is what turns every panel solve of the blocked algorithms into an MFMA GEMM
// (X = A21 * inv(L11)^T), see linalg.hip.
#include "common.h"

namespace gpmp {
namespace {

constexpr int LDS_LD = NB + 1;

__global__ void __launch_bounds__(256) potf2_inv_kernel(double* __restrict__ A, long lda, int n_total,
                                                        double* __restrict__ dinv, int* info,
                                                        int offset, int do_factor) {
  // batched over blockIdx.x: block b works on the diagonal block starting at row/col b * NB
  A += (long)blockIdx.x * NB * (lda + 1);
  dinv += (long)blockIdx.x * NB * NB;
  offset += blockIdx.x * NB;
  const int jb = (n_total - (int)blockIdx.x * NB) < NB ? (n_total - (int)blockIdx.x * NB) : NB;
  extern __shared__ __attribute__((aligned(16))) double S[];  // [NB][NB+1] + col[NB] + dg[NB]
  double* col = S + NB * LDS_LD;
  double* dg = col + NB;
  const int t = threadIdx.x;

  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx / NB, j = idx % NB;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < jb && j <= i) v = A[(long)i * lda + j];
    S[i * LDS_LD + j] = v;
  }
  __syncthreads();

  if (do_factor) {
    const int tx = t & 15, ty = t >> 4;
    for (int j = 0; j < NB; ++j) {
      double d = S[j * LDS_LD + j];
      if (!(d > 0.0)) {  // also true for NaN
        if (t == 0 && j < jb) atomicCAS(info, 0, offset + j + 1);
        d = 1.0;
      }
      const double sd = sqrt(d);
      const double r = 1.0 / sd;
      if (t > j && t < NB) col[t] = S[t * LDS_LD + j] * r;
      __syncthreads();
      if (t > j && t < NB) S[t * LDS_LD + j] = col[t];
      if (t == j) S[j * LDS_LD + j] = sd;
      const int m = NB - 1 - j;
      for (int ib = 0; ib * 16 < m; ++ib) {
        const int i = j + 1 + ib * 16 + ty;
        if (i < NB) {
          const double ci = col[i];
          for (int kb = 0; kb <= ib; ++kb) {
            const int k = j + 1 + kb * 16 + tx;
            if (k <= i) S[i * LDS_LD + k] -= ci * col[k];
          }
        }
      }
      __syncthreads();
    }
    // factor back to global memory (lower triangle only)
    for (int idx = t; idx < NB * NB; idx += 256) {
      const int i = idx / NB, j = idx % NB;
      if (i < jb && j <= i) A[(long)i * lda + j] = S[i * LDS_LD + j];
    }
  }

  // ---- phase 2: T = L^-1.  Column c by thread pair (2c, 2c+1); T[i][c] (i > c) lives at S[c][i].
  {
    const int c = t >> 1, par = t & 1;
    const double tcc = 1.0 / S[c * LDS_LD + c];
    for (int i = c + 1; i < NB; ++i) {
      // sum_{k=c}^{i-1} L[i][k] * T[k][c], split by parity of (k - c)
      double s = par == 0 ? S[i * LDS_LD + c] * tcc : 0.0;
      for (int k = c + 1 + (par == 0 ? 1 : 0); k < i; k += 2) s += S[i * LDS_LD + k] * S[c * LDS_LD + k];
      s += __shfl_xor(s, 1);
      const double v = -s / S[i * LDS_LD + i];
      if (par == 0) S[c * LDS_LD + i] = v;
      // both threads of the pair read S[c][i] in later iterations: same wave, program order +
      // the shuffle above keep them in lockstep; make the LDS write visible before the next read.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (par == 0) dg[c] = tcc;
  }
  __syncthreads();
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx / NB, c = idx % NB;
    double v = 0.0;
    if (c < i) v = S[c * LDS_LD + i];
    else if (c == i) v = dg[i];
    dinv[idx] = v;
  }
}

int launch(double* A, long lda, int n_total, int nblocks, double* dinv, int* info_dev, int offset,
           int do_factor, hipStream_t st) {
  static bool attr_done = false;
  const size_t lds = sizeof(double) * (NB * LDS_LD + 2 * NB);
  if (!attr_done) {
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_inv_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  {
    ProfScope ps(PK_POTF2, st, (double)nblocks);
    hipLaunchKernelGGL(potf2_inv_kernel, dim3(nblocks), dim3(256), lds, st, A, lda, n_total, dinv, info_dev,
                       offset, do_factor);
  }
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

int launch_potf2_inv(double* A, long lda, int jb, double* dinv, int* info_dev, int offset, hipStream_t st) {
  return launch(A, lda, jb, 1, dinv, info_dev, offset, 1, st);
}
int launch_trtri_blocks(const double* L, long ldl, int n, double* dinv, hipStream_t st) {
  if (n <= 0) return 0;
  return launch(const_cast<double*>(L), ldl, n, (n + NB - 1) / NB, dinv, nullptr, 0, 0, st);
}

}  // namespace gpmp
