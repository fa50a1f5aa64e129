// Diagonal-block kernel of the blocked Cholesky: factor one NB x NB (128 x 128) block and produce
// the inverse of its triangular factor, entirely in LDS, by ONE workgroup of 8 waves.
//
// The block is handled as 8 x 8 sub-blocks of 16 x 16 (the v_mfma_f64_16x16x4_f64 tile):
//   phase A  for each of the 8 block columns j
//     A1  wave 0 factors the 16 x 16 diagonal block in registers (one row per lane, pivots and
//         column entries broadcast with v_readlane; 1/sqrt by v_rsq_f64 + Newton, no division)
//     A2  panel rows below: X L_jj^T = A_panel by forward substitution, one row per lane
//     A3  trailing update A_ik -= L_ij L_kj^T on the MFMA pipe, one 16 x 16 block per wave at a time
//   phase B  inverses of the 8 diagonal 16 x 16 factors, one per wave, in registers
//   phase C  T = L^-1 by block forward substitution on MFMA:  T_ij = -T_ii sum_k L_ik T_kj ; the MFMA
//            result layout (row = (lane>>4) + 4 r) IS the B-operand layout of the next MFMA, so the
//            product with T_ii needs no LDS round trip.  T is kept transposed in the upper triangle.
//   phase D  L -> global (lower triangle), T -> dinv (NB x NB row-major, zero above the diagonal)
//
// LDS image: S[128][130] doubles (row stride 130 makes the MFMA fragment reads conflict free),
// dg[128] = 1 / L_ii, Td[8][16][18] = diagonal inverse blocks.  149 KB of the CU's 160 KB.
//
// The inverse is what turns every panel solve of the blocked algorithms into an MFMA GEMM
// (X = A21 * inv(L11)^T), see linalg.hip.
#include "common.h"

namespace gpmp {
namespace {

constexpr int LD = NB + 2;       // 130
constexpr int SB = 16;           // sub-block edge
constexpr int NSB = NB / SB;     // 8
constexpr int TD_LD = 18;
constexpr int THREADS = 512;

__device__ __forceinline__ double bcast_lane(double x, int src) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, src);
  hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}

// 1 / sqrt(d) to full double precision: hardware estimate + two Newton steps.
__device__ __forceinline__ double rsqrt_full(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * fma(-0.5 * d * y, y, 1.5);
  y = y * fma(-0.5 * d * y, y, 1.5);
  return y;
}

__global__ void __launch_bounds__(THREADS) potf2_inv_kernel(double* __restrict__ A, long lda, int n_total,
                                                            double* __restrict__ dinv, int* info, int offset,
                                                            int do_factor) {
  // batched over blockIdx.x: block b works on the diagonal block starting at row/col b * NB
  A += (long)blockIdx.x * NB * (lda + 1);
  dinv += (long)blockIdx.x * NB * NB;
  offset += blockIdx.x * NB;
  const int jb = (n_total - (int)blockIdx.x * NB) < NB ? (n_total - (int)blockIdx.x * NB) : NB;

  extern __shared__ __attribute__((aligned(16))) double S[];   // [NB][LD]
  double* dg = S + NB * LD;                                    // [NB]     1 / L_ii
  double* Td = dg + NB;                                        // [NSB][SB][TD_LD]
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lk = lane >> 4;

  // ---- load: lower triangle of the jb x jb block, identity padding, zeros above the diagonal
  for (int idx = t; idx < NB * NB; idx += THREADS) {
    const int i = idx >> 7, j = idx & (NB - 1);
    double v = (i == j) ? 1.0 : 0.0;
    if (i < jb && j <= i) v = A[(long)i * lda + j];
    S[i * LD + j] = v;
  }
  __syncthreads();

  if (do_factor) {
    for (int j = 0; j < NSB; ++j) {
      const int j0 = j * SB;
      // ---- A1: diagonal 16 x 16 block, wave 0, row (j0 + lane) in registers of lane < 16
      if (wave == 0) {
        double a[SB];
        const int row = j0 + (lane & 15);
#pragma unroll
        for (int c = 0; c < SB; ++c) a[c] = S[row * LD + j0 + c];
#pragma unroll
        for (int c = 0; c < SB; ++c) {
          double d = bcast_lane(a[c], c);
          if (!(d > 0.0)) {  // also NaN
            if (lane == 0 && j0 + c < jb) atomicCAS(info, 0, offset + j0 + c + 1);
            d = 1.0;
          }
          const double y = rsqrt_full(d);
          double s = d * y;
          s = fma(0.5 * y, fma(-s, s, d), s);   // one correction step: sqrt(d) to < 1 ulp
          a[c] = (lane == c) ? s : a[c] * y;
          if (lane == c) dg[j0 + c] = y;
#pragma unroll
          for (int k = c + 1; k < SB; ++k) {
            const double lkc = bcast_lane(a[c], k);
            a[k] = fma(-a[c], lkc, a[k]);
          }
        }
        if (lane < SB) {
#pragma unroll
          for (int c = 0; c < SB; ++c)
            if (c <= lane) S[row * LD + j0 + c] = a[c];
        }
      }
      __syncthreads();
      // ---- A2: panel rows below the diagonal block: x L_jj^T = a, one row per thread
      {
        const int row = j0 + SB + t;
        if (row < NB) {
          double x[SB];
#pragma unroll
          for (int c = 0; c < SB; ++c) x[c] = S[row * LD + j0 + c];
#pragma unroll
          for (int c = 0; c < SB; ++c) {
            double s = x[c];
#pragma unroll
            for (int k = 0; k < c; ++k) s = fma(-x[k], S[(j0 + c) * LD + j0 + k], s);
            x[c] = s * dg[j0 + c];
          }
#pragma unroll
          for (int c = 0; c < SB; ++c) S[row * LD + j0 + c] = x[c];
        }
      }
      __syncthreads();
      // ---- A3: trailing update on MFMA: blocks (bi, bk), j < bk <= bi < 8
      {
        const int rem = NSB - 1 - j;
        const int nblk = rem * (rem + 1) / 2;
        for (int b = wave; b < nblk; b += THREADS / 64) {
          int bi = 0, acc_cnt = 0;
          while (acc_cnt + bi + 1 <= b) { acc_cnt += bi + 1; ++bi; }
          const int bk = b - acc_cnt;
          const int i0 = (j + 1 + bi) * SB, k0 = (j + 1 + bk) * SB;
          d4 acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = S[(i0 + lk + 4 * r) * LD + k0 + lr];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double af = -S[(i0 + lr) * LD + j0 + 4 * s + lk];   // -L_ij[row][k]
            const double bf = S[(k0 + lr) * LD + j0 + 4 * s + lk];    // L_kj[col][k] = (L_kj^T)[k][col]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) S[(i0 + lk + 4 * r) * LD + k0 + lr] = acc[r];
        }
      }
      __syncthreads();
    }
    // factor back to global memory (lower triangle only)
    for (int idx = t; idx < NB * NB; idx += THREADS) {
      const int i = idx >> 7, j = idx & (NB - 1);
      if (i < jb && j <= i) A[(long)i * lda + j] = S[i * LD + j];
    }
  } else {
    if (t < NB) dg[t] = 1.0 / S[t * LD + t];
    __syncthreads();
  }

  // ---- phase B: T_ww = L_ww^-1 for the 8 diagonal 16 x 16 blocks, wave w, column (lane & 15) per lane
  {
    const int w0 = wave * SB;
    double tc[SB];
    const int c = lane & 15;
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) s = fma(S[(w0 + i) * LD + w0 + k], tc[k], s);
      const double ri = dg[w0 + i];
      tc[i] = (c == i) ? ri : -ri * s;   // lanes with c > i get exactly 0 (all their t_k are 0)
    }
    if (lane < SB) {
#pragma unroll
      for (int i = 0; i < SB; ++i) Td[(wave * SB + i) * TD_LD + c] = tc[i];
    }
  }
  __syncthreads();

  // ---- phase C: off-diagonal blocks of T by block rows;  T_ij = -T_ii * sum_{k=j}^{i-1} L_ik T_kj.
  // T_kj (k > j) is stored transposed in the upper triangle: T[r][c] at S[c][r].
  for (int bi = 1; bi < NSB; ++bi) {
    const int i0 = bi * SB;
    for (int bj = wave; bj < bi; bj += THREADS / 64) {
      const int c0 = bj * SB;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      // k = bj term: B operand is the diagonal inverse block T_jj (from Td)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double af = S[(i0 + lr) * LD + c0 + 4 * s + lk];                 // L_ij[row][k]
        const double bf = Td[(bj * SB + 4 * s + lk) * TD_LD + lr];             // T_jj[k][col]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
      }
      for (int bk = bj + 1; bk < bi; ++bk) {
        const int k0 = bk * SB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double af = S[(i0 + lr) * LD + k0 + 4 * s + lk];               // L_ik[row][k]
          const double bf = S[(c0 + lr) * LD + k0 + 4 * s + lk];               // T_kj[k][col] = S[c0+col][k0+k]
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
        }
      }
      // multiply by -T_ii: acc register r is exactly the B fragment of k-step r
      d4 res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double af = -Td[(bi * SB + lr) * TD_LD + 4 * s + lk];            // -T_ii[row][k]
        res = __builtin_amdgcn_mfma_f64_16x16x4f64(af, acc[s], res, 0, 0, 0);
      }
      // store T_ij transposed into the upper triangle: T_ij[row][col] -> S[c0 + col][i0 + row]
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(c0 + lr) * LD + i0 + lk + 4 * r] = res[r];
    }
    __syncthreads();
  }

  // ---- phase D: inverse to global memory (row-major NB x NB, zeros above the diagonal)
  for (int idx = t; idx < NB * NB; idx += THREADS) {
    const int i = idx >> 7, c = idx & (NB - 1);
    double v = 0.0;
    if ((i >> 4) == (c >> 4)) v = Td[i * TD_LD + (c & 15)];   // diagonal block (zeros above its diagonal)
    else if (c < i) v = S[c * LD + i];
    dinv[idx] = v;
  }
}

int launch(double* A, long lda, int n_total, int nblocks, double* dinv, int* info_dev, int offset,
           int do_factor, hipStream_t st) {
  static bool attr_done = false;
  const size_t lds = sizeof(double) * (NB * LD + NB + NSB * SB * TD_LD);
  if (!attr_done) {
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_inv_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  {
    ProfScope ps(PK_POTF2, st, (double)nblocks);
    hipLaunchKernelGGL(potf2_inv_kernel, dim3(nblocks), dim3(THREADS), lds, st, A, lda, n_total, dinv, info_dev,
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
