// Diagonal-block kernel of the blocked Cholesky: factor one NB x NB (128 x 128) block and produce
// the inverse of its triangular factor, by ONE workgroup of 8 waves, in 89 KB of LDS.
//
// Why 89 KB: this kernel sits on the critical path of the look-ahead factorisation and runs while the
// trailing-update GEMM fills the chip with two 64 KB workgroups per CU.  A 149 KB version (full
// 128 x 130 image) had to wait for BOTH GEMM workgroups of one CU to retire together (measured: ~600 us
// queueing for a 57 us kernel); 89 KB fits next to ONE resident GEMM workgroup (160 - 64 = 96 KB free).
//
// The block is handled as 8 x 8 sub-blocks of 16 x 16 (the v_mfma_f64_16x16x4_f64 tile).  Only the
// 36 sub-blocks on/below the diagonal are stored, packed, each as an unpadded 16 x 16 image whose bank
// conflicts are removed by an XOR swizzle of the column index (col ^ 2*((row>>1)&7)).
//   phase A  for each of the 8 block columns j
//     A1  wave 0 factors the 16 x 16 diagonal block in registers (one row per lane, pivots and column
//         entries broadcast with v_readlane; 1/sqrt by v_rsq_f64 + Newton, no division)
//     A2  panel rows below: X L_jj^T = A_panel by forward substitution, one row per thread
//     A3  trailing update A_ik -= L_ij L_kj^T on the MFMA pipe, one 16 x 16 block per wave at a time
//   phase B  inverses of the 8 diagonal 16 x 16 factors, one per wave, in registers -> Td (LDS)
//   phase C  T = L^-1 by block forward substitution, wave w owns block COLUMN w of T and keeps it in
//            registers: T_iw = -T_ii sum_k L_ik T_kw.  The MFMA result layout (row = (lane>>4) + 4 r) IS
//            the B-operand layout of the next MFMA, so neither the running sum nor T_kw ever touch LDS.
//   phase D  L -> global (lower triangle); T -> dinv (NB x NB row-major, zero above the diagonal)
#include "common.h"

namespace gpmp {
namespace {

constexpr int SB = 16;           // sub-block edge
constexpr int NSB = NB / SB;     // 8
constexpr int NPACK = NSB * (NSB + 1) / 2;   // 36 stored sub-blocks
constexpr int THREADS = 512;

// packed offset of element (r, c) of the 128 x 128 block; requires (r >> 4) >= (c >> 4)
__device__ __forceinline__ int pidx(int r, int c) {
  const int bi = r >> 4, bk = c >> 4, rr = r & 15, cc = c & 15;
  return ((bi * (bi + 1) / 2 + bk) << 8) + rr * 16 + (cc ^ (((rr >> 1) & 7) << 1));
}
// offset inside one 16 x 16 swizzled image
__device__ __forceinline__ int sidx(int rr, int cc) { return rr * 16 + (cc ^ (((rr >> 1) & 7) << 1)); }

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

#ifdef GPMP_POTF2_TRACE   // tools/potf2_probe.hip
__device__ long long g_potf2_trace[64];
#define PF_MARK(slot) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_potf2_trace[slot] = (long long)wall_clock64(); } while (0)
#else
#define PF_MARK(slot) do { } while (0)
#endif

__global__ void __launch_bounds__(THREADS) potf2_inv_kernel(double* __restrict__ A, long lda, int n_total,
                                                            double* __restrict__ dinv, int* info, int offset,
                                                            int do_factor, long prob_stride_a, long prob_stride_dinv) {
  // batched over blockIdx.x: block b works on the diagonal block starting at row/col b * NB;
  // batched over blockIdx.y: independent matrices (problems) prob_stride_a / prob_stride_dinv elements apart, one info word each
  A += (long)blockIdx.y * prob_stride_a;
  dinv += (long)blockIdx.y * prob_stride_dinv;
  if (info != nullptr) info += blockIdx.y;
  A += (long)blockIdx.x * NB * (lda + 1);
  dinv += (long)blockIdx.x * NB * NB;
  offset += blockIdx.x * NB;
  const int jb = (n_total - (int)blockIdx.x * NB) < NB ? (n_total - (int)blockIdx.x * NB) : NB;

  extern __shared__ __attribute__((aligned(16))) double S[];   // [NPACK][16][16] packed lower block triangle
  double* Td = S + NPACK * 256;                                // [NSB][16][16] diagonal inverse blocks
  double* dg = Td + NSB * 256;                                 // [NB] 1 / L_ii
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lk = lane >> 4;

  // ---- load: lower triangle of the jb x jb block, identity padding, zeros elsewhere
  PF_MARK(0);
  // (all 32 loads of a thread are issued before the first LDS store: one exposed latency instead of 32 -- the
  //  rolled loop took 16 us of the kernel's 65; phase timeline: tools/potf2_probe.hip)
  {
    double v[NB * NB / THREADS];
#pragma unroll
    for (int q = 0; q < NB * NB / THREADS; ++q) {
      const int idx = t + q * THREADS;
      const int i = idx >> 7, j = idx & (NB - 1);
      v[q] = (i == j) ? 1.0 : 0.0;
      if (i < jb && j <= i) v[q] = A[(long)i * lda + j];
    }
#pragma unroll
    for (int q = 0; q < NB * NB / THREADS; ++q) {
      const int idx = t + q * THREADS;
      const int i = idx >> 7, j = idx & (NB - 1);
      if ((i >> 4) >= (j >> 4)) S[pidx(i, j)] = v[q];   // sub-blocks above the diagonal are not stored
    }
  }
  __syncthreads();

  PF_MARK(1);
  if (do_factor) {
    // A3(j): trailing update on MFMA, A_ik -= L_ij L_kj^T for the sub-blocks j < gk <= gi < 8, enumerated b = 0, 1, ...
    // with b = 0 the NEXT diagonal block (j+1, j+1).  Wave 0 does b = 0 right after A2(j) and goes on to factor that block
    // (A1(j+1)); the other seven waves do the rest meanwhile, so the update hides behind the register factorisation.
    auto a3_blocks = [&](int j, int b_first, int b_step, int b_end_excl) {
      const int rem = NSB - 1 - j;
      const int nblk = rem * (rem + 1) / 2;
      const int bend = b_end_excl < nblk ? b_end_excl : nblk;
      for (int b = b_first; b < bend; b += b_step) {
        int bi = 0, acc_cnt = 0;
        while (acc_cnt + bi + 1 <= b) { acc_cnt += bi + 1; ++bi; }
        const int bk = b - acc_cnt;
        const int gi = j + 1 + bi, gk = j + 1 + bk;                       // global sub-block indices, gi >= gk
        double* Cik = S + ((gi * (gi + 1) / 2 + gk) << 8);
        const double* Lij = S + ((gi * (gi + 1) / 2 + j) << 8);
        const double* Lkj = S + ((gk * (gk + 1) / 2 + j) << 8);
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Cik[sidx(lk + 4 * r, lr)];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double af = -Lij[sidx(lr, 4 * s + lk)];   // -L_ij[row][k]
          const double bf = Lkj[sidx(lr, 4 * s + lk)];    // L_kj[col][k] = (L_kj^T)[k][col]
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Cik[sidx(lk + 4 * r, lr)] = acc[r];
      }
    };
    for (int j = 0; j < NSB; ++j) {
      const int j0 = j * SB;
      double* Djj = S + ((j * (j + 1) / 2 + j) << 8);   // diagonal sub-block image
      // ---- A1: diagonal 16 x 16 block, wave 0, in the MFMA accumulator layout: lane (g, i) = (lane >> 4, lane & 15) holds
      // C[g + 4 r][i], r = 0..3 -- the full symmetric block, read from its lower triangle.  Blocked by 4 columns: by symmetry
      // register s of lane group g IS column 4 s + g (entry of row i on lane i), so
      //   * the four columns of a step are factored in place: pivot by v_readlane, scaling on one lane group, the update of
      //     the (at most three) later columns of the step with the scaled column broadcast across the lane groups by
      //     v_permlane16_swap / v_permlane32_swap (gfx950; 4 VALU instructions per double, no LDS round trip);
      //   * the rank-4 update of the whole block by these columns is ONE v_mfma_f64_16x16x4_f64 whose A and B operands are
      //     that same register (A[row i][k = g] = L[i][4 s + g] = B[k = g][col i]), zeroed above the diagonal so that rows of
      //     earlier steps (which by now hold L) receive exactly 0.
      // The row-per-lane version spent 2.7 us per 16 x 16 block, 21 of the kernel's 46 us, on ~75 instructions per column
      // (two v_readlane per multiplier of the rank-one updates); this one issues ~35 and four MFMAs per block.
      if (wave == 0) {
        const int g = lane >> 4, i16 = lane & 15;
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = g + 4 * r;
          acc[r] = Djj[row >= i16 ? sidx(row, i16) : sidx(i16, row)];
        }
        unsigned badmask = 0;                  // bit c: pivot c was not positive (also NaN); uniform, off the dependency chain
        double ykeep = 0.0;                    // 1 / L_cc of the column this lane owns on its diagonal row (lane (g, 4 s + g))
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          // the four columns of the step on EVERY lane (x[t] = entry of row i16 in column 4 s4 + t): two permlane rounds
          // per 32-bit half, once per step; from here on the step is lane-local except for uniform v_readlane broadcasts
          double x[4];
          {
            const double v = acc[s4];
            const int lo = __double2loint(v), hi = __double2hiint(v);
            const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);   // [0]: v0 v0 v2 v2   [1]: v1 v1 v3 v3
            const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            const auto la = __builtin_amdgcn_permlane32_swap(l16[0], l16[0], false, false);   // [0]: v0 x 4   [1]: v2 x 4
            const auto ha = __builtin_amdgcn_permlane32_swap(h16[0], h16[0], false, false);
            const auto lb = __builtin_amdgcn_permlane32_swap(l16[1], l16[1], false, false);   // [0]: v1 x 4   [1]: v3 x 4
            const auto hb = __builtin_amdgcn_permlane32_swap(h16[1], h16[1], false, false);
            x[0] = __hiloint2double(ha[0], la[0]);
            x[1] = __hiloint2double(hb[0], lb[0]);
            x[2] = __hiloint2double(ha[1], la[1]);
            x[3] = __hiloint2double(hb[1], lb[1]);
          }
          double sq[4], yy[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int c = 4 * s4 + t;
            const double d = bcast_lane(x[t], c);
            badmask |= (!(d > 0.0)) ? (1u << c) : 0u;      // a failed pivot lets NaN / inf run through (the factor is unspecified then)
            // 1 / sqrt(d): hardware estimate + one third-order step y (1 + e/2 + 3 e^2 / 8), e = 1 - d y^2
            double y = __builtin_amdgcn_rsq(d);
            const double e = fma(-(d * y), y, 1.0);
            y = fma(y * e, fma(0.375, e, 0.5), y);
            x[t] *= y;
#pragma unroll
            for (int tp = t + 1; tp < 4; ++tp) x[tp] = fma(-x[t], bcast_lane(x[t], 4 * s4 + tp), x[tp]);
            // diagonal entry sqrt(d) = d y with one correction (< 1 ulp): uniform, off the chain
            double sd = d * y;
            sd = fma(0.5 * y, fma(-sd, sd, d), sd);
            sq[t] = sd;
            yy[t] = y;
          }
          // back to one column per lane group: group g takes column 4 s4 + g; zero above the diagonal
          const double xg = g == 0 ? x[0] : g == 1 ? x[1] : g == 2 ? x[2] : x[3];
          const double sg = g == 0 ? sq[0] : g == 1 ? sq[1] : g == 2 ? sq[2] : sq[3];
          const double yg = g == 0 ? yy[0] : g == 1 ? yy[1] : g == 2 ? yy[2] : yy[3];
          const int cdiag = 4 * s4 + g;
          const double xz = i16 > cdiag ? xg : (i16 == cdiag ? sg : 0.0);
          ykeep = i16 == cdiag ? yg : ykeep;
          if (s4 < 3) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xz, xz, acc, 0, 0, 0);
          acc[s4] = xz;
        }
        if (badmask != 0 && lane == 0) {
          const int badcol = __builtin_ctz(badmask);
          if (j0 + badcol < jb) atomicCAS(info, 0, offset + j0 + badcol + 1);
        }
        if (((i16 - g) & 3) == 0 && i16 >= g) dg[j0 + i16] = ykeep;      // lane (g, 4 s + g) owns 1 / L_cc of column c = 4 s + g
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) Djj[sidx(i16, 4 * s4 + g)] = acc[s4];
      } else if (j >= 1) {
        a3_blocks(j - 1, wave, THREADS / 64 - 1, 1 << 30);   // b = 1, 2, ... of A3(j-1) over waves 1..7 (b = 0 was wave 0's)
      }
      __syncthreads();
      PF_MARK(8 + 3 * j);
      // ---- A2: panel rows below the diagonal block: x L_jj^T = a, one row per thread
      {
        const int row = j0 + SB + t;
        if (row < NB) {
          double x[SB];
#pragma unroll
          for (int c = 0; c < SB; ++c) x[c] = S[pidx(row, j0 + c)];
#pragma unroll
          for (int c = 0; c < SB; ++c) {
            double s = x[c];
#pragma unroll
            for (int k = 0; k < c; ++k) s = fma(-x[k], Djj[sidx(c, k)], s);
            x[c] = s * dg[j0 + c];
          }
#pragma unroll
          for (int c = 0; c < SB; ++c) S[pidx(row, j0 + c)] = x[c];
        }
      }
      __syncthreads();
      PF_MARK(9 + 3 * j);
      // ---- A3(j), first block only: wave 0 updates the next diagonal block and goes straight on to factor it
      if (wave == 0) a3_blocks(j, 0, 1, 1);
      PF_MARK(10 + 3 * j);
    }
    PF_MARK(2);
    // factor back to global memory (lower triangle only)
    for (int idx = t; idx < NB * NB; idx += THREADS) {
      const int i = idx >> 7, j = idx & (NB - 1);
      if (i < jb && j <= i) A[(long)i * lda + j] = S[pidx(i, j)];
    }
  } else {
    if (t < NB) dg[t] = 1.0 / S[pidx(t, t)];
    __syncthreads();
  }

  PF_MARK(3);
  // ---- phase B: T_ww = L_ww^-1 for the 8 diagonal 16 x 16 blocks, wave w, column (lane & 15) per lane
  {
    const double* Dww = S + ((wave * (wave + 1) / 2 + wave) << 8);
    double tc[SB];
    const int c = lane & 15;
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) s = fma(Dww[sidx(i, k)], tc[k], s);
      const double ri = dg[wave * SB + i];
      tc[i] = (c == i) ? ri : -ri * s;   // lanes with c > i get exactly 0 (all their t_k are 0)
    }
    if (lane < SB) {
#pragma unroll
      for (int i = 0; i < SB; ++i) Td[(wave << 8) + sidx(i, c)] = tc[i];
    }
  }
  __syncthreads();

  PF_MARK(4);
  // ---- phase C + D: wave w computes block column w of T = L^-1, keeps it in registers, streams it out.
  {
    const int bj = wave;
    d4 tcol[NSB];                 // tcol[i] = T_{i,bj} (MFMA C layout), i > bj
#pragma unroll
    for (int bi = 1; bi < NSB; ++bi) {
      if (bi > bj) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        // k = bj term: B operand is the diagonal inverse block T_jj (from Td)
        {
          const double* Lij = S + ((bi * (bi + 1) / 2 + bj) << 8);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double af = Lij[sidx(lr, 4 * s + lk)];                    // L_ij[row][k]
            const double bf = Td[(bj << 8) + sidx(4 * s + lk, lr)];         // T_jj[k][col]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int bk = 1; bk < NSB; ++bk) {
          if (bk > bj && bk < bi) {
            const double* Lik = S + ((bi * (bi + 1) / 2 + bk) << 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const double af = Lik[sidx(lr, 4 * s + lk)];                  // L_ik[row][k]
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, tcol[bk][s], acc, 0, 0, 0);   // B = T_kj registers
            }
          }
        }
        // multiply by -T_ii: acc register r is exactly the B fragment of k-step r
        d4 res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double af = -Td[(bi << 8) + sidx(lr, 4 * s + lk)];          // -T_ii[row][k]
          res = __builtin_amdgcn_mfma_f64_16x16x4f64(af, acc[s], res, 0, 0, 0);
        }
        tcol[bi] = res;
      }
    }
    PF_MARK(5);
    // phase D: block column bj of dinv (row-major NB x NB): zeros above, Td on the diagonal, tcol below
#pragma unroll
    for (int bi = 0; bi < NSB; ++bi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = bi * SB + lk + 4 * r, col = bj * SB + lr;
        double v = 0.0;
        if (bi == bj) v = Td[(bj << 8) + sidx(lk + 4 * r, lr)];
        else if (bi > bj) v = tcol[bi][r];
        dinv[row * NB + col] = v;
      }
    }
  }
  PF_MARK(6);
}

int launch(double* A, long lda, int n_total, int nblocks, double* dinv, int* info_dev, int offset,
           int do_factor, hipStream_t st, int nprob = 1, long prob_stride_a = 0, long prob_stride_dinv = 0) {
  static bool attr_done = false;
  const size_t lds = sizeof(double) * (NPACK * 256 + NSB * 256 + NB);   // 91,136 B
  if (!attr_done) {
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_inv_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  {
    ProfScope ps(PK_POTF2, st, (double)nblocks * nprob);
    hipLaunchKernelGGL(potf2_inv_kernel, dim3(nblocks, nprob), dim3(THREADS), lds, st, A, lda, n_total, dinv, info_dev,
                       offset, do_factor, prob_stride_a, prob_stride_dinv);
  }
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

int launch_potf2_inv(double* A, long lda, int jb, double* dinv, int* info_dev, int offset, hipStream_t st) {
  return launch(A, lda, jb, 1, dinv, info_dev, offset, 1, st);
}
// the same diagonal block of `nprob` independent matrices (batched small problems, drivers_batch.hip)
int launch_potf2_inv_batch(double* A, long lda, int jb, double* dinv, int* info_dev, int offset, int nprob, long stride_a,
                           long stride_dinv, hipStream_t st) {
  return launch(A, lda, jb, 1, dinv, info_dev, offset, 1, st, nprob, stride_a, stride_dinv);
}
int launch_trtri_blocks(const double* L, long ldl, int n, double* dinv, hipStream_t st) {
  if (n <= 0) return 0;
  return launch(const_cast<double*>(L), ldl, n, (n + NB - 1) / NB, dinv, nullptr, 0, 0, st);
}

}  // namespace gpmp
