// Many small GP problems at once: B independent criteria (zero-mean NLL, or REML with a mean design of q <= 16 columns)
// with their analytic gradients, every step ONE launch over all problems.  This is the throughput caller of SURVEY 8(f).4:
//   * mini-batch selection criteria -- the weighted mean over the batches of a loader of the per-batch NLL / REML and its
//     gradient (gpmp/num/torch_backend.py:607-718, gpmp/dataloader.py:484-513): B batches, ONE parameter vector;
//   * posterior samplers / multi-chain optimisers -- log_prob = -criterion at many parameter vectors on the same data
//     (gpmp/mcmc/param_posterior.py:229-278): ONE data set, B parameter vectors.
// Per problem the reference runs cdist -> Matern -> cholesky -> 2 solve_triangular (-> autograd backward); on the GPU
// a problem of n <= 4096 points is too small to fill the machine (0.2 - 3 ms each, latency-bound), so problems
// are stacked: one padded n_max x n_max slot each (identity padding: log-det and quadratic form unchanged), and
//   Gram build          B small launches of the fused distance + Matern kernel (lower tiles, own parameters / own points)
//   Cholesky            the blocked right-looking factorisation with every kernel batched over the problems
//                       (diagonal-block kernel: blockIdx.y = problem; fp64 MFMA GEMM: blockIdx.z = problem)
//   L^-1 [z, P]         one workgroup per problem, 128-row blocks, diagonal-block inverses (no MFMA: n^2 work)
//   log-det, W^T W, the q x q algebra, the value              one workgroup per problem
//   gradient            L^-T W batched the same way; L^-1 by doubling and T^T T as batched GEMMs; the trace kernel per problem
#include "common.h"
#include <cfloat>
#include <cmath>
#include <vector>

namespace gpmp {
namespace {

constexpr int BQ = 16;             // mean-design columns supported in the batched path (round 2: 3, then 7; round 5: 16 = what the
                                   // single-problem drivers' mean-space workgroup carries: a linear mean in d <= 15)
constexpr int BR = BQ + 1;         // right-hand sides per problem: [z, P]; the kernels exist for 4 (q <= 3), 8 (q <= 7) and 17 of them
constexpr int BSM = 288;           // doubles of per-problem scalars: [0] logdet K, [1] quad, [2] ln|S|, [3] ln|PtP|, [4] fail,
                                   // [8 + a] c = S^-1 b, [24 + BQ a + b] S^-1
inline long pad16(long v) { return (v + 15) / 16 * 16; }

// identity in the padding of one slot: rows / columns n .. nmax - 1 (lower triangle + diagonal are what the factorisation reads)
__global__ void pad_identity_kernel(double* __restrict__ K, long ld, int n, int nmax) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = n + blockIdx.y;
  if (i >= nmax || j >= nmax || j > i) return;
  K[(long)i * ld + j] = (i == j) ? 1.0 : 0.0;
}
// the same for every slot of a batch in one launch (blockIdx.z = problem, sizes from the device array)
__global__ void pad_identity_batch_kernel(double* __restrict__ Kall, long ld, long stride, const int* __restrict__ ns, int nmax) {
  const int n = ns[blockIdx.z];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (i < n || i >= nmax || j >= nmax || j > i) return;
  Kall[(long)blockIdx.z * stride + (long)i * ld + j] = (i == j) ? 1.0 : 0.0;
}

// Y_b[i] = [z_b[i], P_b[i, :]] for i < n_b, zeros in the padding
__global__ void batch_pack_kernel(const double* __restrict__ z, long sz, const double* __restrict__ P, long ldp, long sp, int q,
                                  const int* __restrict__ ns, int nmax, double* __restrict__ Y, long ldy, long sy) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nmax) return;
  const bool in = i < ns[b];
  double* y = Y + (long)b * sy + (long)i * ldy;
  y[0] = in ? z[(long)b * sz + i] : 0.0;
  for (int a = 0; a < q; ++a) y[1 + a] = in ? P[(long)b * sp + (long)i * ldp + a] : 0.0;
}

// op(L_b)^-1 Y_b in place, one workgroup (256 threads) per problem; blocks of NB rows, diagonal-block inverses from dinv.
// Thread t owns row (t & 127) of a block and half (t >> 7) of the 128-wide inner range; halves meet in LDS.
template <bool TRANS, int BRT>
__global__ void __launch_bounds__(256) batch_trsv_kernel(const double* __restrict__ Lall, long ldl, long sl, const double* __restrict__ dall,
                                                         long sd, double* __restrict__ Yall, long ldy, long sy, int nmax, int r) {
  __shared__ double xs[NB][BRT];
  __shared__ double part[NB][BRT];
  const double* L = Lall + (long)blockIdx.x * sl;
  const double* dinv = dall + (long)blockIdx.x * sd;
  double* Y = Yall + (long)blockIdx.x * sy;
  const int t = threadIdx.x, i = t & 127, half = t >> 7;
  const int nblk = (nmax + NB - 1) / NB;
  for (int q = 0; q < nblk; ++q) {
    const int k = TRANS ? nblk - 1 - q : q;
    const int k0 = k * NB;
    const int kb = (nmax - k0) < NB ? (nmax - k0) : NB;
    // right-hand side rows of block k (already updated by the blocks solved before)
    for (int idx = t; idx < NB * BRT; idx += 256) {
      const int l = idx / BRT, c = idx % BRT;
      xs[l][c] = (l < kb && c < r) ? Y[(long)(k0 + l) * ldy + c] : 0.0;
    }
    __syncthreads();
    // x_k = op(inv(L_kk)) b_k   (dinv block: NB x NB, identity-padded)
    double acc[BRT];
#pragma unroll
    for (int c = 0; c < BRT; ++c) acc[c] = 0.0;
    const double* D = dinv + (size_t)k * NB * NB;
    for (int l = 0; l < 64; ++l) {
      const int ll = half * 64 + l;
      const double m = TRANS ? D[ll * NB + i] : D[i * NB + ll];
#pragma unroll
      for (int c = 0; c < BRT; ++c) acc[c] = fma(m, xs[ll][c], acc[c]);
    }
    if (half == 1) {
#pragma unroll
      for (int c = 0; c < BRT; ++c) part[i][c] = acc[c];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int c = 0; c < BRT; ++c) xs[i][c] = acc[c] + part[i][c];   // x_k replaces b_k in LDS (all reads of b_k are done)
    }
    __syncthreads();
    if (half == 0 && i < kb) {
#pragma unroll
      for (int c = 0; c < BRT; ++c)
        if (c < r) Y[(long)(k0 + i) * ldy + c] = xs[i][c];
    }
    // blocks still to be solved: b_j -= op(L)[j, k] x_k
    for (int qq = q + 1; qq < nblk; ++qq) {
      const int j = TRANS ? nblk - 1 - qq : qq;
      const int j0 = j * NB;
      const int jb = (nmax - j0) < NB ? (nmax - j0) : NB;
#pragma unroll
      for (int c = 0; c < BRT; ++c) acc[c] = 0.0;
      if (i < jb) {
        const int lmax = (kb - half * 64) < 64 ? (kb - half * 64) : 64;
        for (int l = 0; l < lmax; ++l) {
          const int ll = half * 64 + l;
          // forward: L[j0 + i][k0 + ll]; transposed: (L^T)[j0 + i][k0 + ll] = L[k0 + ll][j0 + i]
          const double m = TRANS ? L[(long)(k0 + ll) * ldl + j0 + i] : L[(long)(j0 + i) * ldl + k0 + ll];
#pragma unroll
          for (int c = 0; c < BRT; ++c) acc[c] = fma(m, xs[ll][c], acc[c]);
        }
      }
      if (half == 1) {
#pragma unroll
        for (int c = 0; c < BRT; ++c) part[i][c] = acc[c];
      }
      __syncthreads();
      if (half == 0 && i < jb) {
#pragma unroll
        for (int c = 0; c < BRT; ++c)
          if (c < r) Y[(long)(j0 + i) * ldy + c] -= acc[c] + part[i][c];
      }
      __syncthreads();
    }
    __syncthreads();
  }
}

// One workgroup per problem: log-det of L, G = W^T W, P^T P, then the q x q algebra and the criterion value.
//   q = 0:  value = 1/2 (n ln 2 pi + ln|K| + w^T w)
//   q > 0:  value = 1/2 ((n - q) ln 2 pi + ln|K| + ln|S| - ln|P^T P| + w^T w - b^T S^-1 b),  S = Wp^T Wp, b = Wp^T w
// small (BSM doubles per problem) keeps S^-1 and c = S^-1 b for the gradient; info[b] += n + pivot on a singular mean design.
// The sums: [0] log L_ii, then the upper triangles of W^T W ((QT + 1) x (QT + 1)) and of P^T P (QT x QT), QT = the mean-design
// columns the instantiation carries.  `batch_value_finish` (one thread) turns their totals into the value.
template <int QT>
__device__ void batch_value_finish(const double* tot, int q, int n, int b, double* __restrict__ small, int* __restrict__ info,
                                   double* __restrict__ values) {
  constexpr int RT = QT + 1;
  double G[RT][RT], PtP[QT][QT];
  {
    int k = 1;
    for (int c = 0; c < RT; ++c)
      for (int e = c; e < RT; ++e) { G[c][e] = G[e][c] = tot[k]; ++k; }
    for (int c = 0; c < QT; ++c)
      for (int e = c; e < QT; ++e) { PtP[c][e] = PtP[e][c] = tot[k]; ++k; }
  }
  const double logdetK = 2.0 * tot[0];
  double* sm = small + (long)b * BSM;
  int fail = 0;
  double ldS = 0.0, ldP = 0.0, quad = G[0][0];
  if (q > 0) {
    // Cholesky of a q x q matrix (q <= QT) with the relative pivot test of the single-problem driver
    auto chol = [&](double (*M)[QT], double& logdet) {
      double d0[QT];
      for (int k = 0; k < q; ++k) d0[k] = M[k][k];
      logdet = 0.0;
      for (int k = 0; k < q; ++k) {
        double dkk = M[k][k];
        for (int l = 0; l < k; ++l) dkk -= M[k][l] * M[k][l];
        if (!(dkk > (double)q * 2.220446049250313e-16 * d0[k])) { if (fail == 0) fail = k + 1; dkk = 1.0; }
        const double rkk = sqrt(dkk);
        M[k][k] = rkk;
        logdet += 2.0 * log(rkk);
        for (int i = k + 1; i < q; ++i) {
          double v = M[i][k];
          for (int l = 0; l < k; ++l) v -= M[i][l] * M[k][l];
          M[i][k] = v / rkk;
        }
      }
    };
    double S[QT][QT], Pm[QT][QT], bvec[QT];
    for (int c = 0; c < q; ++c) {
      bvec[c] = G[1 + c][0];
      for (int e = 0; e < q; ++e) { S[c][e] = G[1 + c][1 + e]; Pm[c][e] = PtP[c][e]; }
    }
    chol(Pm, ldP);
    chol(S, ldS);
    // R^-1 (lower), S^-1 = R^-T R^-1, c = S^-1 b
    double Ri[QT][QT];
    for (int j = 0; j < q; ++j)
      for (int i = 0; i < q; ++i) {
        double s = (i == j) ? 1.0 : 0.0;
        for (int l = j; l < i; ++l) s -= S[i][l] * Ri[l][j];
        Ri[i][j] = (i < j) ? 0.0 : s / S[i][i];
      }
    double bsb = 0.0;
    for (int i = 0; i < q; ++i) {
      double ci = 0.0;
      for (int j = 0; j < q; ++j) {
        double s = 0.0;
        for (int l = (i > j ? i : j); l < q; ++l) s += Ri[l][i] * Ri[l][j];
        sm[24 + BQ * i + j] = s;
        ci += s * bvec[j];
      }
      sm[8 + i] = ci;
      bsb += bvec[i] * ci;
    }
    quad -= bsb;
  }
  sm[0] = logdetK; sm[1] = quad; sm[2] = ldS; sm[3] = ldP; sm[4] = (double)fail;
  if (fail != 0) atomicCAS(info + b, 0, n + fail);
  double v = 0.5 * ((double)(n - q) * 1.8378770664093454835606594728112 + logdetK + ldS - ldP + quad);
  if (info[b] != 0 || !(v == v) || v > DBL_MAX || v < -DBL_MAX) v = __builtin_huge_val();
  values[b] = v;
}

// QT = 3 or 7: every thread keeps all sums in registers (17 / 65 of them) over its rows; reduced by wave shuffles, then across
// the four waves through LDS (a 256 x 65 LDS image would not fit)
template <int QT>
__global__ void __launch_bounds__(256) batch_value_kernel(const double* __restrict__ Lall, long ldl, long sl, const double* __restrict__ Wall,
                                                          long ldw, long sw, const double* __restrict__ P, long ldp, long sp, int q,
                                                          const int* __restrict__ ns, int nmax, double* __restrict__ small,
                                                          int* __restrict__ info, double* __restrict__ values) {
  constexpr int RT = QT + 1;
  constexpr int NS = 1 + RT * (RT + 1) / 2 + QT * (QT + 1) / 2;
  __shared__ double red[4][NS];
  const int b = blockIdx.x, t = threadIdx.x;
  const double* L = Lall + (long)b * sl;
  const double* W = Wall + (long)b * sw;
  const int n = ns[b];
  double acc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) acc[k] = 0.0;
  for (int i = t; i < nmax; i += 256) {
    acc[0] += log(L[(long)i * ldl + i]);
    double w[RT], pr[QT];
#pragma unroll
    for (int c = 0; c < RT; ++c) w[c] = (c <= q) ? W[(long)i * ldw + c] : 0.0;
#pragma unroll
    for (int c = 0; c < QT; ++c) pr[c] = (c < q && i < n) ? P[(long)b * sp + (long)i * ldp + c] : 0.0;
    int k = 1;
#pragma unroll
    for (int c = 0; c < RT; ++c)
#pragma unroll
      for (int e = c; e < RT; ++e) acc[k++] += w[c] * w[e];
#pragma unroll
    for (int c = 0; c < QT; ++c)
#pragma unroll
      for (int e = c; e < QT; ++e) acc[k++] += pr[c] * pr[e];
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double v = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((t & 63) == 0) red[t >> 6][k] = v;
  }
  __syncthreads();
  for (int k = t; k < NS; k += 256) red[0][k] += red[1][k] + red[2][k] + red[3][k];
  __syncthreads();
  if (t == 0) batch_value_finish<QT>(red[0], q, n, b, small, info, values);
}

// QT = 16 (7 < q <= 16, round 5): 290 sums do not fit a thread's registers, so the roles turn: rows go through LDS 64 at a time and
// every thread owns at most two (column, column) PAIRS of W^T W / P^T P over ALL rows; the log-diagonal sum rides on one wave.
__global__ void __launch_bounds__(256) batch_value_wide_kernel(const double* __restrict__ Lall, long ldl, long sl, const double* __restrict__ Wall,
                                                               long ldw, long sw, const double* __restrict__ P, long ldp, long sp, int q,
                                                               const int* __restrict__ ns, int nmax, double* __restrict__ small,
                                                               int* __restrict__ info, double* __restrict__ values) {
  constexpr int QT = BQ, RT = QT + 1, CH = 64;
  constexpr int NW = RT * (RT + 1) / 2, NP = QT * (QT + 1) / 2, NS = 1 + NW + NP;
  __shared__ double rows[CH][RT + QT + 1];            // [w_0 .. w_q | p_0 .. p_{q-1}] of 64 rows (odd stride: no bank pattern)
  __shared__ double tot[NS];
  const int b = blockIdx.x, t = threadIdx.x;
  const double* L = Lall + (long)b * sl;
  const double* W = Wall + (long)b * sw;
  const int n = ns[b];
  // the pairs of this thread: k = t and k = t + 256 in the order of the totals (W pairs first, then P pairs), as LDS columns
  int ca[2], cb[2];
  for (int s = 0; s < 2; ++s) {
    int k = t + 256 * s;
    ca[s] = cb[s] = -1;
    if (k < NW) {
      int c = 0;
      while (k >= RT - c) { k -= RT - c; ++c; }
      ca[s] = c; cb[s] = c + k;
    } else if (k < NW + NP) {
      k -= NW;
      int c = 0;
      while (k >= QT - c) { k -= QT - c; ++c; }
      ca[s] = RT + c; cb[s] = RT + c + k;
    }
  }
  double a0 = 0.0, a1 = 0.0, lg = 0.0;
  for (int i0 = 0; i0 < nmax; i0 += CH) {
    for (int idx = t; idx < CH * (RT + QT); idx += 256) {
      const int l = idx / (RT + QT), c = idx % (RT + QT), i = i0 + l;
      double v = 0.0;
      if (i < nmax) {
        if (c < RT) { if (c <= q) v = W[(long)i * ldw + c]; }
        else if (c - RT < q && i < n) v = P[(long)b * sp + (long)i * ldp + (c - RT)];
      }
      rows[l][c] = v;
    }
    if (t < CH && i0 + t < nmax) lg += log(L[(long)(i0 + t) * ldl + i0 + t]);
    __syncthreads();
    if (ca[0] >= 0) {
#pragma unroll 8
      for (int l = 0; l < CH; ++l) a0 = fma(rows[l][ca[0]], rows[l][cb[0]], a0);
    }
    if (ca[1] >= 0) {
#pragma unroll 8
      for (int l = 0; l < CH; ++l) a1 = fma(rows[l][ca[1]], rows[l][cb[1]], a1);
    }
    __syncthreads();
  }
  if (t < 64) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lg += __shfl_xor(lg, o);
    if (t == 0) tot[0] = lg;
  }
  if (ca[0] >= 0) tot[1 + t] = a0;
  if (ca[1] >= 0) tot[1 + t + 256] = a1;
  __syncthreads();
  if (t == 0) batch_value_finish<QT>(tot, q, n, b, small, info, values);
}

// rows of the low-rank part of the gradient trace: F[i] = [U_i S^-1, beta_i], G[i] = [U_i, beta_i], beta = alpha - U c
__global__ void batch_rows_kernel(const double* __restrict__ Xall, long ldx, long sx, const double* __restrict__ small, int q,
                                  const int* __restrict__ ns, double* __restrict__ Fall, double* __restrict__ Gall, long ldf, long sf) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns[b]) return;
  const double* sm = small + (long)b * BSM;
  const double* xr = Xall + (long)b * sx + (long)i * ldx;
  double* F = Fall + (long)b * sf + (long)i * ldf;
  double* G = Gall + (long)b * sf + (long)i * ldf;
  double beta = xr[0];
  for (int a = 0; a < q; ++a) beta -= xr[1 + a] * sm[8 + a];
  for (int a = 0; a < q; ++a) {
    double s = 0.0;
    for (int l = 0; l < q; ++l) s += xr[1 + l] * sm[24 + BQ * l + a];
    F[a] = s;
    G[a] = xr[1 + a];
  }
  F[q] = beta;
  G[q] = beta;
}

// g <- 1/2 g, zeros for a failed problem (same convention as gpmp_nll_grad)
__global__ void batch_grad_finalize_kernel(double* g, int ng, int B, const int* info) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ng * B) return;
  g[idx] = (info[idx / ng] != 0) ? 0.0 : 0.5 * g[idx];
}

__global__ void fill_int_kernel(int* a, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

struct BatchLayout {
  long ld, ldq;
  size_t K, dinv, T, Y, X, F, G, small, ns, gws, pp, total;
  size_t sK, sD, sY, sG;      // per-problem strides (elements)
};

BatchLayout batch_layout(int nmax, int d, int q, int B, bool grad) {
  BatchLayout l;
  l.ld = pad16(nmax);
  l.ldq = pad16(1 + q);
  l.sK = (size_t)nmax * l.ld;
  l.sD = (size_t)pad16((long)((nmax + NB - 1) / NB) * NB * NB);
  l.sY = (size_t)nmax * l.ldq;
  l.sG = (size_t)pad16((long)gpmp_grad_ws_elems(nmax, d));
  size_t o = 0;
  auto take = [&](size_t cnt) { size_t at = o; o += (size_t)pad16((long)cnt); return at; };
  l.K = take(l.sK * B);
  l.dinv = take(l.sD * B);
  l.T = grad ? take(l.sK * B) : 0;
  l.Y = take(l.sY * B);
  l.X = grad ? take(l.sY * B) : 0;
  l.F = grad ? take(l.sY * B) : 0;
  l.G = grad ? take(l.sY * B) : 0;
  l.small = take((size_t)BSM * B);
  l.ns = take((size_t)(B + 1) / 2 + 8);     // B ints
  l.gws = grad ? take(l.sG * B) : 0;
  l.pp = take((size_t)gram_param_block_elems() * B);      // per-problem parameter blocks (sampler pattern)
  l.total = o;
  return l;
}

// op(L_b)^-1 Y_b for every problem.  Two routes: ONE launch of the workgroup-per-problem kernel (its time hardly depends on B up to
// one workgroup per compute unit, but grows with nmax^2: 17 ms for both directions at nmax = 4096), or B launches of the
// single-problem one-launch sweep (trsv_few: 0.1 - 0.2 ms each, all compute units on ONE problem).  Measured crossover
// (profiles/r5/batch_n4096_*_kernel_stats.csv): the per-problem sweeps win while B < nmax / 100 -- the few-large-problems case.
template <bool TRANS>
int batch_solve(const double* K, long ldk, long sK, const double* dinv, long sD, double* Y, long ldy, long sY, int nmax, int r, int q,
                int B, hipStream_t st) {
  if (nmax >= 3 * NB && (long)B * 100 < nmax) {
    for (int b = 0; b < B; ++b)
      for (int c0 = 0; c0 < r; c0 += TRSV_FEW_MAX) {
        const int mc = r - c0 < TRSV_FEW_MAX ? r - c0 : TRSV_FEW_MAX;
        int rc = trsv_few(K + (long)b * sK, nmax, ldk, dinv + (long)b * sD, Y + (long)b * sY + c0, mc, ldy, TRANS ? 1 : 0, st);
        if (rc) return rc;
      }
    return 0;
  }
  if (q <= 3) hipLaunchKernelGGL((batch_trsv_kernel<TRANS, 4>), dim3(B), dim3(256), 0, st, K, ldk, sK, dinv, sD, Y, ldy, sY, nmax, r);
  else if (q <= 7) hipLaunchKernelGGL((batch_trsv_kernel<TRANS, 8>), dim3(B), dim3(256), 0, st, K, ldk, sK, dinv, sD, Y, ldy, sY, nmax, r);
  else hipLaunchKernelGGL((batch_trsv_kernel<TRANS, BR>), dim3(B), dim3(256), 0, st, K, ldk, sK, dinv, sD, Y, ldy, sY, nmax, r);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace
}  // namespace gpmp

using namespace gpmp;

extern "C" size_t gpmp_batch_ws_elems(int nmax, int d, int q, int B, int with_grad) {
  if (nmax <= 0 || nmax > GPMP_BATCH_MAX_N || B <= 0 || q < 0 || q > BQ || d < 1) return 0;
  return batch_layout(nmax, d, q, B, with_grad != 0).total;
}

extern "C" int gpmp_nll_grad_batch(const double* x, long stride_x, const double* z, long stride_z, const double* P, long ldp,
                                   long stride_p, int q, const int* n_host, int nmax, int d, int B, int p,
                                   const double* theta_host, int theta_stride, int noise, double* ws, double* values_dev,
                                   double* grads_dev, int* info_dev, gpmp_stream_t stream) {
  GPMP_ARG(x != nullptr, 1, "x is NULL");
  GPMP_ARG(z != nullptr, 3, "z is NULL");
  GPMP_ARG(q >= 0 && q <= BQ, 8, "q outside [0, 16] (the batched path carries at most 16 mean-design columns)");
  GPMP_ARG(q == 0 || (P != nullptr && ldp >= q), 5, "P is NULL or ldp < q");
  GPMP_ARG(nmax >= 1 && nmax <= GPMP_BATCH_MAX_N, 10, "nmax outside [1, GPMP_BATCH_MAX_N]");
  GPMP_ARG(d >= 1 && d <= GPMP_MAX_DIM, 11, "d outside [1, GPMP_MAX_DIM]");
  GPMP_ARG(B >= 1 && B <= 65535, 12, "B outside [1, 65535]");
  GPMP_ARG(theta_host != nullptr, 14, "theta is NULL");
  GPMP_ARG(ws != nullptr && values_dev != nullptr && info_dev != nullptr, 17, "ws / values / info is NULL");
  for (int b = 0; n_host != nullptr && b < B; ++b) GPMP_ARG(n_host[b] > q && n_host[b] <= nmax, 9, "a problem size outside (q, nmax]");
  hipStream_t st = as_stream(stream);
  const bool grad = grads_dev != nullptr;
  const BatchLayout l = batch_layout(nmax, d, q, B, grad);
  double* K = ws + l.K;
  double* dinv = ws + l.dinv;
  double* Y = ws + l.Y;
  int* ns = reinterpret_cast<int*>(ws + l.ns);
  const int ntheta = 1 + (noise ? 1 : 0) + d;
  // ---- sizes on the device
  if (n_host != nullptr) GPMP_HIP_TRY(hipMemcpyAsync(ns, n_host, sizeof(int) * (size_t)B, hipMemcpyHostToDevice, st));
  else {
    hipLaunchKernelGGL(fill_int_kernel, dim3((B + 255) / 256), dim3(256), 0, st, ns, B, nmax);
    GPMP_HIP_TRY(hipGetLastError());
  }
  // ---- Gram matrices (lower tiles), identity in the padding: ONE launch each when the problems share their parameters
  // (mini-batches of a loader), one small launch per problem otherwise (their parameters travel in the kernel arguments)
  bool ragged = false;
  for (int b = 0; n_host != nullptr && b < B; ++b) ragged = ragged || n_host[b] < nmax;
  if (theta_stride == 0) {
    const double sigma2 = std::exp(theta_host[0]);
    const double diag = noise ? std::exp(theta_host[1]) : 10.0 * sigma2 * DBL_EPSILON;      // matern.py:90
    int rc0 = launch_gram_lower_batch(x, stride_x, ns, nmax, d, p, theta_host, noise, diag, K, l.ld, (long)l.sK, B, st);
    if (rc0) return rc0;
    if (ragged) {
      hipLaunchKernelGGL(pad_identity_batch_kernel, dim3((nmax + 255) / 256, nmax, B), dim3(256), 0, st, K, l.ld, (long)l.sK, ns, nmax);
      GPMP_HIP_TRY(hipGetLastError());
    }
  } else {
    // every problem its own parameters (the chains of a sampler): their blocks go to device memory and the same two batched
    // launches run (round 2, first version: one small Gram launch, one padding launch and two gradient launches per problem --
    // 29 k problems/s at n = 128 against 75 k with shared parameters)
    const int pps = gram_param_block_elems();
    // (the staging buffer lives until the copy has run: freed by a host function enqueued behind it -- the call stays
    //  enqueue-only)
    std::vector<double>* blocks = new std::vector<double>((size_t)pps * B);
    for (int b = 0; b < B; ++b) {
      const double* th = theta_host + (long)b * theta_stride;
      const double diag = noise ? std::exp(th[1]) : 10.0 * std::exp(th[0]) * DBL_EPSILON;      // matern.py:90
      fill_gram_param_block(blocks->data() + (size_t)pps * b, d, p, th, noise, diag);
    }
    hipError_t ce = hipMemcpyAsync(ws + l.pp, blocks->data(), sizeof(double) * blocks->size(), hipMemcpyHostToDevice, st);
    if (ce == hipSuccess) ce = hipLaunchHostFunc(st, [](void* v) { delete static_cast<std::vector<double>*>(v); }, blocks);
    if (ce != hipSuccess) {
      (void)hipStreamSynchronize(st);
      delete blocks;
      set_error("HIP error %s staging the per-problem parameters", hipGetErrorString(ce));
      return -100;
    }
    int rc0 = launch_gram_lower_batch(x, stride_x, ns, nmax, d, p, theta_host, noise, 0.0, K, l.ld, (long)l.sK, B, st, ws + l.pp);
    if (rc0) return rc0;
    if (ragged) {
      hipLaunchKernelGGL(pad_identity_batch_kernel, dim3((nmax + 255) / 256, nmax, B), dim3(256), 0, st, K, l.ld, (long)l.sK, ns, nmax);
      GPMP_HIP_TRY(hipGetLastError());
    }
  }
  // ---- Cholesky of every slot, each kernel batched over the problems
  ProblemBatch pb;
  pb.nprob = B; pb.stride_a = (long)l.sK; pb.stride_dinv = (long)l.sD;
  int rc = potrf_blocked_batch(K, nmax, l.ld, dinv, info_dev, pb, st);
  if (rc) return rc;
  // ---- W = L^-1 [z, P], then log-det, W^T W, q x q algebra, value
  hipLaunchKernelGGL(batch_pack_kernel, dim3((nmax + 255) / 256, B), dim3(256), 0, st, z, stride_z, P, ldp, stride_p, q, ns, nmax, Y,
                     l.ldq, (long)l.sY);
  GPMP_HIP_TRY(hipGetLastError());
  // (three instantiations of the value kernel: q <= 3 -- every reference example --, q <= 7, q <= 16)
  rc = batch_solve<false>(K, l.ld, (long)l.sK, dinv, (long)l.sD, Y, l.ldq, (long)l.sY, nmax, 1 + q, q, B, st);
  if (rc) return rc;
  if (q <= 3)
    hipLaunchKernelGGL(batch_value_kernel<3>, dim3(B), dim3(256), 0, st, K, l.ld, (long)l.sK, Y, l.ldq, (long)l.sY, P, ldp, stride_p, q, ns,
                       nmax, ws + l.small, info_dev, values_dev);
  else if (q <= 7)
    hipLaunchKernelGGL(batch_value_kernel<7>, dim3(B), dim3(256), 0, st, K, l.ld, (long)l.sK, Y, l.ldq, (long)l.sY, P, ldp, stride_p, q, ns,
                       nmax, ws + l.small, info_dev, values_dev);
  else
    hipLaunchKernelGGL(batch_value_wide_kernel, dim3(B), dim3(256), 0, st, K, l.ld, (long)l.sK, Y, l.ldq, (long)l.sY, P, ldp, stride_p, q, ns,
                       nmax, ws + l.small, info_dev, values_dev);
  GPMP_HIP_TRY(hipGetLastError());
  if (!grad) return 0;
  // ---- gradient: X = L^-T W = K^-1 [z, P]; F, G; K^-1 = T^T T over the factor's slot; trace per problem
  double* X = ws + l.X;
  GPMP_HIP_TRY(hipMemcpyAsync(X, Y, sizeof(double) * l.sY * B, hipMemcpyDeviceToDevice, st));
  rc = batch_solve<true>(K, l.ld, (long)l.sK, dinv, (long)l.sD, X, l.ldq, (long)l.sY, nmax, 1 + q, q, B, st);
  if (rc) return rc;
  hipLaunchKernelGGL(batch_rows_kernel, dim3((nmax + 255) / 256, B), dim3(256), 0, st, X, l.ldq, (long)l.sY, ws + l.small, q, ns,
                     ws + l.F, ws + l.G, l.ldq, (long)l.sY);
  GPMP_HIP_TRY(hipGetLastError());
  double* T = ws + l.T;
  rc = trtri_doubling_batch(K, nmax, l.ld, dinv, T, l.ld, pb, (long)l.sK, st);
  if (rc) return rc;
  rc = lauum_lower_batch(T, nmax, l.ld, (long)l.sK, K, l.ld, (long)l.sK, B, st);
  if (rc) return rc;
  if (theta_stride == 0) {
    rc = launch_grad_trace_batch(K, l.ld, (long)l.sK, x, stride_x, ns, nmax, d, p, theta_host, noise, ws + l.F, ws + l.G, q + 1, l.ldq,
                                 (long)l.sY, grads_dev, ws + l.gws, B, st);
    if (rc) return rc;
  } else {
    rc = launch_grad_trace_batch(K, l.ld, (long)l.sK, x, stride_x, ns, nmax, d, p, theta_host, noise, ws + l.F, ws + l.G, q + 1, l.ldq,
                                 (long)l.sY, grads_dev, ws + l.gws, B, st, ws + l.pp);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(batch_grad_finalize_kernel, dim3((ntheta * B + 255) / 256), dim3(256), 0, st, grads_dev, ntheta, B, info_dev);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}
