// Fused anisotropic-distance + Matern Gram kernels for gfx950, and the analytic-gradient trace pass.
//
// One pass replaces gnp.scaled_distance (scipy cdist on pre-scaled points, direct sum of squared
// differences -- gpmp/num/numpy_backend.py:432-436), maternp_kernel's ~2p+3 full-size temporaries
// (gpmp/kernel/matern.py:54-64) and the dense "+ nugget * eye(n)" (matern.py:94).  The distance
// matrix is never stored: algorithmic HBM traffic is the 8 n m bytes of K written.
//
// Tiling (gram_kernel_v3): 128 x 64 output tile per 256-thread workgroup, 8 x 4 outputs per thread.  The x / y
// row blocks are pre-scaled (2c / rho for the covariance, 1 / rho for the plain distance) while staged into LDS
// ([k][128] and [k][64] images, 16 dimensions per chunk); each thread owns column pairs, so every store is 16 bytes
// and the 16 lanes of a row write 256 contiguous bytes.  The gradient-trace pass keeps 64 x 64 tiles (GT).
#include "common.h"
#include <cmath>

namespace gpmp {
namespace {

constexpr int GT = 64;   // tile edge
constexpr int DC = 16;   // dimensions per LDS chunk

struct MaternSpec {
  int p;
  double c;                      // 2 sqrt(p + 1/2)
  double q[GPMP_MAX_P + 1];      // K(h) = exp(-t/2) sum_k q[k] t^k, t = 2 c h
  double s[GPMP_MAX_P + 1];      // (dK/dh)/h = (2c)^2 exp(-t/2) sum_{k>=1} s[k] t^(k-1)   (p >= 1)
};


__device__ __forceinline__ double matern_eval(const MaternSpec& ms, double h) {
  // maternp_kernel, gpmp/kernel/matern.py:54-64 (Horner form of the same polynomial).
  const double t = 2.0 * ms.c * h;
  double poly = ms.q[ms.p];
  for (int k = ms.p - 1; k >= 0; --k) poly = poly * t + ms.q[k];
  return exp(-ms.c * h) * poly;
}

template <int P>
__device__ __forceinline__ double matern_eval_p(const MaternSpec& ms, double h) {
  const double t = 2.0 * ms.c * h;
  double poly = ms.q[P];
#pragma unroll
  for (int k = P - 1; k >= 0; --k) poly = poly * t + ms.q[k];
  return exp(-ms.c * h) * poly;
}

__device__ __forceinline__ double matern_dispatch(const MaternSpec& ms, double h) {
  switch (ms.p) {
    case 0: return matern_eval_p<0>(ms, h);
    case 1: return matern_eval_p<1>(ms, h);
    case 2: return matern_eval_p<2>(ms, h);
    case 3: return matern_eval_p<3>(ms, h);
    case 4: return matern_eval_p<4>(ms, h);
    default: return matern_eval(ms, h);
  }
}

// ---- fp64 helpers tuned for this kernel (VALU-bound: every instruction per entry counts) ---------------------------
// All constants below travel in the kernel-argument struct: they are then loaded once into SGPRs and used as the
// scalar operand of v_fma_f64.  As C++ literals the compiler re-materialised them into VGPRs next to every use
// (22 v_mov per entry in the previous version of this kernel: a quarter of its VALU work).
struct FastExp {
  double nl2e_half;        // -log2(e) / 2
  double ln2x2_hi, ln2x2_lo;  // 2 ln 2 split so that k * hi is exact for |k| < 2^20
  double c[13];            // c[j] = 1 / (2^j j!)  -- exp(r/2) = sum_j c[j] r^j
  double tiny;             // 1e-280, added to the rsq argument (a no-op for every normal a, keeps a == 0 finite)
};

// Per entry (inlined in the kernel, four entries in lock step):
//  * sqrt(a), a >= 0: hardware 1/sqrt estimate of a + tiny (a == 0 then gives 0 without a select; NaN / inf still
//    propagate through a), one coupled Newton step, one residual correction;
//  * exp(-t/2), t >= 0: k = rint(t log2(e) / 2), r = 2 k ln2 - t in [-0.694, 0.694], exp(r/2) by a degree-12 Taylor
//    polynomial in r (remainder 0.347^13 / 13! = 1.7e-16), scaled by 2^-k: < 1 ulp of libm on [0, 745], exact 1 at 0.
__device__ __forceinline__ double fast_sqrt_pos(double a, double tiny) {
  const double y0 = __builtin_amdgcn_rsq(a + tiny);
  double g = a * y0, h = 0.5 * y0;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, a);
  return fma(d, h, g);
}
__device__ __forceinline__ double fast_exp_neg_half(const FastExp& fe, double t) {
  const double nk = rint(t * fe.nl2e_half);                 // -k
  double r = fma(-nk, fe.ln2x2_hi, -t);
  r = fma(-nk, fe.ln2x2_lo, r);
  double e = fe.c[12];
#pragma unroll
  for (int j = 11; j >= 0; --j) e = fma(e, r, fe.c[j]);
  const int k = (int)nk;
  return ldexp(e, k < -1100 ? -1100 : k);
}

struct GramParams {
  const double* x;
  const double* y;
  double* K;
  long ldk;
  int n, m, d;
  int same, lower_only, aligned;
  int mode;                      // 0: covariance, 1: distance only
  int p;
  double diag_add;
  double scale[GPMP_MAX_DIM];    // mode 0: 2 c / rho_j (the tile accumulates t^2 = (2 c h)^2); mode 1: 1 / rho_j
  double q[GPMP_MAX_P + 1];      // sigma^2 q_k:  K(h) = exp(-t/2) sum_k q_k t^k
  FastExp fe;
  // batched over blockIdx.z (many small problems with the SAME parameters): problem b reads x + b * stride_x, writes
  // K + b * stride_k and has ns[b] points (ns == NULL: n)
  int nprob;
  long stride_x, stride_k;
  const int* ns;
  // per-problem PARAMETERS (sampler pattern: the same kernel, every problem its own theta): pp + b * PP_STRIDE holds
  // [scale (GPMP_MAX_DIM) | q (GPMP_MAX_P + 1) | diag_add | sigma2 | noise variance]; nullptr: the values above
  const double* pp;
};
constexpr int PP_Q = GPMP_MAX_DIM, PP_DIAG = GPMP_MAX_DIM + GPMP_MAX_P + 1, PP_SIGMA2 = PP_DIAG + 1, PP_NOISE = PP_DIAG + 2,
              PP_STRIDE = PP_DIAG + 3;

// 128 x 64 output tile per 256-thread workgroup, 8 x 4 outputs per thread.
// Per entry: 2 d VALU instructions of distance + 9 (sqrt) + 19 (exp) + P + 1 (Matern polynomial, sigma^2 folded in).
// MODE 0: covariance, 1: scaled distance only (gnp.scaled_distance).
template <int P, int MODE>
__global__ void __launch_bounds__(256) gram_kernel_v3(GramParams p) {
  __shared__ __attribute__((aligned(16))) double xs[DC][128];
  __shared__ __attribute__((aligned(16))) double ys[DC][GT];
  const int tj = blockIdx.x, ti = blockIdx.y;
  const int row0 = ti * 128, col0 = tj * GT;
  if (p.lower_only && col0 > row0 + 127) return;
  // (the parameter block is NEVER written: one store into the by-value struct and the compiler keeps a private copy of all of
  //  it -- scale[], q[], the exp constants -- in scratch memory instead of scalar registers: 896 bytes per lane, kernel 4.5 x
  //  slower, found by the round-2 rocprof pass)
  const double* __restrict__ px = p.x;
  double* __restrict__ pK = p.K;
  int pn = p.n, pm = p.m;
  if (p.nprob > 1) {
    px += (long)blockIdx.z * p.stride_x;
    pK += (long)blockIdx.z * p.stride_k;
    if (p.ns != nullptr) pn = pm = p.ns[blockIdx.z];
    if (row0 >= pn || col0 >= pm) return;
  }
  const double* __restrict__ ppb = p.pp != nullptr ? p.pp + (long)blockIdx.z * PP_STRIDE : nullptr;   // wave-uniform
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const double* __restrict__ yp = p.same ? px : p.y;

  double acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;

  for (int k0 = 0; k0 < p.d; k0 += DC) {
    if (k0) __syncthreads();
    const int kc = (p.d - k0) < DC ? (p.d - k0) : DC;
    // stage 128 rows of x and 64 rows of y (kc dims), scaled; consecutive threads walk k first
    // (global reads contiguous along k), LDS stores scatter over rows
    for (int idx = t; idx < 128 * kc; idx += 256) {
      const int r = idx / kc, k = idx - r * kc;
      xs[k][r] = (row0 + r < pn) ? (ppb ? ppb[k0 + k] : p.scale[k0 + k]) * px[(long)(row0 + r) * p.d + k0 + k] : 0.0;
    }
    for (int idx = t; idx < GT * kc; idx += 256) {
      const int r = idx / kc, k = idx - r * kc;
      ys[k][r] = (col0 + r < pm) ? (ppb ? ppb[k0 + k] : p.scale[k0 + k]) * yp[(long)(col0 + r) * p.d + k0 + k] : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < kc; ++k) {
      const d4 xa0 = *reinterpret_cast<const d4*>(&xs[k][ty * 8]);
      const d4 xa1 = *reinterpret_cast<const d4*>(&xs[k][ty * 8 + 4]);
      // this thread's columns: {2tx, 2tx+1, 32+2tx, 32+2tx+1} -> each 16-byte store below is contiguous
      // with its 15 neighbours (256 B = two full 128-B lines per row and instruction)
      const d2 y0 = *reinterpret_cast<const d2*>(&ys[k][2 * tx]);
      const d2 y1 = *reinterpret_cast<const d2*>(&ys[k][32 + 2 * tx]);
      const double yb[4] = {y0[0], y0[1], y1[0], y1[1]};
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double d0 = xa0[a] - yb[b], d1 = xa1[a] - yb[b];
          acc[a][b] = fma(d0, d0, acc[a][b]);
          acc[a + 4][b] = fma(d1, d1, acc[a + 4][b]);
        }
    }
  }

  const bool full = p.aligned && (row0 + 128 <= pn) && (col0 + GT <= pm);
  // tiles crossed by the diagonal (ii path): only they test row == col
  const bool diag_tile = p.same && (col0 < row0 + 128) && (col0 + GT > row0);
  double* __restrict__ out = pK + (long)(row0 + ty * 8) * p.ldk + col0 + 2 * tx;
  // leading coefficients of the two Horner chains live in VGPRs (their first fma would otherwise need two SGPR operands
  // and the compiler copies one of them next to every use)
  const int pdeg = (P >= 0) ? P : p.p;
  // Matern coefficients and the diagonal term: from the kernel arguments, or from this problem's block (uniform either way)
  constexpr int NQ = P >= 0 ? P + 1 : 1;
  double qc[NQ];
  if constexpr (P >= 0) {
#pragma unroll
    for (int k = 0; k <= P; ++k) qc[k] = ppb ? ppb[PP_Q + k] : p.q[k];
  }
  const double dadd = ppb ? ppb[PP_DIAG] : p.diag_add;
  double qtop = ppb ? ppb[PP_Q + pdeg] : p.q[pdeg], c12 = p.fe.c[12];
  asm volatile("" : "+v"(qtop), "+v"(c12));
#pragma unroll
  for (int a = 0; a < 8; ++a, out += p.ldk) {
    const int row = row0 + ty * 8 + a;
    double v[4];
    // the four entries of a row advance in lock step: every line below is four independent instructions, so one wave
    // keeps the fp64 pipe fed across the ~26-deep dependent chain of an entry (measured with entry-after-entry code:
    // 2.5 waves per SIMD resident, each waiting half of the time, VALU 60 % busy)
#pragma unroll
    for (int b = 0; b < 4; ++b) v[b] = __builtin_amdgcn_rsq(acc[a][b] + p.fe.tiny);
    double g[4], h[4], r[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { g[b] = acc[a][b] * v[b]; h[b] = 0.5 * v[b]; }
#pragma unroll
    for (int b = 0; b < 4; ++b) r[b] = fma(-h[b], g[b], 0.5);
#pragma unroll
    for (int b = 0; b < 4; ++b) { g[b] = fma(g[b], r[b], g[b]); h[b] = fma(h[b], r[b], h[b]); }
#pragma unroll
    for (int b = 0; b < 4; ++b) r[b] = fma(-g[b], g[b], acc[a][b]);
#pragma unroll
    for (int b = 0; b < 4; ++b) g[b] = fma(r[b], h[b], g[b]);          // g = t = 2 c h (mode 0) or h (mode 1)
    if constexpr (MODE == 0) {
      double nk[4], e[4], poly[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) nk[b] = rint(g[b] * p.fe.nl2e_half);   // -k
#pragma unroll
      for (int b = 0; b < 4; ++b) r[b] = fma(-nk[b], p.fe.ln2x2_hi, -g[b]);
#pragma unroll
      for (int b = 0; b < 4; ++b) r[b] = fma(-nk[b], p.fe.ln2x2_lo, r[b]);
#pragma unroll
      for (int b = 0; b < 4; ++b) { e[b] = c12; poly[b] = qtop; }
      if constexpr (P >= 0) {
#pragma unroll
        for (int k = P - 1; k >= 0; --k)
#pragma unroll
          for (int b = 0; b < 4; ++b) poly[b] = fma(poly[b], g[b], qc[k]);
      } else {
        for (int k = pdeg - 1; k >= 0; --k) {
          const double qk = ppb ? ppb[PP_Q + k] : p.q[k];
#pragma unroll
          for (int b = 0; b < 4; ++b) poly[b] = fma(poly[b], g[b], qk);
        }
      }
#pragma unroll
      for (int j = 11; j >= 0; --j)
#pragma unroll
        for (int b = 0; b < 4; ++b) e[b] = fma(e[b], r[b], p.fe.c[j]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int k = (int)nk[b];                                        // saturating v_cvt_i32_f64
        v[b] = ldexp(e[b], k < -1100 ? -1100 : k) * poly[b];             // exp underflows to 0 well before 2^-1100
      }
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) v[b] = g[b];
    }
    if (diag_tile) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (row == col0 + (b >> 1) * 32 + 2 * tx + (b & 1)) v[b] += dadd;
    }
    if (full) {
      *reinterpret_cast<d2*>(out) = (d2){v[0], v[1]};
      *reinterpret_cast<d2*>(out + 32) = (d2){v[2], v[3]};
    } else if (row < pn) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int cc = (b >> 1) * 32 + 2 * tx + (b & 1);
        if (col0 + cc < pm) out[cc - 2 * tx] = v[b];
      }
    }
  }
}

static void fill_fast_exp(FastExp& fe) {
  fe.nl2e_half = -0.5 * 1.4426950408889634;
  fe.ln2x2_hi = 2.0 * 6.93147180369123816490e-01;
  fe.ln2x2_lo = 2.0 * 1.90821492927058770002e-10;
  fe.tiny = 1e-280;
  double f = 1.0;
  for (int j = 0; j <= 12; ++j) {
    if (j) f *= 2.0 * j;         // 2^j j!  (exact in fp64 up to j = 12: 1.96e12)
    fe.c[j] = 1.0 / f;
  }
}

static int launch_gram(const GramParams& gp, hipStream_t st) {
  dim3 grid((gp.m + GT - 1) / GT, (gp.n + 127) / 128, gp.nprob > 1 ? gp.nprob : 1);
  if (gp.mode != 0) {
    hipLaunchKernelGGL((gram_kernel_v3<0, 1>), grid, dim3(256), 0, st, gp);
    return 0;
  }
  switch (gp.p) {
    case 0: hipLaunchKernelGGL((gram_kernel_v3<0, 0>), grid, dim3(256), 0, st, gp); break;
    case 1: hipLaunchKernelGGL((gram_kernel_v3<1, 0>), grid, dim3(256), 0, st, gp); break;
    case 2: hipLaunchKernelGGL((gram_kernel_v3<2, 0>), grid, dim3(256), 0, st, gp); break;
    case 3: hipLaunchKernelGGL((gram_kernel_v3<3, 0>), grid, dim3(256), 0, st, gp); break;
    default: hipLaunchKernelGGL((gram_kernel_v3<-1, 0>), grid, dim3(256), 0, st, gp); break;
  }
  return 0;
}

struct PairParams {
  const double* x;
  const double* y;
  double* out;
  int n, d, same;
  double sigma2;
  double invrho[GPMP_MAX_DIM];
  MaternSpec ms;
};

__global__ void pairwise_kernel(PairParams p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  double h = 0.0;
  if (!p.same) {
    double s = 0.0;
    for (int k = 0; k < p.d; ++k) {
      const double df = p.invrho[k] * (p.x[(long)i * p.d + k] - p.y[(long)i * p.d + k]);
      s = fma(df, df, s);
    }
    h = sqrt(s);
  }
  p.out[i] = p.sigma2 * matern_dispatch(p.ms, h);
}

__global__ void matern_elementwise_kernel(const double* __restrict__ h, long count, MaternSpec ms,
                                          double* __restrict__ out) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < count; i += stride) {
    double hv = h[i];
    if (isinf(hv)) hv = 1.7976931348623157e308 / 1000.0;  // inftobigf, numpy_backend.py:250-252
    out[i] = matern_dispatch(ms, hv);
  }
}

// ---- gradient trace pass ----------------------------------------------------------------------
// g[0]      = sum M_ik K_ik                       (d/d log sigma^2; diag_add scales with sigma^2
//                                                  when it is the default nugget, see host code)
// g[1 + j]  = sum M_ik sigma2 (K'(h)/h) (xs_ij - xs_kj)^2          (d/d log(1/rho_j))
// g[DT + 1] = trace(M)                            (for the noise-variance parameter)
// with M_ik = Kinv_ik - sum_a F_ia G_ka, summed over the full symmetric matrix by visiting tiles on
// and below the diagonal (off-diagonal entries weigh 2).
struct GradParams {
  const double* Kinv;
  long ldk;
  const double* x;
  const double* F;
  const double* G;
  long ldf;
  int n, d, r;
  int ntiles_side, ntiles;
  double sigma2;
  double* partial;  // [gridDim.x][DT + 2]
  // batched over blockIdx.y (same parameters): problem b = Kinv + b * stride_kinv, x + b * stride_x, F / G + b * stride_f,
  // partial + b * stride_partial, ns[b] points
  int nprob;
  long stride_kinv, stride_x, stride_f, stride_partial;
  const int* ns;
  const double* pp;              // per-problem parameter blocks (see GramParams), or nullptr
  // cross mode (grad_trace_kernel<DT, false, true>): M is a RECTANGULAR n x m block of a larger matrix, its rows belong to the
  // points x and its columns to the points y; every entry weighs 1 (no symmetric doubling), no trace term
  const double* y;
  int m, ntiles_c;
  double invrho[GPMP_MAX_DIM];   // 2 c / rho_j: the tile accumulates t^2 = (2 c h)^2 and the weights are per (2 c delta_j)^2
  MaternSpec ms;
  FastExp fe;
};

__device__ __forceinline__ double matern_dk_over_h(const MaternSpec& ms, double h, double& kval) {
  const double t = 2.0 * ms.c * h;
  const double e = exp(-ms.c * h);
  double poly = ms.q[ms.p];
  for (int k = ms.p - 1; k >= 0; --k) poly = poly * t + ms.q[k];
  kval = e * poly;
  if (ms.p == 0) return h > 0.0 ? -ms.c * e / h : 0.0;
  double s = ms.s[ms.p];
  for (int k = ms.p - 1; k >= 1; --k) s = s * t + ms.s[k];
  return (2.0 * ms.c) * (2.0 * ms.c) * e * s;
}

// PP: length scales from the problem's parameter block in device memory instead of the kernel arguments (a run-time choice
// between the two sources made the compiler copy the argument block to scratch memory: compile-time instead)
template <int DT, bool PP = false, bool CROSS = false>
__global__ void __launch_bounds__(256) grad_trace_kernel(GradParams p) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* xs = sm;                    // [DT][GT]
  double* ys = xs + DT * GT;          // [DT][GT]
  double* fs = ys + DT * GT;          // [r][GT]  F rows of the tile's i block
  double* gs = fs + p.r * GT;         // [r][GT]  G rows of the tile's k block
  __shared__ double red[4][DT + 2];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;

  double gacc[DT];
#pragma unroll
  for (int k = 0; k < DT; ++k) gacc[k] = 0.0;
  double g0 = 0.0, gtr = 0.0;
  // (locals, never a store into the parameter block: see gram_kernel_v3)
  const double* __restrict__ pKinv = p.Kinv;
  const double* __restrict__ px = p.x;
  const double* __restrict__ pF = p.F;
  const double* __restrict__ pG = p.G;
  double* __restrict__ ppartial = p.partial;
  int pn = p.n, pntiles_side = p.ntiles_side, pntiles = p.ntiles;
  const double* __restrict__ ppb = PP ? p.pp + (long)blockIdx.y * PP_STRIDE : nullptr;
  (void)ppb;
  if (p.nprob > 1) {
    const int z = blockIdx.y;
    pKinv += (long)z * p.stride_kinv;
    px += (long)z * p.stride_x;
    pF += (long)z * p.stride_f;
    pG += (long)z * p.stride_f;
    ppartial += (long)z * p.stride_partial;
    if (p.ns != nullptr) {
      pn = p.ns[z];
      pntiles_side = (pn + GT - 1) / GT;
      pntiles = pntiles_side * (pntiles_side + 1) / 2;
    }
  }
  // column points: the second set in cross mode, otherwise the (per-problem) row points themselves
  const double* __restrict__ py = CROSS ? p.y : px;
  const int pm = CROSS ? p.m : 0;
  (void)pm;

  for (int tile = blockIdx.x; tile < pntiles; tile += gridDim.x) {
    int ti, tj;
    if constexpr (CROSS) {
      ti = tile / p.ntiles_c;
      tj = tile - ti * p.ntiles_c;
    } else {
      ti = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
      while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
      while (ti * (ti + 1) / 2 > tile) --ti;
      tj = tile - ti * (ti + 1) / 2;
    }
    const int row0 = ti * GT, col0 = tj * GT;
    const int ncols = CROSS ? pm : pn;            // valid columns
    __syncthreads();
    for (int idx = t; idx < DT * GT; idx += 256) {
      const int r = idx / DT, k = idx % DT;
      double vx = 0.0, vy = 0.0;
      if (k < p.d) {
        double ir;
        if constexpr (PP) ir = ppb[k]; else ir = p.invrho[k];
        if (row0 + r < pn) vx = ir * px[(long)(row0 + r) * p.d + k];
        if (col0 + r < ncols) vy = ir * py[(long)(col0 + r) * p.d + k];
      }
      xs[k * GT + r] = vx;
      ys[k * GT + r] = vy;
    }
    for (int idx = t; idx < p.r * GT; idx += 256) {
      const int r = idx / p.r, a = idx % p.r;
      fs[a * GT + r] = (row0 + r < pn) ? pF[(long)(row0 + r) * p.ldf + a] : 0.0;
      gs[a * GT + r] = (col0 + r < ncols) ? pG[(long)(col0 + r) * p.ldf + a] : 0.0;
    }
    __syncthreads();

    double h2[4][4], w[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) { h2[a][b] = 0.0; w[a][b] = 0.0; }
#pragma unroll
    for (int k = 0; k < DT; ++k) {
      if (k < p.d) {
        const d4 xa = *reinterpret_cast<const d4*>(&xs[k * GT + ty * 4]);
        const d4 yb = *reinterpret_cast<const d4*>(&ys[k * GT + tx * 4]);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const double df = xa[a] - yb[b];
            h2[a][b] = fma(df, df, h2[a][b]);
          }
      }
    }
    // low-rank part of M
    for (int a2 = 0; a2 < p.r; ++a2) {
      const d4 fa = *reinterpret_cast<const d4*>(&fs[a2 * GT + ty * 4]);
      const d4 gb = *reinterpret_cast<const d4*>(&gs[a2 * GT + tx * 4]);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) w[a][b] = fma(fa[a], gb[b], w[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = row0 + ty * 4 + a;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int col = col0 + tx * 4 + b;
        double wt = 0.0;
        if constexpr (CROSS) {
          if (row < pn && col < ncols) wt = 1.0;
        } else {
          if (row < pn && col < pn) wt = col < row ? 2.0 : (col == row ? 1.0 : 0.0);
        }
        double mval = 0.0;
        if (wt != 0.0) mval = wt * (pKinv[(long)row * p.ldk + col] - w[a][b]);
        // K(h) = e^{-t/2} sum q_k t^k and (K'(h)/h) / (2c)^2 = e^{-t/2} sum_{k>=1} s_k t^{k-1} (p >= 1), t = 2 c h
        const double tt = fast_sqrt_pos(h2[a][b], p.fe.tiny);
        const double e = fast_exp_neg_half(p.fe, tt);
        double poly = p.ms.q[p.ms.p];
        for (int k = p.ms.p - 1; k >= 0; --k) poly = fma(poly, tt, p.ms.q[k]);
        const double kval = e * poly;
        double dk;
        if (p.ms.p == 0) {
          dk = tt > 0.0 ? -0.5 * e / tt : 0.0;
        } else {
          double sp = p.ms.s[p.ms.p];
          for (int k = p.ms.p - 1; k >= 1; --k) sp = fma(sp, tt, p.ms.s[k]);
          dk = e * sp;
        }
        g0 = fma(mval, kval, g0);
        if (!CROSS && row == col) gtr += mval;
        w[a][b] = mval * dk;  // weight of (delta_j)^2 for every dimension j
      }
    }
#pragma unroll
    for (int k = 0; k < DT; ++k) {
      if (k < p.d) {
        const d4 xa = *reinterpret_cast<const d4*>(&xs[k * GT + ty * 4]);
        const d4 yb = *reinterpret_cast<const d4*>(&ys[k * GT + tx * 4]);
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const double df = xa[a] - yb[b];
            s = fma(w[a][b], df * df, s);
          }
        gacc[k] += s;
      }
    }
  }

  // block reduction: wave shuffle, then across the 4 waves through LDS
  const int lane = t & 63, wave = t >> 6;
  auto wave_sum = [](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  g0 = wave_sum(g0);
  gtr = wave_sum(gtr);
#pragma unroll
  for (int k = 0; k < DT; ++k) gacc[k] = wave_sum(gacc[k]);
  if (lane == 0) {
    red[wave][0] = g0;
#pragma unroll
    for (int k = 0; k < DT; ++k) red[wave][1 + k] = gacc[k];
    red[wave][DT + 1] = gtr;
  }
  __syncthreads();
  if (t < DT + 2) {
    ppartial[(long)blockIdx.x * (DT + 2) + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
  }
}

__global__ void grad_finalize_kernel(const double* __restrict__ partial, int nblocks, int width,
                                     int d, int noise, double sigma2, double nugget_scale,
                                     double noise_var, double* __restrict__ g, long stride_partial = 0, int stride_g = 0,
                                     const double* __restrict__ pp = nullptr) {
  // one thread per output column of `partial`; blockIdx.x = problem of a batched launch
  partial += (long)blockIdx.x * stride_partial;
  g += (long)blockIdx.x * stride_g;
  if (pp != nullptr) {                        // per-problem parameters
    sigma2 = pp[(long)blockIdx.x * PP_STRIDE + PP_SIGMA2];
    noise_var = pp[(long)blockIdx.x * PP_STRIDE + PP_NOISE];
  }
  const int k = threadIdx.x;
  if (k >= width) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += partial[(long)b * width + k];
  __shared__ double tot[GPMP_MAX_DIM + 2];
  tot[k] = s;
  __syncthreads();
  const double tr = tot[width - 1];
  if (k == 0) {
    // K = sigma2 * Kc (+ nugget 10 eps sigma2 I when there is no noise parameter)
    g[0] = sigma2 * tot[0] + nugget_scale * sigma2 * tr;
    if (noise) g[1] = noise_var * tr;
  }
  if (k >= 1 && k <= d) g[(noise ? 1 : 0) + k] = sigma2 * tot[k];
}

// dK/dtheta_j as a dense matrix (Fisher information, diagnostics): one entry per thread.
//   jparam = 0: d/d log sigma^2 = sigma2 * Kc (+ nugget on the diagonal when it scales with sigma2)
//   jparam = noise_index: sigma_noise^2 * I
//   otherwise: d/d log(1/rho_j) = sigma2 * (K'(h)/h) * (invrho_j (x_ij - x_kj))^2
struct DerivParams {
  const double* x;
  double* out;
  long ld;
  int n, d, jdim, kind;          // kind 0: log sigma2, 1: noise, 2: length-scale jdim
  double sigma2, diag_val;
  double invrho[GPMP_MAX_DIM];
  MaternSpec ms;
};

__global__ void gram_deriv_kernel(DerivParams p) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (k >= p.n) return;
  double v = 0.0;
  if (p.kind == 1) {
    v = (i == k) ? p.diag_val : 0.0;
  } else {
    double s = 0.0, dj = 0.0;
    for (int c = 0; c < p.d; ++c) {
      const double df = p.invrho[c] * (p.x[(long)i * p.d + c] - p.x[(long)k * p.d + c]);
      s = fma(df, df, s);
      if (c == p.jdim) dj = df * df;
    }
    const double h = sqrt(s);
    double kval;
    const double dk = matern_dk_over_h(p.ms, h, kval);
    if (p.kind == 0) v = p.sigma2 * kval + ((i == k) ? p.diag_val : 0.0);
    else v = p.sigma2 * dk * dj;
  }
  p.out[(long)i * p.ld + k] = v;
}

// ---- host-side helpers -------------------------------------------------------------------------
int fill_matern(MaternSpec& ms, int p) {
  if (p < 0 || p > GPMP_MAX_P) return -1;
  ms.p = p;
  ms.c = 2.0 * std::sqrt(p + 0.5);
  for (int k = 0; k <= GPMP_MAX_P; ++k) ms.q[k] = ms.s[k] = 0.0;
  ms.q[0] = 1.0;
  for (int i = 0; i < p; ++i) {  // a_i multiplies t^(p-i), gpmp/kernel/matern.py:59-63
    const double a = std::exp(std::lgamma(p + 1.0) - std::lgamma(2.0 * p + 1.0) + std::lgamma(p + i + 1.0) -
                              std::lgamma(i + 1.0) - std::lgamma(p - i + 1.0));
    ms.q[p - i] = a;
  }
  for (int k = 0; k <= p; ++k) ms.s[k] = (k + 1 <= p ? (k + 1) * ms.q[k + 1] : 0.0) - 0.5 * ms.q[k];
  return 0;
}

}  // namespace
}  // namespace gpmp

using namespace gpmp;

extern "C" int gpmp_matern_gram(const double* x, const double* y, int n, int m, int d, int p,
                                const double* theta_host, int noise, double diag_add, int lower_only,
                                double* K, long ldk, gpmp_stream_t stream) {
  GPMP_ARG(x != nullptr, 1, "x is NULL");
  GPMP_ARG(n >= 0 && n <= GPMP_MAX_EXTENT, 3, "n outside [0, GPMP_MAX_EXTENT]");
  GPMP_ARG(m >= 0 && m <= GPMP_MAX_EXTENT, 4, "m outside [0, GPMP_MAX_EXTENT]");
  GPMP_ARG(d >= 1 && d <= GPMP_MAX_DIM, 5, "d outside [1, GPMP_MAX_DIM]");
  GPMP_ARG(p >= 0 && p <= GPMP_MAX_P, 6, "p outside [0, GPMP_MAX_P]");
  GPMP_ARG(theta_host != nullptr, 7, "theta is NULL");
  GPMP_ARG(K != nullptr, 11, "K is NULL");
  if (y == nullptr) m = n;
  GPMP_ARG(ldk >= m, 12, "ldk < m");
  if (n == 0 || m == 0) return 0;
  GramParams gp;
  gp.nprob = 1; gp.stride_x = gp.stride_k = 0; gp.ns = nullptr; gp.pp = nullptr;
  gp.x = x; gp.y = y; gp.K = K; gp.ldk = ldk;
  gp.n = n; gp.m = m; gp.d = d;
  gp.same = (y == nullptr); gp.lower_only = (y == nullptr) ? lower_only : 0;
  gp.aligned = ((reinterpret_cast<uintptr_t>(K) & 15) == 0) && ((ldk & 1) == 0);
  gp.mode = 0;
  gp.p = p;
  gp.diag_add = diag_add;
  MaternSpec ms;
  fill_matern(ms, p);
  const double sigma2 = std::exp(theta_host[0]);
  const int off = noise ? 2 : 1;
  for (int k = 0; k < d; ++k) gp.scale[k] = 2.0 * ms.c * std::exp(theta_host[off + k]);
  for (int k = 0; k <= GPMP_MAX_P; ++k) gp.q[k] = sigma2 * ms.q[k];
  fill_fast_exp(gp.fe);
  {
    // work = algorithmic bytes written (8 per entry; lower_only writes about half)
    ProfScope ps(PK_GRAM, as_stream(stream), 8.0 * (double)n * (double)m * (gp.lower_only ? 0.5 : 1.0));
    launch_gram(gp, as_stream(stream));
  }
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gpmp_scaled_distance(const double* x, const double* y, int n, int m, int d,
                                    const double* loginvrho_host, double* D, long ldd,
                                    gpmp_stream_t stream) {
  GPMP_ARG(x != nullptr && y != nullptr, 1, "x or y is NULL");
  GPMP_ARG(d >= 1 && d <= GPMP_MAX_DIM, 5, "d outside [1, GPMP_MAX_DIM]");
  GPMP_ARG(loginvrho_host != nullptr, 6, "loginvrho is NULL");
  GPMP_ARG(D != nullptr && ldd >= m, 7, "D is NULL or ldd < m");
  GPMP_ARG(n <= GPMP_MAX_EXTENT && m <= GPMP_MAX_EXTENT, 3, "n or m above GPMP_MAX_EXTENT");
  if (n <= 0 || m <= 0) return 0;
  GramParams gp;
  gp.nprob = 1; gp.stride_x = gp.stride_k = 0; gp.ns = nullptr; gp.pp = nullptr;
  gp.x = x; gp.y = y; gp.K = D; gp.ldk = ldd;
  gp.n = n; gp.m = m; gp.d = d;
  gp.same = 0; gp.lower_only = 0;
  gp.aligned = ((reinterpret_cast<uintptr_t>(D) & 15) == 0) && ((ldd & 1) == 0);
  gp.mode = 1; gp.p = 0; gp.diag_add = 0.0;
  for (int k = 0; k < d; ++k) gp.scale[k] = std::exp(loginvrho_host[k]);
  for (int k = 0; k <= GPMP_MAX_P; ++k) gp.q[k] = 0.0;
  fill_fast_exp(gp.fe);
  launch_gram(gp, as_stream(stream));
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gpmp_matern_pairwise(const double* x, const double* y, int n, int d, int p,
                                    const double* theta_host, int noise, double* out,
                                    gpmp_stream_t stream) {
  GPMP_ARG(x != nullptr, 1, "x is NULL");
  GPMP_ARG(d >= 1 && d <= GPMP_MAX_DIM, 4, "d outside [1, GPMP_MAX_DIM]");
  GPMP_ARG(p >= 0 && p <= GPMP_MAX_P, 5, "p outside [0, GPMP_MAX_P]");
  GPMP_ARG(theta_host != nullptr, 6, "theta is NULL");
  GPMP_ARG(out != nullptr, 8, "out is NULL");
  GPMP_ARG(n <= GPMP_MAX_EXTENT, 3, "n above GPMP_MAX_EXTENT");
  if (n <= 0) return 0;
  PairParams pp;
  pp.x = x; pp.y = y; pp.out = out; pp.n = n; pp.d = d; pp.same = (y == nullptr || y == x);
  pp.sigma2 = std::exp(theta_host[0]);
  const int off = noise ? 2 : 1;
  for (int k = 0; k < d; ++k) pp.invrho[k] = std::exp(theta_host[off + k]);
  fill_matern(pp.ms, p);
  hipLaunchKernelGGL(pairwise_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), pp);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gpmp_maternp_kernel(const double* h, long count, int p, double* out, gpmp_stream_t stream) {
  GPMP_ARG(h != nullptr, 1, "h is NULL");
  GPMP_ARG(p >= 0 && p <= GPMP_MAX_P, 3, "p outside [0, GPMP_MAX_P]");
  GPMP_ARG(out != nullptr, 4, "out is NULL");
  if (count <= 0) return 0;
  MaternSpec ms;
  fill_matern(ms, p);
  long blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(matern_elementwise_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), h, count, ms, out);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

namespace {
constexpr int GRAD_BLOCKS = 1024;
int grad_tier(int d) { return d <= 4 ? 4 : d <= 8 ? 8 : d <= 16 ? 16 : d <= 32 ? 32 : 64; }

template <int DT>
int launch_grad(GradParams& gp, int nblocks, hipStream_t st) {
  const size_t lds = sizeof(double) * (2 * DT * GT + 2 * (size_t)gp.r * GT);
  static DeviceOnce attr_once;
  if (const long long dev_bit = attr_once.need()) {
    if (dev_bit < 0) { set_error("hipGetDevice failed or device ordinal above 62"); return -1; }
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(grad_trace_kernel<DT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_once.done(dev_bit);
  }
  hipLaunchKernelGGL((grad_trace_kernel<DT>), dim3(nblocks, gp.nprob > 1 ? gp.nprob : 1), dim3(256), lds, st, gp);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" size_t gpmp_grad_ws_elems(int n, int d) {
  (void)n;
  return (size_t)GRAD_BLOCKS * (grad_tier(d) + 2);
}

namespace gpmp {
// One problem's parameter block (PP_STRIDE doubles, layout in GramParams) from its theta (host side).
int gram_param_block_elems() { return PP_STRIDE; }
void fill_gram_param_block(double* blk, int d, int p, const double* theta, int noise, double diag_add) {
  MaternSpec ms;
  fill_matern(ms, p);
  const double sigma2 = std::exp(theta[0]);
  const int off = noise ? 2 : 1;
  for (int k = 0; k < GPMP_MAX_DIM; ++k) blk[k] = k < d ? 2.0 * ms.c * std::exp(theta[off + k]) : 0.0;
  for (int k = 0; k <= GPMP_MAX_P; ++k) blk[PP_Q + k] = sigma2 * ms.q[k];
  blk[PP_DIAG] = diag_add;
  blk[PP_SIGMA2] = sigma2;
  blk[PP_NOISE] = noise ? std::exp(theta[1]) : 0.0;
}
// ---- batched over many small problems with the SAME parameters (drivers_batch.hip) ---------------------------------
// Lower-tile Gram matrices of `nprob` problems (points x + b * stride_x, ns[b] of them; matrices K + b * stride_k) in ONE launch.
int launch_gram_lower_batch(const double* x, long stride_x, const int* ns_dev, int nmax, int d, int p, const double* theta_host,
                            int noise, double diag_add, double* K, long ldk, long stride_k, int nprob, hipStream_t st,
                            const double* pp_dev) {
  GramParams gp;
  gp.nprob = 1; gp.stride_x = gp.stride_k = 0; gp.ns = nullptr; gp.pp = nullptr;
  gp.x = x; gp.y = nullptr; gp.K = K; gp.ldk = ldk;
  gp.n = nmax; gp.m = nmax; gp.d = d;
  gp.same = 1; gp.lower_only = 1;
  gp.aligned = ((reinterpret_cast<uintptr_t>(K) & 15) == 0) && ((ldk & 1) == 0) && ((stride_k & 1) == 0);
  gp.mode = 0;
  gp.p = p;
  gp.diag_add = diag_add;
  MaternSpec ms;
  fill_matern(ms, p);
  const double sigma2 = std::exp(theta_host[0]);
  const int off = noise ? 2 : 1;
  for (int k = 0; k < d; ++k) gp.scale[k] = 2.0 * ms.c * std::exp(theta_host[off + k]);
  for (int k = 0; k <= GPMP_MAX_P; ++k) gp.q[k] = sigma2 * ms.q[k];
  fill_fast_exp(gp.fe);
  gp.nprob = nprob > 1 ? nprob : 2;      // (a batch of one still takes the batched addressing: ns is read)
  gp.stride_x = stride_x; gp.stride_k = stride_k; gp.ns = ns_dev;
  gp.pp = pp_dev;
  {
    ProfScope ps(PK_GRAM, st, 4.0 * (double)nmax * (double)nmax * nprob);
    dim3 grid((nmax + GT - 1) / GT, (nmax + 127) / 128, nprob);
    switch (p) {
      case 0: hipLaunchKernelGGL((gram_kernel_v3<0, 0>), grid, dim3(256), 0, st, gp); break;
      case 1: hipLaunchKernelGGL((gram_kernel_v3<1, 0>), grid, dim3(256), 0, st, gp); break;
      case 2: hipLaunchKernelGGL((gram_kernel_v3<2, 0>), grid, dim3(256), 0, st, gp); break;
      case 3: hipLaunchKernelGGL((gram_kernel_v3<3, 0>), grid, dim3(256), 0, st, gp); break;
      default: hipLaunchKernelGGL((gram_kernel_v3<-1, 0>), grid, dim3(256), 0, st, gp); break;
    }
  }
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

// The gradient traces of `nprob` problems with the same parameters in ONE launch (+ one finalisation launch):
// g_dev + b * ntheta <- sum M_b dK_b / dtheta (see gpmp_matern_grad_trace), ws: nprob * gpmp_grad_ws_elems(nmax, d) doubles.
int launch_grad_trace_batch(const double* Kinv, long ldk, long stride_kinv, const double* x, long stride_x, const int* ns_dev, int nmax,
                            int d, int p, const double* theta_host, int noise, const double* F, const double* G, int r, long ldf,
                            long stride_f, double* g_dev, double* ws, int nprob, hipStream_t st, const double* pp_dev) {
  GradParams gp;
  gp.pp = nullptr;
  gp.y = nullptr; gp.m = 0; gp.ntiles_c = 0;
  gp.Kinv = Kinv; gp.ldk = ldk; gp.x = x; gp.F = F; gp.G = G; gp.ldf = ldf;
  gp.n = nmax; gp.d = d; gp.r = r;
  gp.ntiles_side = (nmax + GT - 1) / GT;
  gp.ntiles = gp.ntiles_side * (gp.ntiles_side + 1) / 2;
  gp.sigma2 = std::exp(theta_host[0]);
  gp.partial = ws;
  const int off = noise ? 2 : 1;
  fill_matern(gp.ms, p);
  for (int k = 0; k < d; ++k) gp.invrho[k] = 2.0 * gp.ms.c * std::exp(theta_host[off + k]);
  fill_fast_exp(gp.fe);
  // few blocks per problem: the problems themselves fill the machine
  int nblocks = gp.ntiles < 32 ? gp.ntiles : 32;
  const int dt = grad_tier(d);
  gp.nprob = nprob > 1 ? nprob : 2;
  gp.stride_kinv = stride_kinv; gp.stride_x = stride_x; gp.stride_f = stride_f;
  gp.stride_partial = (long)GRAD_BLOCKS * (dt + 2);
  gp.ns = ns_dev;
  gp.pp = pp_dev;
  int rc = 0;
  {
    GradParams g1 = gp;
    const size_t lds = sizeof(double) * (2 * (size_t)dt * GT + 2 * (size_t)r * GT);
    auto go = [&](auto kern) -> int {
      GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      hipLaunchKernelGGL(kern, dim3(nblocks, nprob), dim3(256), lds, st, g1);
      GPMP_HIP_TRY(hipGetLastError());
      return 0;
    };
    if (pp_dev != nullptr) {
      switch (dt) {
        case 4: rc = go(grad_trace_kernel<4, true>); break;
        case 8: rc = go(grad_trace_kernel<8, true>); break;
        case 16: rc = go(grad_trace_kernel<16, true>); break;
        case 32: rc = go(grad_trace_kernel<32, true>); break;
        default: rc = go(grad_trace_kernel<64, true>); break;
      }
    } else {
      switch (dt) {
        case 4: rc = go(grad_trace_kernel<4>); break;
        case 8: rc = go(grad_trace_kernel<8>); break;
        case 16: rc = go(grad_trace_kernel<16>); break;
        case 32: rc = go(grad_trace_kernel<32>); break;
        default: rc = go(grad_trace_kernel<64>); break;
      }
    }
  }
  if (rc) return rc;
  const double eps = 2.220446049250313e-16;
  const double nugget_scale = noise ? 0.0 : 10.0 * eps;
  const double noise_var = noise ? std::exp(theta_host[1]) : 0.0;
  hipLaunchKernelGGL(grad_finalize_kernel, dim3(nprob), dim3(128), 0, st, ws, nblocks, dt + 2, d, noise, gp.sigma2, nugget_scale, noise_var,
                     g_dev, gp.stride_partial, 1 + (noise ? 1 : 0) + d, pp_dev);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace gpmp

extern "C" int gpmp_matern_grad_trace(const double* Kinv, long ldk, const double* x, int n, int d, int p,
                                      const double* theta_host, int noise, const double* F,
                                      const double* G, int r, long ldf, double* g_dev, double* ws,
                                      gpmp_stream_t stream) {
  GPMP_ARG(Kinv != nullptr, 1, "Kinv is NULL");
  GPMP_ARG(x != nullptr, 3, "x is NULL");
  GPMP_ARG(n >= 1 && n <= GPMP_MAX_EXTENT, 4, "n outside [1, GPMP_MAX_EXTENT]");
  GPMP_ARG(ldk >= n, 2, "ldk < n");
  GPMP_ARG(d >= 1 && d <= GPMP_MAX_DIM, 5, "d outside [1, GPMP_MAX_DIM]");
  GPMP_ARG(p >= 0 && p <= GPMP_MAX_P, 6, "p outside [0, GPMP_MAX_P]");
  GPMP_ARG(theta_host != nullptr, 7, "theta is NULL");
  GPMP_ARG(r >= 0 && r <= GPMP_MAX_RANK, 11, "r outside [0, GPMP_MAX_RANK]");
  GPMP_ARG(r == 0 || (F != nullptr && G != nullptr && ldf >= r), 9, "F/G NULL or ldf < r with r > 0");
  GPMP_ARG(g_dev != nullptr && ws != nullptr, 13, "g or ws is NULL");
  GradParams gp;
  gp.pp = nullptr;
  gp.nprob = 1; gp.stride_kinv = gp.stride_x = gp.stride_f = gp.stride_partial = 0; gp.ns = nullptr;
  gp.y = nullptr; gp.m = 0; gp.ntiles_c = 0;
  gp.Kinv = Kinv; gp.ldk = ldk; gp.x = x; gp.F = F; gp.G = G; gp.ldf = ldf;
  gp.n = n; gp.d = d; gp.r = r;
  gp.ntiles_side = (n + GT - 1) / GT;
  gp.ntiles = gp.ntiles_side * (gp.ntiles_side + 1) / 2;
  gp.sigma2 = std::exp(theta_host[0]);
  gp.partial = ws;
  const int off = noise ? 2 : 1;
  fill_matern(gp.ms, p);
  for (int k = 0; k < d; ++k) gp.invrho[k] = 2.0 * gp.ms.c * std::exp(theta_host[off + k]);
  fill_fast_exp(gp.fe);
  const int nblocks = gp.ntiles < GRAD_BLOCKS ? gp.ntiles : GRAD_BLOCKS;
  const int dt = grad_tier(d);
  int rc = 0;
  hipStream_t st = as_stream(stream);
  switch (dt) {
    case 4: rc = launch_grad<4>(gp, nblocks, st); break;
    case 8: rc = launch_grad<8>(gp, nblocks, st); break;
    case 16: rc = launch_grad<16>(gp, nblocks, st); break;
    case 32: rc = launch_grad<32>(gp, nblocks, st); break;
    default: rc = launch_grad<64>(gp, nblocks, st); break;
  }
  if (rc) return rc;
  const double eps = 2.220446049250313e-16;
  const double nugget_scale = noise ? 0.0 : 10.0 * eps;   // matern.py:90: nugget = 10 sigma2 eps
  const double noise_var = noise ? std::exp(theta_host[1]) : 0.0;
  hipLaunchKernelGGL(grad_finalize_kernel, dim3(1), dim3(128), 0, st, ws, nblocks, dt + 2, d, noise,
                     gp.sigma2, nugget_scale, noise_var, g_dev);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

// cross traces: g[0] = sigma2 sum M Kc, g[1 + j] = sigma2 sum M (K'/h) delta_j^2  (no nugget / noise terms: the caller owns tr(M))
__global__ void grad_cross_finalize_kernel(const double* __restrict__ partial, int nblocks, int width, int d, double sigma2,
                                           double* __restrict__ g) {
  const int k = threadIdx.x;
  if (k > d) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += partial[(long)b * width + k];
  g[k] = sigma2 * s;
}

extern "C" int gpmp_matern_grad_trace_cross(const double* M, long ldm, const double* x, int n, const double* y, int m, int d, int p,
                                            const double* theta_host, int noise, const double* F, const double* G, int r, long ldf,
                                            double* g_dev, double* ws, gpmp_stream_t stream) {
  GPMP_ARG(M != nullptr, 1, "M is NULL");
  GPMP_ARG(x != nullptr && y != nullptr, 3, "x or y is NULL");
  GPMP_ARG(n >= 1 && m >= 1 && n <= GPMP_MAX_EXTENT && m <= GPMP_MAX_EXTENT && ldm >= m, 4, "n or m outside [1, GPMP_MAX_EXTENT], or ldm < m");
  GPMP_ARG(d >= 1 && d <= GPMP_MAX_DIM, 7, "d outside [1, GPMP_MAX_DIM]");
  GPMP_ARG(p >= 0 && p <= GPMP_MAX_P, 8, "p outside [0, GPMP_MAX_P]");
  GPMP_ARG(theta_host != nullptr, 9, "theta is NULL");
  GPMP_ARG(r >= 0 && r <= GPMP_MAX_RANK, 13, "r outside [0, GPMP_MAX_RANK]");
  GPMP_ARG(r == 0 || (F != nullptr && G != nullptr && ldf >= r), 11, "F/G NULL or ldf < r with r > 0");
  GPMP_ARG(g_dev != nullptr && ws != nullptr, 15, "g or ws is NULL");
  GradParams gp;
  gp.pp = nullptr;
  gp.nprob = 1; gp.stride_kinv = gp.stride_x = gp.stride_f = gp.stride_partial = 0; gp.ns = nullptr;
  gp.Kinv = M; gp.ldk = ldm; gp.x = x; gp.y = y; gp.F = F; gp.G = G; gp.ldf = ldf;
  gp.n = n; gp.m = m; gp.d = d; gp.r = r;
  gp.ntiles_side = (n + GT - 1) / GT;
  gp.ntiles_c = (m + GT - 1) / GT;
  const long nt = (long)gp.ntiles_side * gp.ntiles_c;
  GPMP_ARG(nt < 0x7FFFFFFFL, 4, "too many tiles");
  gp.ntiles = (int)nt;
  gp.sigma2 = std::exp(theta_host[0]);
  gp.partial = ws;
  const int off = noise ? 2 : 1;
  fill_matern(gp.ms, p);
  for (int k = 0; k < d; ++k) gp.invrho[k] = 2.0 * gp.ms.c * std::exp(theta_host[off + k]);
  fill_fast_exp(gp.fe);
  const int nblocks = gp.ntiles < GRAD_BLOCKS ? gp.ntiles : GRAD_BLOCKS;
  const int dt = grad_tier(d);
  hipStream_t st = as_stream(stream);
  const size_t lds = sizeof(double) * (2 * (size_t)dt * GT + 2 * (size_t)r * GT);
  auto go = [&](auto kern) -> int {
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, st, gp);
    GPMP_HIP_TRY(hipGetLastError());
    return 0;
  };
  int rc = 0;
  switch (dt) {
    case 4: rc = go(grad_trace_kernel<4, false, true>); break;
    case 8: rc = go(grad_trace_kernel<8, false, true>); break;
    case 16: rc = go(grad_trace_kernel<16, false, true>); break;
    case 32: rc = go(grad_trace_kernel<32, false, true>); break;
    default: rc = go(grad_trace_kernel<64, false, true>); break;
  }
  if (rc) return rc;
  hipLaunchKernelGGL(grad_cross_finalize_kernel, dim3(1), dim3(128), 0, st, ws, nblocks, dt + 2, d, gp.sigma2, g_dev);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gpmp_matern_gram_deriv(const double* x, int n, int d, int p, const double* theta_host, int noise,
                                      int jparam, double* out, long ld, gpmp_stream_t stream) {
  GPMP_ARG(x != nullptr, 1, "x is NULL");
  GPMP_ARG(n >= 1 && n <= 65535, 2, "n outside [1, 65535] (diagnostic-sized problems)");
  GPMP_ARG(d >= 1 && d <= GPMP_MAX_DIM, 3, "d outside [1, GPMP_MAX_DIM]");
  GPMP_ARG(p >= 0 && p <= GPMP_MAX_P, 4, "p outside [0, GPMP_MAX_P]");
  GPMP_ARG(theta_host != nullptr, 5, "theta is NULL");
  const int off = noise ? 2 : 1;
  GPMP_ARG(jparam >= 0 && jparam < off + d, 7, "jparam outside the parameter vector");
  GPMP_ARG(out != nullptr && ld >= n, 8, "out is NULL or ld < n");
  DerivParams dp;
  dp.x = x; dp.out = out; dp.ld = ld; dp.n = n; dp.d = d;
  dp.sigma2 = std::exp(theta_host[0]);
  for (int k = 0; k < d; ++k) dp.invrho[k] = std::exp(theta_host[off + k]);
  fill_matern(dp.ms, p);
  const double eps = 2.220446049250313e-16;
  dp.jdim = -1;
  if (jparam == 0) { dp.kind = 0; dp.diag_val = noise ? 0.0 : 10.0 * dp.sigma2 * eps; }
  else if (noise && jparam == 1) { dp.kind = 1; dp.diag_val = std::exp(theta_host[1]); }
  else { dp.kind = 2; dp.jdim = jparam - off; dp.diag_val = 0.0; }
  hipLaunchKernelGGL(gram_deriv_kernel, dim3((n + 255) / 256, n), dim3(256), 0, as_stream(stream), dp);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}
