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

// The whole diagonal-block algorithm for ONE workgroup of 512 threads; S = 91,136 B of LDS.
__device__ __forceinline__ void potf2_inv_body(double* __restrict__ A, long lda, int jb, double* __restrict__ dinv,
                                               int* info, int offset, int do_factor, double* S) {
  double* Td = S + NPACK * 256;                                // [NSB][16][16] diagonal inverse blocks
  double* dg = Td + NSB * 256;                                 // [NB] 1 / L_ii
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lk = lane >> 4;

  // ---- load: lower triangle of the jb x jb block, identity padding, zeros elsewhere
  for (int idx = t; idx < NB * NB; idx += THREADS) {
    const int i = idx >> 7, j = idx & (NB - 1);
    if ((i >> 4) < (j >> 4)) continue;   // sub-block above the diagonal: not stored
    double v = (i == j) ? 1.0 : 0.0;
    if (i < jb && j <= i) v = A[(long)i * lda + j];
    S[pidx(i, j)] = v;
  }
  __syncthreads();

  if (do_factor) {
    for (int j = 0; j < NSB; ++j) {
      const int j0 = j * SB;
      double* Djj = S + ((j * (j + 1) / 2 + j) << 8);   // diagonal sub-block image
      // ---- A1: diagonal 16 x 16 block, wave 0, row (lane & 15) in registers
      if (wave == 0) {
        double a[SB];
#pragma unroll
        for (int c = 0; c < SB; ++c) a[c] = Djj[sidx(lr, c)];
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
          for (int c = 0; c < SB; ++c) Djj[sidx(lr, c)] = (c <= lane) ? a[c] : 0.0;
        }
      }
      __syncthreads();
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
      // ---- A3: trailing update on MFMA: blocks (bi, bk), j < bk <= bi < 8
      {
        const int rem = NSB - 1 - j;
        const int nblk = rem * (rem + 1) / 2;
        for (int b = wave; b < nblk; b += THREADS / 64) {
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
      }
      __syncthreads();
    }
    // factor back to global memory (lower triangle only)
    for (int idx = t; idx < NB * NB; idx += THREADS) {
      const int i = idx >> 7, j = idx & (NB - 1);
      if (i < jb && j <= i) A[(long)i * lda + j] = S[pidx(i, j)];
    }
  } else {
    if (t < NB) dg[t] = 1.0 / S[pidx(t, t)];
    __syncthreads();
  }

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
}

__global__ void __launch_bounds__(THREADS) potf2_inv_kernel(double* __restrict__ A, long lda, int n_total,
                                                            double* __restrict__ dinv, int* info, int offset,
                                                            int do_factor) {
  // one workgroup on the critical path of the factorisation, usually sharing its CU with a GEMM workgroup:
  // ask the CU's arbiter to issue these waves first
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) double S[];   // [NPACK][16][16] packed lower block triangle
  // batched over blockIdx.x: block b works on the diagonal block starting at row/col b * NB
  const int jb = (n_total - (int)blockIdx.x * NB) < NB ? (n_total - (int)blockIdx.x * NB) : NB;
  potf2_inv_body(A + (long)blockIdx.x * NB * (lda + 1), lda, jb, dinv + (long)blockIdx.x * NB * NB, info,
                 offset + blockIdx.x * NB, do_factor, S);
}

// ================================================================================================
// Cholesky of a whole panel square (nb <= 8 diagonal blocks, w = 128 nb) in ONE launch.
// Under the look-ahead scheme the panel chain runs on a second stream while the trailing update fills
// the machine; every launch of that chain then costs 0.1-0.4 ms of queueing (measured, tools/chain_probe.hip:
// about half a tile time of the big kernel per launch, however small the launched kernel is), and the
// launch-per-block scheme needs 3 launches per 128 columns.  Here one workgroup per lower 128 x 128 tile
// runs the tile algorithm with flags in device memory:
//   tile (i, j):  T = A_ij - sum_{k<j} L_ik L_jk^T        (as soon as L_ik, L_jk are flagged ready)
//        i == j:  L_jj, inv(L_jj) by the LDS algorithm above;  G_jj = inv(L_jj)
//        i >  j:  L_ij = T inv(L_jj)^T;  G_ij = -inv(L_ii) L_ij   (G: see panel_solve_kernel, gemm_f64.hip)
// Workgroup ids are column-major over the lower tiles, so a workgroup only ever waits for workgroups with a
// smaller id (dispatched earlier), except for the final G_ij which needs L_ii: by then every workgroup of the
// launch (<= 36) is resident or will be as soon as one slot frees -- nothing it waits for depends on it.
// Flags hold the launch generation, so they never need clearing.
struct SquareParams {
  double* A;        // top-left of the panel square
  long lda;
  int nb;
  double* dinv;     // [nb][128][128]
  double* G;        // w x w workspace, row-major, ld = ldg
  long ldg;
  int* info;
  int offset;       // global row index of A[0][0] (for info)
  unsigned gen;
  unsigned* flags;  // [64]
};

__device__ unsigned g_square_flags[64];
#ifdef GPMP_SQ_TRACE
__device__ long long g_sq_trace[36 * 32];
#define SQ_MARK(slot) do { if (threadIdx.x == 0) g_sq_trace[blockIdx.x * 32 + (slot)] = (long long)wall_clock64(); } while (0)
#else
#define SQ_MARK(slot) do { } while (0)
#endif

__device__ __forceinline__ int frag_kc(int idx, int k) { return idx * 16 + 2 * ((k >> 1) ^ ((idx >> 1) & 7)) + (k & 1); }
__device__ __forceinline__ int frag_mc(int idx, int k) { return k * 128 + 2 * ((idx >> 1) ^ ((k & 1) << 3)) + (idx & 1); }

// acc (128 x 16 slab of wave `wave`: acc[mt][r] = C[16 mt + (lane>>4) + 4 r][16 wave + (lane&15)]) += sgn * A B,
// A: 128 x 128, k-contiguous (ld lda).  B: BKC ? B(k, n) = Bp[n * ldb + k] : B(k, n) = Bp[k * ldb + n].
// Operands stream HBM/L2 -> LDS (buffer_load ... lds) in 16-wide k-tiles, two stages (2 x 32 KB at sm).
template <bool BKC>
__device__ __forceinline__ void mini_gemm(const double* Ap, long lda, const double* Bp, long ldb, d4 (&acc)[8], bool neg,
                                          double* sm, int wave, int lane) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int lr = lane & 15, lk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Ap), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Bp), 0, 0x7FFFFFFF, 0x00020000);
  int voffA[2], voffB[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    voffA[h] = ((lane >> 3) * (int)lda + 2 * ((lane & 7) ^ ((h * 4 + (lane >> 4)) & 7))) * 8;
    if constexpr (BKC) voffB[h] = ((lane >> 3) * (int)ldb + 2 * ((lane & 7) ^ ((h * 4 + (lane >> 4)) & 7))) * 8;
    else voffB[h] = 2 * (lane ^ (h << 3)) * 8;
  }
  auto issue = [&](int kt, int st) {
    double* sa = sm + st * 4096;
    double* sb = sa + 2048;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int rb = 16 * wave + 8 * q;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(sa + rb * 16), 16, voffA[q], (rb * (int)lda + kt * 16) * 8, 0, 0);
      if constexpr (BKC) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(sb + rb * 16), 16, voffB[q], (rb * (int)ldb + kt * 16) * 8, 0, 0);
      } else {
        const int k = 2 * wave + q;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(sb + k * 128), 16, voffB[q], ((kt * 16 + k) * (int)ldb) * 8, 0, 0);
      }
    }
  };
  __syncthreads();   // the staging area is free (previous phase finished reading / writing LDS)
  issue(0, 0);
  __syncthreads();
#pragma unroll 1
  for (int kt = 0; kt < NB / 16; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < NB / 16) issue(kt + 1, st ^ 1);
    const double* sa = sm + st * 4096;
    const double* sb = sa + 2048;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double b = BKC ? sb[frag_kc(16 * wave + lr, 4 * ks + lk)] : sb[frag_mc(16 * wave + lr, 4 * ks + lk)];
      if (neg) b = -b;
      double a[8];
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) a[mt] = sa[frag_kc(16 * mt + lr, 4 * ks + lk)];
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b, acc[mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);   // keep at most one k-step of fragments live (register budget: 128)
    }
    __syncthreads();   // next k-tile landed (vmcnt(0)); stage st may be overwritten
  }
}

__device__ __forceinline__ void square_wait(const unsigned* flag, unsigned gen) {
  if (threadIdx.x == 0) {
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != gen) __builtin_amdgcn_s_sleep(4);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave: no stale L1 / L2 lines of the producer's data
}
__device__ __forceinline__ void square_post(unsigned* flag, unsigned gen) {
  __threadfence();   // this thread's stores are visible device-wide
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

typedef unsigned int sq_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sq_v2u sq_bits(double v) {   // (__builtin_bit_cast of a vector element stored element 0)
  return (sq_v2u){(unsigned)__double2loint(v), (unsigned)__double2hiint(v)};
}

// Second half of an off-diagonal tile, out of line so that it gets its own register allocation (inlined after
// the update loop the allocator kept the dead tile registers and spilled the accumulators of both products):
//   L_ij = T inv(L_jj)^T (T already stored at At), flag it, then G_ij = -inv(L_ii) L_ij.
__device__ __attribute__((noinline)) void square_offdiag_tail(double* At, long lda, const double* dj, const double* di,
                                                              double* Gt, long ldg, unsigned* f_jj, unsigned* f_ij,
                                                              unsigned* f_ii, unsigned gen, double* S) {
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int lane_off = (lk * (int)lda + 16 * wave + lr) * 8, row_step = (int)lda * 8;
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(At, 0, 0x7FFFFFFF, 0x00020000);
  d4 acc[8];
  SQ_MARK(19);
  square_wait(f_jj, gen);
  SQ_MARK(20);
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) acc[mt] = (d4){0.0, 0.0, 0.0, 0.0};
  mini_gemm<true>(At, lda, dj, NB, acc, false, S, wave, lane);
  SQ_MARK(21);
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      __builtin_amdgcn_raw_buffer_store_b64(sq_bits(acc[mt][r]), rsT, lane_off, (16 * mt + 4 * r) * row_step, 0);
  square_post(f_ij, gen);
  SQ_MARK(22);
  if (Gt == nullptr) return;
  square_wait(f_ii, gen);
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) acc[mt] = (d4){0.0, 0.0, 0.0, 0.0};
  mini_gemm<false>(di, NB, At, lda, acc, true, S, wave, lane);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(Gt, 0, 0x7FFFFFFF, 0x00020000);
  const int g_off = (lk * (int)ldg + 16 * wave + lr) * 8, g_step = (int)ldg * 8;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      __builtin_amdgcn_raw_buffer_store_b64(sq_bits(acc[mt][r]), rsG, g_off, (16 * mt + 4 * r) * g_step, 0);
}

__global__ void __launch_bounds__(THREADS, 4) chol_square_kernel(SquareParams p) {
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) double S[];
  int id = blockIdx.x, tj = 0, cnt = p.nb;
  while (id >= cnt) { id -= cnt; ++tj; --cnt; }
  const int ti = tj + id;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  double* At = p.A + (long)ti * NB * p.lda + tj * NB;
  // tile <-> accumulator slab through buffer addressing: one per-lane offset, row offsets are scalar
  // (32 separate 64-bit per-lane pointers would take 64 VGPRs of the 128 available)
  const int lane_off = (lk * (int)p.lda + 16 * wave + lr) * 8;
  const int row_step = (int)p.lda * 8;
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(At, 0, 0x7FFFFFFF, 0x00020000);
  d4 acc[8];
  auto tile_load = [&]() {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[mt][r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsT, lane_off, (16 * mt + 4 * r) * row_step, 0));
  };
  auto tile_store = [&](bool lower_only) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (lower_only && 16 * wave + lr > 16 * mt + 4 * r + lk) continue;
        __builtin_amdgcn_raw_buffer_store_b64(sq_bits(acc[mt][r]), rsT, lane_off, (16 * mt + 4 * r) * row_step, 0);
      }
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) acc[mt] = (d4){0.0, 0.0, 0.0, 0.0};
  };

  SQ_MARK(0);
  tile_load();
  for (int k = 0; k < tj; ++k) {
    square_wait(p.flags + ti * 8 + k, p.gen);
    if (ti != tj) square_wait(p.flags + tj * 8 + k, p.gen);
    SQ_MARK(1 + 2 * k);
    mini_gemm<true>(p.A + (long)ti * NB * p.lda + k * NB, p.lda, p.A + (long)tj * NB * p.lda + k * NB, p.lda, acc, true, S, wave, lane);
    SQ_MARK(2 + 2 * k);
  }

  double* dj = p.dinv + (size_t)tj * NB * NB;
  if (ti == tj) {
    // updated diagonal tile back to memory (lower part), then the LDS algorithm on it
    tile_store(true);
    __syncthreads();
    SQ_MARK(16);
    potf2_inv_body(At, p.lda, NB, dj, p.info, p.offset + tj * NB, 1, S);
    square_post(p.flags + tj * 8 + tj, p.gen);
    SQ_MARK(17);
    if (p.G != nullptr) {
      double* Gjj = p.G + (long)tj * NB * p.ldg + tj * NB;
      for (int e = t; e < NB * NB / 2; e += THREADS) {
        const int r = e >> 6, c = (e & 63) * 2;
        *reinterpret_cast<d2*>(Gjj + (long)r * p.ldg + c) = *reinterpret_cast<const d2*>(dj + r * NB + c);
      }
    }
    SQ_MARK(18);
    return;
  }

  // ---- off-diagonal tile: T to memory (it is the A operand of the next product), the rest out of line
  tile_store(false);
  square_offdiag_tail(At, p.lda, dj, p.dinv + (size_t)ti * NB * NB,
                      p.G != nullptr ? p.G + (long)ti * NB * p.ldg + tj * NB : nullptr, p.ldg, p.flags + tj * 8 + tj,
                      p.flags + ti * 8 + tj, p.flags + ti * 8 + ti, p.gen, S);
}

int launch(double* A, long lda, int n_total, int nblocks, double* dinv, int* info_dev, int offset,
           int do_factor, hipStream_t st) {
  static bool attr_done = false;
  const size_t lds = sizeof(double) * (NPACK * 256 + NSB * 256 + NB);   // 91,136 B
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
int launch_chol_square(double* A, long lda, int nb, double* dinv_panel, double* G, int* info_dev, int offset, hipStream_t st) {
  static bool attr_done = false;
  static unsigned gen = 0;
  static unsigned* flags = nullptr;
  const size_t lds = sizeof(double) * (NPACK * 256 + NSB * 256 + NB);
  if (!attr_done) {
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_square_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GPMP_HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&flags), HIP_SYMBOL(g_square_flags)));
    attr_done = true;
  }
  ++gen;
  if (gen == 0) ++gen;   // 0 is the value of never-written flags
  SquareParams p{A, lda, nb, dinv_panel, G, (long)nb * NB, info_dev, offset, gen, flags};
  ProfScope ps(PK_POTF2, st, (double)nb);
  hipLaunchKernelGGL(chol_square_kernel, dim3(nb * (nb + 1) / 2), dim3(THREADS), lds, st, p);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}
int launch_trtri_blocks(const double* L, long ldl, int n, double* dinv, hipStream_t st) {
  if (n <= 0) return 0;
  return launch(const_cast<double*>(L), ldl, n, (n + NB - 1) / NB, dinv, nullptr, 0, 0, st);
}

}  // namespace gpmp
