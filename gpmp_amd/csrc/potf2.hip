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
//
// The time of the kernel is the dependency chain of the 128 columns, which ONE wave walks (A1 below, 1.7-1.85 us per
// 16 x 16 diagonal sub-block).  Everything else is arranged in that wave's shadow, on the other seven:
//   chain wave (wave 0), for each of the 8 block columns j
//     A1  factors the 16 x 16 diagonal sub-block in registers (MFMA accumulator layout, v_permlane swaps,
//         1/sqrt by v_rsq_f64 + one cubic step, no division)
//     A2  (all waves, one 16 x 16 block of the panel below each; wave 0 takes the block the next diagonal sub-block needs):
//         X L_jj^T = A_panel in the same register layout as A1 -- the block held transposed, four columns per step brought
//         to every lane by one permlane round, uniform multipliers, one MFMA per step (0.8 us; the row-per-thread forward
//         substitution it replaces needed 120 broadcast LDS reads per thread: 1.2 us, and A2 is on the chain)
//     A3, column j's contribution to the next diagonal sub-block only, then straight on to A1(j+1)
//   shadow waves (1..7), while wave 0 is in A1(j)
//     A3       LEFT-looking: block column j of the panel and the diagonal sub-block (j+1, j+1) receive the contributions
//              of all columns k < j, one block per wave (the right-looking form put 27 block updates = 108 MFMAs behind
//              the first column, and an f64 MFMA occupies its SIMD for 64 cycles: A1(1) waited 1.2 us at the barrier)
//     B(j-1)   T_dd = L_dd^-1 of the diagonal sub-block factored one step ago (wave 7) -> LDS
//     C(j-2)   block row j-2 of T = L^-1: wave w owns block COLUMN w-1 of T and keeps it in registers,
//              T_iw = -T_ii sum_k L_ik T_kw.  The MFMA result layout (row = (lane>>4) + 4 r) IS the B-operand
//              layout of the next MFMA, so neither the running sum nor T_kw ever touch LDS.  Stored to dinv at once.
//     L(j-1)   block column j-1 of the factor and T_(j-2)(j-2) -> global (waves 3..6)
//   tail  B(7), C(6) and the sums of C(7) together, then the last multiplication by -T_77: ~3 us after the chain
//         ends (round 1 ran B, C and the stores after the factorisation: 13 of the kernel's 46 us).
// Beside a machine-filling trailing update the kernel takes 180-260 us instead of 31 (tools/potf2_probe.hip busy,
// profiles/r2/potf2_phase_timeline_beside_gemm.log: every phase ~7 times longer, the register-only A1 included).  On gfx950
// the f64 MFMA and the f64 vector ALU have the same peak rate -- they share the FP64 units -- so each dependent v_fma_f64 of the
// chain waits for a gap between the co-resident GEMM wave's back-to-back 64-cycle MFMAs.  Four ways around it were built and
// measured in round 2, none kept: CU-masked streams, a CU partition, a resident workgroup serving the blocks from a
// mailbox (the blocks then take 45 us, but a workgroup that holds a compute unit for the whole factorisation slows the trailing
// update by 11 %), and a per-compute-unit yield table that makes the co-resident GEMM workgroups sleep (kernel body 36 us beside a
// GEMM, but the check costs the GEMM's k-loop 2 % and the wait for a workgroup slot, ~150 us, stays).
// Barriers inside the loop wait for LDS traffic only (lds_barrier): the stores to global memory ride along.
// Phase timeline: tools/potf2_probe.hip (profiles/r2/potf2_phase_timeline_v2.log): load 4, eight steps of ~3.1 us
// (A1 1.8, A2 1.0, A3 0.3), tail 2.8: 31 us of kernel against 40 (start of round 2) and 46 (round 1).
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

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the wave's global stores
// (vmcnt(0)): with the factor and its inverse streamed out DURING the factorisation, every barrier of the chain would wait
// for an HBM write (measured: 3-5 us per 16-column step instead of 2.9).  Everything the waves hand each other goes
// through LDS; the stores only have to land by the end of the kernel.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef GPMP_POTF2_TRACE   // tools/potf2_probe.hip
__device__ long long g_potf2_trace[192];
#define PF_MARK(slot) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_potf2_trace[slot] = (long long)wall_clock64(); } while (0)
#define PF_MARK_T(slot, tid) do { if (threadIdx.x == (tid) && blockIdx.x == 0) g_potf2_trace[slot] = (long long)wall_clock64(); } while (0)
#else
#define PF_MARK(slot) do { } while (0)
#define PF_MARK_T(slot, tid) do { } while (0)
#endif

// ---- load: lower triangle of the jb x jb block, identity padding; sub-blocks above the diagonal are not stored.
// (all 32 loads of a thread are issued before the first LDS store: one exposed latency instead of 32 -- the
//  rolled loop took 16 us of the kernel's 65; phase timeline: tools/potf2_probe.hip)
__device__ __forceinline__ void load_block(const double* __restrict__ A, long lda, int jb, double* S, int t) {
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
    if ((i >> 4) >= (j >> 4)) S[pidx(i, j)] = v[q];
  }
}

// phase B: T_dd = L_dd^-1 of diagonal sub-block d by ONE wave, column (lane & 15) per lane -> Td (LDS)
// (global stores from here -- 16 per lane, one per row -- sent the kernel from 124 to 256 registers and 81 spills: the block
//  goes to dinv from Td one step later, one element per thread)
__device__ __forceinline__ void invert_diag_block(const double* S, double* Td, const double* dg, int d, int lane) {
  const double* Ddd = S + ((d * (d + 1) / 2 + d) << 8);
  double tc[SB];
  const int c = lane & 15;
#pragma unroll
  for (int i = 0; i < SB; ++i) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) s = fma(Ddd[sidx(i, k)], tc[k], s);
    const double ri = dg[d * SB + i];
    tc[i] = (c == i) ? ri : -ri * s;   // lanes with c > i get exactly 0 (all their t_k are 0)
  }
  if (lane < SB) {
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      Td[(d << 8) + sidx(i, c)] = tc[i];
    }
  }
}

// The factorisation of one block by the 512 threads of a workgroup; S = 91,136 bytes of LDS.  Every wave's stores to A
// and dinv have been ISSUED when it returns (a kernel end, or the caller's vmcnt(0) + fence, completes them).
__device__ __forceinline__ void potf2_body(double* __restrict__ A, long lda, int jb, double* __restrict__ dinv, int* info, int offset,
                                           double* S, const int t) {
  double* Td = S + NPACK * 256;                                // [NSB][16][16] diagonal inverse blocks
  double* dg = Td + NSB * 256;                                 // [NB] 1 / L_ii
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lk = lane >> 4;

  PF_MARK(0);
  if (wave == 0) __builtin_amdgcn_s_setprio(3);     // the chain wave above the shadow waves (its SIMD is shared with wave 4)
  load_block(A, lda, jb, S, t);

  __syncthreads();
  PF_MARK(1);

  // A3(j): trailing update on MFMA, A_ik -= L_ij L_kj^T for the sub-blocks j < gk <= gi < 8, enumerated b = 0, 1, ...
  // with b = 0 the NEXT diagonal block (j+1, j+1).  Wave 0 does b = 0 right after A2(j) and goes on to factor that block
  // (A1(j+1)); the other seven waves do the rest meanwhile, so the update hides behind the register factorisation.
  auto a3_blocks = [&](int j, int b_first, int b_step, int b_end_excl) __attribute__((always_inline)) {
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


  // The rest of the trailing update, LEFT-looking, by the seven shadow waves while wave 0 is in A1(m): block column m of the
  // panel (rows m+1..7) and the diagonal sub-block (m+1, m+1) receive the contributions of ALL columns k < m in one go -- one
  // block per wave, a chain of 4 m MFMAs, accumulator read and written once.  (The right-looking form updated all
  // (7-j)(8-j)/2 trailing blocks right after column j: 27 blocks at j = 0, and the f64 MFMA issues once per 64 cycles per
  // SIMD -- 0.85 us of pure MFMA time on the busiest SIMD, 2.0 us measured, with A1 waiting at the barrier.)
  // Column k = m of the diagonal sub-block (m+1, m+1) is wave 0's (a3_blocks, b = 0), right after A2(m).
  auto a3_left = [&](int m) __attribute__((always_inline)) {
    const int q = wave - 1;
    if (q > 7 - m || m + 1 >= NSB) return;
    const int gi = q < 7 - m ? m + 1 + q : m + 1;      // target block row
    const int gc = q < 7 - m ? m : m + 1;              // target block column
    double* Cb = S + ((gi * (gi + 1) / 2 + gc) << 8);
    d4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = Cb[sidx(lk + 4 * r, lr)];
    for (int k = 0; k < m; ++k) {
      const double* Lik = S + ((gi * (gi + 1) / 2 + k) << 8);
      const double* Lck = S + ((gc * (gc + 1) / 2 + k) << 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lik[sidx(lr, 4 * s + lk)], Lck[sidx(lr, 4 * s + lk)], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Cb[sidx(lk + 4 * r, lr)] = acc[r];
  };

  // A2(j) for ONE 16 x 16 block below the diagonal sub-block, by one wave, in the same register layout as A1: the block is
  // held TRANSPOSED in the MFMA accumulator layout (lane (g, i) = (lane >> 4, lane & 15) holds R[i][g + 4 r]), so register s
  // of lane group g is column 4 s + g of row i -- as in A1, one permlane round brings the four columns of a step to every
  // lane, they are scaled / eliminated with UNIFORM multipliers (entries of L_jj, broadcast LDS reads), and the rank-4 update
  // of the remaining columns is one MFMA whose A operand is the stored image of L_jj (zero above the diagonal, so finished
  // columns receive exactly 0).  0.4 us per block against 1.2 us for the row-per-thread substitution it replaces (120
  // broadcast LDS reads per thread), and A2 is on the kernel's critical chain.
  auto a2_block = [&](int j, int bi, bool invert) __attribute__((always_inline)) {
    const int g = lane >> 4, i16 = lane & 15;
    const double* Djj = S + ((j * (j + 1) / 2 + j) << 8);
    double* Rb = S + ((bi * (bi + 1) / 2 + j) << 8);
    // invert: R = identity, result X = L_jj^-T = T_jj^T goes, transposed, to the image of T_jj in Td (phase B of round 1
    // solved 16 rows one after the other with 120 broadcast LDS reads per lane: 1.2 us, and it ends the kernel's tail)
    d4 acc2;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc2[r] = invert ? (i16 == g + 4 * r ? 1.0 : 0.0) : Rb[sidx(i16, g + 4 * r)];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      double x[4];
      {
        const double v = acc2[s4];
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const auto la = __builtin_amdgcn_permlane32_swap(l16[0], l16[0], false, false);
        const auto ha = __builtin_amdgcn_permlane32_swap(h16[0], h16[0], false, false);
        const auto lb = __builtin_amdgcn_permlane32_swap(l16[1], l16[1], false, false);
        const auto hb = __builtin_amdgcn_permlane32_swap(h16[1], h16[1], false, false);
        x[0] = __hiloint2double(ha[0], la[0]);
        x[1] = __hiloint2double(hb[0], lb[0]);
        x[2] = __hiloint2double(ha[1], la[1]);
        x[3] = __hiloint2double(hb[1], lb[1]);
      }
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const int c = 4 * s4 + t4;
        x[t4] *= dg[j * SB + c];
#pragma unroll
        for (int tp = t4 + 1; tp < 4; ++tp) x[tp] = fma(-x[t4], Djj[sidx(4 * s4 + tp, c)], x[tp]);
      }
      const double xg = g == 0 ? x[0] : g == 1 ? x[1] : g == 2 ? x[2] : x[3];
      if (s4 < 3) {
        const double lz = Djj[sidx(i16, 4 * s4 + g)];      // L_jj[i][4 s + g], zero above the diagonal
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-lz, xg, acc2, 0, 0, 0);
      }
      acc2[s4] = xg;
    }
    if (invert) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Td[(j << 8) + sidx(g + 4 * r, i16)] = acc2[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) Rb[sidx(i16, g + 4 * r)] = acc2[r];
    }
  };

  if (wave == 0) {
    // ================= chain wave =================
#pragma nounroll
    for (int j = 0; j < NSB; ++j) {
      const int j0 = j * SB;
      double* Djj = S + ((j * (j + 1) / 2 + j) << 8);   // diagonal sub-block image
      // ---- A1: diagonal 16 x 16 block in the MFMA accumulator layout: lane (g, i) = (lane >> 4, lane & 15) holds
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
      {
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
      }
      lds_barrier();                         // (1) L_jj, dg: A2(j), B(j) may start
      PF_MARK(8 + 3 * j);
      if (j == NSB - 1) break;
      a2_block(j, j + 1, false);             // A2(j), block row j + 1 (the one the next diagonal sub-block needs)
      lds_barrier();                         // (2) A2(j) done
      PF_MARK(9 + 3 * j);
      a3_blocks(j, 0, 1, 1);                 // the next diagonal sub-block
      PF_MARK(10 + 3 * j);
    }
    PF_MARK(2);
    // tail, first half: the last diagonal sub-block of the factor -> global
    for (int idx = lane; idx < SB * SB; idx += 64) {
      const int i = NB - SB + (idx >> 4), c = NB - SB + (idx & 15);
      if (i < jb && c <= i) A[(long)i * lda + c] = S[pidx(i, c)];
    }
    lds_barrier();                           // (T) B(7), C(6), sums of C(7) done
  } else {
    // ================= shadow waves =================
    const int cw = wave - 1;                 // block column of T owned by this wave (0..6; column 7 is T_77 alone)
    d4 tcol[NSB];                            // tcol[i] = T_{i,cw} (MFMA C layout), i > cw
    // T row i of this wave's column, without the final multiplication by -T_ii
    auto c_row_sum = [&](int i) __attribute__((always_inline)) -> d4 {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      {
        const double* Lij = S + ((i * (i + 1) / 2 + cw) << 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double af = Lij[sidx(lr, 4 * s + lk)];                    // L_i,cw[row][k]
          const double bf = Td[(cw << 8) + sidx(4 * s + lk, lr)];         // T_cw,cw[k][col]
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int bk = 1; bk < NSB - 1; ++bk) {
        if (bk > cw && bk < i) {
          const double* Lik = S + ((i * (i + 1) / 2 + bk) << 8);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double af = Lik[sidx(lr, 4 * s + lk)];                  // L_ik[row][k]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, tcol[bk][s], acc, 0, 0, 0);   // B = T_k,cw registers
          }
        }
      }
      return acc;
    };
    // ... times -T_ii (acc register r is exactly the B fragment of k-step r), kept and stored to dinv
    auto c_row_finish = [&](int i, d4 acc) __attribute__((always_inline)) {
      d4 res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double af = -Td[(i << 8) + sidx(lr, 4 * s + lk)];           // -T_ii[row][k]
        res = __builtin_amdgcn_mfma_f64_16x16x4f64(af, acc[s], res, 0, 0, 0);
      }
#pragma unroll
      for (int q = 1; q < NSB - 1; ++q) {                                  // (row 7 is never an operand)
        const bool hit = (q == i);      // selects, not a branch: "if (q == i) tcol[q] = res" is folded into ONE dynamically
#pragma unroll                          // indexed store, which moves the whole array to scratch memory
        for (int r = 0; r < 4; ++r) tcol[q][r] = hit ? res[r] : tcol[q][r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dinv[(i * SB + lk + 4 * r) * NB + cw * SB + lr] = res[r];
    };
    // block column d of the factor -> global, by the 256 threads of waves 3..6
    auto store_l_col = [&](int d) __attribute__((always_inline)) {
      const int tt = t - 192, rr = tt >> 4, c = tt & 15;
      double* ap = A + (long)(d * SB + rr) * lda + d * SB + c;
      if (rr >= c && d * SB + rr < jb) *ap = S[((d * (d + 1) / 2 + d) << 8) + sidx(rr, c)];     // diagonal sub-block: lower triangle
      for (int bi = d + 1; bi < NSB; ++bi) {
        ap += (long)SB * lda;
        if (bi * SB + rr < jb) *ap = S[((bi * (bi + 1) / 2 + d) << 8) + sidx(rr, c)];
      }
    };
    // T_dd (in Td since the step before) -> dinv, same threads
    auto store_t_diag = [&](int d) __attribute__((always_inline)) {
      const int tt = t - 192;
      dinv[(d * SB + (tt >> 4)) * NB + d * SB + (tt & 15)] = Td[(d << 8) + sidx(tt >> 4, tt & 15)];
    };
    // zeros above the diagonal of dinv (block column bj: rows 0 .. 16 bj - 1), while wave 0 factors the first diagonal sub-block
#pragma unroll
    for (int bj = 1; bj < NSB; ++bj)
      for (int idx = t - 64; idx < bj * 256; idx += THREADS - 64) dinv[(idx >> 4) * NB + bj * SB + (idx & 15)] = 0.0;
    PF_MARK_T(64, 448);
#pragma nounroll
    for (int j = 0; j < NSB; ++j) {
      if (j >= 1) {
        a3_left(j);
        PF_MARK_T(64 + 4 * j, 448); PF_MARK_T(96 + 4 * j, 256);
        if (wave == 7) invert_diag_block(S, Td, dg, j - 1, lane);   // B(j-1)
        PF_MARK_T(65 + 4 * j, 448);
        if (wave >= 3 && wave <= 6) {
          store_l_col(j - 1);
          if (j >= 2) store_t_diag(j - 2);
        }
        PF_MARK_T(97 + 4 * j, 256);
      }
      if (j - 2 > cw) c_row_finish(j - 2, c_row_sum(j - 2));
      PF_MARK_T(66 + 4 * j, 448); PF_MARK_T(98 + 4 * j, 256);
      lds_barrier();                         // (1)
      if (j == NSB - 1) break;
      // ---- A2(j): block row j + 1 + wave (wave 0 has block row j + 1)
      if (j + 1 + wave < NSB) a2_block(j, j + 1 + wave, false);
      lds_barrier();                         // (2)
    }
    // tail, first half: T_77; block row 6 of T (T_66 is in Td since step 7) and the sums of row 7
    if (wave == 7) invert_diag_block(S, Td, dg, NSB - 1, lane);
    if (wave >= 3 && wave <= 6) store_t_diag(6);
    if (6 > cw) c_row_finish(6, c_row_sum(6));
    const d4 sum7 = c_row_sum(7);
    lds_barrier();                           // (T)
    c_row_finish(7, sum7);
    if (wave >= 3 && wave <= 6) store_t_diag(7);
  }
  PF_MARK(6);
}

// (at most 128 registers: two waves per SIMD of this kernel must fit beside ONE resident wave of the 232-register GEMM)
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) potf2_inv_kernel(double* __restrict__ A, long lda, int n_total,
                                                            double* __restrict__ dinv, int* info, int offset,
                                                            long prob_stride_a, long prob_stride_dinv) {
  // batched over blockIdx.y: independent matrices (problems) prob_stride_a / prob_stride_dinv elements apart, one info word each
  extern __shared__ __attribute__((aligned(16))) double S[];   // [NPACK][16][16] packed lower block triangle + Td + dg
  potf2_body(A + (long)blockIdx.y * prob_stride_a, lda, n_total < NB ? n_total : NB, dinv + (long)blockIdx.y * prob_stride_dinv,
             info + blockIdx.y, offset, S, (int)threadIdx.x);
}

// inv(L_dd) of the diagonal blocks of an already factored matrix (one workgroup per block; phases B and C after one another)
__global__ void __launch_bounds__(THREADS) trtri_blocks_kernel(const double* __restrict__ A, long lda, int n_total,
                                                               double* __restrict__ dinv) {
  A += (long)blockIdx.x * NB * (lda + 1);
  dinv += (long)blockIdx.x * NB * NB;
  const int jb = (n_total - (int)blockIdx.x * NB) < NB ? (n_total - (int)blockIdx.x * NB) : NB;
  extern __shared__ __attribute__((aligned(16))) double S[];
  double* Td = S + NPACK * 256;
  double* dg = Td + NSB * 256;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  load_block(A, lda, jb, S, t);
  __syncthreads();
  if (t < NB) dg[t] = 1.0 / S[pidx(t, t)];
  __syncthreads();
  invert_diag_block(S, Td, dg, wave, lane);
  __syncthreads();
  // wave w computes block column w of T = L^-1, keeps it in registers, streams it out
  const int bj = wave;
  d4 tcol[NSB];
#pragma unroll
  for (int bi = 1; bi < NSB; ++bi) {
    if (bi > bj) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      {
        const double* Lij = S + ((bi * (bi + 1) / 2 + bj) << 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double af = Lij[sidx(lr, 4 * s + lk)];
          const double bf = Td[(bj << 8) + sidx(4 * s + lk, lr)];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int bk = 1; bk < NSB; ++bk) {
        if (bk > bj && bk < bi) {
          const double* Lik = S + ((bi * (bi + 1) / 2 + bk) << 8);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double af = Lik[sidx(lr, 4 * s + lk)];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, tcol[bk][s], acc, 0, 0, 0);
          }
        }
      }
      d4 res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double af = -Td[(bi << 8) + sidx(lr, 4 * s + lk)];
        res = __builtin_amdgcn_mfma_f64_16x16x4f64(af, acc[s], res, 0, 0, 0);
      }
      tcol[bi] = res;
    }
  }
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

constexpr size_t POTF2_LDS = sizeof(double) * (NPACK * 256 + NSB * 256 + NB);   // 91,136 B

int launch_factor(double* A, long lda, int jb, double* dinv, int* info_dev, int offset, hipStream_t st, int nprob, long prob_stride_a,
                  long prob_stride_dinv) {
  static DeviceOnce attr_once;
  if (const long long dev_bit = attr_once.need()) {
    if (dev_bit < 0) { set_error("hipGetDevice failed or device ordinal above 62"); return -1; }
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_inv_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTF2_LDS));
    attr_once.done(dev_bit);
  }
  {
    ProfScope ps(PK_POTF2, st, (double)nprob);
    hipLaunchKernelGGL(potf2_inv_kernel, dim3(1, nprob), dim3(THREADS), POTF2_LDS, st, A, lda, jb, dinv, info_dev, offset,
                       prob_stride_a, prob_stride_dinv);
  }
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

int launch_potf2_inv(double* A, long lda, int jb, double* dinv, int* info_dev, int offset, hipStream_t st) {
  return launch_factor(A, lda, jb, dinv, info_dev, offset, st, 1, 0, 0);
}
// the same diagonal block of `nprob` independent matrices (batched small problems, drivers_batch.hip)
int launch_potf2_inv_batch(double* A, long lda, int jb, double* dinv, int* info_dev, int offset, int nprob, long stride_a,
                           long stride_dinv, hipStream_t st) {
  return launch_factor(A, lda, jb, dinv, info_dev, offset, st, nprob, stride_a, stride_dinv);
}
int launch_trtri_blocks(const double* L, long ldl, int n, double* dinv, hipStream_t st) {
  if (n <= 0) return 0;
  static DeviceOnce attr_once;
  if (const long long dev_bit = attr_once.need()) {
    if (dev_bit < 0) { set_error("hipGetDevice failed or device ordinal above 62"); return -1; }
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(trtri_blocks_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTF2_LDS));
    attr_once.done(dev_bit);
  }
  const int nblocks = (n + NB - 1) / NB;
  {
    ProfScope ps(PK_POTF2, st, (double)nblocks);
    hipLaunchKernelGGL(trtri_blocks_kernel, dim3(nblocks), dim3(THREADS), POTF2_LDS, st, L, ldl, n, dinv);
  }
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace gpmp
