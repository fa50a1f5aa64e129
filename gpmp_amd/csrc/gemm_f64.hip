// fp64 MFMA GEMM for gfx950 (MI355X) -- the one dense contraction behind the blocked Cholesky
// (trailing syrk update, panel scaling), the blocked triangular solves, trtri and lauum.
//
// Design (measured on MI355X, see DESIGN.md / tools/mfma_f64_probe2.hip):
//  * v_mfma_f64_16x16x4_f64 issues back-to-back every 64 cycles from ONE wave *only in its VGPR
//    form*; with the accumulator in AGPRs it issues every ~147 cycles.  The library is therefore
//    built with -mllvm -amdgpu-mfma-vgpr-form=1 and keeps accumulators in architectural VGPRs.
//  * 128 x 128 x 16 block tile, 256 threads = 4 waves (2 x 2), each wave a 64 x 64 sub-tile =
//    4 x 4 MFMA tiles = 128 accumulator VGPRs; 2 workgroups per CU.
//  * Operands staged global -> registers -> LDS (double buffered, one barrier per k-tile).
//    LDS images are padded so that the MFMA fragment reads (ds_read_b64) are bank-conflict free:
//      k-contiguous operand:  [128 rows][16 k] with row stride 18 doubles
//      m/n-contiguous operand: [16 k][128 cols] with row stride 144 doubles
//  * Fragment / accumulator lane maps of v_mfma_f64_16x16x4_f64 (verified on hardware):
//      A: lane l holds A[row = l & 15][k = l >> 4];  B: lane l holds B[k = l >> 4][col = l & 15];
//      C/D register r of lane l: row = (l >> 4) + 4 r, col = l & 15.
//  * blockIdx -> tile map is XCD aware: each XCD (own L2) gets a contiguous run of tiles.
#include "common.h"
#include <cstdlib>

namespace gpmp {

int g_machine_busy = 0;   // gpmp_hint_machine_busy: small NT products take the small-footprint kernel

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDK = 18;    // row stride of a k-contiguous tile image
constexpr int LDN = 144;   // row stride of an m/n-contiguous tile image
constexpr int TILE = 2304; // doubles per operand tile image (128*18 == 16*144)
constexpr int NXCD = 8;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

struct GemmParams {
  const double* A;
  const double* B;
  double* C;
  long lda, ldb, ldc;
  int M, N, K;
  double alpha, beta;
  int lower_only, kstart_row, kend_row, kend_col;
  int kstart_col;   // k loop of tile column j starts at col0(j) (B lower triangular: B(l, j) = 0 for l < j)
  int tiles_m, tiles_n, ntiles;
  int aligned;  // 16-byte loads allowed on A and B
  int lean;     // NT, K <= 512: small-footprint kernel (see gemm_nt_lean_kernel)
  int batch;    // gridDim.y independent products
  int batch2;   // gridDim.z independent problems (outer batch)
  int tri_block;  // lower-triangular tile sets in 8 x 8 super-tiles (round 4) instead of row by row
  int bn;         // v2, plain NN launches: tile width 128 / 112 / 96 (launch_t)
  int early;      // v2: request tile kt + 2 right behind the barrier of tile kt (64 MFMAs of cover) instead of at the top of tile kt + 1 (48)
  int stair_num, stair_den, stair_off, stair_sub;   // staircase tile set (GemmOpts): columns of tile-row group g; den > 0 enables
  int kg_rnum, kg_rden, kg_roff, kg_cnum, kg_cden, kg_coff;   // contraction start per group of 8 tile rows / columns (GemmOpts)
  long sa, sb, sc;   // element strides of A, B, C per batch index
  long sa2, sb2, sc2;   // ... per outer batch index (blockIdx.z)
};

// tile columns of the group g of 8 tile rows of a staircase tile set (host and device)
__host__ __device__ __forceinline__ int stair_cols(int num, int den, int off, int sub, int tiles_n, int g) {
  const int a = num * g + off;
  if (a < 0) return 0;
  const int c = 8 * (a / den + 1 - sub);
  return c < 0 ? 0 : (c > tiles_n ? tiles_n : c);
}

// first k of a tile under GemmOpts::kg_*: 1024 max(0, ceil((num g + off) / den)) for its row group and its column group
__device__ __forceinline__ int group_kstart(const GemmParams& p, int ti, int tj) {
  int kb = 0;
  if (p.kg_rden > 0) {
    const int a = p.kg_rnum * (ti >> 3) + p.kg_roff;
    if (a > 0) kb = 1024 * ((a + p.kg_rden - 1) / p.kg_rden);
  }
  if (p.kg_cden > 0) {
    const int a = p.kg_cnum * (tj >> 3) + p.kg_coff;
    if (a > 0) { const int k2 = 1024 * ((a + p.kg_cden - 1) / p.kg_cden); if (k2 > kb) kb = k2; }
  }
  return kb;
}

__device__ __forceinline__ void decode_tile(const GemmParams& p, int bid, int& ti, int& tj) {
  // XCD-aware remap (bijective for any grid size): workgroups are dealt round-robin over the 8
  // XCDs, so give XCD x the x-th contiguous chunk of the tile list.
  const int nwg = p.ntiles;
  const int q = nwg / NXCD, r = nwg % NXCD;
  const int xcd = bid % NXCD;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  // tiles of unequal cost (triangular k ranges) are dealt round-robin instead: a contiguous chunk per
  // XCD would hand one XCD all the long tiles (measured on lauum: 29 -> 50+ TFLOP/s)
  const int v = (p.kstart_row | p.kend_row | p.kstart_col | p.kend_col | p.kg_rden | p.kg_cden) ? bid : base + bid / NXCD;
  if (p.lower_only) {
    const int tn = p.tiles_n < p.tiles_m ? p.tiles_n : p.tiles_m;
    const int t1 = tn * (tn + 1) / 2;
    if (v < t1 && p.tri_block) {
      // Round 4: the triangle in SUPER-TILES of 8 x 8 tiles, super-row by super-row, so that the 64 co-resident workgroups of an
      // XCD (a contiguous run of v) cover an 8 x 8 block of tiles -- 16 panel k-slices per step instead of the 65 of a 1 x 64
      // strip of the row-by-row order (a quarter of the L2-side traffic of the Cholesky's trailing update).  A full super-row I
      // holds 64 I + 36 tiles (I off-diagonal squares + the diagonal triangle); 32 I^2 + 4 I tiles come before it.  Only the
      // last super-row can be ragged (fewer than 8 tile rows).
      int I = (int)((sqrt(16.0 + 128.0 * (double)v) - 4.0) * (1.0 / 64.0));
      while (32 * (I + 1) * (I + 1) + 4 * (I + 1) <= v) ++I;
      while (32 * I * I + 4 * I > v) --I;
      const int w = v - (32 * I * I + 4 * I);
      const int rows = (tn - 8 * I) < 8 ? (tn - 8 * I) : 8;
      const int sq = rows * 8;                       // tiles of one off-diagonal super-tile of this super-row
      if (w < I * sq) {
        const int J = w / sq, x = w - J * sq;
        ti = 8 * I + x % rows;
        tj = 8 * J + x / rows;
      } else {
        const int w2 = w - I * sq;
        int i = (int)((sqrt(8.0 * (double)w2 + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= w2) ++i;
        while (i * (i + 1) / 2 > w2) --i;
        ti = 8 * I + i;
        tj = 8 * I + (w2 - i * (i + 1) / 2);
      }
    } else if (v < t1) {
      int i = (int)((sqrt(8.0 * (double)v + 1.0) - 1.0) * 0.5);
      while ((i + 1) * (i + 2) / 2 <= v) ++i;
      while (i * (i + 1) / 2 > v) --i;
      ti = i;
      tj = v - i * (i + 1) / 2;
    } else if (p.tri_block) {
      // the rectangle below the triangle (M > N): groups of 8 tile rows, row fastest inside a group
      const int rr = v - t1;
      const int g = rr / (8 * tn);
      const int first = tn + 8 * g;
      const int gsize = (p.tiles_m - first) < 8 ? (p.tiles_m - first) : 8;
      const int w = rr - g * 8 * tn;
      ti = first + w % gsize;
      tj = w / gsize;
    } else {
      const int rr = v - t1;
      ti = tn + rr / tn;
      tj = rr % tn;
    }
  } else if (p.stair_den) {
    // staircase: group by group (8 tile rows each), row fastest inside a group, the group's own number of tile columns
    int g = 0, acc = 0, gsize = 8, cols = 0;
    for (;; ++g) {
      gsize = (p.tiles_m - 8 * g) < 8 ? (p.tiles_m - 8 * g) : 8;
      cols = stair_cols(p.stair_num, p.stair_den, p.stair_off, p.stair_sub, p.tiles_n, g);
      if (v < acc + gsize * cols || 8 * (g + 1) >= p.tiles_m) break;
      acc += gsize * cols;
    }
    const int w = v - acc;
    ti = 8 * g + w % gsize;
    tj = w / gsize;
  } else if (p.kstart_col | p.kend_row) {
    // unequal k ranges: longest tiles first, so that the last workgroups to start are the short ones
    // (kstart_col: tile column 0 has the full k range; kend_row: the last tile row has it)
    if (p.kstart_col) {
      tj = v / p.tiles_m;
      ti = v - tj * p.tiles_m;
    } else {
      const int r = v / p.tiles_n;
      ti = p.tiles_m - 1 - r;
      tj = v - r * p.tiles_n;
    }
  } else {
    constexpr int GM = 8;      // tile rows per group: the 64 co-resident workgroups of an XCD cover an 8 x 8 block of tiles
    const int per_group = GM * p.tiles_n;
    const int g = v / per_group;
    const int first = g * GM;
    const int gsize = (p.tiles_m - first) < GM ? (p.tiles_m - first) : GM;
    const int w = v - g * per_group;
    ti = first + w % gsize;
    tj = w / gsize;
  }
}

// ---- global -> register staging --------------------------------------------------------------
// KC: element (idx, k) at G[idx * ld + k].  Thread t loads k pair (t & 7), rows (t >> 3) + 32 s.
template <bool FAST>
__device__ __forceinline__ void load_kc(const double* __restrict__ G, long ld, int idx0, int lim,
                                        int k0, int kend, int t, d2 (&r)[4]) {
  const int kp = (t & 7) * 2;
  const int rr = t >> 3;
  if constexpr (FAST) {
    // buffer loads: uniform descriptor on the tile's first row, ONE per-lane byte offset, the four
    // row groups and the k position go into the scalar offset -- no per-load VALU address math.
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(G + (long)idx0 * ld), 0, 0x7FFFFFFF, 0x00020000);
    const int voff = (rr * (int)ld + kp) * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (k0 + 32 * s * (int)ld) * 8, 0);
      r[s] = __builtin_bit_cast(d2, raw);
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = idx0 + rr + 32 * s;
    const double* ptr = G + (long)row * ld + k0 + kp;
    if constexpr (FAST) {
      r[s] = *reinterpret_cast<const d2*>(ptr);
    } else {
      d2 v = {0.0, 0.0};
      if (row < lim) {
        if (k0 + kp < kend) v[0] = ptr[0];
        if (k0 + kp + 1 < kend) v[1] = ptr[1];
      }
      r[s] = v;
    }
  }
}
// MC: element (idx, k) at G[k * ld + idx].  Thread t loads column pair (t & 63), k rows (t >> 6) + 4 s.
template <bool FAST>
__device__ __forceinline__ void load_mc(const double* __restrict__ G, long ld, int idx0, int lim,
                                        int k0, int kend, int t, d2 (&r)[4]) {
  const int cp = (t & 63) * 2;
  const int kr = t >> 6;
  if constexpr (FAST) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(G + (long)k0 * ld + idx0), 0, 0x7FFFFFFF, 0x00020000);
    const int voff = (kr * (int)ld + cp) * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (4 * s * (int)ld) * 8, 0);
      r[s] = __builtin_bit_cast(d2, raw);
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = k0 + kr + 4 * s;
    const double* ptr = G + (long)k * ld + idx0 + cp;
    if constexpr (FAST) {
      r[s] = *reinterpret_cast<const d2*>(ptr);
    } else {
      d2 v = {0.0, 0.0};
      if (k < kend) {
        if (idx0 + cp < lim) v[0] = ptr[0];
        if (idx0 + cp + 1 < lim) v[1] = ptr[1];
      }
      r[s] = v;
    }
  }
}
__device__ __forceinline__ void store_kc(double* __restrict__ S, int t, const d2 (&r)[4]) {
  const int kp = (t & 7) * 2;
  const int rr = t >> 3;
#pragma unroll
  for (int s = 0; s < 4; ++s) *reinterpret_cast<d2*>(S + (rr + 32 * s) * LDK + kp) = r[s];
}
__device__ __forceinline__ void store_mc(double* __restrict__ S, int t, const d2 (&r)[4]) {
  const int cp = (t & 63) * 2;
  const int kr = t >> 6;
#pragma unroll
  for (int s = 0; s < 4; ++s) *reinterpret_cast<d2*>(S + (kr + 4 * s) * LDN + cp) = r[s];
}

template <bool AKC, bool BKC, bool CACC>
__global__ void __launch_bounds__(256, 2) gemm_f64_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [2 buffers][A tile | B tile]
  int ti, tj;
  p.A += (long)blockIdx.y * p.sa + (long)blockIdx.z * p.sa2;   // batched launch: one independent product per blockIdx.y
  p.B += (long)blockIdx.y * p.sb + (long)blockIdx.z * p.sb2;
  p.C += (long)blockIdx.y * p.sc + (long)blockIdx.z * p.sc2;
  decode_tile(p, blockIdx.x, ti, tj);
  // the tile coordinates are wave-uniform but come out of VALU code (sqrt in the triangular decode):
  // pin them to SGPRs so every tile base address below is scalar
  ti = __builtin_amdgcn_readfirstlane(ti);
  tj = __builtin_amdgcn_readfirstlane(tj);
  const int row0 = ti * BM, col0 = tj * BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int lr = lane & 15, lk = lane >> 4;

  int kbeg = p.kstart_row ? row0 : 0;
  if (p.kstart_col) {
    const int kb = col0 & ~(BK - 1);
    if (kb > kbeg) kbeg = kb;
  }
  if (p.kg_rden | p.kg_cden) {
    const int kb = group_kstart(p, ti, tj);
    if (kb > kbeg) kbeg = kb;
  }
  int kend = p.K;
  if (p.kend_row && row0 + BM < kend) kend = row0 + BM;
  if (p.kend_col && col0 + BN < kend) kend = col0 + BN;
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const bool full_mn = p.aligned && (row0 + BM <= p.M) && (col0 + BN <= p.N);

  // Accumulators start from (beta / alpha) * C (CACC), so the epilogue is a pure store: the C reads
  // overlap the pipeline fill instead of serialising after the last MFMA.  Addressing: one uniform
  // (SGPR) row base per accumulator row + ONE 32-bit per-lane offset shared by all 64 accesses.
  const double alpha = p.alpha, beta = p.beta;
  const bool full_c = (row0 + BM <= p.M) && (col0 + BN <= p.N);
  double* __restrict__ cbase = p.C + (long)row0 * p.ldc + col0;
  const unsigned lane_off = (unsigned)((wm + lk) * (int)p.ldc + wn + lr);
  d4 acc[4][4];
  if constexpr (CACC) {
    const double sc = beta / alpha;
    if (full_c) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double* rp = cbase + (long)(i * 16 + 4 * r) * p.ldc;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j][r] = sc * rp[lane_off + j * 16];
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + wm + i * 16 + lk + 4 * r;
          const double* rp = cbase + (long)(i * 16 + 4 * r) * p.ldc;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int col = col0 + wn + j * 16 + lr;
            acc[i][j][r] = (row < p.M && col < p.N) ? sc * rp[lane_off + j * 16] : 0.0;
          }
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  }

  d2 ra[4], rb[4];
  auto load_tiles = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    if (full_mn && k0 + BK <= kend) {
      if constexpr (AKC) load_kc<true>(p.A, p.lda, row0, p.M, k0, kend, t, ra);
      else load_mc<true>(p.A, p.lda, row0, p.M, k0, kend, t, ra);
      if constexpr (BKC) load_kc<true>(p.B, p.ldb, col0, p.N, k0, kend, t, rb);
      else load_mc<true>(p.B, p.ldb, col0, p.N, k0, kend, t, rb);
    } else {
      if constexpr (AKC) load_kc<false>(p.A, p.lda, row0, p.M, k0, kend, t, ra);
      else load_mc<false>(p.A, p.lda, row0, p.M, k0, kend, t, ra);
      if constexpr (BKC) load_kc<false>(p.B, p.ldb, col0, p.N, k0, kend, t, rb);
      else load_mc<false>(p.B, p.ldb, col0, p.N, k0, kend, t, rb);
    }
  };
  auto store_tiles = [&](int buf) {
    double* sa = smem + buf * 2 * TILE;
    double* sb = sa + TILE;
    if constexpr (AKC) store_kc(sa, t, ra); else store_mc(sa, t, ra);
    if constexpr (BKC) store_kc(sb, t, rb); else store_mc(sb, t, rb);
  };

  if (nk > 0) {
    load_tiles(0);
    store_tiles(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);  // global loads fly while the MFMAs below run
    const double* sa = smem + cur * 2 * TILE;
    const double* sb = sa + TILE;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (AKC) a[i] = sa[(wm + i * 16 + lr) * LDK + ks * 4 + lk];
        else a[i] = sa[(ks * 4 + lk) * LDN + wm + i * 16 + lr];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (BKC) b[j] = sb[(wn + j * 16 + lr) * LDK + ks * 4 + lk];
        else b[j] = sb[(ks * 4 + lk) * LDN + wn + j * 16 + lr];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C = alpha * acc  (+ beta * C when the accumulators did not start from C)
  if (full_c) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* rp = cbase + (long)(i * 16 + 4 * r) * p.ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double v = alpha * acc[i][j][r];
          if constexpr (!CACC) { if (beta != 0.0) v += beta * rp[lane_off + j * 16]; }
          rp[lane_off + j * 16] = v;
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wm + i * 16 + lk + 4 * r;
        double* rp = cbase + (long)(i * 16 + 4 * r) * p.ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = col0 + wn + j * 16 + lr;
          if (row < p.M && col < p.N) {
            double v = alpha * acc[i][j][r];
            if constexpr (!CACC) { if (beta != 0.0) v += beta * rp[lane_off + j * 16]; }
            rp[lane_off + j * 16] = v;
          }
        }
      }
  }
}

// ================================================================================================
// v2 main loop: operands go HBM -> LDS directly (buffer_load ... lds, no staging registers, no
// ds_write), into UNPADDED images whose bank conflicts are removed by an XOR swizzle applied on the
// per-lane SOURCE offset and on the fragment reads (the LDS-DMA destination is lane-linear):
//   k-contiguous operand  [128 rows][16 k]:   16-byte slot s of row r holds k pair  s ^ ((r >> 1) & 7)
//   m/n-contiguous operand [16 k][128 cols]:  16-byte slot s of k-row k holds column pair  s ^ ((k & 1) << 3)
// The barrier of each k-tile sits between k-steps 2 and 3: after it the fragments of the NEXT tile's
// k-steps 0,1 are fetched from LDS while the 16 MFMAs of k-step 3 (fragments already in registers)
// keep the matrix pipe busy, so no ds_read latency is exposed after the barrier.
// Full tiles only (M, N multiples of 128, k range a multiple of 16, 16-byte aligned operands).
constexpr int V2_TILE = 2048;   // doubles per operand image (unpadded)

__device__ __forceinline__ double swap_lane_xor1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <bool KC>
__device__ __forceinline__ int v2_frag_addr(int idx, int k) {
  if constexpr (KC) return idx * 16 + 2 * ((k >> 1) ^ ((idx >> 1) & 7)) + (k & 1);
  else return k * 128 + 2 * ((idx >> 1) ^ ((k & 1) << 3)) + (idx & 1);
}

// BNT = columns of a tile: 128 (waves 2 x 2, 64 x 64 each) or 112 / 96 (waves 4 x 1, 32 x BNT each: the layout of the narrow solve
// leaves).  Narrower tiles exist for ONE reason: a launch lasts ceil(tiles / (2 CUs)) rounds of the machine, and with N = 50000
// (391 tile columns of 128) the solve's updates of 512 ... 4096 rows end on a nearly empty round -- 447 columns of 112 fit better
// (launch_t picks the width that minimises rounds x width).  Everything else -- operand images, request schedule, k loop -- is shared.
template <bool AKC, bool BKC, bool CACC, int BNT = BN>
__global__ void __launch_bounds__(256, 2) gemm_f64_kernel_v2(GemmParams p) {
  constexpr int MI = BNT == BN ? 4 : 2;            // 16-row MFMA tiles per wave
  constexpr int NJ = BNT == BN ? 4 : BNT / 16;     // 16-column MFMA tiles per wave
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [2 buffers][A image | B image] = 64 KB
  int ti, tj;
  p.A += (long)blockIdx.y * p.sa + (long)blockIdx.z * p.sa2;   // batched launch: one independent product per blockIdx.y
  p.B += (long)blockIdx.y * p.sb + (long)blockIdx.z * p.sb2;
  p.C += (long)blockIdx.y * p.sc + (long)blockIdx.z * p.sc2;
  decode_tile(p, blockIdx.x, ti, tj);
  ti = __builtin_amdgcn_readfirstlane(ti);
  tj = __builtin_amdgcn_readfirstlane(tj);
  const int row0 = ti * BM, col0 = tj * BNT;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = BNT == BN ? (wave >> 1) * 64 : wave * 32, wn = BNT == BN ? (wave & 1) * 64 : 0;
  const int lr = lane & 15, lk = lane >> 4;
  // edge tiles (M, N not multiples of 128; both even): operand rows are clamped / clipped so that nothing is read
  // outside the matrices, the main loop is the same, and only valid rows / columns of C are read and written
  const int rows_v = (p.M - row0) < BM ? (p.M - row0) : BM;
  const int cols_v = (p.N - col0) < BNT ? (p.N - col0) : BNT;
  const bool edge = (rows_v < BM) || (cols_v < BNT);

  int kbeg = p.kstart_row ? row0 : 0;
  if (p.kstart_col) {
    const int kb = col0 & ~(BK - 1);
    if (kb > kbeg) kbeg = kb;
  }
  if (p.kg_rden | p.kg_cden) {
    const int kb = group_kstart(p, ti, tj);
    if (kb > kbeg) kbeg = kb;
  }
  int kend = p.K;
  if (p.kend_row && row0 + BM < kend) kend = row0 + BM;
  if (p.kend_col && col0 + BNT < kend) kend = col0 + BNT;
  const int nk = kend > kbeg ? (kend - kbeg) / BK : 0;

  // ---- accumulators (start from (beta/alpha) C, see v1)
  const double alpha = p.alpha, beta = p.beta;
  double* __restrict__ cbase = p.C + (long)row0 * p.ldc + col0;
  const unsigned lane_off = (unsigned)((wm + lk) * (int)p.ldc + wn + lr);
  d4 acc[MI][NJ];
  // C is read either up front (accumulators start from (beta/alpha) C) or, when the k loop is long
  // enough, one 16x16 MFMA tile per k-tile during the first 16 k-tiles: every workgroup of a round
  // starts its tile at the same time, and 512 simultaneous 128 KB reads are an HBM-rate burst that
  // nothing hides (measured: the up-front read costs exactly C / HBM bandwidth at K = 512).
  const bool spread = CACC && nk >= 18 && !edge;
  if (CACC && !spread) {
    const double sc = beta / alpha;
    if (!edge) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double* rp = cbase + (long)(i * 16 + 4 * r) * p.ldc;
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j][r] = sc * rp[lane_off + j * 16];
        }
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double* rp = cbase + (long)(i * 16 + 4 * r) * p.ldc;
          const bool rok = wm + i * 16 + 4 * r + lk < rows_v;
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j][r] = (rok && wn + j * 16 + lr < cols_v) ? sc * rp[lane_off + j * 16] : 0.0;
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  }

  // ---- per-lane source offsets (bytes), one per load instruction.  Rows (k-contiguous operand) or column pairs
  // (m/n-contiguous operand) past the edge are redirected to a valid one: every load stays inside the matrix and the
  // duplicates only feed rows / columns of C that are never stored.
  int voffA[4], voffB[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if constexpr (AKC) {
      int r = wave * 32 + s * 8 + (lane >> 3);
      r = r < rows_v ? r : rows_v - 1;
      voffA[s] = (r * (int)p.lda + 2 * ((lane & 7) ^ (((s & 1) * 4 + (lane >> 4)) & 7))) * 8;
    } else {
      const int cp = lane ^ ((s & 1) << 3);
      voffA[s] = 2 * (2 * cp < rows_v ? cp : 0) * 8;
    }
    if constexpr (BKC) {
      int r = wave * 32 + s * 8 + (lane >> 3);
      r = r < cols_v ? r : cols_v - 1;
      voffB[s] = (r * (int)p.ldb + 2 * ((lane & 7) ^ (((s & 1) * 4 + (lane >> 4)) & 7))) * 8;
    } else {
      const int cp = lane ^ ((s & 1) << 3);
      voffB[s] = 2 * (2 * cp < cols_v ? cp : 0) * 8;
    }
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;

  auto issue = [&](int kt, int buf) {
    const int k0 = kbeg + kt * BK;
    double* sa = smem + buf * 2 * V2_TILE;
    double* sb = sa + V2_TILE;
    if constexpr (AKC) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.A + (long)row0 * p.lda), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(sa + (wave * 32 + s * 8) * 16), 16, voffA[s], k0 * 8, 0, 0);
    } else {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.A + (long)k0 * p.lda + row0), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(sa + (wave * 4 + s) * 128), 16, voffA[s],
                                                 ((wave * 4 + s) * (int)p.lda) * 8, 0, 0);
    }
    if constexpr (BKC) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.B + (long)col0 * p.ldb), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(sb + (wave * 32 + s * 8) * 16), 16, voffB[s], k0 * 8, 0, 0);
    } else {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.B + (long)k0 * p.ldb + col0), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(sb + (wave * 4 + s) * 128), 16, voffB[s],
                                                 ((wave * 4 + s) * (int)p.ldb) * 8, 0, 0);
    }
  };

  double fa[4][MI], fb[4][NJ];   // [k-step][fragment], statically indexed after unrolling
  auto read_frags = [&](int ks, int buf) {
    const double* sa = smem + buf * 2 * V2_TILE;
    const double* sb = sa + V2_TILE;
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[ks][i] = sa[v2_frag_addr<AKC>(wm + i * 16 + lr, ks * 4 + lk)];
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[ks][j] = sb[v2_frag_addr<BKC>(wn + j * 16 + lr, ks * 4 + lk)];
  };
  auto mfma_step = [&](int ks) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
  };

  // early: tile kt + 2 is requested right behind the barrier of tile kt -- buffer `cur` is free there, every fragment of it was read
  // before the barrier -- instead of at the top of tile kt + 1: 64 MFMAs of cover for the global -> LDS loads instead of 48.
  // (Round 4.  A second barrier per k-tile right after the last fragment read would give 96; measured: it loses 4-5 points.)
  const bool early = p.early != 0;
  auto ktile = [&](int kt, bool more, auto&& inject) {
    const int cur = kt & 1;
    if (more && !early) issue(kt + 1, cur ^ 1);
    // at most three k-steps of fragments are live at any point (48 VGPRs)
    read_frags(2, cur);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(0);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(3, cur);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(1);
    mfma_step(2);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();   // next tile landed (vmcnt(0) of every wave) and buffer `cur` is free for tile kt+2
    if (early && kt + 2 < nk) issue(kt + 2, cur);
    if (more) {
      read_frags(0, cur ^ 1);
      read_frags(1, cur ^ 1);
    }
    inject();
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(3);
  };

  auto nothing = [] {};

  if (nk > 0) {
    issue(0, 0);
    if (early && nk > 1) issue(1, 1);
    __syncthreads();
    read_frags(0, 0);
    read_frags(1, 0);
    int kt = 0;
    if constexpr (CACC) {
      if (spread) {
        const double sc = beta / alpha;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, 0x7FFFFFFF, 0x00020000);
        const int cvoff = (int)lane_off * 8;
        const int ldc8 = (int)p.ldc * 8;
#pragma unroll
        for (int T = 0; T < MI * NJ; ++T) {
          const int i = T / NJ, j = T % NJ;
          // the four rows of MFMA tile (i, j) are requested before this k-tile's operand loads and have
          // landed by the k-tile's barrier (vmcnt(0)); they are folded in just before k-step 3
          typedef unsigned int v2u __attribute__((ext_vector_type(2)));
          v2u craw[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            craw[r] = __builtin_amdgcn_raw_buffer_load_b64(rc, cvoff, (i * 16 + 4 * r) * ldc8 + j * 128, 0);
          ktile(T, true, [&] {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = fma(sc, __builtin_bit_cast(double, craw[r]), acc[i][j][r]);
          });
        }
        kt = MI * NJ;
      }
    }
    for (; kt < nk; ++kt) ktile(kt, kt + 1 < nk, nothing);
  }

  if (!edge && (CACC || beta == 0.0)) {
    // Pure store of a full tile with 16-byte stores: in the MFMA layout a lane holds ONE column of four rows (4 apart), so
    // neighbouring lanes (columns c, c + 1) swap half of their registers (DPP quad_perm [1,0,3,2]): the even lane then owns
    // rows r = 0, 1 of both columns, the odd lane rows r = 2, 3 -- 32 store instructions per thread instead of 64.
    const bool odd = lane & 1;
    const unsigned off2 = lane_off - (odd ? 1u : 0u);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const double a0 = alpha * acc[i][j][0], a1 = alpha * acc[i][j][1], a2 = alpha * acc[i][j][2], a3 = alpha * acc[i][j][3];
        const double g0 = swap_lane_xor1(odd ? a0 : a2), g1 = swap_lane_xor1(odd ? a1 : a3);
        const d2 v0 = odd ? (d2){g0, a2} : (d2){a0, g0};
        const d2 v1 = odd ? (d2){g1, a3} : (d2){a1, g1};
        double* rp0 = cbase + (long)(i * 16 + (odd ? 8 : 0)) * p.ldc + off2 + j * 16;
        *reinterpret_cast<d2*>(rp0) = v0;
        *reinterpret_cast<d2*>(rp0 + 4 * p.ldc) = v1;
      }
  } else if (!edge) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* rp = cbase + (long)(i * 16 + 4 * r) * p.ldc;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          double v = alpha * acc[i][j][r];
          if constexpr (!CACC) { if (beta != 0.0) v += beta * rp[lane_off + j * 16]; }
          rp[lane_off + j * 16] = v;
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* rp = cbase + (long)(i * 16 + 4 * r) * p.ldc;
        const bool rok = wm + i * 16 + 4 * r + lk < rows_v;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (rok && wn + j * 16 + lr < cols_v) {
            double v = alpha * acc[i][j][r];
            if constexpr (!CACC) { if (beta != 0.0) v += beta * rp[lane_off + j * 16]; }
            rp[lane_off + j * 16] = v;
          }
        }
      }
  }
}

// ================================================================================================
// Leaf of the recursive forward solve  X = L^-1 B  (many right-hand sides), fused over the leaf's 128-row
// blocks.  With G the block lower-triangular matrix of the leaf
//   G_jj = inv(L_jj),   G_ji = -inv(L_jj) L_ji  (i < j)            (build_leaf_g_kernel below)
// forward substitution reads  X_j = G[j, 0 : 128 (j+1)] * [X_0; ..; X_{j-1}; B_j], i.e. for each j ONE product
// whose right operand is simply the first 128 (j+1) rows of the column strip in memory (solved blocks followed
// by the untouched block j).  One workgroup per 128-column strip runs the v2 main loop nb times, K = 128 (j+1),
// writing block j in place before block j+1 starts; no synchronisation between workgroups.  Same arithmetic as
// the launch-per-block leaf (inverses of the 128 x 128 diagonal blocks only) with K = 128..512 instead of 128
// and one launch instead of seven.
struct LeafSolveParams {
  const double* G;  // (128 nb) x (128 nb), row-major, ld = ldg
  long ldg;
  double* B;        // leaf rows of the right-hand sides
  long ldb;
  int nb;
  int m;            // right-hand sides; the last strip may be narrower than 128
};

// BNW = columns of a strip: 128 (waves 2 x 2, 64 x 64 each) or 112 / 96 / 80 / 64 / 32 / 16 (waves 4 x 1, 32 x BNW each).  The narrow form halves the
// work of a workgroup and doubles their number: a leaf has only m / BNW workgroups, and at m = 10000 (config 2) 79 of them
// leave two thirds of the machine idle for the 0.17 ms the four sequential products take.
template <int BNW>
__global__ void __launch_bounds__(256, 2) trsm_leaf_kernel(LeafSolveParams p) {
  constexpr int MI = BNW == 128 ? 4 : 2;      // 16-row MFMA tiles per wave
  constexpr int NJ = BNW == 128 ? 4 : BNW / 16;   // 16-column MFMA tiles per wave (strips of 64 / 32 / 16 columns: waves 4 x 1)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = BNW == 128 ? (wave >> 1) * 64 : wave * 32, wn = BNW == 128 ? (wave & 1) * 64 : 0;
  const int lr = lane & 15, lk = lane >> 4;
  double* __restrict__ strip = p.B + (long)blockIdx.x * BNW;
  const unsigned lane_off = (unsigned)((wm + lk) * (int)p.ldb + wn + lr);
  // a narrow last strip (m even): column pairs past m are redirected to the strip's first pair, so every load stays
  // inside B, and only columns < m are stored -- columns are independent
  const int ncv = (p.m - (int)blockIdx.x * BNW) < BNW ? (p.m - (int)blockIdx.x * BNW) : BNW;

  int voffA[2], voffB[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    voffA[h] = ((lane >> 3) * (int)p.ldg + 2 * ((lane & 7) ^ ((h * 4 + (lane >> 4)) & 7))) * 8;
    const int cp = lane ^ (h << 3);
    voffB[h] = 2 * (2 * cp < ncv ? cp : 0) * 8;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(strip, 0, 0x7FFFFFFF, 0x00020000);

  double fa[4][MI], fb[4][NJ];
  d4 acc[MI][NJ];
  for (int jb = 0; jb < p.nb; ++jb) {
    const int nk = (jb + 1) * (BM / BK);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(p.G + (long)jb * BM * p.ldg), 0, 0x7FFFFFFF, 0x00020000);
    auto issue = [&](int kt, int buf) {
      const int k0 = kt * BK;
      double* sa = smem + buf * 2 * V2_TILE;
      double* sb = sa + V2_TILE;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(sa + (wave * 32 + s * 8) * 16), 16, voffA[s & 1],
                                                 ((wave * 32 + s * 8) * (int)p.ldg + k0) * 8, 0, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(sb + (wave * 4 + s) * 128), 16, voffB[s & 1],
                                                 ((k0 + wave * 4 + s) * (int)p.ldb) * 8, 0, 0);
    };
    auto read_frags = [&](int ks, int buf) {
      const double* sa = smem + buf * 2 * V2_TILE;
      const double* sb = sa + V2_TILE;
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[ks][i] = sa[v2_frag_addr<true>(wm + i * 16 + lr, ks * 4 + lk)];
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[ks][j] = sb[v2_frag_addr<false>(wn + j * 16 + lr, ks * 4 + lk)];
    };
    auto mfma_step = [&](int ks) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

    issue(0, 0);
    __syncthreads();
    read_frags(0, 0);
    read_frags(1, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      const bool more = kt + 1 < nk;
      if (more) issue(kt + 1, cur ^ 1);
      read_frags(2, cur);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(0);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(3, cur);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(1);
      mfma_step(2);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      if (more) {
        read_frags(0, cur ^ 1);
        read_frags(1, cur ^ 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(3);
    }
    // every wave has consumed block j of the strip (the barrier of the last k-tile): overwrite it
    double* cb = strip + (long)jb * BM * p.ldb;
    if (ncv == BNW) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double* rp = cb + (long)(i * 16 + 4 * r) * p.ldb;
#pragma unroll
          for (int j = 0; j < NJ; ++j) rp[lane_off + j * 16] = acc[i][j][r];
        }
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double* rp = cb + (long)(i * 16 + 4 * r) * p.ldb;
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            if (wn + j * 16 + lr < ncv) rp[lane_off + j * 16] = acc[i][j][r];
        }
    }
    // block j is an operand of block j+1: stores complete and visible to the workgroup before the next loads
    __threadfence_block();
    __syncthreads();
  }
}

// G of one leaf: grid (nb, nb), block (j, i) with i <= j.  One workgroup = one 128 x 128 block, MFMA straight
// from global memory (L2-resident operands, 4 MFLOP per block: latency matters, not rate).
struct LeafGParams {
  const double* L;     // top-left of the leaf's triangular block
  long ldl;
  const double* dinv;  // inverse diagonal blocks of the leaf, [nb][128][128]
  double* G;
  long ldg;
};

__global__ void __launch_bounds__(256) build_leaf_g_kernel(LeafGParams p) {
  const int jb = blockIdx.x, ib = blockIdx.y;
  if (ib > jb) return;
  const int t = threadIdx.x;
  const double* __restrict__ D = p.dinv + (size_t)jb * BM * BM;
  double* __restrict__ out = p.G + (long)jb * BM * p.ldg + ib * BN;
  if (ib == jb) {
    for (int e = t; e < BM * BN / 2; e += 256) {
      const int r = e / (BN / 2), c = (e % (BN / 2)) * 2;
      *reinterpret_cast<d2*>(out + (long)r * p.ldg + c) = *reinterpret_cast<const d2*>(D + r * BM + c);
    }
    return;
  }
  const double* __restrict__ Lb = p.L + (long)jb * BM * p.ldl + ib * BN;   // L_ji
  const int lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int lr = lane & 15, lk = lane >> 4;
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < BM; k0 += 4) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = D[(wm + i * 16 + lr) * BM + k0 + lk];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = Lb[(long)(k0 + lk) * p.ldl + wn + j * 16 + lr];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        out[(long)(wm + i * 16 + lk + 4 * r) * p.ldg + wn + j * 16 + lr] = -acc[i][j][r];
}

// ================================================================================================
// Latency kernel for the panel chain of the Cholesky (NT products with K = 128 on a grid far smaller than the
// machine: A21 <- A21 inv(L_kk)^T and the rank-128 updates inside a panel).  A 128 x 128 x 128 tile is 4.2 MFLOP,
// i.e. 14 us of MFMA on one CU however fast its operands arrive (measured 22 us per launch for ANY tile count up to
// 256); 32 x 128 tiles put four times as many CUs on the same product.  A tile spans 128 columns so that the
// in-place panel scaling (C == A, N == K == 128) stays race-free: a workgroup reads its 32 rows completely before
// it writes them and nobody else touches them.  All operands of up to 8 k-tiles are requested before the first is
// used, so one load latency is exposed per launch.  Edge tiles and lower-only products are predicated.
constexpr int SBM = 32, SBN = 128, SLD = 18;   // tile shape, padded row stride of a [rows][16] operand image

// SBMT = 16: half-height tiles for launches of fewer than ~one workgroup per compute unit (twice the workgroups, half the
// MFMA time of each: the chain-bound tail of the Cholesky)
template <int SBMT>
__global__ void __launch_bounds__(256) gemm_nt_small_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) double sA[2][SBMT * SLD];
  __shared__ __attribute__((aligned(16))) double sB[2][SBN * SLD];
  const int row0 = blockIdx.y * SBMT, col0 = blockIdx.x * SBN;
  if (p.lower_only && col0 > row0 + SBMT - 1) return;
  // these kernels carry the panel chain of the Cholesky: their waves go first where they share a SIMD with the trailing
  // update's (whose 64-cycle MFMAs otherwise take turns with them one for one)
  __builtin_amdgcn_s_setprio(3);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave * 32;
  const int lr = lane & 15, lk = lane >> 4;
  // staging: thread t owns k quad (t & 3) of row (t >> 2) of A (t < 128) and of rows (t >> 2), (t >> 2) + 64 of B
  const int srow = t >> 2, skq = (t & 3) * 4;
  const bool has_a = t < 4 * SBMT;
  int ar = row0 + (has_a ? srow : 0), br0 = col0 + srow, br1 = col0 + srow + 64;
  ar = ar < p.M ? ar : p.M - 1;     // clamped rows only feed outputs that are never stored
  br0 = br0 < p.N ? br0 : p.N - 1;
  br1 = br1 < p.N ? br1 : p.N - 1;
  const double* __restrict__ ap = p.A + (long)ar * p.lda + skq;
  const double* __restrict__ bp0 = p.B + (long)br0 * p.ldb + skq;
  const double* __restrict__ bp1 = p.B + (long)br1 * p.ldb + skq;

  constexpr int MI = SBMT / 16;
  d4 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  const int nk = p.K / BK;
  for (int kc = 0; kc < nk; kc += 8) {
    const int cnt = (nk - kc) < 8 ? (nk - kc) : 8;
    d2 ra[8][2], rb[8][4];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q < cnt) {
        const int ko = (kc + q) * BK;
        if (has_a) {
          ra[q][0] = *reinterpret_cast<const d2*>(ap + ko);
          ra[q][1] = *reinterpret_cast<const d2*>(ap + ko + 2);
        }
        rb[q][0] = *reinterpret_cast<const d2*>(bp0 + ko);
        rb[q][1] = *reinterpret_cast<const d2*>(bp0 + ko + 2);
        rb[q][2] = *reinterpret_cast<const d2*>(bp1 + ko);
        rb[q][3] = *reinterpret_cast<const d2*>(bp1 + ko + 2);
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q < cnt) {
        const int buf = q & 1;
        if (has_a) {
          *reinterpret_cast<d2*>(&sA[buf][srow * SLD + skq]) = ra[q][0];
          *reinterpret_cast<d2*>(&sA[buf][srow * SLD + skq + 2]) = ra[q][1];
        }
        *reinterpret_cast<d2*>(&sB[buf][srow * SLD + skq]) = rb[q][0];
        *reinterpret_cast<d2*>(&sB[buf][srow * SLD + skq + 2]) = rb[q][1];
        *reinterpret_cast<d2*>(&sB[buf][(srow + 64) * SLD + skq]) = rb[q][2];
        *reinterpret_cast<d2*>(&sB[buf][(srow + 64) * SLD + skq + 2]) = rb[q][3];
        __syncthreads();   // image q complete; image q-1 (other buffer) was released by the previous barrier
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          double fa[MI], fb[2];
#pragma unroll
          for (int i = 0; i < MI; ++i) fa[i] = sA[buf][(16 * i + lr) * SLD + 4 * ks + lk];
#pragma unroll
          for (int j = 0; j < 2; ++j) fb[j] = sB[buf][(wn + 16 * j + lr) * SLD + 4 * ks + lk];
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();   // both images free before the next chunk overwrites them
  }

  const double alpha = p.alpha, beta = p.beta;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 16 * i + 4 * r + lk;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = col0 + wn + 16 * j + lr;
        if (row < p.M && col < p.N) {
          double* cp = p.C + (long)row * p.ldc + col;
          double v = alpha * acc[i][j][r];
          if (beta != 0.0) v += beta * *cp;
          *cp = v;
        }
      }
    }
}

// ---- small-footprint NT kernel (K <= 512): the helper-stream products of the look-ahead Cholesky ------------------
// While the trailing update fills the machine, every CU holds two workgroups of gemm_f64_kernel_v2: 2 x 64 KB of LDS
// and 2 x 232 VGPRs per SIMD lane.  A kernel of the ordinary kind (73 KB / 230 VGPRs) launched on the helper stream
// then waits for one of them to retire -- about one 218 us tile time per launch, whatever its size
// (tools/coresident_probe.hip: 131 us per dependent launch, 20 us for a kernel that fits into what is left: 32 KB of
// LDS and 48 VGPRs).  This kernel fits: 32 x 128 tiles (a workgroup reads its 32 rows completely before it writes
// them, so the in-place panel scaling stays race-free), operands HBM -> LDS by LDS-DMA (no staging registers) into
// ONE 20 KB buffer with the swizzle of the v2 kernel, 32 accumulator VGPRs per wave.  It trades the deep prefetch of
// gemm_nt_small_kernel (one exposed load latency per launch) for starting at once beside the big GEMM.
// (waves_per_eu(10, 10) is how the 48-VGPR budget is imposed: 512 / 10 rounded down to the allocation granule.  The hardware
//  runs at most 8 waves per SIMD, so the compiler reports the occupancy target as unreachable; the register budget is what
//  the attribute is for, hence the diagnostic is silenced for this one kernel.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(10, 10))) gemm_nt_lean_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) double sA[SBM * BK];
  __shared__ __attribute__((aligned(16))) double sB[SBN * BK];
  const int row0 = blockIdx.y * SBM, col0 = blockIdx.x * SBN;
  if (p.lower_only && col0 > row0 + SBM - 1) return;
  // these kernels carry the panel chain of the Cholesky: their waves go first where they share a SIMD with the trailing
  // update's (whose 64-cycle MFMAs otherwise take turns with them one for one)
  __builtin_amdgcn_s_setprio(3);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave * 32;
  const int lr = lane & 15, lk = lane >> 4;
  const int rows_v = (p.M - row0) < SBM ? (p.M - row0) : SBM;
  const int cols_v = (p.N - col0) < SBN ? (p.N - col0) : SBN;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  // per-lane source offsets (bytes): lane l of a 1 KiB piece fills 16-byte slot (l & 7) of image row 8 q + (l >> 3);
  // that slot holds k pair (l & 7) ^ ((row >> 1) & 7).  The buffer descriptors end with the last valid row of the
  // tile, so pieces that reach past the edge of the matrix read zeros (they only feed outputs that are never stored).
  const int ra = wave * 8 + (lane >> 3);
  const int voffA = (ra * (int)p.lda + 2 * ((lane & 7) ^ ((ra >> 1) & 7))) * 8;
  // B: the four pieces of a wave are rows 32 w + 8 s + (l >> 3); s and s + 2 share a swizzle, 16 rows apart
  int voffB[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int rb = wave * 32 + s * 8 + (lane >> 3);
    voffB[s] = (rb * (int)p.ldb + 2 * ((lane & 7) ^ ((rb >> 1) & 7))) * 8;
  }
  const int ldb16 = 16 * (int)p.ldb * 8;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.A + (long)row0 * p.lda), 0,
                                                                        (int)((long)rows_v * p.lda * 8), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p.B + (long)col0 * p.ldb), 0,
                                                                        (int)((long)cols_v * p.ldb * 8), 0x00020000);

  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  const int nk = p.K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int k0 = kt * BK;
    if (kt) __syncthreads();        // every wave has finished reading the previous images
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(sA + wave * 8 * 16), 16, voffA, k0 * 8, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(sB + (wave * 32 + s * 8) * 16), 16, voffB[s & 1],
                                               k0 * 8 + (s >> 1) * ldb16, 0, 0);
    __syncthreads();                // vmcnt(0) of every wave: both images have landed
#pragma unroll 1                     // one k-step of fragments live at a time: the kernel must stay within 48 VGPRs
    for (int ks = 0; ks < 4; ++ks) {
      double fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = sA[v2_frag_addr<true>(16 * i + lr, 4 * ks + lk)];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = sB[v2_frag_addr<true>(wn + 16 * j + lr, 4 * ks + lk)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }

  const double alpha = p.alpha, beta = p.beta;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 16 * i + 4 * r + lk;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = col0 + wn + 16 * j + lr;
        if (row < p.M && col < p.N) {
          double* cp = p.C + (long)row * p.ldc + col;
          double v = alpha * acc[i][j][r];
          if (beta != 0.0) v += beta * *cp;
          *cp = v;
        }
      }
    }
}

#pragma clang diagnostic pop

template <bool AKC, bool BKC, bool CACC>
int launch_t(const GemmParams& p, hipStream_t st) {
  static DeviceOnce attr_once;
  const size_t lds = sizeof(double) * 4 * TILE;        // v1: 73,728 B
  const size_t lds2 = sizeof(double) * 4 * V2_TILE;    // v2: 65,536 B
  if (const long long dev_bit = attr_once.need()) {
    if (dev_bit < 0) { set_error("hipGetDevice failed or device ordinal above 62"); return -1; }
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<AKC, BKC, CACC>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_v2<AKC, BKC, CACC>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_v2<AKC, BKC, CACC, 112>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_v2<AKC, BKC, CACC, 96>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    attr_once.done(dev_bit);
  }
  static int use_v2 = -1;
  if (use_v2 < 0) { const char* e = getenv("GPMP_GEMM_V2"); use_v2 = e ? atoi(e) : 1; }
  // v2 (LDS-direct loads) needs aligned operands, even M and N, and a k range made of whole 16-wide tiles
  // (v2 wins from K = 512 up: 94 % vs 88-92 % of peak at K = 4096; below that its longer fill costs more)
  const bool v2ok = use_v2 && p.K >= 512 && p.aligned && (p.M % 2 == 0) && (p.N % 2 == 0) && (p.K % BK == 0) && p.K > 0 &&
                    ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) && (p.ldc % 2 == 0) &&
                    (p.lda % 2 == 0) && (p.ldb % 2 == 0) && (p.ldc >= p.N) &&
                    ((long)BM * p.lda * 8 + (long)p.K * 8 < 0x7FFFFFFFL) && ((long)BN * p.ldb * 8 + (long)p.K * 8 < 0x7FFFFFFFL);
  // tile width of a plain launch on the LDS-direct kernel (round 4): the width among 128 / 112 / 96 that minimises
  // rounds x width, rounds = ceil(tiles / (2 workgroups x CUs))
  GemmParams pf = p;
  if (v2ok && p.batch == 1 && p.batch2 == 1 && !p.lower_only && !p.stair_den && !(p.kstart_row | p.kend_row | p.kstart_col | p.kend_col | p.kg_rden | p.kg_cden) && p.N >= 16 * BN) {
    const long slots = 2L * device_cu_count();
    auto cost = [&](int w) { const long t = (long)p.tiles_m * ((p.N + w - 1) / w); return ((t + slots - 1) / slots) * (long)w; };
    long best = cost(BN);
    for (int w : {112, 96}) {
      const long c = cost(w);
      if (c * 100 < best * 97) { best = c; pf.bn = w; }
    }
    if (pf.bn != BN) {
      pf.tiles_n = (p.N + pf.bn - 1) / pf.bn;
      pf.ntiles = pf.tiles_m * pf.tiles_n;
    }
  }
  // small NT products (K <= 512): the small-footprint kernel beside a machine-filling update (p.lean), the latency kernel on a grid
  // below one round of the machine (at most `small_max` 128 x 128 tiles at K < 512)
  constexpr int small_max = 128;
  const bool lean_nt = p.batch == 1 && p.batch2 == 1 && AKC && BKC && p.lean && p.aligned && (p.K % BK == 0) && p.K >= BK && p.K <= 512 &&
                       !(p.kstart_row | p.kend_row | p.kstart_col | p.kend_col) && ((long)SBN * p.ldb * 8 + (long)p.K * 8 < 0x7FFFFFFFL) &&
                       ((long)SBM * p.lda * 8 + (long)p.K * 8 < 0x7FFFFFFFL);
  const bool small_nt = !lean_nt && p.batch == 1 && p.batch2 == 1 && AKC && BKC && p.aligned && (p.K % BK == 0) && p.K >= BK && p.K <= 512 &&
                        p.ntiles <= (p.K >= 512 ? 384 : small_max) && !(p.kstart_row | p.kend_row | p.kstart_col | p.kend_col);
  {
    // executed flops of this launch (tiles actually visited, k range actually swept)
    const double kavg = (p.kstart_row || p.kend_row || p.kstart_col || p.kend_col || p.kg_rden || p.kg_cden) ? 0.5 * p.K : (double)p.K;
    ProfScope ps((AKC ? (BKC ? PK_GEMM_NT : PK_GEMM_NN) : (BKC ? PK_GEMM_TT : PK_GEMM_TN)) + ((v2ok && !small_nt) ? 8 : 0), st,
                 (!p.lower_only && !p.stair_den && !(p.kstart_row | p.kend_row | p.kstart_col | p.kend_col | p.kg_rden | p.kg_cden))
                     ? 2.0 * (double)p.M * (double)p.N * (double)p.K * p.batch * p.batch2      // a plain product: its own flops
                     : 2.0 * (double)p.ntiles * BM * BN * kavg * p.batch * p.batch2);
    if (lean_nt) hipLaunchKernelGGL(gemm_nt_lean_kernel, dim3((p.N + SBN - 1) / SBN, (p.M + SBM - 1) / SBM), dim3(256), 0, st, p);
    else if (small_nt) {
      // fewer 32-row tiles than ~compute units: 16-row tiles
      constexpr long below16 = 160;
      const long wg32 = (long)((p.N + SBN - 1) / SBN) * ((p.M + SBM - 1) / SBM);
      if (wg32 < below16) hipLaunchKernelGGL(gemm_nt_small_kernel<16>, dim3((p.N + SBN - 1) / SBN, (p.M + 15) / 16), dim3(256), 0, st, p);
      else hipLaunchKernelGGL(gemm_nt_small_kernel<SBM>, dim3((p.N + SBN - 1) / SBN, (p.M + SBM - 1) / SBM), dim3(256), 0, st, p);
    }
    else if (v2ok && pf.bn == 112) hipLaunchKernelGGL((gemm_f64_kernel_v2<AKC, BKC, CACC, 112>), dim3(pf.ntiles, pf.batch, pf.batch2), dim3(256), lds2, st, pf);
    else if (v2ok && pf.bn == 96) hipLaunchKernelGGL((gemm_f64_kernel_v2<AKC, BKC, CACC, 96>), dim3(pf.ntiles, pf.batch, pf.batch2), dim3(256), lds2, st, pf);
    else if (v2ok) hipLaunchKernelGGL((gemm_f64_kernel_v2<AKC, BKC, CACC>), dim3(p.ntiles, p.batch, p.batch2), dim3(256), lds2, st, p);
    else hipLaunchKernelGGL((gemm_f64_kernel<AKC, BKC, CACC>), dim3(p.ntiles, p.batch, p.batch2), dim3(256), lds, st, p);
  }
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

template <bool AKC, bool BKC>
int launch_c(const GemmParams& p, hipStream_t st) {
  if (p.beta != 0.0 && p.alpha != 0.0) return launch_t<AKC, BKC, true>(p, st);
  return launch_t<AKC, BKC, false>(p, st);
}

}  // namespace

int launch_trsm_leaf_forward(const double* L, long ldl, const double* dinv_leaf, int nb, double* B, long ldb, int ncols,
                             double* G, hipStream_t st) {
  // leaf of nb <= 8 full 128-row blocks; ncols even (16-byte rows), any number of strips, the last one may be narrow
  if (ncols <= 0) return 0;
  const long ldg = (long)nb * BN;
  {
    LeafGParams g{L, ldl, dinv_leaf, G, ldg};
    hipLaunchKernelGGL(build_leaf_g_kernel, dim3(nb, nb), dim3(256), 0, st, g);
    GPMP_HIP_TRY(hipGetLastError());
  }
  static DeviceOnce attr_once;
  const size_t lds2 = sizeof(double) * 4 * V2_TILE;
  if (const long long dev_bit = attr_once.need()) {
    if (dev_bit < 0) { set_error("hipGetDevice failed or device ordinal above 62"); return -1; }
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_leaf_kernel<128>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_leaf_kernel<64>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_leaf_kernel<32>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_leaf_kernel<16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_leaf_kernel<112>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_leaf_kernel<96>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    GPMP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_leaf_kernel<80>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    attr_once.done(dev_bit);
  }
  // A leaf has one workgroup per strip, each walking the nb blocks one after the other (MFMA-bound on its compute unit:
  // 68 us for a 64-column strip of a 512-row leaf): with fewer than `narrow_below` strips of 128 columns the strips are narrowed
  // until there are about `min_strips` of them (config 2: 79 strips of 128 left two thirds of the machine idle).
  constexpr int narrow_below = 192, min_strips = 256;
  const int strips128 = (ncols + BN - 1) / BN;
  int bnw = 128;
  if (strips128 < narrow_below) {
    bnw = 64;
    while (bnw > 16 && (ncols + bnw - 1) / bnw < min_strips) bnw >>= 1;
  } else {
    // Round 4: many strips.  A leaf's workgroups are MFMA-bound on their compute unit, so the leaf lasts as long as the busiest
    // CU: with L = ceil(strips / CUs) workgroups on it, pairs run two at a time (each at half rate, together ~0.9 of the pipe) and an
    // odd one alone (~0.8): time ~ width x (2.22 floor(L / 2) + 1.25 (L mod 2)).  m = 50000 in 128-column strips is 391 workgroups
    // (L = 2): 112 columns (waves 4 x 1 over 32 x 112) make 447 strips of 7 / 8 the work each, m = 40000 goes to 80 columns; m = 30000
    // (235 strips, one per CU) stays at 128.  Measured: predict n = 32768: m = 50000 950.4 -> 947.8 ms, m = 40000 801.4 -> 795.5 ms.
    const int ncu = device_cu_count();
    auto cost = [&](int w) {
      const int sw = (ncols + w - 1) / w;
      const int L = (sw + ncu - 1) / ncu;
      return (double)w * (2.22 * (L / 2) + 1.25 * (L % 2));
    };
    double best = cost(128);
    for (int w : {112, 96, 80}) {
      const double c = cost(w);
      if (c < 0.97 * best) { best = c; bnw = w; }
    }
  }
  LeafSolveParams p{G, ldg, B, ldb, nb, ncols};
  ProfScope ps(PK_GEMM_NN, st, (double)ncols * (double)(nb * BM) * (double)((nb + 1) * BM));   // counted with the small-K NN work it replaces
  const dim3 grid((ncols + bnw - 1) / bnw);
  if (bnw == 128) hipLaunchKernelGGL(trsm_leaf_kernel<128>, grid, dim3(256), lds2, st, p);
  else if (bnw == 112) hipLaunchKernelGGL(trsm_leaf_kernel<112>, grid, dim3(256), lds2, st, p);
  else if (bnw == 96) hipLaunchKernelGGL(trsm_leaf_kernel<96>, grid, dim3(256), lds2, st, p);
  else if (bnw == 80) hipLaunchKernelGGL(trsm_leaf_kernel<80>, grid, dim3(256), lds2, st, p);
  else if (bnw == 64) hipLaunchKernelGGL(trsm_leaf_kernel<64>, grid, dim3(256), lds2, st, p);
  else if (bnw == 32) hipLaunchKernelGGL(trsm_leaf_kernel<32>, grid, dim3(256), lds2, st, p);
  else hipLaunchKernelGGL(trsm_leaf_kernel<16>, grid, dim3(256), lds2, st, p);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

static int launch_gemm_one(bool a_kc, bool b_kc, int M, int N, int K, double alpha, const double* A, long lda,
                           const double* B, long ldb, double beta, double* C, long ldc, const GemmOpts& o,
                           hipStream_t st) {
  if (M <= 0 || N <= 0) return 0;
  GemmParams p;
  p.A = A; p.B = B; p.C = C;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  p.alpha = alpha; p.beta = beta;
  p.lower_only = o.lower_only; p.kstart_row = o.kstart_row; p.kend_row = o.kend_row; p.kend_col = o.kend_col;
  p.kstart_col = o.kstart_col;
  p.tiles_m = (M + BM - 1) / BM;
  p.tiles_n = (N + BN - 1) / BN;
  p.stair_num = o.stair_num; p.stair_den = o.stair_den; p.stair_off = o.stair_off; p.stair_sub = o.stair_sub;
  p.kg_rnum = o.kg_rnum; p.kg_rden = o.kg_rden; p.kg_roff = o.kg_roff; p.kg_cnum = o.kg_cnum; p.kg_cden = o.kg_cden; p.kg_coff = o.kg_coff;
  if ((o.kg_rden > 0 || o.kg_cden > 0) && (o.lower_only | o.kstart_row | o.kend_row | o.kstart_col | o.kend_col)) {
    set_error("group contraction starts combined with another tile / k restriction");
    return -1;
  }
  if (o.stair_den > 0) {
    if (o.lower_only | o.kstart_row | o.kend_row | o.kstart_col | o.kend_col) { set_error("staircase tile set combined with another tile / k restriction"); return -1; }
    p.ntiles = 0;
    for (int g = 0; 8 * g < p.tiles_m; ++g) {
      const int gsize = (p.tiles_m - 8 * g) < 8 ? (p.tiles_m - 8 * g) : 8;
      p.ntiles += gsize * stair_cols(o.stair_num, o.stair_den, o.stair_off, o.stair_sub, p.tiles_n, g);
    }
    if (p.ntiles == 0) return 0;
  } else if (o.lower_only) {
    const int tn = p.tiles_n < p.tiles_m ? p.tiles_n : p.tiles_m;
    p.ntiles = tn * (tn + 1) / 2 + (p.tiles_m - tn) * tn;
  } else {
    p.ntiles = p.tiles_m * p.tiles_n;
  }
  p.lean = o.lean | g_machine_busy;
  p.bn = BN;
  // request distance of the LDS-direct kernel's operand tiles: one barrier earlier for long k loops (K >= 2048: solve updates
  // 92.2 -> 93.1 % of peak; neutral at 1024, -0.3 at 512: profiles/r4/gemm_early_issue_ab.log)
  p.early = K >= 2048 ? 1 : 0;
  // lower-triangular tile sets of equal-cost tiles in 8 x 8 super-tiles (fabric-side fetch of the Cholesky's trailing update / 2.24)
  p.tri_block = !(o.kstart_row | o.kend_row | o.kstart_col | o.kend_col | o.kg_rden | o.kg_cden);
  p.batch = o.batch > 1 ? o.batch : 1;
  p.sa = o.stride_a; p.sb = o.stride_b; p.sc = o.stride_c;
  p.batch2 = o.batch2 > 1 ? o.batch2 : 1;
  p.sa2 = o.stride2_a; p.sb2 = o.stride2_b; p.sc2 = o.stride2_c;
  p.aligned = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((lda & 1) == 0) &&
              ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && ((ldb & 1) == 0);
  if (a_kc && b_kc) return launch_c<true, true>(p, st);
  if (a_kc && !b_kc) return launch_c<true, false>(p, st);
  if (!a_kc && !b_kc) return launch_c<false, false>(p, st);
  return launch_c<false, true>(p, st);
}

int launch_gemm(bool a_kc, bool b_kc, int M, int N, int K, double alpha, const double* A, long lda,
                const double* B, long ldb, double beta, double* C, long ldc, const GemmOpts& o,
                hipStream_t st) {
  if (M <= 0 || N <= 0) return 0;
  // The LDS-direct kernel needs even M and N (16-byte clipping at the edges): peel an odd last row / column off a
  // large rectangular product so that everything else runs on it.
  const bool plain = !o.lower_only && !o.stair_den && !o.kg_rden && !o.kg_cden && !o.kstart_row && !o.kend_row && !o.kstart_col && !o.kend_col && o.batch <= 1 && o.batch2 <= 1;
  const int Nr = N % 2, Mr = M % 2;
  if (plain && (Nr || Mr) && (K % BK == 0) && K >= 512 && (M >= 4 * BM || N >= 4 * BN) && C != A && C != B) {
    const int Mf = M - Mr, Nf = N - Nr;
    int rc = 0;
    if (Mf > 0 && Nf > 0) rc = launch_gemm_one(a_kc, b_kc, Mf, Nf, K, alpha, A, lda, B, ldb, beta, C, ldc, o, st);
    if (rc) return rc;
    if (Nr > 0) {  // last column, all rows
      const double* Bn = b_kc ? B + (long)Nf * ldb : B + Nf;
      rc = launch_gemm_one(a_kc, b_kc, M, Nr, K, alpha, A, lda, Bn, ldb, beta, C + Nf, ldc, o, st);
      if (rc) return rc;
    }
    if (Mr > 0 && Nf > 0) {  // last row, full columns
      const double* Am = a_kc ? A + (long)Mf * lda : A + Mf;
      rc = launch_gemm_one(a_kc, b_kc, Mr, Nf, K, alpha, Am, lda, B, ldb, beta, C + (long)Mf * ldc, ldc, o, st);
    }
    return rc;
  }
  return launch_gemm_one(a_kc, b_kc, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, o, st);
}

}  // namespace gpmp
