// gpmp_dist_*: the LOCAL half of the 2-D block-cyclic Cholesky (BASELINE.json configs[4]) behind the C ABI.
//
// The reference has no distributed code (README.md:39-40: large-scale is future work); SURVEY 8(b) lists `gpmp_dist_*` as
// part of the boundary.  The library never links a communication library: a host (gpmp_amd/dist over torch.distributed, or a
// C++ / RCCL program like examples/dist_potrf_rccl.cpp) owns every collective and calls these entry points between them.
// One block-column step k on rank (r, c) of a Pr x Pc grid, block size nb (global block (I, J) lives on rank
// (I mod Pr, J mod Pc) at local block (I div Pr, J div Pc) of ONE dense row-major local matrix):
//
//   owner of (k, k):           gpmp_dist_diag_factor   -> msg = [L_kk | inverses of its 128-blocks | info]
//                              ... broadcast msg down process column k mod Pc ...
//   ranks of that column:      gpmp_dist_panel_solve   -> panel = A_{I>k,k} L_kk^-T   (also written back in place)
//                              ... broadcast panel along every process row ...
//   per process row rp:        gpmp_dist_exchange_pack on the holder (r == rp), broadcast inside the process column,
//                              gpmp_dist_exchange_unpack on every rank   -> colop = L_{J>k,k} for the owned block columns J
//   every rank:                gpmp_dist_trailing_update  A_IJ -= panel_I colop_J^T   for its blocks I >= J > k
//
// All functions only enqueue on `stream`; index arithmetic happens on the host from (n, nb, Pr, Pc, r, c, k).
#include "common.h"

namespace gpmp {
namespace {

struct Layout {
  int n, nb, pr, pc, r, c, nblocks;
  int bs(int I) const { return (I + 1) * nb <= n ? nb : n - I * nb; }
  // number of owned block rows / columns with global index <= k  (= local index of the first owned one > k)
  int first_row_after(int k) const { return k >= r ? (k - r) / pr + 1 : 0; }
  int first_col_after(int k) const { return k >= c ? (k - c) / pc + 1 : 0; }
  int n_row_blocks() const { return nblocks > r ? (nblocks - 1 - r) / pr + 1 : 0; }
  int n_col_blocks() const { return nblocks > c ? (nblocks - 1 - c) / pc + 1 : 0; }
  // element offset of local block li (only the globally last block can be short, so every offset is li * nb)
  long roff(int li) const { const int nr = n_row_blocks(); return li < nr ? (long)li * nb : local_rows(); }
  long coff(int lj) const { const int nc = n_col_blocks(); return lj < nc ? (long)lj * nb : local_cols(); }
  long local_rows() const { const int nr = n_row_blocks(); return nr == 0 ? 0 : (long)(nr - 1) * nb + bs(r + (nr - 1) * pr); }
  long local_cols() const { const int nc = n_col_blocks(); return nc == 0 ? 0 : (long)(nc - 1) * nb + bs(c + (nc - 1) * pc); }
};

int make_layout(Layout& L, int n, int nb, int pr, int pc, int r, int c) {
  GPMP_ARG(n > 0 && n <= GPMP_MAX_EXTENT, 1, "n outside [1, GPMP_MAX_EXTENT]");
  GPMP_ARG(nb > 0 && nb % NB == 0 && nb <= 2 * OUTER_BLOCKS * NB, 2, "block size must be a multiple of 128 in (0, 1024]");
  GPMP_ARG(pr > 0 && pc > 0, 3, "empty process grid");
  GPMP_ARG(r >= 0 && r < pr && c >= 0 && c < pc, 5, "rank coordinates outside the grid");
  L = Layout{n, nb, pr, pc, r, c, (n + nb - 1) / nb};
  return 0;
}

__global__ void info_word_to_double(double* slot) {
  const int v = *reinterpret_cast<const int*>(slot);
  *slot = (double)v;
}

// `count` blocks of up to nb rows x cols doubles: block t goes from src + (s0 + t * sstep) * lds to dst + (d0 + t * dstep) * ldd;
// the last block has `last_rows` rows.  One workgroup per (block, 16-row group): rows are contiguous runs of `cols` doubles.
__global__ void __launch_bounds__(256) copy_row_blocks(const double* __restrict__ src, long lds, long s0, long sstep,
                                                       double* __restrict__ dst, long ldd, long d0, long dstep, int count, int nb,
                                                       int last_rows, int cols) {
  const int t = blockIdx.y;
  const int rows = t == count - 1 ? last_rows : nb;
  const int row0 = blockIdx.x * 16;
  if (row0 >= rows) return;
  const double* s = src + (s0 + (long)t * sstep) * lds;
  double* d = dst + (d0 + (long)t * dstep) * ldd;
  const int rend = row0 + 16 < rows ? row0 + 16 : rows;
  for (int i = row0 + (threadIdx.x >> 6); i < rend; i += 4)
    for (int j = threadIdx.x & 63; j < cols; j += 64) d[(long)i * ldd + j] = s[(long)i * lds + j];
}

// the blocks J > k with J mod Pc == c and J mod Pr == rp: an arithmetic progression (step lcm(Pr, Pc)) or empty
struct Progression { int first, step, count; };
Progression exchange_blocks(const Layout& L, int k, int rp) {
  int a = L.pr, b = L.pc;
  while (b) { const int tmp = a % b; a = b; b = tmp; }
  const int g = a, lcm = L.pr / g * L.pc;
  Progression p{-1, lcm, 0};
  for (int J = k + 1; J < L.nblocks && J <= k + lcm; ++J)
    if (J % L.pc == L.c && J % L.pr == rp) { p.first = J; break; }
  if (p.first >= 0) p.count = (L.nblocks - 1 - p.first) / lcm + 1;
  return p;
}

}  // namespace
}  // namespace gpmp

using namespace gpmp;

extern "C" size_t gpmp_dist_diag_msg_elems(int bk) {
  if (bk <= 0) return 0;
  const size_t ldk = ((size_t)bk + 15) / 16 * 16;
  return (size_t)bk * ldk + (size_t)((bk + NB - 1) / NB) * NB * NB + 1;
}

extern "C" int gpmp_dist_diag_factor(double* D, int bk, long ldd, double* msg, gpmp_stream_t stream) {
  GPMP_ARG(D != nullptr, 1, "D is NULL");
  GPMP_ARG(bk > 0 && bk <= 2 * OUTER_BLOCKS * NB, 2, "diagonal block outside (0, 1024]");
  GPMP_ARG(ldd >= bk, 3, "ldd < bk");
  GPMP_ARG(msg != nullptr, 4, "msg is NULL");
  hipStream_t st = as_stream(stream);
  const long ldk = ((long)bk + 15) / 16 * 16;
  double* dinv = msg + (size_t)bk * ldk;
  double* slot = dinv + (size_t)((bk + NB - 1) / NB) * NB * NB;
  int rc = gpmp_potrf_lower_async(D, bk, ldd, dinv, reinterpret_cast<int*>(slot), stream);   // (bk <= 1024: dinv = block inverses only)
  if (rc) return rc;
  hipLaunchKernelGGL(info_word_to_double, dim3(1), dim3(1), 0, st, slot);
  GPMP_HIP_TRY(hipGetLastError());
  GPMP_HIP_TRY(hipMemcpy2DAsync(msg, (size_t)ldk * 8, D, (size_t)ldd * 8, (size_t)bk * 8, (size_t)bk, hipMemcpyDeviceToDevice, st));
  return 0;
}

extern "C" size_t gpmp_dist_panel_ws_elems(int bk) {
  return bk > 0 ? (size_t)bk * (((size_t)bk + 15) / 16 * 16) : 0;
}

extern "C" int gpmp_dist_panel_solve(const double* msg, int bk, double* P, int rows, long ldp, double* panel, long ldo, double* ws,
                                     gpmp_stream_t stream) {
  GPMP_ARG(msg != nullptr, 1, "msg is NULL");
  GPMP_ARG(bk > 0 && bk <= 2 * OUTER_BLOCKS * NB, 2, "diagonal block outside (0, 1024]");
  GPMP_ARG(rows >= 0 && rows <= GPMP_MAX_EXTENT, 4, "rows outside [0, GPMP_MAX_EXTENT]");
  if (rows == 0) return 0;
  GPMP_ARG(P != nullptr && ldp >= bk, 5, "P is NULL or ldp < bk");
  GPMP_ARG(panel != nullptr && ldo >= bk, 7, "panel is NULL or ldo < bk");
  hipStream_t st = as_stream(stream);
  const long ldk = ((long)bk + 15) / 16 * 16;
  const double* Lkk = msg;
  const double* dinv = msg + (size_t)bk * ldk;
  if (bk % NB == 0 && ws != nullptr) {
    // panel = P T^T with T = L_kk^-1 (doubling from the 128-block inverses; the k loop of tile column j stops at the
    // diagonal): 2 log2(bk / 128) small launches + ONE large one instead of the substitution's 2 bk / 128 - 1 dependent ones,
    // each of which waits for a workgroup slot under the bulk update of the previous step
    int rc = gpmp_trtri_lower(Lkk, bk, ldk, dinv, ws, ldk, stream);
    if (rc) return rc;
    rc = gpmp_dgemm(0, 1, rows, bk, bk, 1.0, P, ldp, ws, ldk, 0.0, panel, ldo, 4, stream);
    if (rc) return rc;
    GPMP_HIP_TRY(hipMemcpy2DAsync(P, (size_t)ldp * 8, panel, (size_t)ldo * 8, (size_t)bk * 8, (size_t)rows, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  int rc = gpmp_trsm_right_lower(Lkk, bk, ldk, dinv, P, rows, ldp, stream);     // ragged last block: substitution in place
  if (rc) return rc;
  GPMP_HIP_TRY(hipMemcpy2DAsync(panel, (size_t)ldo * 8, P, (size_t)ldp * 8, (size_t)bk * 8, (size_t)rows, hipMemcpyDeviceToDevice, st));
  return 0;
}

extern "C" int gpmp_dist_local_shape(int n, int nb, int pr, int pc, int r, int c, long* rows_out, long* cols_out) {
  Layout L;
  if (int rc = make_layout(L, n, nb, pr, pc, r, c)) return rc;
  if (rows_out) *rows_out = L.local_rows();
  if (cols_out) *cols_out = L.local_cols();
  return 0;
}

extern "C" int gpmp_dist_step_shape(int n, int nb, int pr, int pc, int r, int c, int k, long* panel_rows_out, long* colop_rows_out,
                                    long* panel_row0_out, long* colop_col0_out) {
  Layout L;
  if (int rc = make_layout(L, n, nb, pr, pc, r, c)) return rc;
  GPMP_ARG(k >= 0 && k < L.nblocks, 7, "block column outside the matrix");
  const long r0 = L.roff(L.first_row_after(k)), c0 = L.coff(L.first_col_after(k));
  if (panel_rows_out) *panel_rows_out = L.local_rows() - r0;
  if (colop_rows_out) *colop_rows_out = L.local_cols() - c0;
  if (panel_row0_out) *panel_row0_out = r0;
  if (colop_col0_out) *colop_col0_out = c0;
  return 0;
}

extern "C" long gpmp_dist_exchange_rows(int n, int nb, int pr, int pc, int rp, int c, int k) {
  Layout L;
  if (make_layout(L, n, nb, pr, pc, rp, c)) return -1;
  const Progression p = exchange_blocks(L, k, rp);
  if (p.count == 0) return 0;
  return (long)(p.count - 1) * nb + L.bs(p.first + (p.count - 1) * p.step);
}

extern "C" int gpmp_dist_exchange_pack(const double* panel, long ldp, double* piece, long ldq, int n, int nb, int pr, int pc, int r, int c,
                                       int k, int bk, gpmp_stream_t stream) {
  Layout L;
  if (int rc = make_layout(L, n, nb, pr, pc, r, c)) return rc;
  GPMP_ARG(k >= 0 && k < L.nblocks, 11, "block column outside the matrix");
  const Progression p = exchange_blocks(L, k, r);
  if (p.count == 0) return 0;
  GPMP_ARG(panel != nullptr && piece != nullptr, 1, "NULL buffer");
  GPMP_ARG(ldp >= bk && ldq >= bk && bk > 0 && bk <= nb, 12, "bk outside (0, nb] or a leading dimension below it");
  const int i0 = L.first_row_after(k);
  const long s0 = (long)(p.first / pr - i0) * nb;                    // panel row of block p.first (panel row 0 = first owned row > k)
  const int last_rows = L.bs(p.first + (p.count - 1) * p.step);
  hipLaunchKernelGGL(copy_row_blocks, dim3((nb + 15) / 16, p.count), dim3(256), 0, as_stream(stream), panel, ldp, s0,
                     (long)(p.step / pr) * nb, piece, ldq, 0L, (long)nb, p.count, nb, last_rows, bk);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gpmp_dist_exchange_unpack(const double* piece, long ldq, double* colop, long ldc, int n, int nb, int pr, int pc, int rp, int c,
                                         int k, int bk, gpmp_stream_t stream) {
  Layout L;
  if (int rc = make_layout(L, n, nb, pr, pc, rp, c)) return rc;
  GPMP_ARG(k >= 0 && k < L.nblocks, 11, "block column outside the matrix");
  const Progression p = exchange_blocks(L, k, rp);
  if (p.count == 0) return 0;
  GPMP_ARG(piece != nullptr && colop != nullptr, 1, "NULL buffer");
  GPMP_ARG(ldq >= bk && ldc >= bk && bk > 0 && bk <= nb, 12, "bk outside (0, nb] or a leading dimension below it");
  const int j0 = L.first_col_after(k);
  const long d0 = (long)(p.first / pc - j0) * nb;                    // colop row of block p.first (row 0 = first owned column > k)
  const int last_rows = L.bs(p.first + (p.count - 1) * p.step);
  hipLaunchKernelGGL(copy_row_blocks, dim3((nb + 15) / 16, p.count), dim3(256), 0, as_stream(stream), piece, ldq, 0L, (long)nb, colop, ldc,
                     d0, (long)(p.step / pc) * nb, p.count, nb, last_rows, bk);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gpmp_dist_trailing_update(double* A, long lda, int n, int nb, int pr, int pc, int r, int c, int k, const double* panel,
                                         long ldp, const double* colop, long ldc, int jlo, int jhi, int rows_after,
                                         gpmp_stream_t stream) {
  Layout L;
  if (int rc = make_layout(L, n, nb, pr, pc, r, c)) return rc;
  GPMP_ARG(k >= 0 && k < L.nblocks, 9, "block column outside the matrix");
  const int nrb = L.n_row_blocks(), ncb = L.n_col_blocks();
  const int i0 = L.first_row_after(k), j0 = L.first_col_after(k);
  if (jlo < j0) jlo = j0;
  if (jhi < 0 || jhi > ncb) jhi = ncb;
  if (jhi <= jlo || i0 >= nrb) return 0;
  GPMP_ARG(A != nullptr && lda >= L.local_cols(), 1, "A is NULL or lda below the local column count");
  GPMP_ARG(panel != nullptr && colop != nullptr, 10, "NULL operand");
  const int bk = L.bs(k);
  GPMP_ARG(ldp >= bk && ldc >= bk, 11, "operand leading dimension below the block width");
  int first = rows_after >= 0 ? L.first_row_after(rows_after) : i0;
  if (first < i0) first = i0;                                        // rows_after < k: the panel has no rows above block k
  if (first >= nrb) return 0;
  if (nb == 8 * NB && bk == nb) {
    // ONE launch over the staircase (round 5): local block row li = first + g holds the blocks J <= I of its global block row
    // I = r + pr li, i.e. the local columns lj <= (I - c) / pc -- a tile set the GEMM enumerates directly (GemmOpts::stair_*: a
    // block row = one group of 8 tile rows, so the XCD-aware chunking and the 8 x 8 co-residency of the plain order are kept).
    // The staircase of GEMMs below computed up to three wasted blocks per group of four block rows (5 % of the update at
    // config 5) and ended each of its 16 launches on a partial round of the machine.
    GemmOpts st8;
    st8.stair_num = pr; st8.stair_den = pc; st8.stair_off = r + pr * first - c; st8.stair_sub = jlo;
    const long r0 = L.roff(first), r1 = L.roff(nrb), c0 = L.coff(jlo), c1 = L.coff(jhi);
    return launch_gemm(true, true, (int)(r1 - r0), (int)(c1 - c0), bk, -1.0, panel + (r0 - L.roff(i0)) * ldp, ldp,
                       colop + (c0 - L.coff(j0)) * ldc, ldc, 1.0, A + r0 * lda + c0, lda, st8, as_stream(stream));
  }
  const int G = 4;                                                   // block rows per GEMM of the staircase (other block sizes, ragged K)
  for (int lg = first; lg < nrb; lg += G) {
    const int le = lg + G < nrb ? lg + G : nrb;
    const int I_last = r + (le - 1) * pr;
    int jend = L.first_col_after(I_last);                            // local columns J <= I_last
    if (jend > jhi) jend = jhi;
    if (jend <= jlo) continue;
    const long r0 = L.roff(lg), r1 = L.roff(le), c0 = L.coff(jlo), c1 = L.coff(jend);
    int rc = gpmp_dgemm(0, 1, (int)(r1 - r0), (int)(c1 - c0), bk, -1.0, panel + (r0 - L.roff(i0)) * ldp, ldp,
                        colop + (c0 - L.coff(j0)) * ldc, ldc, 1.0, A + r0 * lda + c0, lda, 0, stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int gpmp_dist_inverse_gram(const double* T, long ldt, const double* T2, long ldt2, double* M, long ldm, int n, int nb, int pr,
                                      int pc, int r, int c, int c2, int lower_only, gpmp_stream_t stream) {
  Layout L, L2;
  if (int rc = make_layout(L, n, nb, pr, pc, r, c)) return rc;
  if (int rc = make_layout(L2, n, nb, pr, pc, r, c2)) return rc;
  GPMP_ARG(nb == 8 * NB, 8, "block size must be 1024 (one block = one group of 8 GEMM tiles)");
  const long rows = L.local_rows(), mc = L.local_cols(), mc2 = L2.local_cols();
  if (mc == 0 || mc2 == 0) return 0;
  GPMP_ARG(M != nullptr && ldm >= mc2, 5, "M is NULL or ldm below the column count of the second column set");
  if (rows == 0) {                                  // this process row holds nothing of T: the partial product is zero
    GPMP_HIP_TRY(hipMemset2DAsync(M, sizeof(double) * (size_t)ldm, 0, sizeof(double) * (size_t)mc2, (size_t)mc, as_stream(stream)));
    return 0;
  }
  GPMP_ARG(T != nullptr && ldt >= mc, 1, "T is NULL or ldt below the local column count");
  GPMP_ARG(T2 != nullptr && ldt2 >= mc2, 3, "T2 is NULL or ldt2 below its local column count");
  // block (I, J) of T^T T2, I = c + pc g (my column set), J = c2 + pc h: T[k, I] = 0 for global block rows k < I, so the contraction
  // starts at the first local block row >= max(I, J): local block ceil((I - r) / pr) resp. ceil((J - r) / pr)
  GemmOpts o;
  o.kg_rnum = pc; o.kg_rden = pr; o.kg_roff = c - r;
  o.kg_cnum = pc; o.kg_cden = pr; o.kg_coff = c2 - r;
  if (lower_only) {                                  // the blocks J <= I only: floor((pc g + c - c2) / pc) + 1 leading column blocks
    o.stair_num = pc; o.stair_den = pc; o.stair_off = c - c2; o.stair_sub = 0;
  }
  return launch_gemm(false, false, (int)mc, (int)mc2, (int)rows, 1.0, T, ldt, T2, ldt2, 0.0, M, ldm, o, as_stream(stream));
}

