// Blocked dense linear algebra built on the fp64 MFMA GEMM (gemm_f64.hip) and the LDS diagonal-block
// kernel (potf2.hip): right-looking Cholesky, triangular solves with many right-hand sides, triangular
// inverse and the T^T T product.  Everything is row-major; L lives in the lower triangle.
//
// Two-level blocking: diagonal blocks of NB = 128 (factored + inverted in LDS by one workgroup, which
// turns every panel solve into a GEMM with inv(L_kk)), grouped into outer panels of 4 blocks so that
// the O(n^3) trailing updates are rank-512 GEMMs (arithmetic intensity 32 flop per HBM byte of C).
//
// Replaces numpy.linalg.cholesky / scipy.linalg.solve_triangular as used by gnp.cholesky_solve
// (gpmp/num/numpy_backend.py:465-469) and diag_Kinv_from_chol (gpmp/core/linalg.py:17-46).
#include "common.h"

namespace gpmp {
namespace {

inline int imin(int a, int b) { return a < b ? a : b; }

int potrf_lower(double* A, int n, long lda, double* dinv, int* info_dev, hipStream_t st) {
  GPMP_HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), st));
  const int nblk = (n + NB - 1) / NB;
  GemmOpts lower;
  lower.lower_only = 1;
  GemmOpts plain;
  for (int ob = 0; ob < nblk; ob += OUTER_BLOCKS) {
    const int oe = imin(ob + OUTER_BLOCKS, nblk);
    const int out_end = imin(oe * NB, n);
    for (int c = ob; c < oe; ++c) {
      const int c0 = c * NB;
      const int jb = imin(NB, n - c0);
      double* dc = dinv + (size_t)c * NB * NB;
      int rc = launch_potf2_inv(A + (long)c0 * lda + c0, lda, jb, dc, info_dev, c0, st);
      if (rc) return rc;
      const int r1 = c0 + jb;
      const int mrem = n - r1;
      if (mrem <= 0) break;
      double* A21 = A + (long)r1 * lda + c0;
      // panel: A21 <- A21 * inv(L_cc)^T  (in place: one 128-wide tile column, K = 128)
      rc = launch_gemm(true, true, mrem, jb, jb, 1.0, A21, lda, dc, NB, 0.0, A21, lda, plain, st);
      if (rc) return rc;
      const int ncols_in = out_end - r1;
      if (ncols_in > 0) {
        // remaining columns of this outer panel: rank-128 update, lower tiles only
        rc = launch_gemm(true, true, mrem, ncols_in, jb, -1.0, A21, lda, A21, lda, 1.0,
                         A + (long)r1 * lda + r1, lda, lower, st);
        if (rc) return rc;
      }
    }
    const int mrem = n - out_end;
    if (mrem > 0) {
      // trailing update A22 -= P P^T, P = A[out_end:, ob*NB : out_end]  (rank-512 syrk on MFMA)
      const int kw = out_end - ob * NB;
      const double* P = A + (long)out_end * lda + (long)ob * NB;
      int rc = launch_gemm(true, true, mrem, mrem, kw, -1.0, P, lda, P, lda, 1.0,
                           A + (long)out_end * lda + out_end, lda, lower, st);
      if (rc) return rc;
    }
  }
  return 0;
}

// L X = B (forward).  tri != 0: B starts as the identity and only the lower triangle of X = L^-1 is
// non-zero, so block row c only touches its first (c+1)*NB columns.
int trsm_forward(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, int tri,
                 hipStream_t st) {
  const int nblk = (n + NB - 1) / NB;
  GemmOpts plain;
  for (int ob = 0; ob < nblk; ob += OUTER_BLOCKS) {
    const int oe = imin(ob + OUTER_BLOCKS, nblk);
    const int out_end = imin(oe * NB, n);
    for (int c = ob; c < oe; ++c) {
      const int c0 = c * NB;
      const int jb = imin(NB, n - c0);
      const double* dc = dinv + (size_t)c * NB * NB;
      double* Bc = B + (long)c0 * ldb;
      const int ncol = tri ? imin(m, c0 + jb) : m;
      int rc = launch_gemm(true, false, jb, ncol, jb, 1.0, dc, NB, Bc, ldb, 0.0, Bc, ldb, plain, st);
      if (rc) return rc;
      const int r1 = c0 + jb;
      const int rows_in = out_end - r1;
      if (rows_in > 0) {
        rc = launch_gemm(true, false, rows_in, ncol, jb, -1.0, L + (long)r1 * ldl + c0, ldl, Bc, ldb, 1.0,
                         B + (long)r1 * ldb, ldb, plain, st);
        if (rc) return rc;
      }
    }
    const int mrem = n - out_end;
    if (mrem > 0) {
      const int kw = out_end - ob * NB;
      const int ncol = tri ? imin(m, out_end) : m;
      int rc = launch_gemm(true, false, mrem, ncol, kw, -1.0, L + (long)out_end * ldl + (long)ob * NB, ldl,
                           B + (long)ob * NB * ldb, ldb, 1.0, B + (long)out_end * ldb, ldb, plain, st);
      if (rc) return rc;
    }
  }
  return 0;
}

// L^T X = B (backward).
int trsm_backward(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb,
                  hipStream_t st) {
  const int nblk = (n + NB - 1) / NB;
  GemmOpts plain;
  const int last_ob = ((nblk - 1) / OUTER_BLOCKS) * OUTER_BLOCKS;
  for (int ob = last_ob; ob >= 0; ob -= OUTER_BLOCKS) {
    const int oe = imin(ob + OUTER_BLOCKS, nblk);
    const int out_end = imin(oe * NB, n);
    const int ob0 = ob * NB;
    for (int c = oe - 1; c >= ob; --c) {
      const int c0 = c * NB;
      const int jb = imin(NB, n - c0);
      const double* dc = dinv + (size_t)c * NB * NB;
      double* Bc = B + (long)c0 * ldb;
      // X_c = inv(L_cc)^T B_c
      int rc = launch_gemm(false, false, jb, m, jb, 1.0, dc, NB, Bc, ldb, 0.0, Bc, ldb, plain, st);
      if (rc) return rc;
      const int rows_in = c0 - ob0;
      if (rows_in > 0) {
        // B[ob0:c0] -= L[c0:c0+jb, ob0:c0]^T X_c
        rc = launch_gemm(false, false, rows_in, m, jb, -1.0, L + (long)c0 * ldl + ob0, ldl, Bc, ldb, 1.0,
                         B + (long)ob0 * ldb, ldb, plain, st);
        if (rc) return rc;
      }
    }
    if (ob0 > 0) {
      // B[0:ob0] -= L[ob0:out_end, 0:ob0]^T X[ob0:out_end]
      const int kw = out_end - ob0;
      int rc = launch_gemm(false, false, ob0, m, kw, -1.0, L + (long)ob0 * ldl, ldl, B + (long)ob0 * ldb, ldb,
                           1.0, B, ldb, plain, st);
      if (rc) return rc;
    }
  }
  return 0;
}

}  // namespace
}  // namespace gpmp

using namespace gpmp;

extern "C" size_t gpmp_dinv_elems(int n) {
  if (n <= 0) return 0;
  return (size_t)((n + NB - 1) / NB) * NB * NB;
}

extern "C" int gpmp_potrf_lower_async(double* A, int n, long lda, double* dinv, int* info_dev,
                                      gpmp_stream_t stream) {
  GPMP_ARG(A != nullptr, 1, "A is NULL");
  GPMP_ARG(n >= 0, 2, "n < 0");
  GPMP_ARG(lda >= n, 3, "lda < n");
  GPMP_ARG(dinv != nullptr, 4, "dinv is NULL");
  GPMP_ARG(info_dev != nullptr, 5, "info is NULL");
  if (n == 0) return 0;
  return potrf_lower(A, n, lda, dinv, info_dev, as_stream(stream));
}

extern "C" int gpmp_trtri_diag_blocks(const double* L, int n, long ldl, double* dinv, gpmp_stream_t stream) {
  GPMP_ARG(L != nullptr, 1, "L is NULL");
  GPMP_ARG(ldl >= n, 3, "ldl < n");
  GPMP_ARG(dinv != nullptr, 4, "dinv is NULL");
  return launch_trtri_blocks(L, ldl, n, dinv, as_stream(stream));
}

extern "C" int gpmp_trsm_lower(const double* L, int n, long ldl, const double* dinv, double* B, int m,
                               long ldb, int trans, double* scratch, gpmp_stream_t stream) {
  GPMP_ARG(L != nullptr, 1, "L is NULL");
  GPMP_ARG(n >= 0, 2, "n < 0");
  GPMP_ARG(ldl >= n, 3, "ldl < n");
  GPMP_ARG(B != nullptr, 5, "B is NULL");
  GPMP_ARG(m >= 0 && ldb >= m, 7, "ldb < m");
  GPMP_ARG(dinv != nullptr || scratch != nullptr, 9, "dinv and scratch both NULL");
  if (n == 0 || m == 0) return 0;
  hipStream_t st = as_stream(stream);
  if (dinv == nullptr) {
    int rc = launch_trtri_blocks(L, ldl, n, scratch, st);
    if (rc) return rc;
    dinv = scratch;
  }
  return trans ? trsm_backward(L, n, ldl, dinv, B, m, ldb, st) : trsm_forward(L, n, ldl, dinv, B, m, ldb, 0, st);
}

extern "C" int gpmp_trtri_lower(const double* L, int n, long ldl, const double* dinv, double* T, long ldt,
                                gpmp_stream_t stream) {
  GPMP_ARG(L != nullptr, 1, "L is NULL");
  GPMP_ARG(ldl >= n, 3, "ldl < n");
  GPMP_ARG(dinv != nullptr, 4, "dinv is NULL");
  GPMP_ARG(T != nullptr && ldt >= n, 5, "T is NULL or ldt < n");
  if (n <= 0) return 0;
  hipStream_t st = as_stream(stream);
  int rc = launch_set_identity_lower(T, n, ldt, st);
  if (rc) return rc;
  return trsm_forward(L, n, ldl, dinv, T, n, ldt, 1, st);
}

extern "C" int gpmp_lauum_lower(const double* T, int n, long ldt, double* Kinv, long ldk, gpmp_stream_t stream) {
  GPMP_ARG(T != nullptr && ldt >= n, 1, "T is NULL or ldt < n");
  GPMP_ARG(Kinv != nullptr && ldk >= n, 4, "Kinv is NULL or ldk < n");
  if (n <= 0) return 0;
  GemmOpts o;
  o.lower_only = 1;
  o.kstart_row = 1;  // T[l][i] = 0 for l < i: tile row i only needs l >= row0(i)
  return launch_gemm(false, false, n, n, n, 1.0, T, ldt, T, ldt, 0.0, Kinv, ldk, o, as_stream(stream));
}

extern "C" int gpmp_dgemm(int ta, int tb, int M, int N, int K, double alpha, const double* A, long lda,
                          const double* B, long ldb, double beta, double* C, long ldc, int lower_only,
                          gpmp_stream_t stream) {
  GPMP_ARG(M >= 0 && N >= 0 && K >= 0, 3, "negative size");
  GPMP_ARG(A != nullptr && B != nullptr && C != nullptr, 7, "NULL matrix");
  GPMP_ARG(lda >= (ta ? M : K), 8, "lda too small");
  GPMP_ARG(ldb >= (tb ? K : N), 10, "ldb too small");
  GPMP_ARG(ldc >= N, 13, "ldc < N");
  GemmOpts o;
  o.lower_only = lower_only;
  return launch_gemm(ta == 0, tb != 0, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, o, as_stream(stream));
}
