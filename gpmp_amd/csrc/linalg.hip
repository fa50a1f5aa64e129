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
#include <vector>
#include <cstdlib>
#include <mutex>
#include <map>
#include <memory>
#include <thread>
#include <atomic>

namespace gpmp {
namespace {

inline int imin(int a, int b) { return a < b ? a : b; }
inline int imax(int a, int b) { return a > b ? a : b; }
// switch from the environment, read at every call (A/B runs inside one process, tests that exercise both settings)

// Blocked (two-level) leaf of the recursive factorisation; row0 = global index of A[0][0] (for info).
int potrf_blocked(double* A, int n, long lda, double* dinv, int* info_dev, int row0, hipStream_t st,
                  const ProblemBatch* pb = nullptr) {
  const int nblk = (n + NB - 1) / NB;
  GemmOpts lower;
  lower.lower_only = 1;
  GemmOpts plain;
  GemmOpts scale;          // panel scaling: B operand = the diagonal-block inverses
  if (pb != nullptr && pb->nprob > 1) {
    // every launch below covers all problems: blockIdx.z of the GEMM / blockIdx.y of the diagonal-block kernel
    lower.batch2 = plain.batch2 = scale.batch2 = pb->nprob;
    lower.stride2_a = lower.stride2_b = lower.stride2_c = pb->stride_a;
    plain.stride2_a = plain.stride2_b = plain.stride2_c = pb->stride_a;
    scale.stride2_a = scale.stride2_c = pb->stride_a;
    scale.stride2_b = pb->stride_dinv;
  }
  const bool batched = pb != nullptr && pb->nprob > 1;
  for (int ob = 0; ob < nblk; ob += OUTER_BLOCKS) {
    const int oe = imin(ob + OUTER_BLOCKS, nblk);
    const int out_end = imin(oe * NB, n);
    for (int c = ob; c < oe; ++c) {
      const int c0 = c * NB;
      const int jb = imin(NB, n - c0);
      double* dc = dinv + (size_t)c * NB * NB;
      int rc = batched ? launch_potf2_inv_batch(A + (long)c0 * lda + c0, lda, jb, dc, info_dev, row0 + c0, pb->nprob, pb->stride_a,
                                                pb->stride_dinv, st)
                       : launch_potf2_inv(A + (long)c0 * lda + c0, lda, jb, dc, info_dev, row0 + c0, st);
      if (rc) return rc;
      const int r1 = c0 + jb;
      const int mrem = n - r1;
      if (mrem <= 0) break;
      double* A21 = A + (long)r1 * lda + c0;
      // panel: A21 <- A21 * inv(L_cc)^T  (in place: one 128-wide tile column, K = 128)
      rc = launch_gemm(true, true, mrem, jb, jb, 1.0, A21, lda, dc, NB, 0.0, A21, lda, scale, st);
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

// L X = B (forward), blocked leaf.
int trsm_forward_blocked(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, hipStream_t st) {
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
      const int ncol = m;
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
      const int ncol = m;
      int rc = launch_gemm(true, false, mrem, ncol, kw, -1.0, L + (long)out_end * ldl + (long)ob * NB, ldl,
                           B + (long)ob * NB * ldb, ldb, 1.0, B + (long)out_end * ldb, ldb, plain, st);
      if (rc) return rc;
    }
  }
  return 0;
}

// L^T X = B (backward), blocked leaf.
int trsm_backward_blocked(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb,
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

// ---- recursive drivers -------------------------------------------------------------------------
// Splitting in halves turns almost all of the work into GEMMs whose inner dimension is a large
// fraction of n: the per-tile fill / drain of the MFMA pipeline (about 1.7 k-tiles of 16) is then
// amortised over hundreds of k-tiles instead of 32 (measured: 81 % of peak at K = 512, 87 % at
// K = 4096), and C is read and written log2(n / leaf) times instead of n / 512 times.
constexpr int LEAF_TRSM = 512;   // triangular solves: one outer panel (4 diagonal blocks) per leaf

inline int split_point(int n) {
  // first half size: a multiple of the outer panel (4 * NB), roughly n / 2
  const int unit = OUTER_BLOCKS * NB;
  int n1 = ((n / 2 + unit - 1) / unit) * unit;
  if (n1 >= n) n1 = n - (n > unit ? unit : NB);
  return n1;
}

// B (M x k) <- B * L^-T for a lower-triangular k x k L with diagonal-block inverses dinv (right side;
// the panel solve of a distributed Cholesky step).  Recursive halving over 512-column leaves.
int trsm_right(const double* L, int k, long ldl, const double* dinv, double* B, int M, long ldb, hipStream_t st) {
  if (M <= 0 || k <= 0) return 0;
  GemmOpts plain;
  if (k <= LEAF_TRSM) {
    const int nblk = (k + NB - 1) / NB;
    for (int c = 0; c < nblk; ++c) {
      const int c0 = c * NB, jb = imin(NB, k - c0);
      double* Bc = B + c0;
      int rc = launch_gemm(true, true, M, jb, jb, 1.0, Bc, ldb, dinv + (size_t)c * NB * NB, NB, 0.0, Bc, ldb, plain, st);
      if (rc) return rc;
      const int rest = k - (c0 + jb);
      if (rest > 0) {
        rc = launch_gemm(true, true, M, rest, jb, -1.0, Bc, ldb, L + (long)(c0 + jb) * ldl + c0, ldl, 1.0,
                         B + c0 + jb, ldb, plain, st);
        if (rc) return rc;
      }
    }
    return 0;
  }
  const int k1 = split_point(k);
  int rc = trsm_right(L, k1, ldl, dinv, B, M, ldb, st);
  if (rc) return rc;
  rc = launch_gemm(true, true, M, k - k1, k1, -1.0, B, ldb, L + (long)k1 * ldl, ldl, 1.0, B + k1, ldb, plain, st);
  if (rc) return rc;
  return trsm_right(L + (long)k1 * ldl + k1, k - k1, ldl, dinv + (size_t)(k1 / NB) * NB * NB, B + k1, M, ldb, st);
}

// ---- Cholesky with look-ahead --------------------------------------------------------------------
// Right-looking over outer panels of w = 4 * NB columns.  After panel k is factored, the update of the
// NEXT panel's columns and that panel's factorisation (latency-bound: LDS diagonal kernels, 128-wide
// GEMMs) run on a high-priority helper stream while the main stream applies the rank-w update to the
// rest of the trailing matrix, so the MFMA pipe never waits for a panel except at the very end.
struct LookAhead {
  hipStream_t helper = nullptr;
  hipStream_t side = nullptr;      // later pieces of a split look-ahead update (see potrf_lookahead)
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  hipEvent_t next() {
    if (used == pool.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[used++];
  }
};
// The helper streams, the event pool and the solve stream belong to the device that was current when they were created, and
// the pool is reset at every factorisation: ONE look-ahead factorisation being enqueued at a time PER DEVICE.  State is kept
// per device ordinal (round 4; rounds 1-3 kept one set and refused a second device), so a host that drives several GPUs from one
// process -- one thread per GPU, each with its device current -- uses one copy of the library: threads on different devices do
// not wait for each other, a second thread on the same device waits at that device's mutex (enqueueing takes microseconds).
struct DeviceState {
  LookAhead la;
  hipStream_t solve_stream = nullptr;
  std::mutex mu;
};
std::mutex g_dev_table_mu;
std::map<int, std::shared_ptr<DeviceState>> g_dev_table;
// (shared ownership: a thread that has looked its device's state up keeps it alive across a concurrent gpmp_device_release on
//  another thread -- the release then only drops the table's reference and waits at the state's mutex)
std::shared_ptr<DeviceState> device_state(int dev) {
  std::lock_guard<std::mutex> lk(g_dev_table_mu);
  auto it = g_dev_table.find(dev);
  if (it == g_dev_table.end()) it = g_dev_table.emplace(dev, std::make_shared<DeviceState>()).first;
  return it->second;
}

// Factor the panel of columns [p0, p1) (p0, p1 multiples of NB; rows p0 .. n): diagonal blocks in LDS,
// panel scaling, rank-128 updates inside 512-column sub-panels and rank-512 updates between them.
// `ready` (optional): columns >= ready[i].col of the panel may only be touched after ready[i].ev (pieces of a split
// look-ahead update that are still running on another stream), in increasing order of col.
struct ColsReady { int col; hipEvent_t ev; };
int factor_panel(double* A, int n, long lda, double* dinv, int* info_dev, int p0, int p1, hipStream_t st, int lean = 0,
                 const ColsReady* ready = nullptr, int nready = 0) {
  int iready = 0;
  auto need_cols = [&](int hi) -> int {       // the next kernel touches columns < hi
    while (iready < nready && ready[iready].col < hi) {
      GPMP_HIP_TRY(hipStreamWaitEvent(st, ready[iready].ev, 0));
      ++iready;
    }
    return 0;
  };
  // lean: a machine-filling trailing update is running on the other stream -- the K = 128 products take the
  // small-footprint kernel that starts beside its resident workgroups instead of queueing for a slot
  GemmOpts lower, plain;
  lower.lower_only = 1;
  lower.lean = plain.lean = lean;
  const int sub = OUTER_BLOCKS * NB;
  for (int s0 = p0; s0 < p1; s0 += sub) {
    const int s1 = imin(s0 + sub, p1);   // sub-panel [s0, s1)
    for (int c0 = s0; c0 < s1; c0 += NB) {
      const int jb = imin(NB, n - c0);
      double* dc = dinv + (size_t)(c0 / NB) * NB * NB;
      int rc = launch_potf2_inv(A + (long)c0 * lda + c0, lda, jb, dc, info_dev, c0, st);
      if (rc) return rc;
      const int r1 = c0 + jb;
      const int mrem = n - r1;
      if (mrem <= 0) return 0;
      double* A21 = A + (long)r1 * lda + c0;
      rc = launch_gemm(true, true, mrem, jb, jb, 1.0, A21, lda, dc, NB, 0.0, A21, lda, plain, st);
      if (rc) return rc;
      const int ncols_in = imin(s1, n) - r1;
      if (ncols_in > 0) {
        rc = need_cols(r1 + ncols_in);
        if (rc) return rc;
        rc = launch_gemm(true, true, mrem, ncols_in, jb, -1.0, A21, lda, A21, lda, 1.0, A + (long)r1 * lda + r1, lda,
                         lower, st);
        if (rc) return rc;
      }
    }
    // Between sub-panels: binary blocking.  After sub-panel j (0-based inside the panel) the aligned block of W = sub << ctz(j + 1)
    // columns that ends here is complete, and the next W columns receive it in ONE update of rank W:
    //   1024-column panel:  S0 -> S1 (rank 512)                                  [what rounds 1-3 did]
    //   2048-column panel:  S0 -> S1 (512), [S0 S1] -> [S2 S3] (rank 1024), S2 -> S3 (512)
    // (round 1 updated ALL remaining columns of a wide panel after every sub-panel: rank 512 throughout, which is why 2048-wide
    //  panels did not pay then.)
    const int j = (s0 - p0) / sub;
    int W = sub;
    for (int t = j + 1; (t & 1) == 0; t >>= 1) W <<= 1;
    const int k0 = s1 - W;                     // >= p0 by construction (s1 - p0 is a multiple of W)
    const int rest_cols = imin(imin(p1, n), s1 + W) - s1;
    if (rest_cols > 0 && k0 >= p0) {
      { int rcw = need_cols(s1 + rest_cols); if (rcw) return rcw; }
      // A[s1:, s1:s1+W] -= A[s1:, k0:s1] A[s1:s1+W, k0:s1]^T
      int rc = launch_gemm(true, true, n - s1, rest_cols, s1 - k0, -1.0, A + (long)s1 * lda + k0, lda,
                           A + (long)s1 * lda + k0, lda, 1.0, A + (long)s1 * lda + s1, lda, lower, st);
      if (rc) return rc;
    }
  }
  return need_cols(p1 + 1);                    // (every piece is waited for: the panel event stands for the whole panel)
}

int trsm_forward(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, double* gws, hipStream_t st);

// Solve along (chain-bound sizes, n <= 8192: the factorisation leaves most of the machine idle): B (n x m) <- L^-1 B with the rows
// of every piece of panels solved as soon as those panels are factored, on a third stream -- a block solve
// B_p <- L_pp^-1 (B_p - L_p,<p X_<p) with RIGHT-looking updates (behind the solve of a piece ALL later rows receive its
// contribution at once), so that when the factorisation ends only the last piece's diagonal solve is left.
struct SolveAlong {
  double* B = nullptr;
  int m = 0;
  long ldb = 0;
  double* gws = nullptr;
};

// Schedule constants of the look-ahead factorisation, each the outcome of an A/B in one process (logs: profiles/r1 ... r4,
// HISTORY section 4 "Switches"; rounds 1-4 kept them as environment switches):
constexpr int LA_WIDE_ABOVE = 4096;          // 1024-column panels while more rows than this are left (rank-1024 updates: ~89 % of peak), 256-column panels below: in the chain-bound tail the in-panel updates then ride in the trailing update
constexpr int LA_LEAN_ABOVE = 4096;          // panel products take the small-footprint kernel while the trailing update is at least this large
constexpr int LA_SPLIT_ABOVE = 8192;         // look-ahead update of a 1024-column panel in three column pieces while more rows than this are left
constexpr int LA_MAIN_AFTER_LA_BELOW = 4096; // at or below: a step's trailing update starts only after the next panel's look-ahead update

int potrf_lookahead(double* A, int n, long lda, double* dinv, int* info_dev, hipStream_t s0, const SolveAlong* sa = nullptr) {
  int dev = 0;
  GPMP_HIP_TRY(hipGetDevice(&dev));
  const std::shared_ptr<DeviceState> ds = device_state(dev);
  std::lock_guard<std::mutex> la_lock(ds->mu);
  LookAhead& g_la = ds->la;
  hipStream_t& g_solve_stream = ds->solve_stream;
  std::vector<int> pb;
  for (int p = 0; p < n;) {
    pb.push_back(p);
    p += (n - p) > LA_WIDE_ABOVE ? 2 * OUTER_BLOCKS * NB : 2 * NB;
  }
  pb.push_back(n);
  const int np = (int)pb.size() - 1;
  if (g_la.helper == nullptr) {
    int lo = 0, hi = 0;
    GPMP_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    GPMP_HIP_TRY(hipStreamCreateWithPriority(&g_la.helper, hipStreamNonBlocking, hi));
    GPMP_HIP_TRY(hipStreamCreateWithPriority(&g_la.side, hipStreamNonBlocking, hi));
  }
  hipStream_t s1 = g_la.helper;
  g_la.used = 0;
  GemmOpts lower;
  lower.lower_only = 1;
  // helper starts after everything already queued on the caller's stream (Gram build, memset of info)
  hipEvent_t e = g_la.next();
  GPMP_HIP_TRY(hipEventRecord(e, s0));
  GPMP_HIP_TRY(hipStreamWaitEvent(s1, e, 0));
  int rc = factor_panel(A, n, lda, dinv, info_dev, pb[0], pb[1], s1);
  if (rc) return rc;
  hipEvent_t e_f = g_la.next();                 // panel k factored (on s1)
  GPMP_HIP_TRY(hipEventRecord(e_f, s1));
  hipEvent_t e_u2 = nullptr;                    // trailing update k-1 finished (on s0)
  int n1_solved = 0;
  if (sa != nullptr && g_solve_stream == nullptr) GPMP_HIP_TRY(hipStreamCreateWithFlags(&g_solve_stream, hipStreamNonBlocking));
  if (sa != nullptr) {   // the solve stream starts after everything already queued by the caller (B is built there)
    hipEvent_t eb = g_la.next();
    GPMP_HIP_TRY(hipEventRecord(eb, s0));
    GPMP_HIP_TRY(hipStreamWaitEvent(g_solve_stream, eb, 0));
  }
  // solve along: the rows of B are solved in pieces of >= along_rows rows (whole panels).  A piece [r0, r1) needs
  //   (a) its share of the updates with the rows solved before: issued right behind the previous piece's solve, for ALL later rows
  //       at once, B[r0:n] -= L[r0:n, s0:r0] X[s0:r0] (K = one piece), BEFORE the wait for this piece's panels, and
  //   (b) the triangular solve with L[r0:r1, r0:r1]: once the panels up to r1 are factored.
  // (round 1 issued (a) and (b) together behind the piece's last panel: kernel trace at n = 4096, m = 10000: 1.75 ms of solve
  //  after the last panel, of which 0.83 ms was that piece's (a).)
  int along_rows = 0, upd_end = 0;              // rows < upd_end belong to the piece whose (b) comes next
  auto piece_end = [&](int r0) {
    for (size_t j = 0; j < pb.size(); ++j)
      if (pb[j] > r0 && pb[j] - r0 >= along_rows) return pb[j];
    return n;
  };
  int solved_from = 0;                           // start of the piece solved last
  auto early_update = [&](int r0) -> int {       // (a) behind the piece that ends at r0 = n1_solved
    if (r0 >= n) return 0;
    GemmOpts plain;
    const int rcu = launch_gemm(true, false, n - r0, sa->m, r0 - solved_from, -1.0, A + (long)r0 * lda + solved_from, lda,
                                sa->B + (long)solved_from * sa->ldb, sa->ldb, 1.0, sa->B + (long)r0 * sa->ldb, sa->ldb, plain, g_solve_stream);
    solved_from = r0;
    upd_end = piece_end(r0);
    return rcu;
  };
  if (sa != nullptr) {
    // (a quarter of the matrix at a time measured best: 2048 -> 512, 4096 -> 1024, 8192 -> 2048 rows per piece)
    along_rows = imax(OUTER_BLOCKS * NB, (n / 4) / (OUTER_BLOCKS * NB) * (OUTER_BLOCKS * NB));
    GPMP_HIP_TRY(hipStreamWaitEvent(g_solve_stream, e_f, 0));
    rc = trsm_forward(A, pb[1], lda, dinv, sa->B, sa->m, sa->ldb, sa->gws, g_solve_stream);
    if (rc) return rc;
    n1_solved = pb[1];
    rc = early_update(n1_solved);
    if (rc) return rc;
  }
  for (int k = 0; k + 1 < np; ++k) {
    const int p0 = pb[k], p1 = pb[k + 1], p2 = pb[k + 2];   // panel k = [p0, p1), next panel = [p1, p2)
    const int w = p1 - p0;
    hipStream_t sside = g_la.side;
    // -- helper: update next panel's columns with P_k, then factor it
    if (e_u2) GPMP_HIP_TRY(hipStreamWaitEvent(s1, e_u2, 0));
    const int lean_panel = (n - p2 >= LA_LEAN_ABOVE) ? 1 : 0;
    hipEvent_t e_main_go = e_f;
    // Wide panels: the look-ahead update is cut in three column pieces.  The chain stream does the first 128 columns and
    // starts the diagonal block at once; the other two ([128, 512) and [512, 1024), the second sub-panel) follow on a side
    // stream while the first diagonal blocks are factored, and factor_panel waits for each just before it touches those
    // columns.  (At n = 16384 the panel, not the trailing update, is the longer of the two in EVERY step -- kernel trace: a
    // 1024-column panel = 0.8-1.0 ms of look-ahead update + 8 x 0.3 ms -- so the 0.8 ms in front of the first potf2 were
    // on the critical path.  Below LA_SPLIT_ABOVE rows the pieces are too small to be worth two more events: n = 8192 loses 2 %.)
    ColsReady ready[2];
    int nready = 0;
    if (p2 - p1 == 2 * OUTER_BLOCKS * NB && p2 <= n && n - p1 > LA_SPLIT_ABOVE) {
      const int cuts[4] = {p1, p1 + NB, p1 + OUTER_BLOCKS * NB, p2};
      // the side stream reads panel k: it waits for an event recorded HERE, behind that panel on the chain stream (e_f is
      // only renewed where somebody else waits for it -- need_ef below -- and may be an older panel's)
      hipEvent_t e_panel = g_la.next();
      GPMP_HIP_TRY(hipEventRecord(e_panel, s1));
      GPMP_HIP_TRY(hipStreamWaitEvent(sside, e_panel, 0));
      if (e_u2) GPMP_HIP_TRY(hipStreamWaitEvent(sside, e_u2, 0));
      for (int q = 0; q < 3; ++q) {
        const int ca = cuts[q], cb = cuts[q + 1];
        hipStream_t sq = q == 0 ? s1 : sside;
        rc = launch_gemm(true, true, n - ca, cb - ca, w, -1.0, A + (long)ca * lda + p0, lda, A + (long)ca * lda + p0, lda,
                         1.0, A + (long)ca * lda + ca, lda, lower, sq);
        if (rc) return rc;
        if (q > 0) {
          ready[nready].col = ca;
          ready[nready].ev = g_la.next();
          GPMP_HIP_TRY(hipEventRecord(ready[nready].ev, sside));
          ++nready;
        }
      }
    } else {
      rc = launch_gemm(true, true, n - p1, p2 - p1, w, -1.0, A + (long)p1 * lda + p0, lda, A + (long)p1 * lda + p0, lda,
                       1.0, A + (long)p1 * lda + p1, lda, lower, s1);
      if (rc) return rc;
    }
    // chain-bound tail: the trailing update of this step starts only when the look-ahead update above has finished, so
    // that the latter -- on the critical chain -- does not share the machine with it (kernel trace, n = 4096: 12 us alone,
    // 37 us when both start together); the trailing update has slack there
    if (n - p1 <= LA_MAIN_AFTER_LA_BELOW) {
      e_main_go = g_la.next();
      GPMP_HIP_TRY(hipEventRecord(e_main_go, s1));
    }
    // (the main stream's update of this iteration covers (n - p2)^2 / 2: with at least two rounds of tiles it holds every
    //  workgroup slot of the machine while this panel is factored; with the solve along, the solve stream's GEMMs hold the slots)
    rc = factor_panel(A, n, lda, dinv, info_dev, p1, p2, s1, lean_panel, ready, nready);
    if (rc) return rc;
    // "panel k+1 factored": only recorded where somebody waits for it -- the trailing update of the next step (unless that
    // one starts behind its look-ahead update anyway), the solve stream, the final join.  An event between two kernels of
    // the chain stream costs ~5 us of packet processing (kernel trace: 9-11 us gaps around the look-ahead update against
    // 0-1 us between kernels that follow each other directly).
    const bool need_ef = sa != nullptr || k + 2 >= np || (n - p2 > LA_MAIN_AFTER_LA_BELOW);
    hipEvent_t e_f_next = e_f;
    if (need_ef) {
      e_f_next = g_la.next();
      GPMP_HIP_TRY(hipEventRecord(e_f_next, s1));
    }
    if (sa != nullptr && p2 == upd_end) {
      // rows [r0, p2) of B: their update with the rows solved before was issued early (above); the diagonal part goes
      // behind the factorisation of the last of these panels, followed at once by the update of everything below
      const int r0 = n1_solved;
      GPMP_HIP_TRY(hipStreamWaitEvent(g_solve_stream, e_f_next, 0));
      rc = trsm_forward(A + (long)r0 * lda + r0, p2 - r0, lda, dinv + (size_t)(r0 / NB) * NB * NB, sa->B + (long)r0 * sa->ldb,
                        sa->m, sa->ldb, sa->gws, g_solve_stream);
      if (rc) return rc;
      n1_solved = p2;
      rc = early_update(n1_solved);
      if (rc) return rc;
    }
    // -- main: rank-w update of the rest of the trailing matrix with P_k
    GPMP_HIP_TRY(hipStreamWaitEvent(s0, e_main_go, 0));
    if (p2 < n) {
      rc = launch_gemm(true, true, n - p2, n - p2, w, -1.0, A + (long)p2 * lda + p0, lda, A + (long)p2 * lda + p0, lda,
                       1.0, A + (long)p2 * lda + p2, lda, lower, s0);
      if (rc) return rc;
    }
    e_u2 = g_la.next();
    GPMP_HIP_TRY(hipEventRecord(e_u2, s0));
    e_f = e_f_next;
  }
  GPMP_HIP_TRY(hipStreamWaitEvent(s0, e_f, 0));  // join: everything visible to the caller's stream
  if (sa != nullptr) {                           // n1_solved == n: every piece's rows were solved behind its panels
    hipEvent_t e_rows = g_la.next();
    GPMP_HIP_TRY(hipEventRecord(e_rows, g_solve_stream));
    GPMP_HIP_TRY(hipStreamWaitEvent(s0, e_rows, 0));
  }
  return 0;
}

// One stream, no look-ahead, up to 2048 columns: nothing runs beside the chain's kernels there, and the look-ahead's second
// stream only adds event packets and a trailing update that slows the chain (measured in one process, both routes:
// n = 1536: 0.69 vs 0.77 ms, 2048: 0.96 vs 1.01, 3072: 1.68 vs 1.57, 4096: 2.47 vs 2.17).
constexpr int POTRF_ONE_STREAM_MAX = 4 * OUTER_BLOCKS * NB;

int potrf_lower(double* A, int n, long lda, double* dinv, int* info_dev, hipStream_t st) {
  GPMP_HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), st));
  if (n <= POTRF_ONE_STREAM_MAX) return potrf_blocked(A, n, lda, dinv, info_dev, 0, st);
  return potrf_lookahead(A, n, lda, dinv, info_dev, st);
}

// gws: optional scratch of at least (LEAF_TRSM)^2 doubles; enables the fused leaf (gemm_f64.hip: trsm_leaf_kernel)
int trsm_forward(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, double* gws, hipStream_t st) {
  if (n <= LEAF_TRSM) {
    const bool ok = gws != nullptr && n % NB == 0 && m >= 4 * NB && (m % 2 == 0) && (ldb % 2 == 0) &&
                    ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && ((long)n * ldb * 8 < 0x7FFFFFFFL);
    if (ok) return launch_trsm_leaf_forward(L, ldl, dinv, n / NB, B, ldb, m, gws, st);
    return trsm_forward_blocked(L, n, ldl, dinv, B, m, ldb, st);
  }
  const int n1 = split_point(n);
  int rc = trsm_forward(L, n1, ldl, dinv, B, m, ldb, gws, st);
  if (rc) return rc;
  GemmOpts plain;
  // B2 -= L21 * X1
  rc = launch_gemm(true, false, n - n1, m, n1, -1.0, L + (long)n1 * ldl, ldl, B, ldb, 1.0, B + (long)n1 * ldb, ldb, plain, st);
  if (rc) return rc;
  return trsm_forward(L + (long)n1 * ldl + n1, n - n1, ldl, dinv + (size_t)(n1 / NB) * NB * NB, B + (long)n1 * ldb, m, ldb, gws, st);
}

int trsm_backward(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, hipStream_t st) {
  if (n <= LEAF_TRSM) return trsm_backward_blocked(L, n, ldl, dinv, B, m, ldb, st);
  const int n1 = split_point(n);
  int rc = trsm_backward(L + (long)n1 * ldl + n1, n - n1, ldl, dinv + (size_t)(n1 / NB) * NB * NB, B + (long)n1 * ldb, m,
                         ldb, st);
  if (rc) return rc;
  GemmOpts plain;
  // B1 -= L21^T * X2
  rc = launch_gemm(false, false, n1, m, n - n1, -1.0, L + (long)n1 * ldl, ldl, B + (long)n1 * ldb, ldb, 1.0, B, ldb, plain, st);
  if (rc) return rc;
  return trsm_backward(L, n1, ldl, dinv, B, m, ldb, st);
}

// T = L^-1 by doubling: the diagonal 128-blocks come from dinv; at level s = 128, 256, ... every pair of adjacent
// blocks [T11 0; T21 T22] of sizes (s, len2 <= s) gets T21 = -T22 (L21 T11).  All pairs of a level have the same shape
// (except a ragged last one), so a level is TWO batched launches whatever n is: 2 log2(n / 128) launches in total instead
// of the 31 large + 224 small ones of the forward solve on the identity at n = 16384.  The intermediate W = L21 T11 is
// kept transposed in the pair's T12 block (s x len2: always fits), which is otherwise zero; the triangular structure of
// T11 (first product, as the transposed left operand) and of T22 (second product) is skipped tile-wise, so the W blocks
// above the diagonal are never read as part of a triangle; they are zeroed at the end.
int trtri_doubling(const double* L, int n, long ldl, const double* dinv, double* T, long ldt, hipStream_t st,
                   const ProblemBatch* pb = nullptr, long stride_t = 0) {
  const int nprob = pb != nullptr ? pb->nprob : 1;
  const long sl = pb != nullptr ? pb->stride_a : 0;
  int rc = launch_diag_blocks(T, n, ldt, dinv, st, nprob, stride_t, pb != nullptr ? pb->stride_dinv : 0);
  if (rc) return rc;
  for (long s = NB; s < n; s *= 2) {
    const int npairs = (int)(n / (2 * s));                  // pairs with two full halves
    const long tail0 = (long)npairs * 2 * s;                // a ragged pair starts here if tail0 + s < n
    for (int part = 0; part < 2; ++part) {
      const long o = part == 0 ? 0 : tail0;
      const int len2 = part == 0 ? (int)s : (int)(n - (tail0 + s));
      const int batch = part == 0 ? npairs : 1;
      if (batch <= 0 || len2 <= 0) continue;
      const double* T11 = T + o * ldt + o;
      const double* L21 = L + (o + s) * ldl + o;
      double* Wt = T + o * ldt + (o + s);                    // s x len2, W^T = T11^T L21^T
      double* T21 = T + (o + s) * ldt + o;
      const double* T22 = T + (o + s) * ldt + (o + s);
      if (part == 0) {
        // full pairs (len2 == s): W = L21 T11 kept UNtransposed in the T12 block, so that both products are of the NN kind --
        // the LDS-direct kernel's best operand layout (k-contiguous left operand, n-contiguous right operand: 91-92 % of peak
        // at K >= 2048 against 82 % for the transposed-left kind the W^T form needs for its first product)
        double* W = Wt;
        GemmOpts g1;
        g1.kstart_col = 1;                                   // T11(l, j) = 0 for l < j
        g1.batch = batch; g1.stride_a = 2 * s * (ldl + 1); g1.stride_b = 2 * s * (ldt + 1); g1.stride_c = 2 * s * (ldt + 1);
        g1.batch2 = nprob; g1.stride2_a = sl; g1.stride2_b = stride_t; g1.stride2_c = stride_t;
        rc = launch_gemm(true, false, len2, (int)s, (int)s, 1.0, L21, ldl, T11, ldt, 0.0, W, ldt, g1, st);
        if (rc) return rc;
        GemmOpts g2;
        g2.kend_row = 1;                                     // T22(i, l) = 0 for l > i
        g2.batch = batch; g2.stride_a = g2.stride_b = g2.stride_c = 2 * s * (ldt + 1);
        g2.batch2 = nprob; g2.stride2_a = g2.stride2_b = g2.stride2_c = stride_t;
        rc = launch_gemm(true, false, len2, (int)s, len2, -1.0, T22, ldt, W, ldt, 0.0, T21, ldt, g2, st);
        if (rc) return rc;
        continue;
      }
      GemmOpts g1;
      g1.kstart_row = 1;                                     // (T11^T)(i, l) = T11(l, i) = 0 for l < i
      g1.batch = batch; g1.stride_a = 2 * s * (ldt + 1); g1.stride_b = 2 * s * (ldl + 1); g1.stride_c = 2 * s * (ldt + 1);
      g1.batch2 = nprob; g1.stride2_a = stride_t; g1.stride2_b = sl; g1.stride2_c = stride_t;
      rc = launch_gemm(false, true, (int)s, len2, (int)s, 1.0, T11, ldt, L21, ldl, 0.0, Wt, ldt, g1, st);
      if (rc) return rc;
      GemmOpts g2;
      g2.kend_row = 1;                                       // T22(i, l) = 0 for l > i
      g2.batch = batch; g2.stride_a = g2.stride_b = g2.stride_c = 2 * s * (ldt + 1);
      g2.batch2 = nprob; g2.stride2_a = g2.stride2_b = g2.stride2_c = stride_t;
      rc = launch_gemm(true, true, len2, (int)s, len2, -1.0, T22, ldt, Wt, ldt, 0.0, T21, ldt, g2, st);
      if (rc) return rc;
    }
  }
  return launch_tril(T, n, ldt, st, nprob, stride_t);
}

}  // namespace

// ---- batched small problems (drivers_batch.hip) ---------------------------------------------------------------------
int potrf_blocked_batch(double* A, int n, long lda, double* dinv, int* info_dev, const ProblemBatch& pb, hipStream_t st) {
  GPMP_HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int) * (size_t)pb.nprob, st));
  return potrf_blocked(A, n, lda, dinv, info_dev, 0, st, &pb);
}
int trtri_doubling_batch(const double* L, int n, long ldl, const double* dinv, double* T, long ldt, const ProblemBatch& pb,
                         long stride_t, hipStream_t st) {
  return trtri_doubling(L, n, ldl, dinv, T, ldt, st, &pb, stride_t);
}
int lauum_lower_batch(const double* T, int n, long ldt, long stride_t, double* Kinv, long ldk, long stride_k, int nprob,
                      hipStream_t st) {
  GemmOpts o;
  o.lower_only = 1;
  o.kstart_row = 1;
  o.batch2 = nprob; o.stride2_a = o.stride2_b = stride_t; o.stride2_c = stride_k;
  return launch_gemm(false, false, n, n, n, 1.0, T, ldt, T, ldt, 0.0, Kinv, ldk, o, st);
}
}  // namespace gpmp

using namespace gpmp;

// ---- per-device state of the look-ahead factorisation (helper streams, event pool, solve stream) ----------------------
namespace gpmp {
namespace {
int release_device_state(int dev) {
  std::shared_ptr<DeviceState> ds;
  {
    std::lock_guard<std::mutex> lk(g_dev_table_mu);
    auto it = g_dev_table.find(dev);
    if (it == g_dev_table.end()) return 0;
    ds = std::move(it->second);
    g_dev_table.erase(it);
  }
  std::lock_guard<std::mutex> lk(ds->mu);          // an enqueue section still running on another thread finishes first
  // (streams are synchronised before they go: their kernels may still be running.)  Every handle is released whatever the
  // earlier ones answered; the first error is what the caller gets.
  hipError_t first = hipSuccess;
  auto note = [&](hipError_t e) { if (e != hipSuccess && first == hipSuccess) first = e; };
  for (hipStream_t* st : {&ds->la.helper, &ds->la.side, &ds->solve_stream})
    if (*st != nullptr) {
      note(hipStreamSynchronize(*st));
      note(hipStreamDestroy(*st));
      *st = nullptr;
    }
  for (hipEvent_t e : ds->la.pool) note(hipEventDestroy(e));
  ds->la.pool.clear();
  ds->la.used = 0;
  return first == hipSuccess ? 0 : hip_fail(first, "gpmp_device_release");
}
}  // namespace
}  // namespace gpmp

extern "C" int gpmp_device_state_count(void) {
  std::lock_guard<std::mutex> lk(g_dev_table_mu);
  return (int)g_dev_table.size();
}

extern "C" int gpmp_device_release(void) {
  int dev = 0;
  GPMP_HIP_TRY(hipGetDevice(&dev));
  return release_device_state(dev);
}

// Host-only self-test of the device table (no HIP call: the entries it makes hold no stream): `threads` host threads look up
// `ordinals` made-up device ordinals (1000, 1001, ...) `iters` times each, every thread must see ONE state object per ordinal,
// then the entries are released.  0 = consistent.  Run under the host-side sanitizer build (tests/test_asan_cpu.py).
extern "C" int gpmp_debug_device_table_selftest(int threads, int ordinals, int iters) {
  GPMP_ARG(threads > 0 && threads <= 64, 1, "threads outside [1, 64]");
  GPMP_ARG(ordinals > 0 && ordinals <= 64, 2, "ordinals outside [1, 64]");
  const int before = gpmp_device_state_count();
  std::vector<std::vector<DeviceState*>> seen(threads, std::vector<DeviceState*>(ordinals, nullptr));
  std::atomic<int> bad{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      for (int it = 0; it < iters; ++it)
        for (int o = 0; o < ordinals; ++o) {
          const std::shared_ptr<DeviceState> ds = device_state(1000 + (o + t) % ordinals);
          std::lock_guard<std::mutex> lk(ds->mu);           // what an enqueue section does
          ds->la.used = 0;
          DeviceState*& slot = seen[t][(o + t) % ordinals];
          if (slot == nullptr) slot = ds.get();
          else if (slot != ds.get()) bad.fetch_add(1);
        }
    });
  for (auto& th : pool) th.join();
  for (int t = 1; t < threads; ++t)
    for (int o = 0; o < ordinals; ++o)
      if (seen[t][o] != seen[0][o]) bad.fetch_add(1);
  if (gpmp_device_state_count() != before + ordinals) bad.fetch_add(1);
  for (int o = 0; o < ordinals; ++o)
    if (release_device_state(1000 + o)) bad.fetch_add(1);
  if (gpmp_device_state_count() != before) bad.fetch_add(1);
  return bad.load();
}

extern "C" size_t gpmp_dinv_elems(int n) {
  if (n <= 0) return 0;
  // inverse diagonal blocks + (n > 1024) one 1024 x 1024 scratch area for the fused solve leaves
  return (size_t)((n + NB - 1) / NB) * NB * NB + (n > 2 * OUTER_BLOCKS * NB ? (size_t)4 * OUTER_BLOCKS * OUTER_BLOCKS * NB * NB : 0);
}

extern "C" int gpmp_potrf_lower_async(double* A, int n, long lda, double* dinv, int* info_dev,
                                      gpmp_stream_t stream) {
  GPMP_ARG(A != nullptr, 1, "A is NULL");
  GPMP_ARG(n >= 0 && n <= GPMP_MAX_EXTENT, 2, "n outside [0, GPMP_MAX_EXTENT]");
  GPMP_ARG(lda >= n, 3, "lda < n");
  GPMP_ARG(dinv != nullptr, 4, "dinv is NULL");
  GPMP_ARG(info_dev != nullptr, 5, "info is NULL");
  if (n == 0) return 0;
  return potrf_lower(A, n, lda, dinv, info_dev, as_stream(stream));
}

extern "C" int gpmp_potrf_trsm_lower_async(double* A, int n, long lda, double* dinv, int* info_dev, double* B, int m, long ldb,
                                           gpmp_stream_t stream) {
  GPMP_ARG(A != nullptr, 1, "A is NULL");
  GPMP_ARG(n >= 0 && n <= GPMP_MAX_EXTENT, 2, "n outside [0, GPMP_MAX_EXTENT]");
  GPMP_ARG(lda >= n, 3, "lda < n");
  GPMP_ARG(dinv != nullptr, 4, "dinv is NULL");
  GPMP_ARG(info_dev != nullptr, 5, "info_dev is NULL");
  GPMP_ARG(B != nullptr, 6, "B is NULL");
  GPMP_ARG(m >= 0 && m <= GPMP_MAX_EXTENT && ldb >= m, 8, "m outside [0, GPMP_MAX_EXTENT] or ldb < m");
  if (n == 0) return 0;
  hipStream_t st = as_stream(stream);
  GPMP_HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), st));
  double* gws = (n > 2 * OUTER_BLOCKS * NB) ? dinv + (size_t)((n + NB - 1) / NB) * NB * NB : nullptr;
  // Chain-bound sizes: the factorisation of n <= 8192 leaves most of the machine idle (3.2 ms for 0.33 ms of MFMA work at
  // n = 4096), so the rows of every piece of panels are solved behind those panels' factorisation on a third stream (SolveAlong).
  // Above that the solve follows the factorisation on the caller's stream: overlapping its leading half with the trailing half of
  // the factorisation gained 5-7 ms of a 950 ms call (two GEMM streams lose ~7 % to each other) and was dropped.
  constexpr int along_above = 2 * OUTER_BLOCKS * NB, along_below = 8192;
  if (m > TRSV_FEW_MAX && n > along_above && n <= along_below) {
    SolveAlong sa;
    sa.B = B; sa.m = m; sa.ldb = ldb; sa.gws = gws;
    return potrf_lookahead(A, n, lda, dinv, info_dev, st, &sa);
  }
  int rc = (n <= POTRF_ONE_STREAM_MAX) ? potrf_blocked(A, n, lda, dinv, info_dev, 0, st) : potrf_lookahead(A, n, lda, dinv, info_dev, st);
  if (rc || m == 0) return rc;
  if (m <= TRSV_FEW_MAX) return trsv_few(A, n, lda, dinv, B, m, ldb, 0, st);
  return trsm_forward(A, n, lda, dinv, B, m, ldb, gws, st);
}

extern "C" int gpmp_trtri_diag_blocks(const double* L, int n, long ldl, double* dinv, gpmp_stream_t stream) {
  GPMP_ARG(L != nullptr, 1, "L is NULL");
  GPMP_ARG(n <= GPMP_MAX_EXTENT, 2, "n above GPMP_MAX_EXTENT");
  GPMP_ARG(ldl >= n, 3, "ldl < n");
  GPMP_ARG(dinv != nullptr, 4, "dinv is NULL");
  return launch_trtri_blocks(L, ldl, n, dinv, as_stream(stream));
}

extern "C" int gpmp_trsm_lower(const double* L, int n, long ldl, const double* dinv, double* B, int m,
                               long ldb, int trans, double* scratch, gpmp_stream_t stream) {
  GPMP_ARG(L != nullptr, 1, "L is NULL");
  GPMP_ARG(n >= 0 && n <= GPMP_MAX_EXTENT, 2, "n outside [0, GPMP_MAX_EXTENT]");
  GPMP_ARG(ldl >= n, 3, "ldl < n");
  GPMP_ARG(B != nullptr, 5, "B is NULL");
  GPMP_ARG(m >= 0 && m <= GPMP_MAX_EXTENT && ldb >= m, 7, "m outside [0, GPMP_MAX_EXTENT] or ldb < m");
  GPMP_ARG(dinv != nullptr || scratch != nullptr, 9, "dinv and scratch both NULL");
  if (n == 0 || m == 0) return 0;
  hipStream_t st = as_stream(stream);
  if (dinv == nullptr) {
    int rc = launch_trtri_blocks(L, ldl, n, scratch, st);
    if (rc) return rc;
    dinv = scratch;
  }
  if (m <= TRSV_FEW_MAX) return trsv_few(L, n, ldl, dinv, B, m, ldb, trans, st);   // HBM-bound fused sweep
  // the panel scratch behind the block inverses (n > 1024, see gpmp_dinv_elems) of `scratch` enables the fused leaf
  double* gws = (scratch != nullptr && n > 2 * OUTER_BLOCKS * NB) ? scratch + (size_t)((n + NB - 1) / NB) * NB * NB : nullptr;
  return trans ? trsm_backward(L, n, ldl, dinv, B, m, ldb, st) : trsm_forward(L, n, ldl, dinv, B, m, ldb, gws, st);
}

extern "C" int gpmp_trsm_right_lower(const double* L, int k, long ldl, const double* dinv, double* B, int M, long ldb,
                                     gpmp_stream_t stream) {
  GPMP_ARG(L != nullptr, 1, "L is NULL");
  GPMP_ARG(k >= 0 && k <= GPMP_MAX_EXTENT && ldl >= k, 3, "k outside [0, GPMP_MAX_EXTENT] or ldl < k");
  GPMP_ARG(dinv != nullptr, 4, "dinv is NULL");
  GPMP_ARG(B != nullptr, 5, "B is NULL");
  GPMP_ARG(M >= 0 && M <= GPMP_MAX_EXTENT && ldb >= k, 7, "M outside [0, GPMP_MAX_EXTENT] or ldb < k");
  return trsm_right(L, k, ldl, dinv, B, M, ldb, as_stream(stream));
}

extern "C" int gpmp_trtri_lower(const double* L, int n, long ldl, const double* dinv, double* T, long ldt,
                                gpmp_stream_t stream) {
  GPMP_ARG(L != nullptr, 1, "L is NULL");
  GPMP_ARG(n <= GPMP_MAX_EXTENT, 2, "n above GPMP_MAX_EXTENT");
  GPMP_ARG(ldl >= n, 3, "ldl < n");
  GPMP_ARG(dinv != nullptr, 4, "dinv is NULL");
  GPMP_ARG(T != nullptr && ldt >= n, 5, "T is NULL or ldt < n");
  if (n <= 0) return 0;
  return trtri_doubling(L, n, ldl, dinv, T, ldt, as_stream(stream));
}

extern "C" int gpmp_lauum_lower(const double* T, int n, long ldt, double* Kinv, long ldk, gpmp_stream_t stream) {
  GPMP_ARG(T != nullptr && ldt >= n, 1, "T is NULL or ldt < n");
  GPMP_ARG(n <= GPMP_MAX_EXTENT, 2, "n above GPMP_MAX_EXTENT");
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
  GPMP_ARG(M >= 0 && N >= 0 && K >= 0 && M <= GPMP_MAX_EXTENT && N <= GPMP_MAX_EXTENT && K <= GPMP_MAX_EXTENT, 3, "size outside [0, GPMP_MAX_EXTENT]");
  GPMP_ARG(A != nullptr && B != nullptr && C != nullptr, 7, "NULL matrix");
  GPMP_ARG(lda >= (ta ? M : K), 8, "lda too small");
  GPMP_ARG(ldb >= (tb ? K : N), 10, "ldb too small");
  GPMP_ARG(ldc >= N, 13, "ldc < N");
  GemmOpts o;
  o.lower_only = lower_only & 1;
  o.lean = (lower_only >> 1) & 1;
  o.kend_col = (lower_only >> 2) & 1;
  return launch_gemm(ta == 0, tb != 0, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, o, as_stream(stream));
}
