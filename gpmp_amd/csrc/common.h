// Shared declarations for libgpmp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include "../../include/gpmp_hip.h"

namespace gpmp {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int NB = GPMP_NB;       // diagonal block (potf2 / Dinv granularity) == GEMM tile edge
constexpr int OUTER_BLOCKS = 4;   // outer panel = 4 diagonal blocks (rank-512 trailing updates)

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define GPMP_HIP_TRY(expr)                                   \
  do {                                                       \
    hipError_t e__ = (expr);                                 \
    if (e__ != hipSuccess) return ::gpmp::hip_fail(e__, #expr); \
  } while (0)

#define GPMP_ARG(cond, k, msg)                                          \
  do {                                                                  \
    if (!(cond)) { ::gpmp::set_error("argument %d: %s", (k), (msg)); return -(k); } \
  } while (0)

inline hipStream_t as_stream(gpmp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// compute units of the CURRENT device (cached per device ordinal; 256 when the query fails): launch shapes that are fitted to the
// machine (tile widths, strip widths, persistent grids) ask here, so a host thread per GPU sees its own device's count
int device_cu_count();

// Per-device one-time setup (kernel attributes belong to the device that was current when they were set): one bit per device
// ordinal in a function-local mask.  `done` is set by the caller AFTER the setup succeeded; two host threads racing on the same
// device both run the (idempotent) setup.  A thread-per-GPU host drives every device of the node through one copy of the library.
struct DeviceOnce {
  std::atomic<unsigned long long> mask{0};
  // > 0: this device still needs its setup (the value is the bit to pass to done()); 0: already done; < 0: HIP error
  long long need() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 62) return -1;
    const unsigned long long bit = 1ull << dev;
    return (mask.load(std::memory_order_acquire) & bit) ? 0 : (long long)bit;
  }
  void done(long long bit) { mask.fetch_or((unsigned long long)bit, std::memory_order_release); }
};

// ---- opt-in per-kernel timing with HIP events on the launch stream (capi.cpp) -------------------
// Kinds index the table returned by gpmp_profile_end().
enum ProfKind { PK_GEMM_NT = 0, PK_GEMM_NN = 1, PK_GEMM_TN = 2, PK_GEMM_TT = 3, PK_POTF2 = 4, PK_GRAM = 5,
                PK_COLDOTS = 6, PK_GRAD = 7, PK_GEMM2_NT = 8, PK_GEMM2_NN = 9, PK_GEMM2_TN = 10, PK_GEMM2_TT = 11,
                PK_COUNT = 12 };
extern bool g_prof_on;
void prof_start(int kind, hipStream_t st);
void prof_stop(int kind, hipStream_t st, double work);
struct ProfScope {
  int kind; hipStream_t st; double work; bool on;
  ProfScope(int k, hipStream_t s, double w) : kind(k), st(s), work(w), on(g_prof_on) { if (on) prof_start(kind, st); }
  ~ProfScope() { if (on) prof_stop(kind, st, work); }
};

// ---- GEMM (gemm_f64.hip) --------------------------------------------------------------------
// C (M x N, row-major) = alpha * A(M x K) * B(K x N) + beta * C.
//   a_kc != 0: A(i,l) = A[i*lda + l]   (k-contiguous)   else A(i,l) = A[l*lda + i]
//   b_kc != 0: B(l,j) = B[j*ldb + l]   (k-contiguous)   else B(l,j) = B[l*ldb + j]
// lower_only: tiles strictly above the diagonal are skipped.  kstart_row: the k loop of tile row i
// starts at row0(i) (lauum; K and M index the same axis).  kend_row: the k loop of tile row i ends
// at row0(i) + 128 (triangular A: A(i,l) = 0 for l > i).
struct GemmOpts {
  int lower_only = 0;
  int kstart_row = 0;
  int kend_row = 0;
  int kstart_col = 0;       // B(l, j) = 0 for l < j (B lower triangular): tile column j starts its k loop at col0(j)
  int kend_col = 0;         // B(l, j) = 0 for l > j (B upper triangular as K x N, e.g. the transpose of a lower T given as N x K): tile
                            // column j only needs l < col0(j) + 128
  int batch = 1;            // independent products of one shape: operand / result pointers advance by the strides below
  long stride_a = 0, stride_b = 0, stride_c = 0;   // (elements) per batch index (blockIdx.y)
  int batch2 = 1;           // a second, outer batch dimension (blockIdx.z): independent PROBLEMS, each with `batch` products
  long stride2_a = 0, stride2_b = 0, stride2_c = 0;
  int lean = 0;             // NT products with K <= 512 issued next to a machine-filling GEMM on another stream: take the
                            // small-footprint kernel that starts beside the two resident workgroups of that GEMM on every CU
  // Staircase tile set (the trailing update of a 2-D block-cyclic factor, dist.hip): the group g of 8 tile rows (1024 rows) only has
  // its first  min(tiles_n, 8 (floor((stair_num g + stair_off) / stair_den) + 1 - stair_sub))  tile columns (none when that is
  // <= 0 or the numerator is negative); stair_den > 0 enables it.  Tiles are enumerated group by group, row fastest -- the plain
  // order with a per-group column count -- so the XCD-aware chunking and the 8 x 8 co-residency are kept.
  int stair_num = 0, stair_den = 0, stair_off = 0, stair_sub = 0;
  // Contraction start per GROUP of 8 tile rows / 8 tile columns (the blocks of T^T T on a block-cyclic inverse factor, dist.hip):
  // the k loop of tile (ti, tj) starts at 1024 max(0, ceil((kg_rnum (ti / 8) + kg_roff) / kg_rden), ceil((kg_cnum (tj / 8) + kg_coff)
  // / kg_cden)) -- what lies before is structurally zero in both operands' block columns.  den > 0 enables the respective term; tiles
  // of unequal cost are dealt round-robin over the XCDs.  Combines with the staircase tile set, with nothing else.
  int kg_rnum = 0, kg_rden = 0, kg_roff = 0, kg_cnum = 0, kg_cden = 0, kg_coff = 0;
};
int launch_gemm(bool a_kc, bool b_kc, int M, int N, int K, double alpha, const double* A, long lda,
                const double* B, long ldb, double beta, double* C, long ldc, const GemmOpts& o,
                hipStream_t st);

// Fused leaf of the forward solve (gemm_f64.hip): X = L^-1 B in place for a leaf of nb <= 8 full 128-row blocks and
// ncols (multiple of 128) right-hand sides; G = (128 nb)^2 doubles of scratch.
int launch_trsm_leaf_forward(const double* L, long ldl, const double* dinv_leaf, int nb, double* B, long ldb, int ncols,
                             double* G, hipStream_t st);

// ---- diagonal block kernels (potf2.hip) -------------------------------------------------------
// Factor the jb x jb block at A (lower, in place) and write inv(L) (NB x NB, ld NB, zero padded /
// identity padded) to dinv.  info_dev: set to (offset + k + 1) at the first non-positive pivot.
int launch_potf2_inv(double* A, long lda, int jb, double* dinv, int* info_dev, int offset, hipStream_t st);
int launch_potf2_inv_batch(double* A, long lda, int jb, double* dinv, int* info_dev, int offset, int nprob, long stride_a,
                           long stride_dinv, hipStream_t st);
// inv(L_kk) of every NB diagonal block of an already factored n x n lower L (one launch).
int launch_trtri_blocks(const double* L, long ldl, int n, double* dinv, hipStream_t st);

// ---- few right-hand sides (trsv.hip): in-place op(L)^-1 B, m <= TRSV_FEW_MAX (in passes of at most 8 columns)
constexpr int TRSV_FEW_MAX = 16;      // right-hand sides the HBM-bound sweep takes in one pass
int trsv_few(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, int trans,
             hipStream_t st);

// ---- misc kernels (reduce.hip) -----------------------------------------------------------------
int launch_tril(double* A, int n, long lda, hipStream_t st, int nprob = 1, long prob_stride = 0);
int launch_symmetrize(double* A, int n, long lda, hipStream_t st);
int launch_diag_blocks(double* T, int n, long ldt, const double* dinv, hipStream_t st, int nprob = 1, long prob_stride_t = 0,
                       long prob_stride_dinv = 0);

// ---- batched small problems (linalg.hip): `nprob` independent n x n matrices (n <= GPMP_BATCH_MAX_N), a fixed number of
// elements apart; every step is ONE launch over all problems (blockIdx.y / .z = problem)
struct ProblemBatch {
  int nprob = 1;
  long stride_a = 0;      // between the matrices (factor / inverse factor / inverse), elements
  long stride_dinv = 0;   // between the diagonal-block inverse buffers
};
int potrf_blocked_batch(double* A, int n, long lda, double* dinv, int* info_dev, const ProblemBatch& pb, hipStream_t st);
int trtri_doubling_batch(const double* L, int n, long ldl, const double* dinv, double* T, long ldt, const ProblemBatch& pb,
                         long stride_t, hipStream_t st);
// Gram matrices (lower tiles) / gradient traces of `nprob` problems that share ONE parameter vector, one launch each (gram.hip)
int launch_gram_lower_batch(const double* x, long stride_x, const int* ns_dev, int nmax, int d, int p, const double* theta_host,
                            int noise, double diag_add, double* K, long ldk, long stride_k, int nprob, hipStream_t st,
                            const double* pp_dev = nullptr);
// per-problem parameter blocks of the batched Gram / gradient-trace kernels (pp_dev: nprob blocks in device memory)
int gram_param_block_elems();
void fill_gram_param_block(double* blk, int d, int p, const double* theta, int noise, double diag_add);
int launch_grad_trace_batch(const double* Kinv, long ldk, long stride_kinv, const double* x, long stride_x, const int* ns_dev, int nmax,
                            int d, int p, const double* theta_host, int noise, const double* F, const double* G, int r, long ldf,
                            long stride_f, double* g_dev, double* ws, int nprob, hipStream_t st, const double* pp_dev = nullptr);
int lauum_lower_batch(const double* T, int n, long ldt, long stride_t, double* Kinv, long ldk, long stride_k, int nprob,
                      hipStream_t st);

}  // namespace gpmp
