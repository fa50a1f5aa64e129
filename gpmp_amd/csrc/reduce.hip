// HBM-bound helper kernels: column dots / sums of squares over row-major matrices (the einsum
// "i..., i..." reductions of gpmp/core/kriging.py:194, model.py:298-300, linalg.py:44), log-det of
// a Cholesky factor (likelihood.py:50), triangle utilities.
#include "common.h"

namespace gpmp {
namespace {

constexpr int CD_R = 8;          // Y columns per pass
constexpr int CD_MAX_CHUNKS = 32;

// partial[(chunk * (CD_R + 1) + k) * m + j] = sum_{i in chunk} V[i,j] * Y[i,k0+k]  (k < rk)
// partial[(chunk * (CD_R + 1) + CD_R) * m + j] = sum_{i in chunk} V[i,j]^2          (if want_sq)
__global__ void __launch_bounds__(256) coldots_partial_kernel(const double* __restrict__ V, int n, int m,
                                                              long ldv, const double* __restrict__ Y,
                                                              int k0, int rk, long ldy, int rows_per,
                                                              int want_sq, double* __restrict__ partial) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int chunk = blockIdx.y;
  const int i0 = chunk * rows_per;
  const int i1 = (i0 + rows_per) < n ? (i0 + rows_per) : n;
  double acc[CD_R];
#pragma unroll
  for (int k = 0; k < CD_R; ++k) acc[k] = 0.0;
  double sq = 0.0;
  if (j < m) {
#pragma unroll 4
    for (int i = i0; i < i1; ++i) {
      const double v = V[(long)i * ldv + j];
      const double* yr = Y + (long)i * ldy + k0;  // wave-uniform address -> scalar loads
#pragma unroll
      for (int k = 0; k < CD_R; ++k) {
        const double yk = (k < rk) ? yr[k] : 0.0;
        acc[k] = fma(v, yk, acc[k]);
      }
      sq = fma(v, v, sq);
    }
    double* out = partial + (long)chunk * (CD_R + 1) * m + j;
#pragma unroll
    for (int k = 0; k < CD_R; ++k)
      if (k < rk) out[(long)k * m] = acc[k];
    if (want_sq) out[(long)CD_R * m] = sq;
  }
}

__global__ void coldots_final_kernel(const double* __restrict__ partial, int m, int nchunks, int k0, int rk,
                                     int r_total, int want_sq, double* __restrict__ out, long ldo) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  for (int k = 0; k < rk; ++k) {
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += partial[((long)c * (CD_R + 1) + k) * m + j];
    out[(long)(k0 + k) * ldo + j] = s;
  }
  if (want_sq) {
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += partial[((long)c * (CD_R + 1) + CD_R) * m + j];
    out[(long)r_total * ldo + j] = s;
  }
}

// partial[chunk * m + j] = sum_{i in chunk} A[i,j] * B[i,j]: the matrix x matrix form of einsum("i..., i...") -- the reference's
// own variance reduction sum_i lambda_t[i,j] Kit[i,j] (kriging.py:194).  Two coalesced streams, 16 B per flop pair: HBM-bound.
__global__ void __launch_bounds__(256) colpair_partial_kernel(const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                              long ldb, int n, int m, int rows_per, double* __restrict__ partial) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * rows_per;
  const int i1 = (i0 + rows_per) < n ? (i0 + rows_per) : n;
  if (j >= m) return;
  double acc0 = 0.0, acc1 = 0.0;
  int i = i0;
  for (; i + 1 < i1; i += 2) {
    acc0 = fma(A[(long)i * lda + j], B[(long)i * ldb + j], acc0);
    acc1 = fma(A[(long)(i + 1) * lda + j], B[(long)(i + 1) * ldb + j], acc1);
  }
  if (i < i1) acc0 = fma(A[(long)i * lda + j], B[(long)i * ldb + j], acc0);
  partial[(long)blockIdx.y * m + j] = acc0 + acc1;
}

__global__ void colpair_final_kernel(const double* __restrict__ partial, int m, int nchunks, double* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  double s = 0.0;
  for (int c = 0; c < nchunks; ++c) s += partial[(long)c * m + j];
  out[j] = s;
}

__global__ void __launch_bounds__(1024) logdet_kernel(const double* __restrict__ L, int n, long ldl,
                                                      double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += log(L[(long)i * ldl + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < 16; ++w) tot += red[w];
    out[0] = 2.0 * tot;
  }
}

__global__ void tril_kernel(double* __restrict__ A, int n, long lda, long prob_stride) {
  A += (long)blockIdx.z * prob_stride;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  for (int i = blockIdx.y; i < n; i += gridDim.y)
    if (j > i) A[(long)i * lda + j] = 0.0;
}

// A[j][i] = A[i][j] for i > j, through a 32x33 LDS tile so both sides stay coalesced.
__global__ void __launch_bounds__(256) symmetrize_kernel(double* __restrict__ A, int n, long lda) {
  __shared__ double tile[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    tile[r][tx] = (i < n && j < n) ? A[(long)i * lda + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int jj = bj * 32 + r, ii = bi * 32 + tx;  // writes A[jj][ii] = A[ii][jj]
    if (jj < n && ii < n && ii > jj) A[(long)jj * lda + ii] = tile[tx][r];
  }
}

// T's diagonal 128 x 128 blocks <- the block inverses (dinv: [block][128][128], identity-padded last block)
__global__ void diag_blocks_kernel(double* __restrict__ T, int n, long ldt, const double* __restrict__ dinv, long prob_stride_t,
                                   long prob_stride_dinv) {
  T += (long)blockIdx.y * prob_stride_t;
  dinv += (long)blockIdx.y * prob_stride_dinv;
  const int b = blockIdx.x, b0 = b * NB;
  const int jb = (n - b0) < NB ? (n - b0) : NB;
  const double* D = dinv + (size_t)b * NB * NB;
  for (int idx = threadIdx.x; idx < NB * NB; idx += blockDim.x) {
    const int i = idx >> 7, j = idx & (NB - 1);
    if (i < jb && j < jb) T[(long)(b0 + i) * ldt + b0 + j] = D[idx];
  }
}

}  // namespace

int launch_tril(double* A, int n, long lda, hipStream_t st, int nprob, long prob_stride) {
  if (n <= 1) return 0;
  hipLaunchKernelGGL(tril_kernel, dim3((n + 255) / 256, n < 16384 ? n : 16384, nprob), dim3(256), 0, st, A, n, lda, prob_stride);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}
int launch_symmetrize(double* A, int n, long lda, hipStream_t st) {
  if (n <= 1) return 0;
  const int nb = (n + 31) / 32;
  hipLaunchKernelGGL(symmetrize_kernel, dim3(nb, nb), dim3(256), 0, st, A, n, lda);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}
int launch_diag_blocks(double* T, int n, long ldt, const double* dinv, hipStream_t st, int nprob, long prob_stride_t,
                       long prob_stride_dinv) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(diag_blocks_kernel, dim3((n + NB - 1) / NB, nprob), dim3(256), 0, st, T, n, ldt, dinv, prob_stride_t,
                     prob_stride_dinv);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace gpmp

using namespace gpmp;

extern "C" int gpmp_coldots_ws_rows(int n) {
  // (128 rows per chunk until the cap: at n = 4096, m = 10000 the pass ran as 320 workgroups of 512 rows each -- 1.4 TB/s;
  //  with 1280 workgroups of 128 rows it is bandwidth-bound like the large cases)
  int chunks = (n + 127) / 128;
  if (chunks < 1) chunks = 1;
  if (chunks > CD_MAX_CHUNKS) chunks = CD_MAX_CHUNKS;
  return chunks * (CD_R + 1);
}

extern "C" int gpmp_coldots(const double* V, int n, int m, long ldv, const double* Y, int r, long ldy,
                            double* out, long ldo, double* ws, gpmp_stream_t stream) {
  GPMP_ARG(V != nullptr, 1, "V is NULL");
  GPMP_ARG(n >= 0 && m >= 0 && n <= GPMP_MAX_EXTENT && m <= GPMP_MAX_EXTENT, 2, "size outside [0, GPMP_MAX_EXTENT]");
  GPMP_ARG(ldv >= m, 4, "ldv < m");
  GPMP_ARG(r >= 0 && r <= GPMP_MAX_RANK, 6, "r outside [0, GPMP_MAX_RANK]");
  GPMP_ARG(r == 0 || (Y != nullptr && ldy >= r), 5, "Y is NULL or ldy < r");
  GPMP_ARG(out != nullptr && ws != nullptr && ldo >= m, 8, "out or ws is NULL, or ldo < m");
  if (m == 0) return 0;
  hipStream_t st = as_stream(stream);
  const int nchunks = gpmp_coldots_ws_rows(n) / (CD_R + 1);
  const int rows_per = (n + nchunks - 1) / nchunks > 0 ? (n + nchunks - 1) / nchunks : 1;
  int k0 = 0;
  bool first = true;
  ProfScope ps(PK_COLDOTS, st, 8.0 * (double)n * (double)m);
  do {
    const int rk = (r - k0) < CD_R ? (r - k0) : CD_R;
    const int want_sq = first ? 1 : 0;
    hipLaunchKernelGGL(coldots_partial_kernel, dim3((m + 255) / 256, nchunks), dim3(256), 0, st, V, n, m, ldv,
                       Y, k0, rk, ldy, rows_per, want_sq, ws);
    GPMP_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(coldots_final_kernel, dim3((m + 255) / 256), dim3(256), 0, st, ws, m, nchunks, k0, rk, r,
                       want_sq, out, ldo);
    GPMP_HIP_TRY(hipGetLastError());
    k0 += CD_R;
    first = false;
  } while (k0 < r);
  return 0;
}

extern "C" int gpmp_coldots_pair(const double* A, long lda, const double* B, long ldb, int n, int m, double* out, double* ws,
                                 gpmp_stream_t stream) {
  GPMP_ARG(A != nullptr && B != nullptr, 1, "A or B is NULL");
  GPMP_ARG(n >= 0 && m >= 0 && n <= GPMP_MAX_EXTENT && m <= GPMP_MAX_EXTENT, 5, "size outside [0, GPMP_MAX_EXTENT]");
  GPMP_ARG(lda >= m && ldb >= m, 2, "leading dimension < m");
  GPMP_ARG(out != nullptr && ws != nullptr, 7, "out or ws is NULL");
  if (m == 0) return 0;
  hipStream_t st = as_stream(stream);
  const int nchunks = gpmp_coldots_ws_rows(n) / (CD_R + 1);
  const int rows_per = (n + nchunks - 1) / nchunks > 0 ? (n + nchunks - 1) / nchunks : 1;
  ProfScope ps(PK_COLDOTS, st, 16.0 * (double)n * (double)m);
  hipLaunchKernelGGL(colpair_partial_kernel, dim3((m + 255) / 256, nchunks), dim3(256), 0, st, A, lda, B, ldb, n, m, rows_per, ws);
  GPMP_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(colpair_final_kernel, dim3((m + 255) / 256), dim3(256), 0, st, ws, m, nchunks, out);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gpmp_logdet_chol(const double* L, int n, long ldl, double* out_dev, gpmp_stream_t stream) {
  GPMP_ARG(L != nullptr, 1, "L is NULL");
  GPMP_ARG(n >= 0 && n <= GPMP_MAX_EXTENT && ldl >= n, 3, "n outside [0, GPMP_MAX_EXTENT] or ldl < n");
  GPMP_ARG(out_dev != nullptr, 4, "out is NULL");
  hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(1024), 0, as_stream(stream), L, n, ldl, out_dev);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int gpmp_tril(double* A, int n, long lda, gpmp_stream_t stream) {
  GPMP_ARG(A != nullptr, 1, "A is NULL");
  GPMP_ARG(lda >= n && n <= GPMP_MAX_EXTENT, 3, "lda < n or n above GPMP_MAX_EXTENT");
  if (n <= 0) return 0;
  return launch_tril(A, n, lda, as_stream(stream));
}
extern "C" int gpmp_symmetrize_from_lower(double* A, int n, long lda, gpmp_stream_t stream) {
  GPMP_ARG(A != nullptr, 1, "A is NULL");
  GPMP_ARG(lda >= n && n <= GPMP_MAX_EXTENT, 3, "lda < n or n above GPMP_MAX_EXTENT");
  if (n <= 0) return 0;
  return launch_symmetrize(A, n, lda, as_stream(stream));
}
