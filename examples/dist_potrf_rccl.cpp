// A C++ / RCCL host of the 2-D block-cyclic Cholesky (BASELINE.json configs[4]) on the C ABI: one PROCESS per GPU, this
// program owns every collective (RCCL broadcasts over xGMI), libgpmp_hip.so does the local arithmetic of each block-column
// step through gpmp_dist_* (include/gpmp_hip.h).  No torch, no Python.  The reference has no counterpart
// (README.md:39-40); the schedule is the one of gpmp_amd/dist/cholesky.py without its look-ahead (plain right-looking, one
// stream), kept short on purpose: it shows WHICH call goes between WHICH collectives.
//
//   build:  hipcc -O2 -std=c++17 -Iinclude examples/dist_potrf_rccl.cpp -Lgpmp_amd -lgpmp_hip -lrccl -o examples/dist_potrf_rccl.bin
//   run  :  for r in 0 1 ... N-1:  ./dist_potrf_rccl.bin <rank> <N> <Pr> <Pc> <n> <nb> <id-file> &     (rank 0 writes the id file)
//           rank r uses GPU r.  With N = 1 (a 1 x 1 grid) it runs on a one-GPU box and checks log|K| against the single-GPU
//           factorisation of the same matrix (gpmp_potrf_lower_async).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "gpmp_hip.h"

#define HIP_OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define NCCL_OK(e) do { ncclResult_t r_ = (e); if (r_ != ncclSuccess) { fprintf(stderr, "RCCL %s at %s:%d\n", ncclGetErrorString(r_), __FILE__, __LINE__); exit(1); } } while (0)
#define GPMP_OK(e) do { int r_ = (e); if (r_ != 0) { fprintf(stderr, "gpmp %d (%s) at %s:%d\n", r_, gpmp_last_error(), __FILE__, __LINE__); exit(1); } } while (0)

static double* dmalloc(size_t elems) { double* p = nullptr; HIP_OK(hipMalloc(reinterpret_cast<void**>(&p), (elems ? elems : 1) * sizeof(double))); return p; }
static long ld16(long c) { return (c + 15) / 16 * 16; }

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: %s rank world Pr Pc n nb id-file\n", argv[0]); return 2; }
  const int rank = atoi(argv[1]), world = atoi(argv[2]), pr = atoi(argv[3]), pc = atoi(argv[4]), n = atoi(argv[5]), nb = atoi(argv[6]);
  const char* idfile = argv[7];
  if (pr * pc != world || nb % 128 != 0 || nb > 1024) { fprintf(stderr, "need Pr * Pc == world and nb a multiple of 128, <= 1024\n"); return 2; }
  const int r = rank / pc, c = rank % pc, d = 8, nblk = (n + nb - 1) / nb;
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  HIP_OK(hipSetDevice(rank % ndev));
  hipStream_t st;
  HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

  // ---- communicators: world, one per process row, one per process column (the host's business, not the library's)
  ncclUniqueId id;
  if (rank == 0) {
    NCCL_OK(ncclGetUniqueId(&id));
    FILE* f = fopen(idfile, "wb");
    if (!f || fwrite(&id, sizeof(id), 1, f) != 1) { fprintf(stderr, "cannot write %s\n", idfile); return 1; }
    fclose(f);
  } else {
    for (int tries = 0;; ++tries) {
      FILE* f = fopen(idfile, "rb");
      if (f) { const size_t got = fread(&id, sizeof(id), 1, f); fclose(f); if (got == 1) break; }
      if (tries > 600) { fprintf(stderr, "no id file\n"); return 1; }
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  }
  ncclComm_t comm_world, comm_row, comm_col;
  NCCL_OK(ncclCommInitRank(&comm_world, world, id, rank));
  NCCL_OK(ncclCommSplit(comm_world, r, c, &comm_row, nullptr));      // rank inside a row communicator = process column
  NCCL_OK(ncclCommSplit(comm_world, pr + c, r, &comm_col, nullptr)); // rank inside a column communicator = process row

  // ---- synthetic inputs of SURVEY 8(d) (replicated) and the LOCAL matrix K(x[rows owned], x[cols owned]) + 1e-4 I
  std::vector<double> x((size_t)n * d);
  unsigned long long s = 1234;
  for (auto& v : x) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; v = (double)(s >> 11) / 9007199254740992.0; }
  double theta[9];
  theta[0] = 0.0;
  for (int j = 0; j < d; ++j) theta[1 + j] = -std::log(0.5 * (1.0 + (double)j / d));
  long lrows = 0, lcols = 0;
  GPMP_OK(gpmp_dist_local_shape(n, nb, pr, pc, r, c, &lrows, &lcols));
  auto gather = [&](int first, int step) {          // the points of the owned block rows (or columns), in local order
    std::vector<double> g;
    for (int I = first; I < nblk; I += step)
      for (int i = I * nb; i < n && i < (I + 1) * nb; ++i) g.insert(g.end(), x.begin() + (size_t)i * d, x.begin() + (size_t)(i + 1) * d);
    return g;
  };
  const std::vector<double> xr = gather(r, pr), xc = gather(c, pc);
  double *xr_d = dmalloc(xr.size()), *xc_d = dmalloc(xc.size());
  HIP_OK(hipMemcpy(xr_d, xr.data(), xr.size() * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(xc_d, xc.data(), xc.size() * 8, hipMemcpyHostToDevice));
  const long lda = ld16(lcols);
  double* A = dmalloc((size_t)lrows * lda);
  const double noise = 1e-4;
  if (lrows && lcols) GPMP_OK(gpmp_matern_gram(xr_d, xc_d, (int)lrows, (int)lcols, d, 2, theta, 0, 0.0, 0, A, lda, st));   // cross-covariance: no diagonal term
  for (int I = r; I < nblk; I += pr)                // the diagonal blocks this rank owns get theirs from the ii path
    if (I % pc == c) {
      const int bs = (I + 1) * nb <= n ? nb : n - I * nb;
      const long li = I / pr, lj = I / pc;
      GPMP_OK(gpmp_matern_gram(xr_d + li * nb * d, nullptr, bs, bs, d, 2, theta, 0, noise, 0, A + li * nb * lda + lj * nb, lda, st));
    }

  // ---- buffers of a step
  const size_t msg_elems = gpmp_dist_diag_msg_elems(nb);
  double *msg = dmalloc(msg_elems), *panel = dmalloc((size_t)lrows * nb), *colop = dmalloc((size_t)lcols * nb), *piece = dmalloc((size_t)lcols * nb),
         *ws = dmalloc(gpmp_dist_panel_ws_elems(nb)), *logdet_part = dmalloc(2);
  HIP_OK(hipMemsetAsync(logdet_part, 0, 16, st));
  std::vector<double> info_host(nblk, 0.0);
  double logdet_local = 0.0;
  HIP_OK(hipStreamSynchronize(st));
  const auto t0 = std::chrono::steady_clock::now();

  for (int k = 0; k < nblk; ++k) {
    const int rd = k % pr, cd = k % pc, bk = (k + 1) * nb <= n ? nb : n - k * nb;
    long prow = 0, crow = 0, r0 = 0, c0 = 0;
    GPMP_OK(gpmp_dist_step_shape(n, nb, pr, pc, r, c, k, &prow, &crow, &r0, &c0));
    const size_t mk = gpmp_dist_diag_msg_elems(bk);
    if (c == cd) {
      // 1. the owner factors the diagonal block; 2. (L_kk | block inverses | info) down the process column
      if (r == rd) GPMP_OK(gpmp_dist_diag_factor(A + (long)(k / pr) * nb * lda + (long)(k / pc) * nb, bk, lda, msg, st));
      if (pr > 1) NCCL_OK(ncclBroadcast(msg, msg, mk, ncclDouble, rd, comm_col, st));
      HIP_OK(hipMemcpyAsync(&info_host[k], msg + mk - 1, 8, hipMemcpyDeviceToHost, st));
      if (r == rd) {      // log-det contribution of the block, accumulated on its owner
        GPMP_OK(gpmp_logdet_chol(msg, bk, ld16(bk), logdet_part + 1, st));
        double v = 0.0;
        HIP_OK(hipMemcpyAsync(&v, logdet_part + 1, 8, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        logdet_local += v;
      }
      // 3. panel solve on the owning process column
      GPMP_OK(gpmp_dist_panel_solve(msg, bk, A + r0 * lda + (long)(k / pc) * nb, (int)prow, lda, panel, nb, ws, st));
    }
    // 4. the panel along every process row
    if (pc > 1 && prow > 0) NCCL_OK(ncclBroadcast(panel, panel, (size_t)prow * nb, ncclDouble, cd, comm_row, st));
    // 5. column operand: per process row rp the holder packs, the process column broadcasts, everybody unpacks
    for (int rp = 0; rp < pr; ++rp) {
      const long rows = gpmp_dist_exchange_rows(n, nb, pr, pc, rp, c, k);
      if (rows <= 0) continue;
      if (r == rp) GPMP_OK(gpmp_dist_exchange_pack(panel, nb, piece, nb, n, nb, pr, pc, r, c, k, bk, st));
      if (pr > 1) NCCL_OK(ncclBroadcast(piece, piece, (size_t)rows * nb, ncclDouble, rp, comm_col, st));
      GPMP_OK(gpmp_dist_exchange_unpack(piece, nb, colop, nb, n, nb, pr, pc, rp, c, k, bk, st));
    }
    // 6. trailing update of the local blocks I >= J > k
    if (lrows && lcols) GPMP_OK(gpmp_dist_trailing_update(A, lda, n, nb, pr, pc, r, c, k, panel, nb, colop, nb, 0, -1, -1, st));
  }
  HIP_OK(hipStreamSynchronize(st));
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // ---- agree on info and log|K| (two scalars, one all-reduce each)
  double first_bad = 1e300;
  for (int k = 0; k < nblk; ++k)
    if (info_host[k] != 0.0) { first_bad = (double)k * nb + info_host[k]; break; }
  double* red = dmalloc(2);
  const double mine[2] = {-first_bad, logdet_local};      // max(-x) = -min(x)
  HIP_OK(hipMemcpyAsync(red, mine, 16, hipMemcpyHostToDevice, st));
  NCCL_OK(ncclAllReduce(red, red, 1, ncclDouble, ncclMax, comm_world, st));
  NCCL_OK(ncclAllReduce(red + 1, red + 1, 1, ncclDouble, ncclSum, comm_world, st));
  double out[2];
  HIP_OK(hipMemcpyAsync(out, red, 16, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  const long info = -out[0] > 1e299 ? 0 : (long)(-out[0]);
  if (rank == 0) {
    printf("dist_potrf_rccl: n=%d nb=%d grid=%dx%d info=%ld logdet=%.12f seconds=%.4f tflops=%.2f\n", n, nb, pr, pc, info, out[1], secs,
           (double)n * n * n / 3.0 / secs / 1e12);
    if (n <= 16384) {       // single-GPU reference on rank 0: the same matrix through gpmp_potrf_lower_async
      double* xd = dmalloc(x.size());
      HIP_OK(hipMemcpy(xd, x.data(), x.size() * 8, hipMemcpyHostToDevice));
      const long ldk = ld16(n);
      double *K = dmalloc((size_t)n * ldk), *dinv = dmalloc(gpmp_dinv_elems(n)), *ld = dmalloc(1);
      int* inf = nullptr;
      HIP_OK(hipMalloc(reinterpret_cast<void**>(&inf), 4));
      GPMP_OK(gpmp_matern_gram(xd, nullptr, n, n, d, 2, theta, 0, noise, 1, K, ldk, st));
      GPMP_OK(gpmp_potrf_lower_async(K, n, ldk, dinv, inf, st));
      GPMP_OK(gpmp_logdet_chol(K, n, ldk, ld, st));
      double ref = 0.0;
      HIP_OK(hipMemcpyAsync(&ref, ld, 8, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      printf("single-GPU logdet=%.12f rel_diff=%.3e\n", ref, std::fabs(out[1] - ref) / std::fabs(ref));
    }
  }
  NCCL_OK(ncclCommDestroy(comm_row));
  NCCL_OK(ncclCommDestroy(comm_col));
  NCCL_OK(ncclCommDestroy(comm_world));
  return info == 0 ? 0 : 3;
}
