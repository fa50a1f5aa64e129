// Micro-benchmark of libgpmp_hip's fp64 MFMA GEMM on shapes the blocked algorithms use.
// Diagnostic tool (not product code):  ./tools/gemm_bench.bin [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/gpmp_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void fill(double* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) { unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; p[i] = ((double)(h & 0xFFFFFF) / 8388608.0) - 1.0; }
}

static void run(const char* name, int ta, int tb, int M, int N, int K, int lower, double beta, int reps) {
  size_t ea = (size_t)(ta ? K : M) * (ta ? M : K), eb = (size_t)(tb ? N : K) * (tb ? K : N), ec = (size_t)M * N;
  double *A, *B, *C; CK(hipMalloc(&A, ea * 8)); CK(hipMalloc(&B, eb * 8)); CK(hipMalloc(&C, ec * 8));
  fill<<<2048, 256>>>(A, ea, 1); fill<<<2048, 256>>>(B, eb, 2); fill<<<2048, 256>>>(C, ec, 3);
  long lda = ta ? M : K, ldb = tb ? K : N, ldc = N;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gpmp_dgemm(ta, tb, M, N, K, -1.0, A, lda, B, ldb, beta, C, ldc, lower, nullptr); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) gpmp_dgemm(ta, tb, M, N, K, -1.0, A, lda, B, ldb, beta, C, ldc, lower, nullptr);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  double tm = (M + 127) / 128, tn = (N + 127) / 128;
  double tiles = lower ? (tn * (tn + 1) / 2 + (tm - tn) * tn) : tm * tn;
  double flops = 2.0 * tiles * 128 * 128 * K;
  printf("%-34s ta=%d tb=%d M=%6d N=%6d K=%5d lower=%d beta=%g : %8.3f ms  %6.2f TFLOP/s (%.1f%% of 78.6)\n", name, ta, tb, M, N, K, lower, beta, ms,
         flops / ms / 1e9, flops / ms / 1e9 / 78.6 * 100);
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
}

int main(int argc, char** argv) {
  int reps = argc > 1 ? atoi(argv[1]) : 5;
  int which = argc > 2 ? atoi(argv[2]) : -1;
  if (getenv("BUSY")) gpmp_hint_machine_busy(atoi(getenv("BUSY")));
  if (which < 0 || which == 0) run("syrk trailing (potrf)", 0, 1, 32256, 32256, 512, 1, 1.0, reps);
  if (which < 0 || which == 1) run("syrk trailing mid", 0, 1, 16384, 16384, 512, 1, 1.0, reps);
  if (which < 0 || which == 2) run("trsm update (predict)", 0, 0, 16384, 49920, 512, 0, 1.0, reps);
  if (which < 0 || which == 3) run("trsm update K=128", 0, 0, 16384, 49920, 128, 0, 1.0, reps);
  if (which < 0 || which == 4) run("square NN K=4096", 0, 0, 8192, 8192, 4096, 0, 0.0, reps);
  if (which < 0 || which == 5) run("square NT K=4096", 0, 1, 8192, 8192, 4096, 0, 0.0, reps);
  if (which < 0 || which == 6) run("square TN K=4096", 1, 0, 8192, 8192, 4096, 0, 0.0, reps);
  if (which < 0 || which == 7) run("small NT L2-resident K=4096", 0, 1, 2048, 2048, 4096, 0, 0.0, reps);
  if (which < 0 || which == 8) run("panel scale N=128 K=128", 0, 1, 32640, 128, 128, 0, 0.0, reps);
  if (which == 30) { for (int M : {128, 1024, 4096, 32640}) { run("panel scale NT N=128 K=128 b0", 0, 1, M, 128, 128, 0, 0.0, reps); run("rank-128 update NT N=384 b1", 0, 1, M, 384, 128, 0, 1.0, reps); run("NN K=128 N=4096 b1", 0, 0, M, 4096, 128, 0, 1.0, reps); } }
  if (which == 40) { for (int M : {3584, 3072, 2560, 2048, 1792, 1536, 1024, 512}) run("syrk trailing, chain-bound tail K=256", 0, 1, M, M, 256, 1, 1.0, reps); }
  // the triangular solve's updates of the headline step (B2 -= L21 X1: M = K = R rows, N = 50000) and the Cholesky's trailing
  // updates at rank 1024 / 2048 -- the shapes whose L2-side traffic DESIGN section 4 discusses (run under rocprofv3 --pmc FETCH_SIZE)
  if (which == 50) { for (int R : {16384, 8192, 4096, 2048}) run("solve update NN M=K=R N=50000", 0, 0, R, 50000, R, 0, 1.0, reps); }
  if (which == 53) { for (int R : {512, 1024, 2048, 4096, 8192}) run("solve update NN M=K=R N=50000", 0, 0, R, 50000, R, 0, 1.0, reps); }
  if (which == 52) run("solve update NN M=K=8192 N=50000", 0, 0, 8192, 50000, 8192, 0, 1.0, reps);
  if (which == 51) { for (int K : {1024, 2048}) run("potrf trailing NT lower M=N=24576", 0, 1, 24576, 24576, K, 1, 1.0, reps); }
  if (which == 20) { for (int K : {128, 256, 512, 1024, 2048, 4096}) { run("NN beta=1 K sweep", 0, 0, 16384, 16384, K, 0, 1.0, reps); run("NN beta=0 K sweep", 0, 0, 16384, 16384, K, 0, 0.0, reps); } }
  return 0;
}
