// Diagnostic: phase timeline of potf2_inv_kernel on one 128 x 128 SPD block (100 MHz wall clock).
#define GPMP_POTF2_TRACE 1
#include "../gpmp_amd/csrc/potf2.hip"
#include <vector>
#include <cmath>
#include <time.h>
using namespace gpmp;
int main(int argc, char** argv) {
  const bool busy = argc > 1 && argv[1][0] == 'b';   // any argument: a machine-filling trailing-update GEMM runs on another stream meanwhile
  const int n = 128;
  std::vector<double> h((size_t)n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) h[(size_t)i * n + j] = (i == j ? n + 1.0 : std::cos(0.37 * i * j + 0.11 * (i + j)));
  double *A, *dinv; int* info;
  if (hipMalloc(&A, h.size() * 8) != hipSuccess || hipMalloc(&dinv, (size_t)n * n * 8) != hipSuccess || hipMalloc(&info, 4) != hipSuccess) return 1;
  hipStream_t sg = nullptr, sp = nullptr;
  double *GA = nullptr, *GC = nullptr;
  const int gn = 16384, gk = 1024;
  if (busy) {
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    (void)hipStreamCreateWithFlags(&sg, hipStreamNonBlocking); (void)hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi);
    if (hipMalloc(&GA, (size_t)gn * gk * 8) != hipSuccess || hipMalloc(&GC, (size_t)gn * gn * 8) != hipSuccess) return 1;
    (void)hipMemset(GA, 0, (size_t)gn * gk * 8); (void)hipMemset(GC, 0, (size_t)gn * gn * 8);
  }
  for (int rep = 0; rep < 3; ++rep) {
    if (busy) {
      GemmOpts o; o.lower_only = 1;
      for (int q = 0; q < 2; ++q) launch_gemm(true, true, gn, gn, gk, -1.0, GA, gk, GA, gk, 1.0, GC, gn, o, sg);
      struct timespec ts = {0, 3000000}; nanosleep(&ts, nullptr);     // the GEMM is in full swing
    }
    (void)hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice); (void)hipMemset(info, 0, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, sp);
    launch_potf2_inv(A, n, n, dinv, info, 0, sp);
    (void)hipEventRecord(e1, sp); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); printf("rep %d: %.1f us\n", rep, ms * 1e3);
  }
  long long tr[192];
  (void)hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_potf2_trace), sizeof(tr));
  auto us = [&](int s) { return (tr[s] - tr[0]) * 0.01; };
  printf("load done %.1f | chain done %.1f | end (wave 0) %.1f\n", us(1), us(2), us(6));
  for (int j = 0; j < 8; ++j) printf("  j=%d: A1 done %.1f  A2 done %.1f  A3(first block) done %.1f\n", j, us(8 + 3 * j), j < 7 ? us(9 + 3 * j) : 0.0, j < 7 ? us(10 + 3 * j) : 0.0);
  printf("shadow: zero fill done (wave 7) %.1f\n", us(64));
  for (int j = 1; j < 8; ++j) printf("  j=%d: wave 7: a3 %.1f invert %.1f crow %.1f | wave 4: a3 %.1f stores %.1f crow %.1f\n", j, us(64 + 4 * j), us(65 + 4 * j), us(66 + 4 * j), us(96 + 4 * j), us(97 + 4 * j), us(98 + 4 * j));
  return 0;
}
