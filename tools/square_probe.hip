// Diagnostic: phase timeline of chol_square_kernel on one 1024 x 1024 SPD matrix (100 MHz wall clock).
#define GPMP_SQ_TRACE 1
#include "../gpmp_amd/csrc/potf2.hip"
#include <vector>
#include <cmath>
using namespace gpmp;
int run(int n, int kind) {
  const int nb = n / 128;
  std::vector<double> h((size_t)n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) h[(size_t)i * n + j] = kind == 0 ? std::exp(-0.001 * (i - j) * (i - j)) + (i == j ? 1e-3 : 0.0) : (i == j ? n + 1.0 : std::cos(0.37 * i * j + 0.11 * (i + j)));
  double *A, *dinv, *G; int* info;
  hipMalloc(&A, h.size() * 8); hipMalloc(&dinv, (size_t)nb * 128 * 128 * 8); hipMalloc(&G, (size_t)n * n * 8); hipMalloc(&info, 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemset(info, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    launch_chol_square(A, n, nb, dinv, G, info, 0, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("rep %d: %.1f us\n", rep, ms * 1e3);
  }
  // check against a host Cholesky
  std::vector<double> L(h), Ld(h.size());
  for (int j = 0; j < n; ++j) {
    double d = L[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    d = std::sqrt(d); L[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double v = L[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = v / d;
    }
  }
  hipMemcpy(Ld.data(), A, h.size() * 8, hipMemcpyDeviceToHost);
  int hinfo; hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost);
  double maxerr = 0; int bi = -1, bj = -1;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double e = std::fabs(L[(size_t)i * n + j] - Ld[(size_t)i * n + j]); if (!(e <= maxerr)) { maxerr = e; bi = i; bj = j; } }
  printf("n=%d info=%d max |L - L_host| = %.3e at (%d,%d)\n", n, hinfo, maxerr, bi, bj);
  std::vector<long long> tr(36 * 32);
  hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_sq_trace), tr.size() * 8);
  long long t0 = tr[0];
  int id = 0;
  for (int j = 0; j < nb; ++j) for (int i = j; i < nb; ++i, ++id) {
    if (!(i <= j + 1 || (i == 7))) continue;
    printf("tile (%d,%d):", i, j);
    for (int s = 0; s < 23; ++s) { long long v = tr[id * 32 + s]; if (v) printf(" [%d]%.1f", s, (v - t0) * 0.01); }
    printf("\n");
  }
  hipFree(A); hipFree(dinv); hipFree(G); hipFree(info);
  return 0;
}
int main() { run(256, 0); run(256, 1); run(1024, 1); return 0; }
