// Probe 2 (gfx950): issue rate of v_mfma_f64_16x16x4_f64 versus waves per SIMD, with the
// loop written in inline asm so no compiler AGPR<->VGPR shuffling pollutes the count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, bool AG>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* stamps, int iters, double av, double bv) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = av + threadIdx.x * 1e-3, b = bv - threadIdx.x * 1e-3;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (AG) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { int w = blockIdx.x * 4 + (threadIdx.x >> 6); stamps[2 * w] = c1 - c0; stamps[2 * w + 1] = r1 - r0; }
}

template <int NACC, bool AG>
static void run(int blocks, int iters) {
  double* out; CK(hipMalloc(&out, sizeof(double) * blocks * 256));
  unsigned long long* st; CK(hipMalloc(&st, 16 * blocks * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<NACC, AG><<<blocks, 256>>>(out, st, iters, 1.0, 1.0); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); k<NACC, AG><<<blocks, 256>>>(out, st, iters, 1.0, 1.0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * blocks * 4);
  CK(hipMemcpy(h.data(), st, 16 * blocks * 4, hipMemcpyDeviceToHost));
  double cs = 0, rs = 0; for (int w = 0; w < blocks * 4; ++w) { cs += h[2 * w]; rs += h[2 * w + 1]; }
  cs /= blocks * 4; rs /= blocks * 4;
  double n = (double)iters * NACC;
  printf("NACC=%d %s blocks=%4d: %.1f cyc/MFMA/wave  clock %.3f GHz  wall %.3f ms  %.2f TFLOP/s\n", NACC, AG ? "agpr" : "vgpr", blocks,
         cs / n, cs / (rs * 10.0), ms, (double)blocks * 4 * n * 2048 / ms / 1e9);
  CK(hipFree(out)); CK(hipFree(st));
}

int main() {
  for (int m = 1; m <= 6; ++m) run<8, false>(256 * m, 4000);
  for (int m = 1; m <= 4; ++m) run<8, true>(256 * m, 4000);
  for (int m = 1; m <= 4; ++m) run<4, false>(256 * m, 8000);
  for (int m = 1; m <= 4; ++m) run<2, false>(256 * m, 16000);
  for (int m = 1; m <= 3; ++m) run<16, false>(256 * m, 2000);
  return 0;
}
