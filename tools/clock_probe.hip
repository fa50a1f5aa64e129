// What does s_memtime count on gfx950, and at what core clock does a ONE-WAVE kernel (the diagonal-block chain) run?
// A chain of dependent v_fma_f64 bracketed by s_memtime (shader clock counter) AND wall_clock64 (100 MHz constant);
// once on an otherwise idle GPU, once while a second stream keeps every compute unit busy with fp64 FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void chain(double* out, long long* t, double seed) {
  double a = seed + threadIdx.x * 1e-3;
  const double b = 1.0000001, c = 1e-9;
  long long m0, m1;
  const long long w0 = wall_clock64();
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(m0), "+v"(a) : : "memory");
#pragma unroll 64
  for (int i = 0; i < N; ++i) a = fma(a, b, c);
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(m1), "+v"(a) : : "memory");
  const long long w1 = wall_clock64();
  if (threadIdx.x == 0) { t[0] = m1 - m0; t[1] = w1 - w0; }
  out[threadIdx.x] = a;
}
__global__ void busy(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9, d = a + 1.0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) { a = fma(a, b, c); d = fma(d, b, c); }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + d;
}
int main() {
  double* out; long long* t; double* big;
  hipMalloc(&out, 64 * 8); hipMalloc(&t, 16); hipMalloc(&big, 1024 * 256 * 8);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  long long h[2];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, out, t, 1.0); hipStreamSynchronize(s1);
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("idle GPU : %d dependent v_fma_f64: s_memtime delta %lld, wall clock (100 MHz) delta %lld = %.2f us -> %.2f memtime ticks per fma, %.2f ns per fma, memtime rate %.1f MHz\n",
           N, h[0], h[1], h[1] * 0.01, (double)h[0] / N, h[1] * 10.0 / N, h[0] / (h[1] * 0.01));
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, s2, big, 20000);      // ~ms of fp64 FMAs on every CU
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, out, t, 1.0); hipStreamSynchronize(s1);
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    hipStreamSynchronize(s2);
    printf("busy GPU : %d dependent v_fma_f64: s_memtime delta %lld, wall clock (100 MHz) delta %lld = %.2f us -> %.2f memtime ticks per fma, %.2f ns per fma, memtime rate %.1f MHz\n",
           N, h[0], h[1], h[1] * 0.01, (double)h[0] / N, h[1] * 10.0 / N, h[0] / (h[1] * 0.01));
  }
  // back-to-back one-wave kernels (the chain-bound regime): does the clock differ after a burst of them?
  for (int rep = 0; rep < 200; ++rep) hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, out, t, 1.0);
  hipStreamSynchronize(s1);
  hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
  printf("after 200 back-to-back one-wave kernels: %.2f ns per fma, memtime rate %.1f MHz\n", h[1] * 10.0 / N, h[0] / (h[1] * 0.01));
  return 0;
}
