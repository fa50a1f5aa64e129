// Can a small-footprint kernel start NEXT TO the two resident workgroups of the trailing-update GEMM on every CU
// (no waiting for a workgroup slot)?  The LDS-direct GEMM holds 2 x 64 KB of LDS and 2 x 232 VGPRs per SIMD lane:
// 32 KB of LDS and 48 VGPRs stay free.  Chain of dependent 1-workgroup kernels on a high-priority stream while
// gpmp_dgemm (NT, K = 1024, lower tiles of a 32768 x 32768 matrix) runs on another stream.
//   hipcc --offload-arch=gfx950 -O3 tools/coresident_probe.hip -Iinclude -Lgpmp_amd -lgpmp_hip -Wl,-rpath,$PWD/gpmp_amd -o tools/coresident_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include "gpmp_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// <= 48 VGPRs: 10 waves per SIMD would fit
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(10, 10))) tiny_lean(double* buf) {
  extern __shared__ double sm[];
  sm[threadIdx.x] = buf[threadIdx.x];
  __syncthreads();
  double v = sm[(threadIdx.x + 1) & 255];
  for (int i = 0; i < 200; ++i) v = v * 1.0000001 + 1e-9;
  buf[threadIdx.x] = v;
}
// ~128 VGPRs per wave
__global__ void __launch_bounds__(256) tiny_fat(double* buf) {
  extern __shared__ double sm[];
  sm[threadIdx.x] = buf[threadIdx.x];
  __syncthreads();
  double v[60];
#pragma unroll
  for (int i = 0; i < 60; ++i) v[i] = sm[(threadIdx.x + i) & 255];
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int i = 0; i < 60; ++i) v[i] = v[i] * 1.0000001 + v[(i + 1) % 60];
  double s = 0;
#pragma unroll
  for (int i = 0; i < 60; ++i) s += v[i];
  buf[threadIdx.x] = s;
}

template <typename K>
static double chain(K kern, hipStream_t hs, double* buf, int n, size_t lds) {
  CK(hipStreamSynchronize(hs));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, hs, buf);
  CK(hipStreamSynchronize(hs));
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
  hipStream_t ms, hs;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&ms, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, hi));
  const int n = 32768, k = 1024;
  double *A, *C, *buf;
  CK(hipMalloc(&A, (size_t)n * k * 8)); CK(hipMalloc(&C, (size_t)n * n * 8)); CK(hipMalloc(&buf, 4096 * 8));
  CK(hipMemset(A, 0, (size_t)n * k * 8)); CK(hipMemset(C, 0, (size_t)n * n * 8)); CK(hipMemset(buf, 0, 4096 * 8));
  CK(hipFuncSetAttribute((const void*)tiny_lean, hipFuncAttributeMaxDynamicSharedMemorySize, 92160));
  CK(hipFuncSetAttribute((const void*)tiny_fat, hipFuncAttributeMaxDynamicSharedMemorySize, 92160));
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)tiny_lean)); printf("tiny_lean: %d VGPRs\n", fa.numRegs);
  CK(hipFuncGetAttributes(&fa, (const void*)tiny_fat)); printf("tiny_fat : %d VGPRs\n", fa.numRegs);
  // warm the GEMM once
  if (gpmp_dgemm(0, 1, n, n, k, -1.0, A, k, A, k, 1.0, C, n, 1, ms)) { fprintf(stderr, "gemm: %s\n", gpmp_last_error()); return 1; }
  CK(hipStreamSynchronize(ms));
  struct V { const char* name; int fat; size_t lds; } vs[] = {
      {"lean (<=48 VGPR),  8 KB LDS", 0, 8192}, {"lean (<=48 VGPR), 31 KB LDS", 0, 31744}, {"lean (<=48 VGPR), 40 KB LDS", 0, 40960},
      {"lean (<=48 VGPR), 89 KB LDS", 0, 91136}, {"fat (~128 VGPR),   8 KB LDS", 1, 8192}, {"fat (~128 VGPR),  89 KB LDS", 1, 91136}};
  for (auto& v : vs) {
    double alone = v.fat ? chain(tiny_fat, hs, buf, 100, v.lds) : chain(tiny_lean, hs, buf, 100, v.lds);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, ms));
    gpmp_dgemm(0, 1, n, n, k, -1.0, A, k, A, k, 1.0, C, n, 1, ms);
    CK(hipEventRecord(e1, ms));
    auto t = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count() < 2.0) {}
    double per = v.fat ? chain(tiny_fat, hs, buf, 20, v.lds) : chain(tiny_lean, hs, buf, 20, v.lds);
    const bool still = hipEventQuery(e1) == hipErrorNotReady;
    CK(hipStreamSynchronize(ms));
    float gm; CK(hipEventElapsedTime(&gm, e0, e1));
    printf("%-30s: %6.1f us per kernel alone, %7.1f us under the GEMM (GEMM still running after the chain: %d; GEMM %.2f ms)\n",
           v.name, alone, per, (int)still, gm);
  }
  return 0;
}
