// Diagnostic: latency of a chain of tiny dependent kernels on a high-priority stream while a heavy kernel
// fills the machine on another stream.  Separates "waiting for a workgroup slot" from "kernel-boundary cost".
//   ./chain_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) heavy(double* out, int iters, int do_write, int stagger) {
  extern __shared__ double sm[];
  if (stagger && blockIdx.x < 512 && (blockIdx.x & 1) == 0) {
    const int steps = (int)(((blockIdx.x >> 1) * 40503u) & 255u) * iters >> 8;
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(127);
  }
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it)          // 64 MFMAs per iteration ~ one k-tile of the GEMM
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sm[threadIdx.x] = s;
  if (do_write) {
    double* o = out + (size_t)blockIdx.x * 16384;
    for (int i = threadIdx.x; i < 16384; i += 256) o[i] = s + i;   // 128 KB per workgroup, like a C tile
  } else if (s == 12345.678) out[0] = s;
}

__global__ void __launch_bounds__(512) tiny(double* buf, int lds_touch) {
  extern __shared__ double sm[];
  sm[threadIdx.x] = buf[threadIdx.x];
  __syncthreads();
  double v = sm[(threadIdx.x + 1) & 511];
  for (int i = 0; i < 200; ++i) v = v * 1.0000001 + 1e-9;
  buf[threadIdx.x] = v;
}

static double chain(hipStream_t hs, double* buf, int n, size_t lds) {
  CK(hipStreamSynchronize(hs));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(512), lds, hs, buf, 0);
  CK(hipStreamSynchronize(hs));
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
  hipStream_t ms, hs;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&ms, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, hi));
  double *out, *buf; CK(hipMalloc(&out, (size_t)20000 * 16384 * 8)); CK(hipMalloc(&buf, 4096 * 8));
  CK(hipMemset(buf, 0, 4096 * 8));
  CK(hipFuncSetAttribute((const void*)heavy, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)tiny, hipFuncAttributeMaxDynamicSharedMemorySize, 92160));
  const size_t tl = 91136;   // like potf2_inv_kernel
  printf("chain alone: %.1f us per kernel\n", chain(hs, buf, 200, tl));
  printf("chain alone: %.1f us per kernel\n", chain(hs, buf, 200, tl));
  struct Cfg { const char* name; size_t lds; int write; int stagger; int iters; };
  Cfg cfgs[] = {{"heavy 64KB LDS, no writes", 65536, 0, 0, 64}, {"heavy 64KB LDS, writes 128KB/wg", 65536, 1, 0, 64},
                {"heavy 64KB LDS, writes, staggered", 65536, 1, 1, 64}, {"heavy 32KB LDS (slot always free), writes", 32768, 1, 0, 64},
                {"heavy 64KB, no writes, short tiles (16 it)", 65536, 0, 0, 16}, {"heavy 64KB, writes, short tiles (16 it)", 65536, 1, 0, 16}};
  for (auto& c : cfgs) {
    const int tiles = 512 * 40 * 64 / c.iters;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, ms));
    hipLaunchKernelGGL(heavy, dim3(tiles > 20000 ? 20000 : tiles), dim3(256), c.lds, ms, out, c.iters, c.write, c.stagger);
    CK(hipEventRecord(e1, ms));
    // wait until the heavy kernel is certainly running
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count() < 1.0) {}
    double per = chain(hs, buf, 20, tl);
    bool still = hipEventQuery(e1) == hipErrorNotReady;
    CK(hipStreamSynchronize(ms));
    float hm; CK(hipEventElapsedTime(&hm, e0, e1));
    printf("%-45s: chain %.1f us per kernel (heavy still running afterwards: %d, heavy total %.2f ms)\n", c.name, per, (int)still, hm);
  }
  return 0;
}
