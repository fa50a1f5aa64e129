// Does ONE resident workgroup that does nothing slow the machine-filling fp64 GEMM down?  (Round 2 measured 11-13 % slower
// trailing updates with a resident diagonal-block server and dropped it; this separates residency from polling.)
// A one-workgroup kernel holds `lds_kb` of LDS until a wall-clock deadline (it always exits: the deadline is absolute) and, by mode,
//   0: only sleeps            1: also polls a device word with a RELAXED load (sc1, no invalidate) every ~1 us
//   2: polls with ACQUIRE semantics (buffer_inv sc1 per poll)
// while the trailing-update GEMM of the look-ahead Cholesky (syrk, K = 1024, lower) runs on another stream.
//   ./tools/resident_probe.bin [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../include/gpmp_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void fill(double* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) { unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; p[i] = ((double)(h & 0xFFFFFF) / 8388608.0) - 1.0; }
}

__global__ void __launch_bounds__(512) resident(int* word, long long ticks, int mode, int* out) {
  extern __shared__ double hold[];
  const long long t_end = (long long)wall_clock64() + ticks;       // 100 MHz ticks
  if (threadIdx.x == 0) hold[0] = 1.0;
  int seen = 0;
  while ((long long)wall_clock64() < t_end) {
    __builtin_amdgcn_s_sleep(127);
    if (mode == 1) seen += __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mode == 2) seen += __hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) out[0] = seen + (int)hold[0];
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 6;
  const int M = 16384, K = 1024;
  double *A, *C; CK(hipMalloc(&A, (size_t)M * K * 8)); CK(hipMalloc(&C, (size_t)M * M * 8));
  int *word, *out; CK(hipMalloc(&word, 64)); CK(hipMalloc(&out, 64)); CK(hipMemset(word, 0, 64));
  fill<<<2048, 256>>>(A, (size_t)M * K, 1); fill<<<2048, 256>>>(C, (size_t)M * M, 3);
  hipStream_t sg, sr; CK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sr, hipStreamNonBlocking));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(resident), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipDeviceSynchronize());
  auto gemm_ms = [&]() {
    gpmp_dgemm(0, 1, M, M, K, -1.0, A, K, A, K, 1.0, C, M, 1, sg);
    CK(hipEventRecord(e0, sg));
    for (int r = 0; r < reps; ++r) gpmp_dgemm(0, 1, M, M, K, -1.0, A, K, A, K, 1.0, C, M, 1, sg);
    CK(hipEventRecord(e1, sg)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
  };
  const double flops = 2.0 * (128.0 * 129 / 2) * 128 * 128 * K;
  for (int round = 0; round < 2; ++round) {
    float base = gemm_ms();
    printf("no resident workgroup              : %7.3f ms  %5.1f TFLOP/s\n", base, flops / base / 1e9);
    struct { int mode, lds_kb, threads; const char* name; } cases[] = {
      {0, 100, 512, "sleeping, 100 KB LDS, 8 waves     "}, {0, 1, 64, "sleeping, 1 KB LDS, 1 wave        "},
      {1, 100, 512, "relaxed poll / us, 100 KB, 8 waves"}, {2, 100, 512, "ACQUIRE poll / us, 100 KB, 8 waves"}};
    for (auto& c : cases) {
      // resident for 60 ms (the GEMM loop below takes reps x ~6 ms); absolute deadline => it always leaves
      hipLaunchKernelGGL(resident, dim3(1), dim3(c.threads), c.lds_kb * 1024, sr, word, 6000000LL, c.mode, out);
      float ms = gemm_ms();
      CK(hipStreamSynchronize(sr));
      printf("%s : %7.3f ms  %5.1f TFLOP/s  (%+.1f %%)\n", c.name, ms, flops / ms / 1e9, 100.0 * (ms / base - 1.0));
    }
  }
  return 0;
}
