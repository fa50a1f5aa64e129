// What does a cross-stream dependency cost on the critical chain?  A chain of tiny kernels on stream A, where every
// kernel additionally (a) nothing, (b) waits for an event recorded on stream B long ago (already complete), (c) waits for
// an event that stream B records right after a kernel it launched in the same iteration, (d) hipStreamWaitValue64 on a
// word written by hipStreamWriteValue64 on stream B, (e) polls a device flag inside the kernel (agent-scope atomic) that a
// kernel on stream B sets.  Microseconds per link.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(int* p) { if (threadIdx.x == 0) p[0] += 1; }
__global__ void setflag(unsigned* f, unsigned v) { if (threadIdx.x == 0) __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void waitflag(unsigned* f, unsigned v, int* p) {
  if (threadIdx.x == 0) { unsigned spins = 0; while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < v && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(1); p[0] += 1; }
}
int main() {
  hipStream_t A, B; int lo, hi; CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CHECK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, hi)); CHECK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  int *pa, *pb; CHECK(hipMalloc(&pa, 64)); CHECK(hipMalloc(&pb, 64)); CHECK(hipMemset(pa, 0, 64)); CHECK(hipMemset(pb, 0, 64));
  unsigned* flag; CHECK(hipMalloc(&flag, 64)); CHECK(hipMemset(flag, 0, 64));
  const int N = 200;
  hipEvent_t ev[N + 1]; for (int i = 0; i <= N; ++i) CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  auto run = [&](int mode) -> double {
    hipMemset(flag, 0, 64); hipDeviceSynchronize();
    if (mode == 1) { tiny<<<1, 64, 0, B>>>(pb); hipEventRecord(ev[N], B); hipStreamSynchronize(B); }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
      if (mode == 2) { tiny<<<1, 64, 0, B>>>(pb); hipEventRecord(ev[i], B); }
      if (mode == 4) setflag<<<1, 64, 0, B>>>(flag, (unsigned)(i + 1));
      if (mode == 1) hipStreamWaitEvent(A, ev[N], 0);
      if (mode == 2) hipStreamWaitEvent(A, ev[i], 0);
      if (mode == 4) waitflag<<<1, 64, 0, A>>>(flag, (unsigned)(i + 1), pa); else tiny<<<1, 64, 0, A>>>(pa);
    }
    hipStreamSynchronize(A); hipStreamSynchronize(B);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
  };
  const char* names[5] = {"plain chain on A", "A waits for an old, complete event of B", "A waits for B's kernel of this iteration (event)", "(unused)", "A's kernel polls a flag set by B's kernel"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode : {0, 1, 2, 4}) { double us = run(mode); if (rep) printf("%-55s %7.2f us per link\n", names[mode], us); }
  // stream memory operations
  int can = 0; hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  if (can) {
    uint64_t* sig = nullptr;
    if (hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory) == hipSuccess) {
      hipMemset(sig, 0, 64); hipDeviceSynchronize();
      for (int rep = 0; rep < 2; ++rep) {
        hipMemset(sig, 0, 8); hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
          tiny<<<1, 64, 0, B>>>(pb);
          hipStreamWriteValue64(B, sig, (uint64_t)(i + 1), 0);
          hipStreamWaitValue64(A, sig, (uint64_t)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull);
          tiny<<<1, 64, 0, A>>>(pa);
        }
        hipStreamSynchronize(A); hipStreamSynchronize(B);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        if (rep) printf("%-55s %7.2f us per link\n", "A waits by hipStreamWaitValue64 on B's WriteValue64", us);
      }
    } else printf("hipExtMallocWithFlags(hipMallocSignalMemory) failed\n");
  }
  return 0;
}
