// Dependent-instruction latencies of ONE wave on gfx950 (the pivot chain of the 16 x 16 diagonal factorisation is a
// single wave issuing dependent fp64 ops): cycles per dependent op for v_fma_f64, v_rsq_f64, v_readlane -> v_fma,
// v_permlane16/32_swap, and v_cmp + uniform branch.  s_memtime around N-long chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
// s_memtime ordered after `dep` has been computed (input operand) and before anything that uses the returned `dep`
__device__ __forceinline__ long long tick(double& dep) {
  long long t;
  asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) : : "memory");
  return t;
}
__device__ __forceinline__ double bcast_lane(double x, int src) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, src); hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}
__global__ void probe(double* out, long long* cyc, double seed) {
  double a = seed + threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9;
  long long t0, t1;
  // (0) dependent fma
  t0 = tick(a);
#pragma unroll
  for (int i = 0; i < N; ++i) a = fma(a, b, c);
  t1 = tick(a); cyc[0] = t1 - t0;
  // (1) dependent rsq
  double r = a * a + 1.0;
  t0 = tick(r);
#pragma unroll
  for (int i = 0; i < N; ++i) r = __builtin_amdgcn_rsq(r) + 1.0;   // rsq + add per step
  t1 = tick(r); cyc[1] = t1 - t0;
  // (2) readlane -> fma dependent
  double q = a;
  t0 = tick(q);
#pragma unroll
  for (int i = 0; i < N; ++i) q = fma(q, bcast_lane(q, i & 15), c);
  t1 = tick(q); cyc[2] = t1 - t0;
  // (3) permlane16 + permlane32 swap of a double, dependent
  double p = q;
  t0 = tick(p);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    int lo = __double2loint(p), hi = __double2hiint(p);
    auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    auto l32 = __builtin_amdgcn_permlane32_swap(l16[0], l16[0], false, false);
    auto h32 = __builtin_amdgcn_permlane32_swap(h16[0], h16[0], false, false);
    p = __hiloint2double(h32[0], l32[0]) + c;
  }
  t1 = tick(p); cyc[3] = t1 - t0;
  // (4) readlane -> compare -> uniform branch -> fma
  double u = p;
  t0 = tick(u);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double d = bcast_lane(u, i & 15);
    if (!(d > 0.0)) d = 1.0;
    u = fma(u, d, c);
  }
  t1 = tick(u); cyc[4] = t1 - t0;
  // (5) dependent mfma f64 16x16x4 (accumulator chain)
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc = {u, u, u, u};
  t0 = tick(u);
  acc[0] = u;
#pragma unroll
  for (int i = 0; i < 64; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
  double a0 = acc[0];
  t1 = tick(a0); cyc[5] = t1 - t0;
  acc[0] = a0;
  // (6) fma feeding mfma feeding fma (VALU <-> MFMA turnaround)
  double w = acc[0];
  t0 = tick(w);
#pragma unroll
  for (int i = 0; i < 64; ++i) { d4 z = {w, w, w, w}; z = __builtin_amdgcn_mfma_f64_16x16x4f64(w, 1e-3, z, 0, 0, 0); w = z[0] * 0.5; }
  t1 = tick(w); cyc[6] = t1 - t0;
  // (7) LDS write -> read round trip, dependent
  __shared__ double sh[64];
  double v = w;
  t0 = tick(v);
#pragma unroll
  for (int i = 0; i < 64; ++i) { sh[threadIdx.x] = v; v = sh[(threadIdx.x + 1) & 63] + c; }
  t1 = tick(v); cyc[7] = t1 - t0;
  out[threadIdx.x] = a + r + q + p + u + acc[1] + w + v;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8 * 8);
  for (int rep = 0; rep < 2; ++rep) probe<<<1, 64>>>(out, cyc, 1.0);
  long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[8] = {"v_fma_f64 dependent", "v_rsq_f64 + v_add_f64", "2 v_readlane + v_fma_f64", "permlane16+32 swap of a double + add",
                          "2 readlane + cmp + branch + fma", "mfma_f64_16x16x4 accumulator chain", "mul -> mfma -> mul turnaround", "LDS write -> read -> add"};
  const int cnt[8] = {N, N, N, N, N, 64, 64, 64};
  // s_memtime counts shader-clock cycles (2.4 GHz here: tools/clock_probe.hip compares it with the 100 MHz wall clock)
  for (int k = 0; k < 8; ++k) printf("%-40s %8lld cycles / %3d = %7.3f cycles per dependent step (%.1f ns at 2.4 GHz)\n", names[k], h[k], cnt[k], (double)h[k] / cnt[k], (double)h[k] / cnt[k] / 2.4);
  return 0;
}
