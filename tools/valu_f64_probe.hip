// Issue cost of the fp64 VALU instructions of the Gram epilogue on gfx950: cycles per wave64 instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_f64_probe.hip -o tools/valu_f64_probe.bin && tools/valu_f64_probe.bin
// Each kernel runs ITER iterations of 8 independent dependent-chains of one instruction; 4 waves per SIMD resident.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITER = 4096;

#define PROBE(NAME, STMT)                                                          \
  __global__ void __launch_bounds__(256) NAME(double* out, double seed, int n) {  \
    double a[8];                                                                   \
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x * 1e-3;               \
    int e = n;                                                                     \
    for (int it = 0; it < ITER; ++it) {                                            \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) { STMT; }                      \
    }                                                                              \
    double s = 0;                                                                  \
    for (int i = 0; i < 8; ++i) s += a[i];                                         \
    if (s == 12345.678) out[0] = s + e;                                            \
  }

PROBE(k_fma, a[i] = __builtin_fma(a[i], 0.999999, seed))
PROBE(k_mul, a[i] = a[i] * seed)
PROBE(k_add, a[i] = a[i] + seed)
PROBE(k_rsq, a[i] = __builtin_amdgcn_rsq(a[i]))
PROBE(k_rcp, a[i] = __builtin_amdgcn_rcp(a[i]))
PROBE(k_rndne, a[i] = __builtin_rint(a[i]) + 0.0)
PROBE(k_ldexp, a[i] = __builtin_amdgcn_ldexp(a[i], e))
PROBE(k_cvt, { int q = (int)a[i]; asm volatile("" : "+v"(q)); a[i] = __hiloint2double(q, __double2hiint(a[i])); })
PROBE(k_max, a[i] = __builtin_fmax(a[i], seed))

template <typename K>
void run(const char* name, K kern, double* d, int extra_per_op) {
  const int blocks = 256 * 4 * 4;   // 4 workgroups per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.25, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.25, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4;
  const double insts = waves * ITER * 8.0;
  const double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;
  printf("%-8s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz; %d helper ops per probe op not subtracted)\n",
         name, ms, simd_cycles / insts, extra_per_op);
}

int main() {
  double* d; hipMalloc(&d, 64);
  run("fma", k_fma, d, 0); run("mul", k_mul, d, 0); run("add", k_add, d, 0); run("rsq", k_rsq, d, 0); run("rcp", k_rcp, d, 0);
  run("rndne", k_rndne, d, 1); run("ldexp", k_ldexp, d, 0); run("cvt_i32", k_cvt, d, 0); run("max", k_max, d, 0);
  return 0;
}
