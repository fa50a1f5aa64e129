// Probe for gfx950: (1) lane/register layout of v_mfma_f64_16x16x4_f64,
// (2) its sustained issue rate (-> fp64 matrix peak actually reachable),
// (3) HBM copy bandwidth. Diagnostic tool only; not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double* A, const double* B, double* C) {
  // A: 16x4 row-major, B: 4x16 row-major. Hypothesis: lane l holds A[l&15][l>>4], B[l>>4][l&15].
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_kernel(double* out, int iters, double seed) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void copy_kernel(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}

__global__ void write_kernel(double2* __restrict__ dst, size_t n, double v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  double2 x = {v, v + 1};
  for (; i < n; i += stride) dst[i] = x;
}

// fp64 VALU transcendental cost probe: exp + sqrt per element, compute-only.
__global__ void expsqrt_kernel(double* out, int iters, double seed) {
  double x = seed + threadIdx.x * 1e-6, s = 0;
  for (int it = 0; it < iters; ++it) {
    double h = sqrt(x + it * 1e-3);
    s += exp(-h) * (1.0 + h + h * h * (1.0 / 3.0));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// Cycle-stamped MFMA loop: per-wave shader cycles (s_memtime) and 100 MHz realtime ticks.
template <int NACC>
__global__ void __launch_bounds__(256) cyc_kernel(double* out, unsigned long long* stamps, int iters, double av, double bv) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = av == 0 ? 0.0 : av + threadIdx.x * 1e-3, b = bv == 0 ? 0.0 : bv - threadIdx.x * 1e-3;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = c1 - c0; stamps[2 * w + 1] = r1 - r0;
  }
}

__global__ void __launch_bounds__(256) vfma_kernel(double* out, int iters, double seed) {
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = seed + i + threadIdx.x * 1e-3;
  double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run_cyc(int blocks, int iters, double av, double bv) {
  double* out; CK(hipMalloc(&out, sizeof(double) * blocks * 256));
  unsigned long long* st; CK(hipMalloc(&st, 16 * blocks * 4));
  for (int rep = 0; rep < 3; ++rep) cyc_kernel<NACC><<<blocks, 256>>>(out, st, iters, av, bv);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(2 * blocks * 4);
  CK(hipMemcpy(h.data(), st, 16 * blocks * 4, hipMemcpyDeviceToHost));
  double cs = 0, rs = 0; for (int w = 0; w < blocks * 4; ++w) { cs += h[2 * w]; rs += h[2 * w + 1]; }
  cs /= blocks * 4; rs /= blocks * 4;
  double n = (double)iters * NACC;
  printf("cyc NACC=%d blocks=%d a=%g: %.1f shader-cycles/MFMA/wave, %.2f ns/MFMA/wave, eff clock %.3f GHz\n",
         NACC, blocks, av, cs / n, rs * 10.0 / n, cs / (rs * 10.0));
  CK(hipFree(out)); CK(hipFree(st));
}

template <int NACC>
static void run_rate(int blocks, int iters) {
  double* out; CK(hipMalloc(&out, sizeof(double) * blocks * 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate_kernel<NACC><<<blocks, 256>>>(out, iters / 10, 1.0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  rate_kernel<NACC><<<blocks, 256>>>(out, iters, 1.0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double nmfma = (double)blocks * 4 * iters * NACC;
  double flops = nmfma * 2.0 * 16 * 16 * 4;
  printf("rate NACC=%d blocks=%d iters=%d : %.3f ms  %.2f TFLOP/s  (%.1f ns per MFMA per wave)\n",
         NACC, blocks, iters, ms, flops / ms / 1e9, ms * 1e6 / ((double)iters * NACC) / (blocks > 256 ? (blocks / 256.0) : 1.0));
  CK(hipFree(out));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s arch %s CUs %d clock %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
  // ---- layout
  std::vector<double> A(64), B(64), C(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i + 100 * k;      // asymmetric
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k == 0 ? 1 : 0) * (j + 1) + (k == 1 ? 1000.0 * (j + 1) : 0) + (k >= 2 ? 1e6 * (k - 1) * (j + 3) : 0);
  double *dA, *dB, *dC; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 2048));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost));
  // reference
  double R[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; R[i][j] = s; }
  int bad1 = 0, bad2 = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    double v = C[l * 4 + r];
    if (v != R[(l >> 4) + 4 * r][l & 15]) ++bad1;          // hypothesis 1: row=(l>>4)+4r
    if (v != R[(l >> 4) * 4 + r][l & 15]) ++bad2;          // hypothesis 2: row=4(l>>4)+r (f32 map)
  }
  printf("layout: hyp1 row=(l>>4)+4r mismatches=%d ; hyp2 row=4(l>>4)+r mismatches=%d\n", bad1, bad2);

  // ---- MFMA rate: 1 block/CU (1 wave/SIMD), 2 blocks/CU
  int cus = p.multiProcessorCount;
  run_rate<1>(cus, 20000); run_rate<2>(cus, 20000); run_rate<4>(cus, 10000); run_rate<8>(cus, 5000); run_rate<16>(cus, 2500);
  run_rate<4>(2 * cus, 10000); run_rate<16>(2 * cus, 2500);
  run_rate<16>(1, 2500);

  run_cyc<4>(cus, 10000, 1.0, 1.0); run_cyc<16>(cus, 2500, 1.0, 1.0); run_cyc<4>(2 * cus, 10000, 1.0, 1.0); run_cyc<16>(2 * cus, 2500, 1.0, 1.0);
  run_cyc<4>(cus, 10000, 0.0, 0.0); run_cyc<4>(2 * cus, 10000, 0.0, 0.0); run_cyc<4>(1, 10000, 1.0, 1.0); run_cyc<2>(1, 10000, 1.0, 1.0); run_cyc<1>(1, 10000, 1.0, 1.0);
  run_cyc<8>(4 * cus, 2500, 1.0, 1.0);
  { double* out; CK(hipMalloc(&out, 8 * 2048 * 256)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    vfma_kernel<<<2048, 256>>>(out, 1000, 0.5); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); vfma_kernel<<<2048, 256>>>(out, 20000, 0.5); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("v_fma_f64: %.3f ms  %.2f TFLOP/s\n", ms, 2048.0 * 256 * 20000 * 16 * 2 / ms / 1e9); }
  // ---- HBM copy / write
  size_t bytes = (size_t)4 << 30; double2 *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes));
  CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0)); copy_kernel<<<2048, 256>>>(s, d, bytes / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("copy 4GiB: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e9);
    CK(hipEventRecord(e0)); write_kernel<<<2048, 256>>>(d, bytes / 16, 1.0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("write 4GiB: %.3f ms  %.2f TB/s\n", ms, 1.0 * bytes / ms / 1e9);
  }
  // ---- exp+sqrt fp64 throughput
  { double* out; CK(hipMalloc(&out, 8 * 2048 * 256));
    expsqrt_kernel<<<2048, 256>>>(out, 100, 0.5); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); expsqrt_kernel<<<2048, 256>>>(out, 2000, 0.5); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("exp+sqrt+poly fp64: %.3f ms  %.3f G elem/s\n", ms, 2048.0 * 256 * 2000 / ms / 1e6); }
  return 0;
}
