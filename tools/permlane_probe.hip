#include <hip/hip_runtime.h>
__global__ void k(unsigned* o) {
  unsigned x = threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  auto q = __builtin_amdgcn_permlane32_swap(r[0], r[0], false, false);
  o[threadIdx.x] = q[0]; o[64 + threadIdx.x] = q[1]; o[128 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 192 * 4); k<<<1, 64>>>(d); unsigned h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int s = 0; s < 3; ++s) { for (int i = 0; i < 64; i += 4) printf("%u ", h[64 * s + i]); printf("\n"); }
}
