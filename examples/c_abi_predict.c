/* Host program on the C ABI alone (no Python, no torch): zero-mean Matern-5/2 kriging at m points and the
 * negative log-likelihood through the fused drivers of include/gpmp_hip.h, then -- with a constant mean, P = ones(n, 1) --
 * the restricted likelihood with its analytic gradient (gpmp_nll_grad) and leave-one-out (gpmp_loo).
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_predict.c -Lgpmp_amd -lgpmp_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/gpmp_amd -o c_abi_predict
 *   ./c_abi_predict [n] [m]        prints  nll  and the first posterior means / variances
 *
 * The synthetic data are those of bench.py: x ~ U[0,1]^d from a fixed LCG, z = sin(2 pi x_0) + sum x_j.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "gpmp_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define GK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "gpmp: rc=%d %s (%s:%d)\n", rc_, gpmp_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static double lcg(unsigned long long* s) { *s = *s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(*s >> 11) / 9007199254740992.0; }

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2048, m = argc > 2 ? atoi(argv[2]) : 1000, d = 4, p = 2;
  double theta[1 + 4];
  theta[0] = 0.0;
  for (int j = 0; j < d; ++j) theta[1 + j] = -log(0.5 * (1.0 + (double)j / d));
  double *xi = malloc(sizeof(double) * n * d), *zi = malloc(sizeof(double) * n), *xt = malloc(sizeof(double) * m * d);
  unsigned long long seed = 1234;
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int j = 0; j < d; ++j) { xi[i * d + j] = lcg(&seed); if (j) s += xi[i * d + j]; }
    zi[i] = sin(6.283185307179586 * xi[i * d]) + s;
  }
  for (int i = 0; i < m * d; ++i) xt[i] = lcg(&seed);

  double *dxi, *dzi, *dxt, *ws, *zpm, *zpv, *nll;
  int* info;
  const size_t wn = gpmp_predict_ws_elems(n, m), wl = gpmp_nll_ws_elems(n);
  CK(hipMalloc((void**)&dxi, sizeof(double) * n * d)); CK(hipMalloc((void**)&dzi, sizeof(double) * n));
  CK(hipMalloc((void**)&dxt, sizeof(double) * m * d)); CK(hipMalloc((void**)&ws, sizeof(double) * (wn > wl ? wn : wl)));
  CK(hipMalloc((void**)&zpm, sizeof(double) * m)); CK(hipMalloc((void**)&zpv, sizeof(double) * m));
  CK(hipMalloc((void**)&nll, sizeof(double))); CK(hipMalloc((void**)&info, sizeof(int)));
  CK(hipMemcpy(dxi, xi, sizeof(double) * n * d, hipMemcpyHostToDevice));
  CK(hipMemcpy(dzi, zi, sizeof(double) * n, hipMemcpyHostToDevice));
  CK(hipMemcpy(dxt, xt, sizeof(double) * m * d, hipMemcpyHostToDevice));

  GK(gpmp_nll_zero_mean(dxi, dzi, n, d, p, theta, 0, ws, nll, info, NULL));
  double hnll; int hinfo;
  CK(hipMemcpy(&hnll, nll, sizeof(double), hipMemcpyDeviceToHost)); CK(hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost));
  printf("n=%d m=%d d=%d  nll=%.12e  info=%d\n", n, m, d, hnll, hinfo);

  GK(gpmp_predict_zero_mean(dxi, dzi, dxt, n, m, d, p, theta, 0, 1, ws, zpm, zpv, info, NULL));
  double hm[4], hv[4];
  const int k = m < 4 ? m : 4;
  CK(hipMemcpy(hm, zpm, sizeof(double) * k, hipMemcpyDeviceToHost)); CK(hipMemcpy(hv, zpv, sizeof(double) * k, hipMemcpyDeviceToHost));
  for (int i = 0; i < k; ++i) printf("xt[%d]: mean %.12e  var %.6e\n", i, hm[i], hv[i]);

  /* constant-mean model: REML value + gradient (what each optimiser evaluation needs) and leave-one-out */
  const int q = 1;
  double *P = malloc(sizeof(double) * n), *dP, *ws2, *val, *grad, *zl, *s2, *el;
  for (int i = 0; i < n; ++i) P[i] = 1.0;
  const size_t wg = gpmp_nll_grad_ws_elems(n, d, q), wo = gpmp_loo_ws_elems(n, q);
  CK(hipMalloc((void**)&dP, sizeof(double) * n)); CK(hipMemcpy(dP, P, sizeof(double) * n, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&ws2, sizeof(double) * (wg > wo ? wg : wo)));
  CK(hipMalloc((void**)&val, sizeof(double))); CK(hipMalloc((void**)&grad, sizeof(double) * (1 + d)));
  CK(hipMalloc((void**)&zl, sizeof(double) * n)); CK(hipMalloc((void**)&s2, sizeof(double) * n)); CK(hipMalloc((void**)&el, sizeof(double) * n));
  GK(gpmp_nll_grad(dxi, dzi, dP, 1, n, d, q, p, theta, 0, ws2, val, grad, info, NULL));
  double hval, hgrad[1 + 4];
  CK(hipMemcpy(&hval, val, sizeof(double), hipMemcpyDeviceToHost)); CK(hipMemcpy(hgrad, grad, sizeof(double) * (1 + d), hipMemcpyDeviceToHost));
  CK(hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost));
  printf("reml=%.12e  info=%d\n", hval, hinfo);
  for (int j = 0; j < 1 + d; ++j) printf("dreml[%d] %.12e\n", j, hgrad[j]);
  GK(gpmp_loo(dxi, dzi, dP, 1, n, d, q, p, theta, 0, ws2, zl, s2, el, info, NULL));
  double hz[3], hs[3];
  CK(hipMemcpy(hz, zl, sizeof(double) * 3, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, s2, sizeof(double) * 3, hipMemcpyDeviceToHost));
  for (int i = 0; i < 3; ++i) printf("loo[%d]: zloo %.12e  s2loo %.6e\n", i, hz[i], hs[i]);

  /* prediction with that constant mean of unknown level (universal kriging): the mean design at the prediction points too */
  double *Pt = malloc(sizeof(double) * m), *dPt, *ws3;
  for (int i = 0; i < m; ++i) Pt[i] = 1.0;
  CK(hipMalloc((void**)&dPt, sizeof(double) * m)); CK(hipMemcpy(dPt, Pt, sizeof(double) * m, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&ws3, sizeof(double) * gpmp_predict_mean_ws_elems(n, m, q)));
  GK(gpmp_predict_mean(dxi, dzi, dP, 1, dxt, dPt, 1, n, m, d, q, p, theta, 0, 1, ws3, zpm, zpv, info, NULL));
  CK(hipMemcpy(hm, zpm, sizeof(double) * k, hipMemcpyDeviceToHost)); CK(hipMemcpy(hv, zpv, sizeof(double) * k, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost));
  for (int i = 0; i < k; ++i) printf("uk[%d]: ukm %.12e  ukv %.6e\n", i, hm[i], hv[i]);
  return hinfo != 0;
}
