// hwmath.hip — measures the hardware transcendental units against double precision, exhaustively over the 2^23 values
// GetRand01 can return (k * 2^-23): v_sin_f32 / v_cos_f32 (argument in revolutions) and exp2(y * log2(x)) as a pow.
// Build: hipcc --offload-arch=gfx950 -O2 -o hwmath hwmath.hip ; prints max abs / rel errors.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k_sincos(float* s, float* c, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float u = (float)i * (1.0f / 8388608.0f);
  s[i] = __builtin_amdgcn_sinf(u);
  c[i] = __builtin_amdgcn_cosf(u);
}
__global__ void k_pow(float* out, uint32_t n, float y) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = (float)i * (1.0f / 8388608.0f);
  out[i] = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
}
int main() {
  const uint32_t n = 1u << 23;
  float *ds, *dc;
  if (hipMalloc(&ds, n * 4) != hipSuccess || hipMalloc(&dc, n * 4) != hipSuccess) return 1;
  std::vector<float> s(n), c(n);
  hipLaunchKernelGGL(k_sincos, dim3(n / 256), dim3(256), 0, 0, ds, dc, n);
  if (hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  if (hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  double es = 0, ec = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const double a = 2.0 * M_PI * (double)i / 8388608.0;
    es = std::fmax(es, std::fabs((double)s[i] - std::sin(a)));
    ec = std::fmax(ec, std::fabs((double)c[i] - std::cos(a)));
  }
  printf("v_sin_f32 max abs err %.3e   v_cos_f32 max abs err %.3e   (over all 2^23 u)\n", es, ec);
  for (float y : {5.0f, 1.0f / 34.0f, 1.0f / 102.0f, 32.0f, 100.0f, 1000.0f}) {
    hipLaunchKernelGGL(k_pow, dim3(n / 256), dim3(256), 0, 0, ds, n, y);
    if (hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    double er = 0; uint32_t worst = 0;
    for (uint32_t i = 1; i < n; ++i) {
      const double want = std::pow((double)i / 8388608.0, (double)y);
      if (want < 1e-30) continue;
      const double r = std::fabs((double)s[i] - want) / want;
      if (r > er) { er = r; worst = i; }
    }
    printf("exp2(y*log2(x)) y=%-10g max rel err %.3e at x=%.6g (results >= 1e-30)\n", y, er, worst / 8388608.0);
  }
  return 0;
}
