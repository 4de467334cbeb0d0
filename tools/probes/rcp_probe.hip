// Accuracy of the fp64 reciprocal estimate (v_rcp_f64) on gfx950, raw and after one / two Newton steps, against IEEE 1/x.
// hipcc --offload-arch=gfx950 -O2 tools/probes/rcp_probe.hip -o /tmp/rcp_probe && /tmp/rcp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* x, double* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double y = __builtin_amdgcn_rcp(v);
  o[3 * i] = y;
  y = fma(fma(-v, y, 1.0), y, y);
  o[3 * i + 1] = y;
  y = fma(fma(-v, y, 1.0), y, y);
  o[3 * i + 2] = y;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> h(n), r(3 * (size_t)n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(-40.0, 40.0), m(1.0, 2.0);
  for (int i = 0; i < n; ++i) h[i] = std::ldexp(m(g), (int)u(g));
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 3 * (size_t)n * 8);
  hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(r.data(), dout, 3 * (size_t)n * 8, hipMemcpyDeviceToHost);
  double e[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) for (int k2 = 0; k2 < 3; ++k2) e[k2] = std::fmax(e[k2], std::fabs(r[3 * (size_t)i + k2] * h[i] - 1.0));
  std::printf("v_rcp_f64 max |x y - 1|: raw %.3e, one Newton step %.3e, two %.3e (eps = %.3e)\n", e[0], e[1], e[2], 2.2e-16);
  return 0;
}
