// readlane_chain_probe.hip -- cycles per link of the dependency chain the reduced solve's back-substitution is made of (gfx950, ONE wave):
//   x_j = lane j of z  (two v_readlane_b32 into an SGPR pair)  ->  z = fma(-L_j, x_j, z)  ->  x_(j-1) = lane j - 1 of z ...
// variants: 0 the chain as pba_solve.h writes it (runtime lane index in an SGPR)   1 lane index a compile-time constant (fully unrolled)
//           2 the broadcast through ds_bpermute (__shfl) instead of v_readlane      3 chain of plain dependent v_fma_f64 (no cross-lane step)
//           4 variant 0 with the select between two row registers in front of the read (the two-rows-per-lane form, n > 64)
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)__double2loint(v), src_lane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)__double2hiint(v), src_lane);
  return __hiloint2double((int)hi, (int)lo);
}

template <int V>
__global__ __launch_bounds__(64) void k_chain(double* out, unsigned long long* cyc, int links, double l0) {
  double z = 1.0 + 1e-3 * threadIdx.x, z1 = 2.0 + 1e-3 * threadIdx.x;
  const double L = l0 * (1.0 + 1e-6 * threadIdx.x);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (V == 1) {
    for (int i = 0; i < links; i += 64) {
#pragma unroll
      for (int j = 63; j >= 0; --j) { const double xj = readlane_f64(z, j); z = fma(-L, xj, z); }
    }
  } else {
    for (int i = links - 1; i >= 0; --i) {
      const int j = __builtin_amdgcn_readfirstlane(i & 63);
      double xj;
      if (V == 0) xj = readlane_f64(z, j);
      if (V == 2) xj = __shfl(z, j);
      if (V == 3) xj = z;
      if (V == 4) { const double zs = (i & 64) ? z1 : z; xj = readlane_f64(zs, j); z1 = fma(-L, xj, z1); }
      z = fma(-L, xj, z);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = z + z1;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
  const int links = 64 * 64;
  const char* names[] = {"v_readlane pair (runtime lane) + v_fma_f64", "v_readlane pair (constant lane) + v_fma_f64", "ds_bpermute broadcast + v_fma_f64",
                         "dependent v_fma_f64 alone", "select of two row registers + v_readlane pair + 2 v_fma_f64"};
  for (int v = 0; v < 5; ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      if (v == 0) hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, out, cyc, links, 1e-9);
      if (v == 1) hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, out, cyc, links, 1e-9);
      if (v == 2) hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(64), 0, 0, out, cyc, links, 1e-9);
      if (v == 3) hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(64), 0, 0, out, cyc, links, 1e-9);
      if (v == 4) hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(64), 0, 0, out, cyc, links, 1e-9);
      (void)hipDeviceSynchronize();
    }
    unsigned long long h = 0;
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-62s %7.1f cycles per link (s_memtime, %d links)\n", names[v], (double)h / links, links);
  }
  return 0;
}
