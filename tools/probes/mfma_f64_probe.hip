// mfma_f64_probe.hip -- measured issue rates of v_mfma_f64_16x16x4_f64 against v_fma_f64 on gfx950, and of the two
// running side by side on one SIMD (different waves).  Answers the DESIGN question "does the fp64 matrix pipe buy
// anything for the 6x6x3 Schur products": build with  hipcc --offload-arch=gfx950 -O3 mfma_f64_probe.hip -o mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: MFMA only, 1: VALU fma only, 2: even waves MFMA / odd waves VALU
__global__ __launch_bounds__(256) void k_probe(double* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = MODE == 0 || (MODE == 2 && (wave & 1) == 0);
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double4_t acc[4];
  for (int k = 0; k < 4; ++k) acc[k] = double4_t{0.0, 0.0, 0.0, 0.0};
  double v[16];
  for (int k = 0; k < 16; ++k) v[k] = k;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (do_mfma) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
  } else {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = fma(v[k], a, b);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0.0;
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  for (int k = 0; k < 16; ++k) s += v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  const int iters = 4096, grid = 256 * waves_per_simd;   // 256-thread workgroups = one wave per SIMD each
  double* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(double) * grid * 256);
  hipMalloc(&cyc, sizeof(unsigned long long) * grid * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
  double c_m = 0, c_v = 0; int n_m = 0, n_v = 0;
  for (int i = 0; i < grid * 4; ++i) {
    const bool m = MODE == 0 || (MODE == 2 && (i & 1) == 0);
    if (m) { c_m += h[i]; ++n_m; } else { c_v += h[i]; ++n_v; }
  }
  // flops: MFMA 16x16x4 = 2 * 1024 per wave-instruction; fma = 2 * 64
  const double n_waves = (double)grid * 4;
  const double mf = (MODE == 1 ? 0.0 : (MODE == 0 ? n_waves : n_waves / 2)) * iters * 4 * 2048.0;
  const double vf = (MODE == 0 ? 0.0 : (MODE == 1 ? n_waves : n_waves / 2)) * iters * 16 * 128.0;
  std::printf("%-34s waves/SIMD %d  %.3f ms  MFMA %.2f TFLOP/s (%.1f cycles/instr/wave)  VALU %.2f TFLOP/s (%.1f cycles/instr/wave)\n", name,
              waves_per_simd, ms, mf / ms * 1e-9, n_m ? c_m / n_m / (iters * 4.0) : 0.0, vf / ms * 1e-9, n_v ? c_v / n_v / (iters * 16.0) : 0.0);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_mfma_f64_16x16x4_f64 only", w);
    run<1>("v_fma_f64 only", w);
    run<2>("MFMA waves + VALU waves, same SIMDs", w);
  }
  return 0;
}
