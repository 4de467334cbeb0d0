// valu_rate_probe.hip -- issue rates of the instruction kinds the exact bilinear walk is made of (gfx950):
// v_fma_f64, v_mul_f64, v_add_f64, v_cvt_f64_f32, v_cvt_f32_f64, v_fma_f32, v_cvt_f32_i32, v_bfe_i32, ds_read_b32.
// One workgroup of 256 threads per CU x OCC, every wave runs 16 independent chains of one instruction kind.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, unsigned long long* cyc, int iters) {
  double d[16]; float f[16]; int n[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { d[k] = 1.0 + 1e-3 * (threadIdx.x + k); f[k] = 1.0f + 1e-3f * (threadIdx.x + k); n[k] = threadIdx.x * 977 + k; }
  const double a = 1.0000001, b = 1e-9;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (KIND == 0) d[k] = fma(d[k], a, b);
      if (KIND == 1) d[k] = d[k] * a;
      if (KIND == 2) d[k] = d[k] + b;
      if (KIND == 3) { d[k] = (double)f[k]; asm volatile("" : "+v"(d[k])); f[k] += 1.0f; }            // cvt_f64_f32 (+ one f32 add)
      if (KIND == 4) { f[k] = (float)d[k]; asm volatile("" : "+v"(f[k])); d[k] = d[k] + b; }          // cvt_f32_f64 (+ one f64 add)
      if (KIND == 5) f[k] = fmaf(f[k], 1.0000001f, 1e-9f);
      if (KIND == 6) { f[k] = (float)n[k]; asm volatile("" : "+v"(f[k])); n[k] += 3; }                // cvt_f32_i32 (+ int add)
      if (KIND == 7) { n[k] = ((n[k] << 14) >> 22) + i; }                                              // shift pair (bfe)
      if (KIND == 8) f[k] = f[k] + 1e-9f;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += (float)d[k] + f[k] + (float)n[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
  for (int occ : {1, 2, 3}) {
    const int iters = 2048, grid = 256 * occ;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * grid * 256);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * grid * 4);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 8);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions of the probed kind per SIMD per microsecond
    const double wave_instr = (double)grid * 4 * iters * 16;
    std::printf("%-34s waves/SIMD %d  %.3f ms  %.1f ns per wave-instruction per SIMD (%.2f G wave-instr/s/SIMD)\n", name, occ, ms,
                ms * 1e6 / (wave_instr / 1024.0), wave_instr / 1024.0 / (ms * 1e6));
    (void)hipFree(out); (void)hipFree(cyc);
  }
}

int main() {
  run<0>("v_fma_f64");
  run<1>("v_mul_f64");
  run<2>("v_add_f64");
  run<3>("v_cvt_f64_f32 + v_add_f32");
  run<4>("v_cvt_f32_f64 + v_add_f64");
  run<5>("v_fma_f32");
  run<6>("v_cvt_f32_i32 + v_add_u32");
  run<7>("v_lshl + v_ashr + v_add (i32)");
  run<8>("v_add_f32");
  return 0;
}
