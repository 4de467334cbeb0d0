// schur_pair_mfma_probe.hip -- VERDICT r3/r4: "k_schur pair blocks on v_mfma_f64_16x16x4_f64 for n_frames >= 12: build and MEASURE".
// The pair-block phase of k_schur in isolation, in its 16-frame shape (15 free cameras, n = 90, tiles of 8 points x 16 observations,
// two waves per workgroup, four workgroups per CU), two ways over the SAME LDS-resident tile:
//   VALU : the production loop (pba_kernels.h, P3a): a thread owns one 6x6 block (camera pair a <= b, 120 pairs), per point 32 LDS
//          reads of the rank-2 factors + 92 FMAs (N = Q_a MAp_b^T, Z = N Ac_b, T -= Ac_a^T Z)
//   MFMA : S (96 x 96, padded) -= Y W^T per point with Y, W (90 x 3, k padded to 4) staged per point in LDS; the 21 upper 16x16
//          tiles are split over the two waves (11 + 10 accumulator tiles = 44 / 40 doubles per lane), per point and wave 6 + 6
//          operand reads and 11 / 10 v_mfma_f64_16x16x4_f64.  (Forming W = Ac^T MAp, Y = Ac^T Q per observation -- 72 FMAs instead of
//          the 18 of the rank-2 record -- is NOT charged to the MFMA side here.)
// Both variants run `tiles` tiles per workgroup back to back; reported: cycles (s_memtime) per tile and wave, and the wall time of the launch.
// Build: hipcc --offload-arch=gfx950 -O3 schur_pair_mfma_probe.hip -o schur_pair_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int kNF = 15, kPairs = kNF * (kNF + 1) / 2, kPts = 8, kStride = 37;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// factors per observation lane (point q, camera a): lane = 16 q + a + 1 (camera slot 0 is the constant one)
__global__ __launch_bounds__(128, 2) void k_valu(const double* __restrict__ fac, double* out, unsigned long long* cyc, int tiles) {
  __shared__ double s_obs[128 * kStride];
  const int tid = threadIdx.x;
  for (int k = tid; k < 128 * kStride; k += 128) s_obs[k] = fac[k];
  int pa = 0, pb = 0;
  { int a = 0, rem = tid < kPairs ? tid : 0; while (rem >= kNF - a) { rem -= kNF - a; ++a; } pa = a; pb = a + rem; }
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  lds_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    if (tid < kPairs) {
      for (int q = 0; q < kPts; ++q) {
        const double* Fa = s_obs + (16 * q + pa + 1) * kStride;
        const double* Fb = s_obs + (16 * q + pb + 1) * kStride;
        double aa[10], qa[6], ab[10], mb[6];
#pragma unroll
        for (int k = 0; k < 10; ++k) { aa[k] = Fa[k]; ab[k] = Fb[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { qa[k] = Fa[16 + k]; mb[k] = Fb[10 + k]; }
        double N[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 2; ++c) N[r][c] = fma(qa[3 * r + 2], mb[3 * c + 2], fma(qa[3 * r + 1], mb[3 * c + 1], qa[3 * r] * mb[3 * c]));
        double Z[2][6];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
          for (int j = 0; j < 3; ++j) Z[r][j] = fma(N[r][1], ab[3 + j], N[r][0] * ab[j]);
          Z[r][3] = N[r][0] * ab[6]; Z[r][4] = N[r][1] * ab[8]; Z[r][5] = fma(N[r][1], ab[9], N[r][0] * ab[7]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[6 * i + j] = fma(-aa[3 + i], Z[1][j], fma(-aa[i], Z[0][j], acc[6 * i + j]));
          acc[18 + j] = fma(-aa[6], Z[0][j], acc[18 + j]);
          acc[24 + j] = fma(-aa[8], Z[1][j], acc[24 + j]);
          acc[30 + j] = fma(-aa[9], Z[1][j], fma(-aa[7], Z[0][j], acc[30 + j]));
        }
      }
    }
    lds_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid < kPairs) for (int k = 0; k < 36; ++k) out[((size_t)blockIdx.x * kPairs + tid) * 36 + k] = acc[k];
  if ((tid & 63) == 0) cyc[blockIdx.x * 2 + (tid >> 6)] = t1 - t0;
}

// yw: per point [2][96][4] doubles: Y rows (k = 3 zero), then W rows
__global__ __launch_bounds__(128, 2) void k_mfma(const double* __restrict__ yw, double* out, unsigned long long* cyc, int tiles) {
  __shared__ double s_yw[kPts * 2 * 96 * 4];      // 48 KB: (the production tile would stage one or two points at a time)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < kPts * 2 * 96 * 4; k += 128) s_yw[k] = yw[k];
  // upper tiles (I <= J), enumerated row by row; wave 0 takes the even ones, wave 1 the odd ones
  double4_t acc[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) acc[k] = double4_t{0.0, 0.0, 0.0, 0.0};
  lds_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    for (int q = 0; q < kPts; ++q) {
      const double* Y = s_yw + (size_t)q * 2 * 96 * 4;
      const double* W = Y + 96 * 4;
      double av[6], bv[6];
#pragma unroll
      for (int I = 0; I < 6; ++I) { av[I] = -Y[(16 * I + (lane & 15)) * 4 + (lane >> 4)]; bv[I] = W[(16 * I + (lane & 15)) * 4 + (lane >> 4)]; }
      // (compile-time accumulator indices: a uniform branch per wave)
      if (wave == 0) {
        int e = 0;
#pragma unroll
        for (int I = 0; I < 6; ++I)
#pragma unroll
          for (int J = I; J < 6; ++J) { if ((e & 1) == 0) acc[e >> 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I], bv[J], acc[e >> 1], 0, 0, 0); ++e; }
      } else {
        int e = 0;
#pragma unroll
        for (int I = 0; I < 6; ++I)
#pragma unroll
          for (int J = I; J < 6; ++J) { if ((e & 1) == 1) acc[e >> 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I], bv[J], acc[e >> 1], 0, 0, 0); ++e; }
      }
    }
    lds_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < 11; ++k)
    for (int i = 0; i < 4; ++i) out[(((size_t)blockIdx.x * 2 + wave) * 11 + k) * 256 + 4 * lane + i] = acc[k][i];
  if (lane == 0) cyc[blockIdx.x * 2 + wave] = t1 - t0;
}

int main() {
  const int grid = 1024, tiles = 64;
  std::vector<double> fac(128 * kStride), yw(kPts * 2 * 96 * 4, 0.0);
  for (size_t i = 0; i < fac.size(); ++i) fac[i] = std::sin(0.37 * (double)i) * 1e-2;
  // W = Ac^T MAp (6 x 3), Y = Ac^T Q per observation from the same factors (Ac: rotation 2x3 | ju0 0 ju2 ; 0 jv1 jv2)
  for (int q = 0; q < kPts; ++q)
    for (int a = 0; a < kNF; ++a) {
      const double* F = &fac[(16 * q + a + 1) * kStride];
      double Ac[2][6] = {{F[0], F[1], F[2], F[6], 0.0, F[7]}, {F[3], F[4], F[5], 0.0, F[8], F[9]}};
      for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 3; ++k) {
          yw[((size_t)q * 2 + 0) * 96 * 4 + (6 * a + i) * 4 + k] = Ac[0][i] * F[16 + k] + Ac[1][i] * F[19 + k];      // Y = Ac^T Q
          yw[((size_t)q * 2 + 1) * 96 * 4 + (6 * a + i) * 4 + k] = Ac[0][i] * F[10 + k] + Ac[1][i] * F[13 + k];      // W = Ac^T MAp
        }
    }
  double *d_fac, *d_yw, *d_o1, *d_o2; unsigned long long* d_c;
  hipMalloc(&d_fac, fac.size() * 8); hipMalloc(&d_yw, yw.size() * 8);
  hipMalloc(&d_o1, (size_t)grid * kPairs * 36 * 8); hipMalloc(&d_o2, (size_t)grid * 2 * 11 * 256 * 8); hipMalloc(&d_c, grid * 2 * 8);
  hipMemcpy(d_fac, fac.data(), fac.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_yw, yw.data(), yw.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto report = [&](const char* name, float ms) {
    std::vector<unsigned long long> h(grid * 2);
    hipMemcpy(h.data(), d_c, h.size() * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += (double)v;
    std::printf("%-6s %4d workgroups x %d tiles of 8 points, 15 free cameras: %.3f ms per launch, %.0f cycles per tile and wave, %.2f us per tile per CU-slot\n",
                name, grid, tiles, ms, c / h.size() / tiles, 1e3 * ms / tiles / (grid / 1024.0));
  };
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_valu, dim3(grid), dim3(128), 0, 0, d_fac, d_o1, d_c, tiles); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k_valu, dim3(grid), dim3(128), 0, 0, d_fac, d_o1, d_c, tiles); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); report("VALU", ms);
    hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(128), 0, 0, d_yw, d_o2, d_c, tiles); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(128), 0, 0, d_yw, d_o2, d_c, tiles); hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1); report("MFMA", ms);
  }
  // cross-check: block (0, 1) entry (0, 0) of both (tiles x the same tile): T = - sum_q Y_0q[0,:] . W_1q[0,:]
  std::vector<double> o1(36), o2(256);
  hipMemcpy(o1.data(), d_o1 + (size_t)1 * 36, 36 * 8, hipMemcpyDeviceToHost);      // pair index 1 = (0, 1)
  hipMemcpy(o2.data(), d_o2, 256 * 8, hipMemcpyDeviceToHost);                      // wave 0, first tile (I = 0, J = 0): rows 0-15 x cols 0-15
  // D layout of v_mfma_f64_16x16x4: lane l holds rows 4 (l / 16) + i, column l % 16; entry (row 0, col 6) = block (0, 1) entry (0, 0)
  std::printf("cross-check block (0,1)[0][0]: VALU % .12e  MFMA % .12e\n", o1[0], o2[4 * 6 + 0]);
  return 0;
}
