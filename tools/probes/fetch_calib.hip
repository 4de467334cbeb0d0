// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the engine uses
// (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a 16-byte/lane stream; other widths are uncalibrated).
// Every kernel streams a buffer far larger than L2 + Infinity Cache exactly once:
//     rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib      and      rocprofv3 --pmc WRITE_SIZE -- ./fetch_calib
// then divide the counter (KiB) by the known byte count printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int DW>
__global__ void k_read(const uint32_t* __restrict__ src, uint32_t* __restrict__ sink, size_t n_vec) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t v[DW];
    __builtin_memcpy(v, src + i * DW, sizeof(v));
#pragma unroll
    for (int k = 0; k < DW; ++k) acc ^= v[k];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// the engine's footprint pattern: two lanes fetch one 24-byte row segment (dwordx3 each), 6 rows `pitch` apart per site
__global__ void k_read_rows(const uint32_t* __restrict__ src, uint32_t* __restrict__ sink, size_t n_sites, int pitch) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  const size_t site = t >> 1;
  const int half = (int)(t & 1);
  if (site < n_sites) {
    const uint32_t* base = src + (site / 200) * (size_t)pitch * 6 + (site % 200) * 6 + half * 3;   // 200 disjoint sites per 6-row band
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      uint32_t v[3];
      __builtin_memcpy(v, base + (size_t)r * pitch, sizeof(v));
      acc ^= v[0] ^ v[1] ^ v[2];
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int DW>
__global__ void k_write(uint32_t* __restrict__ dst, size_t n_vec) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t v[DW];
#pragma unroll
    for (int k = 0; k < DW; ++k) v[k] = (uint32_t)i + k;
    __builtin_memcpy(dst + i * DW, v, sizeof(v));
  }
}

int main() {
  const size_t bytes = (size_t)3 << 30;             // 3 GiB: one pass never re-hits L2 / the 256 MiB Infinity Cache
  uint32_t *buf, *sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
  (void)hipMemset(buf, 1, bytes);
  (void)hipDeviceSynchronize();
  const int grid = 256 * 16;
  hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, 0, buf, sink, bytes / 4);
  hipLaunchKernelGGL(k_read<2>, dim3(grid), dim3(256), 0, 0, buf, sink, bytes / 8);
  hipLaunchKernelGGL(k_read<3>, dim3(grid), dim3(256), 0, 0, buf, sink, bytes / 12);
  hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, buf, sink, bytes / 16);
  const int pitch = 1241;                            // dwords per image row, as in the engine's frames
  const size_t bands = bytes / 4 / ((size_t)pitch * 6) - 1, n_sites = bands * 200;
  hipLaunchKernelGGL(k_read_rows, dim3((unsigned)((2 * n_sites + 255) / 256)), dim3(256), 0, 0, buf, sink, n_sites, pitch);
  hipLaunchKernelGGL(k_write<2>, dim3(grid), dim3(256), 0, 0, buf, bytes / 8);
  hipLaunchKernelGGL(k_write<4>, dim3(grid), dim3(256), 0, 0, buf, bytes / 16);
  (void)hipDeviceSynchronize();
  std::printf("known bytes: k_read<1..4> %zu each; k_read_rows %zu requested (24-byte segments: %zu sites x 6 rows), "
              "k_write<2|4> %zu each\n", bytes, n_sites * 6 * 24, n_sites, bytes);
  return 0;
}
