// grid_sync_probe.hip -- cost of the cross-workgroup hand-overs a resident (one launch per solve) LM kernel is made of, gfx950:
//   mode 0  counter barrier: every workgroup adds 1 to a counter (agent scope) and polls it until all G have arrived
//   mode 1  ticket + flag : every workgroup takes a ticket; the LAST arriver stores a flag word, everybody polls the flag
//   mode 3  flag per workgroup (u32, contiguous), no read-modify-write: every workgroup stores its own epoch, all 256 threads of every
//           workgroup poll one flag each until all G show the epoch (all-to-all barrier)
//   mode 4  mode 3 with the flags 64 bytes apart
//   mode 5  gather + broadcast: workgroup 0 polls the G per-workgroup flags, then stores ONE go-flag everybody else polls (the shape of
//           "partials -> fixed serial workgroup -> decision")
//   mode 6  cooperative_groups::this_grid().sync() (the runtime's grid barrier)
//   mode 7  mode 6 with the 4 KB payload of mode 2 around it
//   mode 2  mode 1 with a 4 KB write-through (sc1) payload stored and drained (s_waitcnt vmcnt(0)) in front of the ticket and
//           read back (sc1 loads) by every workgroup behind the flag -- the shape of "partials -> last arriver -> broadcast"
// Launched cooperatively (hipLaunchCooperativeKernel) with G = CUs x {1, 2} workgroups of 256 threads.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void store_agent(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__global__ __launch_bounds__(256) void k_sync(unsigned* cnt, unsigned* flag, double* payload, double* sink, unsigned long long* ticks, int iters, int mode, int sleep) {
  const int tid = threadIdx.x, G = gridDim.x;
  __shared__ int s_last;
  double acc = 0.0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (mode == 2) {
      // 512 doubles per workgroup, write-through, drained
      store_agent(payload + ((size_t)(it & 1) * G + blockIdx.x) * 512 + tid, (double)(it + tid));
      store_agent(payload + ((size_t)(it & 1) * G + blockIdx.x) * 512 + 256 + tid, (double)(it - tid));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (mode >= 6) {
      if (mode == 7) {
        store_agent(payload + ((size_t)(it & 1) * G + blockIdx.x) * 512 + tid, (double)(it + tid));
        store_agent(payload + ((size_t)(it & 1) * G + blockIdx.x) * 512 + 256 + tid, (double)(it - tid));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      cooperative_groups::this_grid().sync();
      if (mode == 7) {
        const int src = (blockIdx.x + 1 + it) % G;
        acc += load_agent(payload + ((size_t)(it & 1) * G + src) * 512 + tid) + load_agent(payload + ((size_t)(it & 1) * G + src) * 512 + 256 + tid);
      }
    } else if (mode >= 3) {
      const int stride = (mode == 4) ? 16 : 1;
      const unsigned ep = (unsigned)(it + 1);
      unsigned* flags = reinterpret_cast<unsigned*>(payload);      // (reused as the flag array)
      if (tid == 0) __hip_atomic_store(flags + (size_t)blockIdx.x * stride, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (mode == 5 && blockIdx.x != 0) {
        if (tid == 0) while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ep) { if (sleep) __builtin_amdgcn_s_sleep(1); }
        __syncthreads();
      } else {
        for (;;) {
          int ok = 1;
          for (int g = tid; g < G; g += 256) ok &= (__hip_atomic_load(flags + (size_t)g * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ep) ? 1 : 0;
          if (__syncthreads_and(ok)) break;
          if (sleep) __builtin_amdgcn_s_sleep(1);
        }
        if (mode == 5 && tid == 0) __hip_atomic_store(flag, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (mode == 0) {
      if (tid == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)(it + 1) * (unsigned)G;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { if (sleep) __builtin_amdgcn_s_sleep(1); }
      }
      __syncthreads();
    } else {
      if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (unsigned)(it + 1) * (unsigned)G - 1u);
        if (s_last) __hip_atomic_store(flag, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1)) { if (sleep) __builtin_amdgcn_s_sleep(1); }
      }
      __syncthreads();
    }
    if (mode == 2) {
      // everybody reads one other workgroup's payload
      const int src = (blockIdx.x + 1 + it) % G;
      acc += load_agent(payload + ((size_t)(it & 1) * G + src) * 512 + tid) + load_agent(payload + ((size_t)(it & 1) * G + src) * 512 + 256 + tid);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 256 + tid] = acc;
}

int main() {
  int dev = 0;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, dev);
  const int cus = prop.multiProcessorCount;
  unsigned *cnt, *flag; double *payload, *sink; unsigned long long* ticks;
  hipMalloc(&cnt, 4); hipMalloc(&flag, 4); hipMalloc(&payload, sizeof(double) * 2 * 1024 * 512); hipMalloc(&sink, sizeof(double) * 1024 * 256); hipMalloc(&ticks, 8 * 1024);
  int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
  printf("device %s, %d CUs, wall clock %d kHz, cooperative launch %d\n", prop.name, cus, khz, prop.cooperativeLaunch);
  for (int per_cu : {1, 2})
    for (int mode : {5, 2, 6, 7})
      for (int sleep : {0, 1}) {
        int G = cus * per_cu, iters = 2000;
        hipMemset(cnt, 0, 4); hipMemset(flag, 0, 4); hipMemset(payload, 0, sizeof(double) * 2 * 1024 * 512);
        void* args[] = {&cnt, &flag, &payload, &sink, &ticks, &iters, &mode, &sleep};
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipError_t r = hipLaunchCooperativeKernel((const void*)k_sync, dim3(G), dim3(256), args, 0, 0);
        hipEventRecord(e1);
        hipError_t r2 = hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(G);
        hipMemcpy(h.data(), ticks, 8 * G, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
        printf("G %4d (x%d per CU) mode %d sleep %d: %s %s  %.3f us per hand-over (device ticks), launch+run %.3f ms\n", G, per_cu, mode, sleep,
               hipGetErrorString(r), hipGetErrorString(r2), (double)mx / (khz * 1e-3) / iters, ms);
      }
  return 0;
}
