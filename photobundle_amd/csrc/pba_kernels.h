// pba_kernels.h -- hand-written HIP kernels of the photometric BA hot path for gfx950 (CDNA4, wave64).
//
// Replaces, on the device, what the reference evaluates through Ceres autodiff:
//   DescriptorError::operator()  reference src/photobundle.cc:696-727
//   SampleWithDerivative / SampleLinear / LinearInitAxis  reference src/sample_eigen.h:33-126
//   Chain::Rule (derivative injection)  reference src/jet_extras.h:87-111 -> analytic: J_i = -w [gx gy] A
//   Calibration::project  reference src/calibration.h:34-38
//   HuberLoss + Corrector, SchurEliminator, LevenbergMarquardtStrategy (Ceres, absent dependency)
#pragma once
#include "pba_device.h"

#include <cfloat>
#include <climits>
#include <type_traits>

// Per-phase cycle stamps inside the kernels (PBA_SCHUR_TIMING=1|2 at run time) cost registers and a device printf, so
// they are compiled in only on request: make TIMING=1.
#ifndef PBA_PHASE_TIMING
#define PBA_PHASE_TIMING 0
#endif

// Resident waves per SIMD the Jacobian-pass sampling kernel is compiled for at patch radius <= 2 (register budget
// 512 / N VGPRs; 38.6 KB of LDS per 256-thread workgroup allows 4).  Measured at configs[1]: 4 -> 49.3 us, 3 -> 52.1 us.
#ifndef PBA_SAMPLE_WAVES_PER_SIMD
#define PBA_SAMPLE_WAVES_PER_SIMD 4
#endif
// ... and at patch radius >= 4 (register budget 512 / N)
#ifndef PBA_SAMPLE_WAVES_LARGE
#define PBA_SAMPLE_WAVES_LARGE 2
#endif
// Timing experiment only (results are WRONG): irregular observations contribute nothing
#ifndef PBA_EXPERIMENT_SKIP_IRREGULAR
#define PBA_EXPERIMENT_SKIP_IRREGULAR 0
#endif
// Order of the exact patch walk inside one footprint row (see k_sample): experiment switch
#ifndef PBA_WALK_ROWWISE
#define PBA_WALK_ROWWISE 0
#endif

// How the Jacobian-pass records / the Schur partials leave the CU (A/B switches): 0 nontemporal (records) or plain (partials),
// 1 agent-scope write-through (`sc1`: the line does not stay dirty in the XCD's L2, so the kernel does not end with its
// write-back -- MI355X_MICROARCH.md "boundary": + B / 6 TB/s for B dirty bytes)
#ifndef PBA_REC_SC1
#define PBA_REC_SC1 1
#endif
#ifndef PBA_PARTIAL_SC1
#define PBA_PARTIAL_SC1 1
#endif
// Pair blocks inside a Schur partial: 0 block-major [pair][36] (a thread's 36 stores are 288 B apart from its neighbour's),
// 1 entry-major [36][n_pairs] (one store instruction writes n_pairs consecutive doubles)
#ifndef PBA_PARTIAL_T
#define PBA_PARTIAL_T 1
#endif

namespace pba {

// Agent-scope relaxed 8-byte store / load (gfx950: write-through `sc1` store, L1-bypassing `sc1` load).  Used for the
// tiny per-workgroup partials that another workgroup of the SAME launch consumes: no release/acquire fence is
// needed (an agent-scope release fence is a whole-L2 write-back, ~us per workgroup).
__device__ __forceinline__ void store_agent(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Fixed-order (butterfly) wave reductions: no LDS, no barriers, reproducible.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
// N wave sums at once, level by level: the 2 N cross-lane moves of a level are in flight together (N calls of wave_sum
// leave the compiler with N dependent chains it only partly interleaves).  Same butterfly, same bits as wave_sum.
template <int N>
__device__ __forceinline__ void wave_sum_n(double (&v)[N]) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double t[N];
#pragma unroll
    for (int k = 0; k < N; ++k) t[k] = __shfl_xor(v[k], off);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += t[k];
  }
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}

// One lane's double for the whole wave (the lane index must be wave-uniform): lands in scalar registers.
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)__double2loint(v), src_lane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)__double2hiint(v), src_lane);
  return __hiloint2double((int)hi, (int)lo);
}

// Cooperative copy of the per-camera geometry table into LDS (8-byte words, coalesced): the per-lane gathers of
// ~50 doubles per observation then hit LDS instead of going through the vector memory path.
// AG: the table was written by another workgroup of the SAME launch (resident solve): agent-scope loads
template <int NT, bool AG = false>
__device__ __forceinline__ void stage_geom(const CamGeom* __restrict__ src, CamGeom* dst, int n_frames, int tid) {
  static_assert(sizeof(CamGeom) % 8 == 0, "CamGeom is copied as 8-byte words");
  const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
  unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
  const int n = n_frames * (int)(sizeof(CamGeom) / 8);
  // all loads of the thread in flight together (the rolled loop paid one L2 round trip per trip: 2 at 8 frames, 4 at 16)
  constexpr int IT = (kMaxFrames * (int)(sizeof(CamGeom) / 8) + NT - 1) / NT;
  unsigned long long v[IT];
#pragma unroll
  for (int u = 0; u < IT; ++u) {
    const int k = tid + u * NT;
    if (AG) v[u] = (k < n) ? __hip_atomic_load(s + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    else v[u] = (k < n) ? s[k] : 0ull;
  }
#pragma unroll
  for (int u = 0; u < IT; ++u) { const int k = tid + u * NT; if (k < n) d[k] = v[u]; }
}

// XCD-aware workgroup order (MI355X: 8 XCDs with private 4 MiB L2s; the dispatcher places workgroup b on XCD b % 8).
// Returns the LOGICAL workgroup index for physical index b so that each XCD works on one contiguous eighth of the
// logical range: observations are point-major and points are spatially sorted, so an XCD's L2 then only has to hold
// its own band of every frame instead of all of it.  Placement is a speed hint only; any mapping is correct.
__device__ __forceinline__ int xcd_logical_block(int b, int grid) {
  const int q = grid >> 3, r = grid & 7;
  const int x = b & 7, i = b >> 3;
  return x * q + (x < r ? x : r) + i;
}

// Device time stamps of the PIPELINED iteration (pba_set_profiling(e, 2)): the three kernels of an LM iteration run back
// to back, so the interval between the END of one and the END of the next is that kernel's share of the iteration
// (launch latency included) -- without the event records that make a kernel end with a cache write-back.  Every kernel
// only STORES the 100 MHz counter into the iteration's record (fire and forget: nothing is loaded or awaited on the
// critical path); the host forms the intervals afterwards.  Record of one iteration (u64):
//   [0] end of the sampling kernel (its last workgroup, after the step finalisation)   [1] unused
//   [2] end of the reduced solve (the solving workgroup)   [3 + b] end of k_schur's workgroup b (the host takes the maximum)
enum StampSlot { kStampEndSample = 0, kStampEndSolve = 2, kStampSchur0 = 3, kStampSchurBlocks = 1024, kStampRecord = 3 + 1024, kStampMaxIters = 256 };

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it waits for every
// outstanding global STORE of the wave (~1 us round trip) although nothing after the barrier depends on it.
// Global LOADS stay correct: the compiler still waits for a load's result before its first use.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Orders the LDS traffic of ONE wave (its LDS operations execute in order; this only stops the compiler from moving
// them): used where a single wave exchanges data with itself through LDS.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// =====================================================================================================
// frames
// =====================================================================================================
// One thread per pixel; 4 pixels per thread along the row would be the next step if this ever mattered
// (once per frame, 466k pixels).  imgproc.cc:27-95 semantics, packed as described in pba_device.h.
__global__ void k_pack_frame(const uint8_t* __restrict__ img, uint32_t* __restrict__ tex, int rows, int cols) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= cols) return;
  const size_t i = (size_t)y * cols + x;
  const int I = img[i];
  int gx2 = 0, gy2 = 0;
  if (y >= 1 && y < rows - 1 && x >= 1 && x < cols - 1) {
    gx2 = (int)img[i + 1] - (int)img[i - 1];
    gy2 = (int)img[i + cols] - (int)img[i - cols];
  }
  tex[i] = pack_texel(I, gx2, gy2);
}

__device__ __forceinline__ float tex_I(uint32_t t) { return (float)(t & 0xffu); }
__device__ __forceinline__ float tex_gx2(uint32_t t) { return (float)(((int32_t)(t << 14)) >> 22); }
__device__ __forceinline__ float tex_gy2(uint32_t t) { return (float)(((int32_t)(t << 4)) >> 22); }

__global__ void k_unpack_frame(const uint32_t* __restrict__ tex, float* I, float* Gx, float* Gy, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t t = tex[i];
  I[i] = tex_I(t);
  Gx[i] = 0.5f * tex_gx2(t);
  Gy[i] = 0.5f * tex_gy2(t);
}

// fp64 reciprocal / reciprocal square root from the hardware estimate + two Newton steps (~1 ulp): the IEEE division
// and sqrt sequences are ~10x longer and sit on latency-critical paths of the solve kernels.  Not used where the
// reference's rounding matters (pixel coordinates).
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}

// =====================================================================================================
// camera geometry
// =====================================================================================================
// Sine and cosine of the rotation angle.  Window cameras rotate by a fraction of a radian, and for |x| <= pi / 4 the
// library entry points spend most of their ~2 x 150 instructions on an argument reduction that is the identity there:
// this is the reduced-range core alone (the fdlibm __kernel_sin / __kernel_cos minimax polynomials, error < 1 ulp),
// with the library as the fallback for larger angles.  The candidate-camera geometry sits on the serial stretch of every
// LM iteration (solve epilogue).
__device__ __forceinline__ void sincos_angle(double x, double& sn, double& cs) {
  const double ax = fabs(x);
  if (ax <= 0.78539816339744828) {
    const double z = x * x;
    const double v = z * x;
    const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    sn = x + v * (-1.66666666666666324348e-01 + z * rs);
    const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    if (ax < 0.3) {
      cs = 1.0 - (0.5 * z - z * rc);
    } else {
      // qx ~ |x| / 4 with a zeroed low word, so that 1 - qx and 0.5 z - qx are exact
      const double qx = (ax > 0.78125) ? 0.28125 : __hiloint2double(__double2hiint(ax) - 0x00200000, 0);
      const double hz = 0.5 * z - qx;
      const double a = 1.0 - qx;
      cs = a - (hz - z * rc);
    }
  } else {
    sn = sin(x);
    cs = cos(x);
  }
}

__device__ inline void cam_geom_one(const double* __restrict__ cams, CamGeom* __restrict__ geom, int c, int fixed_slot) {
  CamGeom g;
  const double* p = cams + 6 * c;
  for (int k = 0; k < 3; ++k) { g.aa[k] = p[k]; g.t[k] = p[3 + k]; }
  const double wx = p[0], wy = p[1], wz = p[2];
  const double theta2 = wx * wx + wy * wy + wz * wz;
  g.rodrigues = theta2 > DBL_EPSILON;
  g.is_free = (c != fixed_slot);
  g.free_index = (c == fixed_slot) ? -1 : (fixed_slot >= 0 && c > fixed_slot ? c - 1 : c);
  g.pad = 0;
  if (g.rodrigues) {
    const double theta = sqrt(theta2);
    double ct, st;
    sincos_angle(theta, st, ct);
    const double ti = 1.0 / theta;
    const double ax = wx * ti, ay = wy * ti, az = wz * ti;
    g.w[0] = ax; g.w[1] = ay; g.w[2] = az; g.ct = ct; g.st = st;
    const double oc = 1.0 - ct;
    double R[9] = {ct + ax * ax * oc,      ax * ay * oc - az * st, ay * st + ax * az * oc,
                   az * st + ax * ay * oc, ct + ay * ay * oc,      -ax * st + ay * az * oc,
                   -ay * st + ax * az * oc, ax * st + ay * az * oc, ct + az * az * oc};
    for (int k = 0; k < 9; ++k) g.R[k] = R[k];
    // B = (w w^T + (R^T - I) [w]x) / theta^2   (Gallego & Yezzi 2015);  dR_k = R [B_k]x
    const double W[3] = {wx, wy, wz};
    const double Wx[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double B[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double acc = W[i] * W[j];
        for (int k = 0; k < 3; ++k) acc += (R[3 * k + i] - (i == k ? 1.0 : 0.0)) * Wx[3 * k + j];
        B[3 * i + j] = acc / theta2;
      }
    for (int k = 0; k < 3; ++k) {
      const double b0 = B[k], b1 = B[3 + k], b2 = B[6 + k];   // column k of B
      const double Bx[9] = {0, -b2, b1, b2, 0, -b0, -b1, b0, 0};
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double acc = 0;
          for (int m = 0; m < 3; ++m) acc += R[3 * i + m] * Bx[3 * m + j];
          g.dR[9 * k + 3 * i + j] = acc;
        }
    }
  } else {
    g.w[0] = g.w[1] = g.w[2] = 0.0; g.ct = 1.0; g.st = 0.0;
    const double R[9] = {1, -wz, wy, wz, 1, -wx, -wy, wx, 1};
    for (int k = 0; k < 9; ++k) g.R[k] = R[k];
    for (int k = 0; k < 3; ++k) {
      const double e0 = (k == 0), e1 = (k == 1), e2 = (k == 2);
      const double Ex[9] = {0, -e2, e1, e2, 0, -e0, -e1, e0, 0};
      for (int m = 0; m < 9; ++m) g.dR[9 * k + m] = Ex[m];
    }
  }
  geom[c] = g;
}

__global__ void k_cam_geom(const double* __restrict__ cams, CamGeom* __restrict__ geom, int n_frames, int fixed_slot) {
  const int c = threadIdx.x;
  if (c < n_frames) cam_geom_one(cams, geom, c, fixed_slot);
}

// xw = R(aa) X + t in the operation order of ceres::AngleAxisRotatePoint, then Calibration::project.
// Contraction is off so that (u, v) round exactly like the (FMA-free) reference build.
__device__ __forceinline__ void transform_point(const CamGeom& g, const double X[3], double xw[3]) {
#pragma clang fp contract(off)
  if (g.rodrigues) {
    const double w0 = g.w[0], w1 = g.w[1], w2 = g.w[2];
    const double c0 = w1 * X[2] - w2 * X[1];
    const double c1 = w2 * X[0] - w0 * X[2];
    const double c2 = w0 * X[1] - w1 * X[0];
    const double tmp = (w0 * X[0] + w1 * X[1] + w2 * X[2]) * (1.0 - g.ct);
    xw[0] = X[0] * g.ct + c0 * g.st + w0 * tmp;
    xw[1] = X[1] * g.ct + c1 * g.st + w1 * tmp;
    xw[2] = X[2] * g.ct + c2 * g.st + w2 * tmp;
  } else {
    xw[0] = X[0] + (g.aa[1] * X[2] - g.aa[2] * X[1]);
    xw[1] = X[1] + (g.aa[2] * X[0] - g.aa[0] * X[2]);
    xw[2] = X[2] + (g.aa[0] * X[1] - g.aa[1] * X[0]);
  }
  xw[0] += g.t[0];
  xw[1] += g.t[1];
  xw[2] += g.t[2];
}

__device__ __forceinline__ void project_point(const double xw[3], double fx, double fy, double cx, double cy,
                                              double& u, double& v) {
#pragma clang fp contract(off)
  u = ((xw[0] * fx) / xw[2]) + cx;
  v = ((xw[1] * fy) / xw[2]) + cy;
}

// Analytic 2x6 / 2x3 Jacobians of (u, v) w.r.t. camera [w, t] and point.
__device__ __forceinline__ void projection_jacobians(const CamGeom& g, const double X[3], const double xw[3],
                                                     double fx, double fy, double Ac[2][6], double Ap[2][3]) {
  const double iz = fast_rcp(xw[2]);
  const double ju0 = fx * iz, ju2 = -fx * xw[0] * iz * iz;
  const double jv1 = fy * iz, jv2 = -fy * xw[1] * iz * iz;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double* D = g.dR + 9 * k;
    const double d0 = D[0] * X[0] + D[1] * X[1] + D[2] * X[2];
    const double d1 = D[3] * X[0] + D[4] * X[1] + D[5] * X[2];
    const double d2 = D[6] * X[0] + D[7] * X[1] + D[8] * X[2];
    Ac[0][k] = ju0 * d0 + ju2 * d2;
    Ac[1][k] = jv1 * d1 + jv2 * d2;
    Ap[0][k] = ju0 * g.R[k] + ju2 * g.R[6 + k];
    Ap[1][k] = jv1 * g.R[3 + k] + jv2 * g.R[6 + k];
  }
  Ac[0][3] = ju0; Ac[0][4] = 0.0; Ac[0][5] = ju2;
  Ac[1][3] = 0.0; Ac[1][4] = jv1; Ac[1][5] = jv2;
}


// Point parameterisation.  rays == nullptr: the reference's free world point (photobundle.cc:692, :795: the three
// parameters ARE X).  rays != nullptr: the inverse-depth variant named by the north star (no reference counterpart,
// never parity-graded): the point lives on the fixed world ray rays[pt] = {origin o, direction d} and its parameters are
// (rho, 0, 0), X = o + d / rho.  Only the first parameter is free: d(u, v)/d(params) = [Ap (-d / rho^2) | 0 | 0], so the
// 3x3 point blocks degenerate to the scalar rho block plus two parameters that nothing ever moves (zero gradient, zero
// coupling; their LM diagonal is the min_lm_diagonal floor) -- every kernel downstream runs unchanged.
__device__ __forceinline__ void point_world(const double* __restrict__ rays, int pt, const double prm[3], double X[3], double q[3]) {
  if (!rays) {
    X[0] = prm[0]; X[1] = prm[1]; X[2] = prm[2];
    q[0] = q[1] = q[2] = 0.0;
  } else {
    const double* r = rays + 6 * (size_t)pt;
    const double inv = 1.0 / prm[0];
#pragma unroll
    for (int k = 0; k < 3; ++k) { X[k] = r[k] + r[3 + k] * inv; q[k] = -r[3 + k] * inv * inv; }
  }
}
__device__ __forceinline__ void point_jacobian(const double* __restrict__ rays, const double q[3], double Ap[2][3]) {
  if (rays) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double a = Ap[r][0] * q[0] + Ap[r][1] * q[1] + Ap[r][2] * q[2];
      Ap[r][0] = a; Ap[r][1] = 0.0; Ap[r][2] = 0.0;
    }
  }
}

// =====================================================================================================
// device-side trust-region bookkeeping (asynchronous driver): the decisions of pba_lm.cpp for ONE step
// =====================================================================================================
__device__ __forceinline__ void lm_step_rejected(LmState* st) {   // LevenbergMarquardtStrategy::StepRejected
  st->radius = st->radius / st->decrease_factor;
  st->decrease_factor *= 2.0;
}

__device__ inline void lm_log(LmState* st, pba_iteration_summary* log, int max_log, const pba_iteration_summary& it) {
  if (st->n_log < max_log) log[st->n_log] = it;
  st->n_log++;
}

// `s` = the step's (fully reduced) scalar block.  grad_only: only the gradient norms of the current point are valid.
// Gate of the final (gradient-only) pass, evaluated on the device so that the host can enqueue that pass without first
// reading the state back: it is needed for iteration zero of a zero-iteration solve and when the iteration limit was
// reached right after an accepted step (whose gradient norms are still to be reported).
__device__ __forceinline__ bool lm_final_pass_needed(const LmState* st) {
  return st->first || (st->pending_grad >= 0 && (st->done == kLmRunning || st->done == kLmMaxIterations));
}

__device__ inline void lm_decide(LmState* st, const double* s, pba_iteration_summary* log, int max_log, int grad_only) {
  const double gmax = fmax(s[kGmaxPts], s[kGmaxCams]);
  const double gnorm = sqrt(s[kGnorm2Pts] + s[kGnorm2Cams]);
  if (st->done) {
    // final pass after the iteration limit: only report the gradient norms of the last accepted point
    if (grad_only && st->done == kLmMaxIterations && st->pending_grad >= 0 && !st->first) {
      if (st->pending_grad < max_log) { log[st->pending_grad].gradient_max_norm = gmax; log[st->pending_grad].gradient_norm = gnorm; }
      st->pending_grad = -1;
    }
    return;
  }
  pba_iteration_summary it;
  memset(&it, 0, sizeof(it));
  it.eta = 1e-1;
  if (st->first) {
    // IterationZero
    if (s[kEvalFailLin] > 0.5) { st->done = kLmEvalFailure; return; }
    st->first = 0;
    st->x_cost = s[kCostLin];
    st->initial_cost = st->x_cost;
    st->minimum_cost = st->x_cost;
    it.iteration = 0; it.cost = st->x_cost; it.gradient_max_norm = gmax; it.gradient_norm = gnorm;
    it.step_is_valid = 1; it.step_is_successful = 1; it.trust_region_radius = st->radius;
    st->num_successful = 1;
    lm_log(st, log, max_log, it);
    if (0 >= st->max_num_iterations) st->done = kLmMaxIterations;
    else if (gmax <= st->gradient_tolerance) { st->done = kLmGradientTolerance; st->last_value[0] = gmax; }
    else if (st->radius <= st->min_radius) st->done = kLmMinRadius;
    if (st->done) return;
    // the record of iteration zero is out; iteration one starts from a clean one (its step_is_successful / step_is_valid
    // flags used to survive into a REJECTED first step's log entry)
    memset(&it, 0, sizeof(it));
    it.eta = 1e-1;
  } else if (st->pending_grad >= 0) {
    // gradient norms of the point accepted by the previous iteration + its deferred termination checks
    if (st->pending_grad < max_log) { log[st->pending_grad].gradient_max_norm = gmax; log[st->pending_grad].gradient_norm = gnorm; }
    st->pending_grad = -1;
    if (s[kEvalFailLin] > 0.5) { st->done = kLmEvalFailure; return; }
    if (gmax <= st->gradient_tolerance) { st->done = kLmGradientTolerance; st->last_value[0] = gmax; return; }
    if (st->radius <= st->min_radius) { st->done = kLmMinRadius; return; }
  }
  if (grad_only) return;

  const int iteration = st->iteration + 1;
  it.iteration = iteration;
  it.gradient_max_norm = gmax; it.gradient_norm = gnorm;
  it.linear_solver_iterations = 1;
  it.model_cost_change = s[kMccPts] + s[kMccCams];
  const bool solver_ok = s[kSolveOk] > 0.5 && s[kSchurFail] < 0.5;
  const bool step_is_valid = solver_ok && it.model_cost_change > 0.0;
  bool successful = false;
  if (!step_is_valid) {
    // HandleInvalidStep
    st->num_invalid++;
    it.cost = st->x_cost;
    if (st->num_invalid >= st->max_invalid) {
      it.trust_region_radius = st->radius;
      lm_log(st, log, max_log, it);
      st->iteration = iteration;
      st->done = kLmInvalidSteps;
      return;
    }
    lm_step_rejected(st);
  } else {
    it.step_is_valid = 1;
    st->num_invalid = 0;
    const bool eval_ok = s[kEvalFailCand] < 0.5 && isfinite(s[kCandCost]);
    const double candidate_cost = eval_ok ? s[kCandCost] : DBL_MAX;
    it.candidate_cost = candidate_cost;
    it.step_norm = sqrt(s[kStep2Pts] + s[kStep2Cams]);
    const double x_norm = sqrt(s[kX2Pts] + s[kX2Cams]);
    if (it.step_norm <= st->parameter_tolerance * (x_norm + st->parameter_tolerance)) {   // ParameterToleranceReached
      st->done = kLmParameterTolerance;
      st->last_value[0] = it.step_norm / (x_norm + st->parameter_tolerance);
      return;
    }
    it.cost_change = st->x_cost - candidate_cost;
    if (fabs(it.cost_change) <= st->function_tolerance * st->x_cost) {                    // FunctionToleranceReached
      st->done = kLmFunctionTolerance;
      st->last_value[0] = fabs(it.cost_change) / st->x_cost;
      return;
    }
    it.relative_decrease = it.cost_change / it.model_cost_change;
    if (it.relative_decrease > st->min_relative_decrease) {
      // HandleSuccessfulStep + LevenbergMarquardtStrategy::StepAccepted
      successful = true;
      st->cur ^= 1;
      st->x_cost = candidate_cost;
      const double t = 2.0 * it.relative_decrease - 1.0;
      st->radius = st->radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
      st->radius = fmin(st->max_radius, st->radius);
      st->decrease_factor = 2.0;
      it.step_is_successful = 1;
      it.cost = st->x_cost;
      st->pending_grad = st->n_log;
    } else {
      lm_step_rejected(st);
      it.cost = candidate_cost;
    }
  }
  // FinalizeIterationAndCheckIfMinimizerCanContinue (gradient tolerance deferred to the next decision)
  if (successful) { st->num_successful++; st->minimum_cost = st->x_cost; }
  else st->num_unsuccessful++;
  it.trust_region_radius = st->radius;
  lm_log(st, log, max_log, it);
  st->iteration = iteration;
  if (iteration >= st->max_num_iterations) st->done = kLmMaxIterations;
  else if (!successful && st->radius <= st->min_radius) st->done = kLmMinRadius;
}

// Publishes state + scalars to the host mirror, then the sequence number.
// Host-mapped (fine-grained, uncached) destinations: the stores go straight to the host, so waiting for their
// acknowledgement (vmcnt) orders them before the sequence number; a system-scope release fence would instead write
// back every dirty line of this XCD's L2 (megabytes of Jacobian records) from a single workgroup.
__device__ __forceinline__ void store_system_u32(void* dst, unsigned v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(dst), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void publish_seq(unsigned long long* host_seq, unsigned long long seq, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(host_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__device__ inline void publish_scal(const double* scal, double* host_scal, int tid, int nthreads) {
  const unsigned* src = reinterpret_cast<const unsigned*>(scal);
  for (int k = tid; k < 2 * kNumScal; k += nthreads) store_system_u32(reinterpret_cast<unsigned*>(host_scal) + k, src[k]);
}
__device__ inline void lm_publish(const LmState* st, LmState* host_state, const double* scal, double* host_scal,
                                  unsigned long long* host_seq, unsigned long long seq, int tid, int nthreads) {
  if (host_scal) publish_scal(scal, host_scal, tid, nthreads);
  if (host_state) {
    static_assert(sizeof(LmState) % 4 == 0, "word copy");
    const unsigned* src = reinterpret_cast<const unsigned*>(st);
    for (int k = tid; k < (int)(sizeof(LmState) / 4); k += nthreads) store_system_u32(reinterpret_cast<unsigned*>(host_state) + k, src[k]);
  }
  publish_seq(host_seq, seq, tid);
}

// ---- multi-rank exchange of the step scalars in ONE sum all-reduce -----------------------------------------------
// xchg = [ sum group (kSumBCount) | world x max group (kMaxCount) ]: every rank writes its max-group values into its
// own slot and zeros into the others, so the SUM all-reduce doubles as an all-gather; the max is taken afterwards.
// System-scope (write-through, cache-bypassing) 8-byte store / load: mailboxes that PEER devices read / that live on a peer.
__device__ __forceinline__ void store_system_f64(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double load_system_f64(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
// sys: the buffer is this rank's peer-exchange mailbox (pba_comm.h)
__device__ inline void xchg_pack(const double* scal, double* xchg, int rank, int world, int tid, int nthreads, bool sys = false) {
  for (int i = tid; i < kSumBCount; i += nthreads) { if (sys) store_system_f64(xchg + i, scal[kCandCost + i]); else xchg[i] = scal[kCandCost + i]; }
  for (int k = tid; k < kMaxCount * world; k += nthreads) {
    const int r = k / kMaxCount, j = k - r * kMaxCount;
    const double v = (r == rank) ? scal[kGmaxPts + j] : 0.0;
    if (sys) store_system_f64(xchg + kSumBCount + k, v); else xchg[kSumBCount + k] = v;
  }
}
__global__ void k_xchg_pack(const double* __restrict__ scal, double* __restrict__ xchg, int rank, int world, int sys) {
  xchg_pack(scal, xchg, rank, world, threadIdx.x, blockDim.x, sys != 0);
}

// ---- peer exchange: sum all-reduce of n doubles over the ranks' mailboxes, without a collective launch ----------------
// Every rank's producer kernel (earlier on the stream) has written its contribution into ITS OWN mailbox slot with
// system-scope stores.  This kernel raises the rank's flag for exchange `seq` (the producer has retired, so its stores
// have), waits until every peer's flag shows `seq`, and sums the contributions in RANK ORDER -- the same order on every
// rank, so all ranks end with bit-identical sums.  Spin-waits are bounded: after `timeout_ticks` of the 100 MHz counter the
// kernel reports through a host-mapped word and carries on (the host turns that into PBA_ERR_COMM).
struct PeerParams {
  const double* mb[8];         // every rank's mailbox as mapped into this process (own at [rank])
  double* own;
  int32_t world, rank;
};

// Peer exchange folded into a consumer kernel: raise nothing (the producer kernel's last workgroup raised this rank's flag
// when its stores had left), wait -- bounded -- until every rank's flag shows `seq`.  A timeout is reported through the
// host-mapped word AND poisons the result (the caller writes NaNs), so that a missed flag can never look like a step.
__device__ __forceinline__ bool peer_wait_all(const PeerParams& pp, int world, int flag_idx, unsigned long long seq,
                                              unsigned long long timeout_ticks, unsigned int* host_err, int tid) {
  __shared__ int s_peer_bad;
  if (tid == 0) s_peer_bad = 0;
  __syncthreads();
  if (tid < world) {
    // (select chain: a runtime index moves the whole kernel-parameter block to scratch)
    const double* mbq = tid == 1 ? pp.mb[1] : tid == 2 ? pp.mb[2] : tid == 3 ? pp.mb[3] : tid == 4 ? pp.mb[4] : tid == 5 ? pp.mb[5]
                      : tid == 6 ? pp.mb[6] : tid == 7 ? pp.mb[7] : pp.mb[0];
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(mbq) + flag_idx;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
        if (host_err) __hip_atomic_store(host_err, 1u + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_peer_bad = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  return s_peer_bad == 0;
}
__device__ __forceinline__ double peer_sum(const PeerParams& pp, int world, unsigned long long off, int idx) {
  double v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = (q < world) ? load_system_f64(pp.mb[q] + off + idx) : 0.0;
  double acc = v[0];
#pragma unroll
  for (int q = 1; q < 8; ++q) if (q < world) acc += v[q];      // rank order: the same bits on every rank
  return acc;
}

__global__ __launch_bounds__(256) void k_peer_allreduce(PeerParams pp, int flag_idx, unsigned long long data_off, int n,
                                                         unsigned long long seq, double* __restrict__ out,
                                                         unsigned long long timeout_ticks, unsigned int* host_err) {
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  // the producer kernel (earlier on this stream) has retired, so its system-scope stores have; release ordering on the flag
  if (blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(pp.own) + flag_idx, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if ((int)threadIdx.x < pp.world) {
    const int t = threadIdx.x;
    const double* mbq = t == 1 ? pp.mb[1] : t == 2 ? pp.mb[2] : t == 3 ? pp.mb[3] : t == 4 ? pp.mb[4] : t == 5 ? pp.mb[5]
                      : t == 6 ? pp.mb[6] : t == 7 ? pp.mb[7] : pp.mb[0];
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(mbq) + flag_idx;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
        if (host_err) __hip_atomic_store(host_err, 1u + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_bad = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
  }
  __syncthreads();
  // a peer that never showed up: the sums are poisoned (NaN), so that a missed error word can never look like a step
  const bool bad = s_bad != 0;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (q < pp.world) ? load_system_f64(pp.mb[q] + data_off + e) : 0.0;
    double acc = v[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) if (q < pp.world) acc += v[q];
    out[e] = bad ? __longlong_as_double(0x7ff8000000000000ll) : acc;
  }
}
__device__ inline void xchg_unpack(const double* xchg, double* scal, int world) {   // one thread
  for (int i = 0; i < kSumBCount; ++i) scal[kCandCost + i] = xchg[i];
  for (int j = 0; j < kMaxCount; ++j) {
    double m = xchg[kSumBCount + j];
    for (int r = 1; r < world; ++r) m = fmax(m, xchg[kSumBCount + kMaxCount * r + j]);
    scal[kGmaxPts + j] = m;
  }
}

struct DecideParams {
  LmState* lm; LmState* host_state;
  double* scal; double* host_scal;
  const double* xchg; int32_t world;
  pba_iteration_summary* log; int32_t max_log; int32_t grad_only;
  unsigned long long* host_seq; unsigned long long seq;
  // peer exchange of the step scalars folded into this kernel (peer_world > 0): the producer -- the last workgroup of the
  // fused sampling kernel -- wrote this rank's contribution into its mailbox slot and raised the rank's flag; here every
  // rank's flag is awaited and the slots are summed in rank order (no exchange kernel in between)
  PeerParams peer; int32_t peer_world, peer_flag;
  unsigned long long peer_off, peer_seq, peer_timeout;
  unsigned int* peer_err;
};

__global__ void k_decide(DecideParams p) {
  __shared__ double s_x[kSumBCount + kMaxCount * 8];
  const double* xchg = p.xchg;
  if (p.peer_world > 0) {
    // (a terminated solve: every rank's producer was a no-op and raised nothing -- nothing to wait for)
    if (p.lm->done) return;
    const bool ok = peer_wait_all(p.peer, p.peer_world, p.peer_flag, p.peer_seq, p.peer_timeout, p.peer_err, threadIdx.x);
    const int n = kSumBCount + kMaxCount * p.peer_world;
    for (int e = threadIdx.x; e < n; e += blockDim.x) s_x[e] = ok ? peer_sum(p.peer, p.peer_world, p.peer_off, e) : __longlong_as_double(0x7ff8000000000000ll);
    __syncthreads();
    xchg = s_x;
  }
  if (threadIdx.x == 0) {
    if (xchg) xchg_unpack(xchg, p.scal, p.world);
    lm_decide(p.lm, p.scal, p.log, p.max_log, p.grad_only);
    if (p.lm->done && p.lm->done_seq == 0) p.lm->done_seq = p.seq;
  }
  __syncthreads();
  if (p.host_seq) lm_publish(p.lm, p.host_state, p.scal, p.host_scal, p.host_seq, p.seq, threadIdx.x, blockDim.x);   // null: deferred to k_schur / k_flush
}

// =====================================================================================================
// sampling (sample_eigen.h) -- exact mixed fp32/fp64 arithmetic of the reference
// =====================================================================================================
__device__ __forceinline__ int trunc_x86(float x) {
  // static_cast<int>(float) with the x86 cvttss2si convention for NaN / out-of-range (INT_MIN)
  return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : INT_MIN;
}

// sample_eigen.h:34-52
__device__ __forceinline__ void linear_init_axis(float x, int size, int& x1, int& x2, float& dx) {
  const int ix = trunc_x86(x);
  if (ix < 0) { x1 = 0; x2 = 0; dx = 1.0f; }
  else if (ix > size - 2) { x1 = size - 1; x2 = size - 1; dx = 1.0f; }
  else { x1 = ix; x2 = ix + 1; dx = __fsub_rn((float)x2, x); }
}

// sample_eigen.h:82-83: dx*a11 is a float product; (1.0 - dx) and everything downstream is double.
// (1.0 - dx) * a2 is EXACT in double (dx: 24-bit significand, a2: integer of <= 9 bits), so one fma gives the same
// bits as the reference's separate multiply and add.
__device__ __forceinline__ double hlerp_exact(float dx, double omdx, float a1, float a2) {
  return __fma_rn(omdx, (double)a2, (double)__fmul_rn(dx, a1));
}
// The same holds for the FLOAT channel images of the multi-channel descriptors (a2 with 24 significant bits): dx comes out
// of LinearInitAxis as x2 - x (or 1), a multiple of 2^-24 in (0, 1], so (1.0 - dx) has at most 24 significant bits and
// the product at most 48 -- exact, one fma again gives the reference's bits.
//
// dy * top + (1 - dy) * bot: two double products, ONE double sum, then float (sample_eigen.h:82-83 assigned to the float
// sample; the reference is built without FMA instructions: CMakeLists.txt:25, -msse4.1).  Contraction must stay off here:
// fma(dy, top, omdy * bot) skips the rounding of the first product and flips the float result about once in 5e8 samples
// on float channel images (tests/golden/sampler_double_rounding.json holds such inputs).  The __dmul_rn / __dadd_rn
// intrinsics are plain operators in inlined bodies to this compiler and do NOT prevent the fusion; the pragma does.
__device__ __forceinline__ float vlerp_exact(float dy, float omdy, double top, double bot) {
#pragma clang fp contract(off)
  const double a = (double)dy * top;
  const double b = (double)omdy * bot;
  return (float)(a + b);
}
// One instruction less, for the regular walk over PACKED u8 frames only: there |texel| <= 255, and for a footprint whose
// origin has bx >= 4 or by >= 4 both products are EXACT in double, so the fused form returns the same bits.  Proof: for
// x in [2^k, 2^(k+1)) dx (and 1 - dx) are multiples of 2^(k-23) (k = -1 for x < 1), so top / bot -- sums of multiples
// of 2^(k-23) below 2^8 -- have at most 31 - k significant bits; dy and 1 - dy are multiples of 2^(j-23) in [0, 1] with
// at most 23 - j bits (y in [2^j, 2^(j+1))); a product needs at most 54 - k - j <= 53 bits once k + j >= 1, which
// max(k, j) >= 2 guarantees (k, j >= -1).  k_sample routes the 4 x 4 corner to the per-tap path (vlerp_exact).
__device__ __forceinline__ float vlerp_u8_interior(float dy, float omdy, double top, double bot) {
  return (float)fma((double)dy, top, (double)omdy * bot);
}

// Generic (irregular) tap: any position, straight from the packed frame in global memory.
template <bool JAC>
__device__ __forceinline__ void sample_generic(const uint32_t* __restrict__ frame, int rows, int cols, float yf,
                                               float xf, float& sI, float& sgx, float& sgy) {
  int x1, x2, y1, y2; float dx, dy;
  linear_init_axis(yf, rows, y1, y2, dy);
  linear_init_axis(xf, cols, x1, x2, dx);
  const uint32_t t11 = frame[(size_t)y1 * cols + x1], t12 = frame[(size_t)y1 * cols + x2];
  const uint32_t t21 = frame[(size_t)y2 * cols + x1], t22 = frame[(size_t)y2 * cols + x2];
  const double omdx = __dsub_rn(1.0, (double)dx);
  const float omdy = __fsub_rn(1.0f, dy);
  sI = vlerp_exact(dy, omdy, hlerp_exact(dx, omdx, tex_I(t11), tex_I(t12)), hlerp_exact(dx, omdx, tex_I(t21), tex_I(t22)));
  if (JAC) {
    // 2*G is blended and the exact power-of-two scale is applied at the end
    sgx = 0.5f * vlerp_exact(dy, omdy, hlerp_exact(dx, omdx, tex_gx2(t11), tex_gx2(t12)), hlerp_exact(dx, omdx, tex_gx2(t21), tex_gx2(t22)));
    sgy = 0.5f * vlerp_exact(dy, omdy, hlerp_exact(dx, omdx, tex_gy2(t11), tex_gy2(t12)), hlerp_exact(dx, omdx, tex_gy2(t21), tex_gy2(t22)));
  }
}

// debug / test: the engine's sampler at arbitrary float positions of one frame (pba_sample_frame)
__global__ void k_sample_probe(const uint32_t* __restrict__ frame, int rows, int cols, int n, const float* __restrict__ y,
                               const float* __restrict__ x, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sI, sgx, sgy;
  sample_generic<true>(frame, rows, cols, y[i], x[i], sI, sgx, sgy);
  out[3 * i] = sI; out[3 * i + 1] = sgx; out[3 * i + 2] = sgy;
}

struct SampleParams {
  const uint32_t* frames;     // [n_frames][rows*cols] packed texels
  const CamGeom* geom;        // [n_frames]
  const double* xyz;          // [n_points][3] point parameters (world XYZ, or (rho, 0, 0) in the inverse-depth variant)
  const double* rays;         // null, or [n_points][6] fixed world rays of the inverse-depth variant (point_world)
  const float* desc;          // [n_points][P]
  const double* w2;           // [P] squared patch weights
  const int32_t* obs_point;   // [n_obs]
  const uint8_t* obs_slot;    // [n_obs]
  double* rec;                // SoA [6][rec_stride]: rho'*M11, M12, M22, rho'*b1, b2, rho/2   (JAC only)
  double* block_cost;         // [gridDim.x] per-block cost partial
  int32_t* block_fail;        // [gridDim.x] non-finite flag
  int64_t rec_stride;
  int32_t n_obs, n_frames;
  int32_t rows, cols;
  double fx, fy, cx, cy;
  double huber;
  // ---- FUSED only: back-substitution of the step that leads to the point being sampled, and step finalisation ----
  const int4* tile_info;      // [n_tiles] whole-point tiles of <= 128 observations (shared with k_schur)
  const int2* lane_rec;       // [n_tiles][128] per tile lane: {point, slot | first lane of the point << 8 | observations of the point << 16}
                              // (indexed by tile and lane only: no dependency on the tile descriptor, one round trip less)
  const CamGeom* geom_prev;   // geometry at the CURRENT point (the linearisation the step was computed from)
  const double* xyz_prev;     // current points; `xyz` is then the OUTPUT (candidate points)
  const double* rec_prev;     // Jacobian-pass records of the current point
  const double* sp;           // [n_points][3]
  const double* ptrec;        // [n_points][12]
  const double* delta_c;      // [n_frames][6]
  double* block_bs;           // [gridDim.x][3] mcc, step^2, x^2 partials
  unsigned int* ticket;       // arrival counter (zero between launches)
  unsigned long long* stamp;  // null, or the device time-stamp block of pba_set_profiling(e, 2) (kStamp* below)
  double* scal;               // device scalar block
  double* host_scal;          // host-mapped copy (null: multi-rank, published later)
  unsigned long long* host_seq;
  unsigned long long seq;
  int32_t n_tiles;
  int32_t skip_backsub;       // FUSED kernel used as a plain (first) linearisation: no step to back-substitute
  // ---- asynchronous driver (lm != null): parity resolved on the device, decisions by the last workgroup ----
  LmState* lm;
  LmState* host_state;
  pba_iteration_summary* log;
  int32_t max_log;
  int32_t enq_cur;            // parity the host assumed when it filled xyz/xyz_prev, rec/rec_prev, geom/geom_prev
  double* block_cost_alt;     // the other parity's block arrays
  int32_t* block_fail_alt;
  int32_t decide;             // run lm_decide in the last workgroup (single rank)
  int32_t prec;               // 0: reference-exact sampler (default); 1: fp32 walk; 2: fp32 walk with bf16 operands (sweep)
  double* xchg;               // multi-rank: exchange buffer of the step scalars, packed by the last workgroup
  int32_t xchg_rank, xchg_world;
  int32_t xchg_sys;           // the exchange buffer is this rank's peer mailbox: system-scope stores
  unsigned long long* xchg_flag;   // null, or this rank's mailbox flag of the exchange: raised (xchg_seq) by the last workgroup
  unsigned long long xchg_seq;     // once its contribution has left, for the consumer kernel (k_decide) of every rank
  unsigned long long* dbg;    // optional [gridDim.x][8] per-phase cycle stamps of thread 0 (diagnostics)
  // first linearisation of an asynchronous solve: workgroup 0 copies the initial trust-region state from the host-mapped mirror
  // to the device (it was a launch of its own, k_lm_init, in front of every solve); null otherwise
  LmState* lm_init_dst; const LmState* lm_init_src;
  // resident solve (pba_resident.h): the workgroup's four block partials ALSO go out as one line of self-validating words -- eight u64
  // = epoch << 32 | one half of a double -- that the deciding workgroup polls directly: no drain of the stores in front of a flag, no flag,
  // no separate load of the partials behind it (three dependent round trips become one).  null on the three-kernel path.
  unsigned long long* res_tag; unsigned res_epoch;
};

// epoch-tagged halves of a double (the self-validating words of the resident solve's hand-overs)
__device__ __forceinline__ void store_tagged(unsigned long long* w, unsigned ep, double v) {
  const unsigned long long e = (unsigned long long)ep << 32;
  __hip_atomic_store(w, e | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(w + 1, e | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool load_tagged(const unsigned long long* w, unsigned ep, double& v) {
  const unsigned long long a = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __hiloint2double((int)(unsigned)a, (int)(unsigned)b);
  return (unsigned)(a >> 32) == ep && (unsigned)(b >> 32) == ep;
}

// ---- pieces of the FUSED sampling kernels (k_sample, k_sample_mc): the step that leads to the point being sampled -------
// Compact table of the PREVIOUS cameras for the back-substitution, per slot: R (9) | t (3) | Omega (9) | dt (3) |
// free index (1): with Omega_a = sum_k dw_k dR_k (the step's rotation part applied to the stored derivative matrices)
// the camera step enters every observation as  Ac dc = dpi (Omega_a X + dt_a)  and the point side as
// Ap^T u = R^T (dpi^T u): ~60 fp64 operations per observation instead of the full 2x6 / 2x3 Jacobians
constexpr int kBk = 26;
static_assert(kBk * sizeof(double) <= sizeof(CamGeom), "the compact table fits the second camera-table slot");

__device__ __forceinline__ bool fused_resolve_parity(SampleParams& p) {
  if (p.lm) {
    if (p.lm->done) return false;
    if (p.lm->cur != p.enq_cur) {     // the host's parity guess was off by an odd number of accepted steps
      const double* tx = p.xyz; p.xyz = p.xyz_prev; p.xyz_prev = tx;
      double* tr = p.rec; p.rec = const_cast<double*>(p.rec_prev); p.rec_prev = tr;
      const CamGeom* tg = p.geom; p.geom = p.geom_prev; p.geom_prev = tg;
      double* tc = p.block_cost; p.block_cost = p.block_cost_alt; p.block_cost_alt = tc;
      int32_t* tf = p.block_fail; p.block_fail = p.block_fail_alt; p.block_fail_alt = tf;
    }
  }
  return true;
}

// whole-point tiles of <= 128 observations, two per 256-thread workgroup
struct FusedIdx { int4 ti; int pt, slot, l0, cnt; };
template <int NT>
__device__ __forceinline__ FusedIdx fused_prefetch_indices(const SampleParams& p, int bid, const int tix) {
  FusedIdx f{make_int4(0, 0, 0, 0), 0, 0, 0, 0};
  const int tile = bid * (NT / 128) + (int)(tix >> 7);
  const int lt = tix & 127;
  if (tile < p.n_tiles) {
    f.ti = p.tile_info[tile];
    const int2 r = p.lane_rec[(size_t)tile * 128 + lt];      // (lanes beyond the tile's observations hold zeros)
    f.pt = r.x;
    f.slot = r.y & 0xff; f.l0 = (r.y >> 8) & 0xff; f.cnt = (r.y >> 16) & 0xff;
  }
  return f;
}

template <int NT, bool AG = false>
__device__ __forceinline__ void fused_stage_step_table(const SampleParams& p, double* s_bk, const int tix) {
  if (!p.skip_backsub) {
    // Branch-free: every entry issues the same seven loads (three table words, the rotation part of the camera step, one translation
    // component, the free index) and selects afterwards.  The five-way branch on the entry kind this replaces was divergent inside
    // every wave, so its global loads went out one kind after the other: four dependent L2 round trips per trip of this loop, 6.7 k
    // (8 frames) / 13 k (16 frames) cycles of every workgroup's start (in-kernel stamps).
    constexpr int GW = (int)(sizeof(CamGeom) / 8);
    constexpr int oT = (int)(offsetof(CamGeom, t) / 8), oR = (int)(offsetof(CamGeom, R) / 8), oD = (int)(offsetof(CamGeom, dR) / 8);
    constexpr int oF = (int)(offsetof(CamGeom, free_index) / 4);
    static_assert(offsetof(CamGeom, t) % 8 == 0 && offsetof(CamGeom, R) % 8 == 0 && offsetof(CamGeom, dR) % 8 == 0, "double words");
    const double* G = reinterpret_cast<const double*>(p.geom_prev);
    constexpr int IT = (kBk * kMaxFrames + NT - 1) / NT;
    const int n_e = kBk * p.n_frames;
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int e_raw = tix + u * NT;
      const int e = e_raw < n_e ? e_raw : 0;        // (idle lanes recompute entry 0: no exec-mask branch around the loads)
      const int a = e / kBk, k = e - a * kBk;
      const double* g = G + (size_t)a * GW;
      const double* dc = p.delta_c + 6 * a;
      const bool rot = (k >= 12) && (k < 21);
      const int j = rot ? k - 12 : 0;
      const int o0 = (k < 9) ? oR + k : ((k < 12) ? oT + (k - 9) : oD + j);
      // (AG: geometry and camera step were written by the solving workgroup of the SAME launch -- agent-scope loads)
      auto ld = [](const double* q) { return AG ? load_agent(q) : *q; };
      const double w0 = ld(g + o0), w1 = ld(g + oD + 9 + j), w2 = ld(g + oD + 18 + j);
      const double d0 = ld(dc), d1 = ld(dc + 1), d2 = ld(dc + 2);
      const double dt = ld(dc + ((k >= 21 && k < 24) ? 3 + (k - 21) : 3));
      const int32_t fidx = AG ? __hip_atomic_load(reinterpret_cast<const int32_t*>(g) + oF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              : reinterpret_cast<const int32_t*>(g)[oF];
      double val = rot ? d0 * w0 + d1 * w1 + d2 * w2 : w0;
      if (k >= 21) val = (k < 24) ? dt : (double)fidx;
      if (e_raw < n_e) s_bk[e] = val;
    }
  }
}

// Phase 0 of the fused kernels: back-substitution (SchurEliminator::BackSubstitute) for the workgroup's whole points; on
// return X is the CANDIDATE point parameters of the lane's observation.  s_bs: [NT][3] doubles of LDS scratch.
// RL = ResLane<R> of the resident solve (point, records, scale and P | g_p | D^2 come from it, the candidate goes back there), or
// `const void` on the three-kernel path (everything through global memory)
template <int NT, class RL = const void>
__device__ __forceinline__ void fused_backsub(const SampleParams& p, const double* rays, const FusedIdx& fi, const double* s_bk,
                                              double* s_bs, int& pt, int& slot, int& obs, bool& active, double (&X)[3],
                                              double& bs_mcc, double& bs_st2, double& bs_x2, const int tix, RL* rl = nullptr) {
  constexpr bool RES = !std::is_void<RL>::value;
  // delta_p = -P (g_p + sum_l W_l^T delta_c[slot_l]),  W_l^T delta_c = Ap^T M' (Ac delta_c)
  const int half = tix >> 7, lt = tix & 127;
  const int4 ti = fi.ti;
  active = lt < ti.y;
  obs = ti.x + lt;
  int l0 = 0, cnt = 0;
  double c3[3] = {0.0, 0.0, 0.0};
  double pr[12], spk[3] = {1.0, 1.0, 1.0};
#pragma unroll
  for (int k = 0; k < 12; ++k) pr[k] = 0.0;
  if (active) {
    pt = fi.pt;
    slot = fi.slot;
    l0 = fi.l0; cnt = fi.cnt;
    if constexpr (RES) {
      X[0] = rl->X[0]; X[1] = rl->X[1]; X[2] = rl->X[2];
      if (!p.skip_backsub) {
#pragma unroll
        for (int k = 0; k < 12; ++k) pr[k] = rl->pr[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) spk[k] = rl->sp[k];
      }
    } else {
    const double* xsrc = p.skip_backsub ? p.xyz : p.xyz_prev;
    X[0] = xsrc[3 * (size_t)pt]; X[1] = xsrc[3 * (size_t)pt + 1]; X[2] = xsrc[3 * (size_t)pt + 2];
    // issued here (same dependency level as X) so that they are in flight across the barrier below
    if (!p.skip_backsub) {
#pragma unroll
    for (int k = 0; k < 12; ++k) pr[k] = p.ptrec[12 * (size_t)pt + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) spk[k] = p.sp[3 * (size_t)pt + k];
    }
    }
    const double* bk = s_bk + kBk * slot;
    if (!p.skip_backsub && bk[24] >= 0.0) {
      // world point of the parameters (inverse-depth variant: X = o + d / rho, dX/drho = q)
      double Xw[3] = {X[0], X[1], X[2]}, qd[3] = {0.0, 0.0, 0.0};
      if (rays) point_world(rays, pt, X, Xw, qd);
      // xw = R X + t (plain order: this point only feeds the derivative of the projection, not a sampled position)
      const double xw0 = bk[0] * Xw[0] + bk[1] * Xw[1] + bk[2] * Xw[2] + bk[9];
      const double xw1 = bk[3] * Xw[0] + bk[4] * Xw[1] + bk[5] * Xw[2] + bk[10];
      const double xw2 = bk[6] * Xw[0] + bk[7] * Xw[1] + bk[8] * Xw[2] + bk[11];
      const double iz = fast_rcp(xw2);
      const double ju0 = p.fx * iz, ju2 = -p.fx * xw0 * iz * iz;
      const double jv1 = p.fy * iz, jv2 = -p.fy * xw1 * iz * iz;
      // Ac dc = dpi (Omega X + dt)
      const double s0 = bk[12] * Xw[0] + bk[13] * Xw[1] + bk[14] * Xw[2] + bk[21];
      const double s1 = bk[15] * Xw[0] + bk[16] * Xw[1] + bk[17] * Xw[2] + bk[22];
      const double s2 = bk[18] * Xw[0] + bk[19] * Xw[1] + bk[20] * Xw[2] + bk[23];
      const double t0 = ju0 * s0 + ju2 * s2, t1 = jv1 * s1 + jv2 * s2;
      double m0, m1, m2;
      if constexpr (RES) { m0 = rl->rec[0]; m1 = rl->rec[1]; m2 = rl->rec[2]; }
      else { m0 = p.rec_prev[0 * p.rec_stride + obs]; m1 = p.rec_prev[1 * p.rec_stride + obs]; m2 = p.rec_prev[2 * p.rec_stride + obs]; }
      const double u0 = m0 * t0 + m1 * t1, u1 = m1 * t0 + m2 * t1;
      // W_l^T dc = Ap^T u = R^T (dpi^T u)
      const double w0 = ju0 * u0, w1 = jv1 * u1, w2 = ju2 * u0 + jv2 * u1;
#pragma unroll
      for (int k = 0; k < 3; ++k) c3[k] = bk[k] * w0 + bk[3 + k] * w1 + bk[6 + k] * w2;
      if (rays) { c3[0] = qd[0] * c3[0] + qd[1] * c3[1] + qd[2] * c3[2]; c3[1] = 0.0; c3[2] = 0.0; }   // Ap -> Ap q (point_jacobian)
    }
  }
  s_bs[tix * 3 + 0] = c3[0]; s_bs[tix * 3 + 1] = c3[1]; s_bs[tix * 3 + 2] = c3[2];
  lds_barrier();
  if (active && !p.skip_backsub) {
    double acc[3] = {0.0, 0.0, 0.0};
    const double* src = s_bs + (half * 128 + l0) * 3;
    // four observations per trip, their twelve LDS reads in flight together (one dependent read per trip otherwise);
    // the adds keep the lane order
    for (int lb = 0; lb < cnt; lb += 4) {
      double x[4][3];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const int li = (lb + l < cnt) ? lb + l : cnt - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) x[l][k] = src[3 * li + k];
      }
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const bool in = lb + l < cnt;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += in ? x[l][k] : 0.0;
      }
    }
    const double q0 = pr[6] + acc[0], q1 = pr[7] + acc[1], q2 = pr[8] + acc[2];
    const double d[3] = {-(pr[0] * q0 + pr[1] * q1 + pr[2] * q2), -(pr[1] * q0 + pr[3] * q1 + pr[4] * q2),
                         -(pr[2] * q0 + pr[4] * q1 + pr[5] * q2)};
    const bool head = (lt == l0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (head) {
        const double sk = spk[k];
        const double yk = -d[k] * fast_rcp(sk);             // step in Jacobi-scaled coordinates is -y
        bs_mcc += 0.5 * yk * (sk * pr[6 + k]) + 0.5 * pr[9 + k] * yk * yk;
        bs_st2 += d[k] * d[k];
        bs_x2 += X[k] * X[k];
        if constexpr (!RES) const_cast<double*>(p.xyz)[3 * (size_t)pt + k] = X[k] + d[k];
      }
      X[k] = X[k] + d[k];
    }
  }
  if constexpr (RES) {      // (the candidate of the lane's point; a first linearisation samples at the current one)
    rl->Xc[0] = X[0]; rl->Xc[1] = X[1]; rl->Xc[2] = X[2];
  }
}

// Per-workgroup partials of the fused kernels: cost of the block and the point part of the step statistics, fixed-order
// sums (butterfly inside each wave, then the waves in order), write-through stores for the last workgroup of the launch.
// The three step statistics of a wave's points, reduced (same butterfly as the block cost) as soon as the back-substitution
// has produced them and kept in SCALAR registers from there on: as per-lane doubles they were six vector registers live across
// the whole walk.  PBA_STEP_SUMS_EARLY=0: reduced with the block cost at the end (rounds 2-4).
#ifndef PBA_STEP_SUMS_EARLY
#define PBA_STEP_SUMS_EARLY 1
#endif
__device__ __forceinline__ void fused_wave_step_sums(double& bs_mcc, double& bs_st2, double& bs_x2) {
  double q3[3] = {bs_mcc, bs_st2, bs_x2};
  wave_sum_n<3>(q3);
  bs_mcc = readlane_f64(q3[0], 0); bs_st2 = readlane_f64(q3[1], 0); bs_x2 = readlane_f64(q3[2], 0);
}
// bs_*: already summed over the wave (fused_wave_step_sums)
template <int WAVES>
__device__ __forceinline__ void fused_block_partials(const SampleParams& p, int bid, int lane, int wave, double cost_obs,
                                                     double bs_mcc, double bs_st2, double bs_x2, double* s_red, const int32_t& s_fail, const int tix) {
  double q4[4] = {wave_sum(cost_obs), bs_mcc, bs_st2, bs_x2};
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) s_red[q * WAVES + wave] = q4[q];
  }
  lds_barrier();
  double red_out[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double a = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) a += s_red[q * WAVES + w];
    red_out[q] = a;
  }
  if (tix == 0) {
    // consumed by the last workgroup of THIS launch: write-through stores; a non-finite block poisons its cost
    store_agent(p.block_cost + bid, s_fail ? __longlong_as_double(0x7ff8000000000000ll) : red_out[0]);
    store_agent(p.block_bs + 3 * bid, red_out[1]);
    store_agent(p.block_bs + 3 * bid + 1, red_out[2]);
    store_agent(p.block_bs + 3 * bid + 2, red_out[3]);
    __hip_atomic_store(p.block_fail + bid, (int32_t)s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p.res_tag) {
      unsigned long long* w = p.res_tag + 8 * (size_t)bid;
      store_tagged(w, p.res_epoch, s_fail ? __longlong_as_double(0x7ff8000000000000ll) : red_out[0]);
      store_tagged(w + 2, p.res_epoch, red_out[1]);
      store_tagged(w + 4, p.res_epoch, red_out[2]);
      store_tagged(w + 6, p.res_epoch, red_out[3]);
    }
  }
}

// Step finalisation by the last workgroup to arrive (agent-scope release / acquire around the ticket): fixed-order sums
// of all block partials, the trust-region decision (single rank), the exchange buffer (multi-rank), publication.
// s_f: [NTH] ints, s_r4: [4][NTH] doubles of LDS scratch that is free by now.
// Fixed-order sums of the per-workgroup partials of a fused sampling pass (by ONE workgroup of WAVES waves): s_r4[q * WAVES] = model cost
// change | step^2 | x^2 (point parts) | candidate cost, s_f[0] = non-finite flag.  Shared by fused_finalize and the deciding workgroup of
// the resident solve (pba_resident.h), so that both paths add the same numbers in the same order.
// AGL = false: the partials are already in this workgroup's LDS (resident solve: gathered from the tagged lines) -- plain loads.
template <int WAVES, bool AGL = true>
__device__ __forceinline__ void fused_sum_partials(const double* block_bs, const double* block_cost, const int n_blocks, const int lane, const int wave,
                                                   int* s_f, double* s_r4, unsigned long long* t_loaded, const int tix) {
  constexpr int NTH = WAVES * 64;
  auto load_agent = [](const double* q) { return AGL ? pba::load_agent(q) : *q; };
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0; int f = 0;
  // 8 blocks' partials in flight per thread (every load misses this XCD's L2), summed in block order
  for (int b0 = tix; b0 < n_blocks; b0 += 8 * NTH) {
    double v0[8], v1[8], v2[8], v3[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = b0 + k * NTH;
      const bool ok = b < n_blocks;
      v0[k] = ok ? load_agent(block_bs + 3 * b) : 0.0;
      v1[k] = ok ? load_agent(block_bs + 3 * b + 1) : 0.0;
      v2[k] = ok ? load_agent(block_bs + 3 * b + 2) : 0.0;
      v3[k] = ok ? load_agent(block_cost + b) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { a0 += v0[k]; a1 += v1[k]; a2 += v2[k]; a3 += v3[k]; f |= (v3[k] != v3[k]) ? 1 : 0; }
  }
  if (t_loaded) *t_loaded = __builtin_amdgcn_s_memrealtime();
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
  f = __any(f) ? 1 : 0;
  if (lane == 0) { s_r4[wave] = a0; s_r4[WAVES + wave] = a1; s_r4[2 * WAVES + wave] = a2; s_r4[3 * WAVES + wave] = a3; s_f[wave] = f; }
  __syncthreads();
  if (tix == 0) {
    for (int w = 1; w < WAVES; ++w) {
      s_r4[0] += s_r4[w]; s_r4[WAVES] += s_r4[WAVES + w]; s_r4[2 * WAVES] += s_r4[2 * WAVES + w]; s_r4[3 * WAVES] += s_r4[3 * WAVES + w];
      s_f[0] |= s_f[w];
    }
  }
  __syncthreads();
}

template <int WAVES>
__device__ __forceinline__ void fused_finalize(const SampleParams& p, int lane, int wave, int* s_f, double* s_r4, unsigned long long t_begin, const int tix) {
  constexpr int NTH = WAVES * 64;
  __shared__ int s_last;
  if (tix == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the write-through partials have left this CU
    const unsigned t = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (s_last) {
    const unsigned long long t_fin0 = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
    unsigned long long t_fin1 = 0;
    fused_sum_partials<WAVES>(p.block_bs, p.block_cost, (int)gridDim.x, lane, wave, s_f, s_r4, p.dbg ? &t_fin1 : nullptr, tix);
    unsigned long long t_fin2 = 0, t_fin3 = 0;
    if (tix == 0) {
      t_fin2 = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
      p.scal[kMccPts] = s_r4[0]; p.scal[kStep2Pts] = s_r4[WAVES]; p.scal[kX2Pts] = s_r4[2 * WAVES];
      p.scal[kCandCost] = s_r4[3 * WAVES]; p.scal[kEvalFailCand] = (double)s_f[0];
      *p.ticket = 0;
      if (p.lm && p.decide) lm_decide(p.lm, p.scal, p.log, p.max_log, 0);
      if (p.lm && p.decide && p.lm->done && p.lm->done_seq == 0) p.lm->done_seq = p.seq;
      t_fin3 = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
    }
    if (p.xchg) {
      __syncthreads();
      xchg_pack(p.scal, p.xchg, p.xchg_rank, p.xchg_world, tix, NTH, p.xchg_sys != 0);
      if (p.xchg_flag) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every thread's system-scope stores have left
        __syncthreads();
        if (tix == 0) __hip_atomic_store(p.xchg_flag, p.xchg_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (p.host_scal) {
      __syncthreads();
      lm_publish(p.lm, (p.lm && p.decide) ? p.host_state : nullptr, p.scal, p.host_scal, p.host_seq, p.seq, tix, NTH);
    }
    if (p.stamp && tix == 0) p.stamp[kStampEndSample] = __builtin_amdgcn_s_memrealtime();
    if (p.dbg && tix == 0) {
      p.dbg[(size_t)gridDim.x * 8] = __builtin_amdgcn_s_memrealtime() - t_fin0;
      p.dbg[(size_t)gridDim.x * 8 + 1] = t_fin0 - t_begin;
      p.dbg[(size_t)gridDim.x * 8 + 2] = t_fin1 - t_fin0;      // partial loads
      p.dbg[(size_t)gridDim.x * 8 + 3] = t_fin2 - t_fin1;      // block reduction
      p.dbg[(size_t)gridDim.x * 8 + 4] = t_fin3 - t_fin2;      // decision
    }
  }
}

// One LANE per observation (residual block); each wave stages the (2R+2)^2 texel footprints of its 64
// observations through LDS with cooperative row-segment loads (consecutive lanes read consecutive texels of a
// footprint row), then every lane walks its own patch with the separable form of the reference's bilinear rule:
// horizontal lerps H[r][j] are shared by the two pixel rows that touch footprint row r, which is exact because
// bot(i, j) and top(i+1, j) are the same expression (sample_eigen.h:82-83).
//   JAC = true : Jacobian pass. Emits per observation M = sum w^2 g g^T, b = sum w^2 g e (rho'-scaled) and rho/2.
//   JAC = false: cost pass (intensity only).
//   FUSED      : the workgroup first back-substitutes the step for its own (whole) points
//                (SchurEliminator::BackSubstitute), samples at the candidate it just formed, and the last workgroup
//                to finish reduces the per-block partials in a fixed order and publishes the step's scalar block.
//   FAST       : opt-in reduced-precision walk (SampleParams::prec = 1 | 2, unit weights only): NOT reference-exact.
// Rows of the footprints go through LDS in batches of RB rows (the walk only ever needs ONE footprint row at a time: the
// horizontal lerps of the previous row live in registers), so that a wave's LDS share stays ~10 KB at every patch radius
// and 11x11 patches run at the same occupancy and in the same fused form as 5x5 ones.
constexpr int sample_rows_per_batch(int R) { return R <= 2 ? 2 * R + 2 : (R == 4 ? 5 : 4); }
constexpr int sample_stage_groups(int ng, int per_group) {
  int best = 1;
  for (int g = 1; g <= ng; ++g) if (ng % g == 0 && g * per_group <= 40) best = g;
  return best;
}

// LDS of one sampling workgroup (k_sample; the sampling phase of the resident solve, pba_resident.h, aliases it with the other phases')
// texel-major LDS layout [t][lane]: the walk reads stride-1 across lanes; the odd stride keeps the staging stores of
// the vector row segments at the 2-way minimum (64 lanes on 32 banks) and four workgroups within 160 KB of LDS.
constexpr int kSampleLdsStride = 65;
struct SampleIrrRec { double u, v; int32_t wyx, pt, src, pad; };
constexpr int kSampleIrrCap = 32;
template <int R, int WAVES>
struct SampleSmem {
  static constexpr int F = 2 * R + 2;
  static constexpr int RB = sample_rows_per_batch(R);
  static constexpr int FF = RB * F;                      // texels of one batch
  static constexpr size_t kTexBytes = sizeof(uint32_t) * WAVES * FF * kSampleLdsStride;
  static constexpr size_t kPreBytes = sizeof(double) * 3 * WAVES * 64 + 2 * kMaxFrames * sizeof(CamGeom);
  static_assert(kTexBytes >= sizeof(double) * 4 * WAVES * 64, "the finalisation reuses the texel region");
  alignas(16) char raw[kTexBytes > kPreBytes ? kTexBytes : kPreBytes];
  alignas(8) int32_t bi[2][WAVES][64];
  // r4: the windowed irregular observations of the WHOLE workgroup go through one queue and are dealt round robin to its four
  // waves (the wave that owned three border patches used to hold the other three -- and their LDS -- for its per-tap passes)
  SampleIrrRec q[(RB == F) ? kSampleIrrCap : 1];
  int32_t icnt[WAVES];
  uint32_t win[(RB == F) ? 1 : WAVES][(RB == F) ? 1 : F * F];   // large radii: the clamped window of ONE irregular observation per wave
  double red[4 * WAVES];
  int32_t fail;
};
static_assert(sizeof(double) * 6 * kSampleIrrCap <= sizeof(int32_t) * 2 * 4 * 64, "the six sums of every queued patch fit s_base | s_irr");

// Per-lane state of the RESIDENT solve (pba_resident.h: one launch per solve, every workgroup owns a fixed pair of whole-point
// tiles): what the three-kernel path re-reads from global memory in every kernel of every iteration -- the observation's indices,
// its point, the Jacobian-pass records, the point's Jacobi scale / damped inverse / gradient, the descriptor -- stays in registers.
template <int R>
struct ResLane {
  int4 ti;                     // tile descriptor {first observation, observations, first point, points}
  int pt, slot, l0, cnt;       // lane_rec of the lane's observation
  double X[3], Xc[3];          // parameters of the lane's point: current | candidate
  double rec[6], recc[6];      // rho' M11, M12, M22, rho' b1, b2, rho / 2 at the current | candidate point
  double sp[3];                // Jacobi scale of the point's columns
  double pr[12];               // P (6) | g_p (3) | D_p^2 (3) of the point at the current linearisation
  float desc[(2 * R + 1) * (2 * R + 1)];
};

// The body of the sampling kernel for ONE 256-thread workgroup `bid` (see k_sample below for the template switches).
//   RES: phase of the resident solve -- indices, points, previous records and descriptors come from `rl`, the candidate point and
//        its records go back there; tables produced by other workgroups of the SAME launch are read with agent-scope loads; the
//        step finalisation (ticket, fixed-order reduction, decision) is the caller's.
template <int R, bool JAC, int WAVES, bool FUSED, bool UNITW, bool FAST, bool RES>
// tix = threadIdx.x (the resident solve passes it through an opaque copy per trip of its loop: with the plain builtin the compiler
// hoists every per-lane address and mask of every phase out of the loop -- hundreds of registers live across all of it)
__device__ __forceinline__ void sample_wg(SampleParams& p, SampleSmem<R, WAVES>& sm, ResLane<R>& rl, const int bid, const int n_blocks, const int tix, CamGeom* geom_lds = nullptr) {
  static_assert(!FUSED || (WAVES * 64) % 128 == 0, "fused tiles are 128 observations");
  static_assert(!FAST || UNITW, "the reduced-precision walk assumes unit patch weights");
  static_assert(!RES || (FUSED && !FAST), "the resident phase is the fused exact kernel");
  // inverse-depth variant (point_world): null for the reference's free world points
  const double* rays = p.rays;
  constexpr int W = 2 * R + 1;      // patch side
  constexpr int F = 2 * R + 2;      // footprint side
  constexpr int RB = sample_rows_per_batch(R);   // footprint rows staged per batch
  constexpr int NB = (F + RB - 1) / RB;
  constexpr int FF = RB * F;                      // texels of one batch
  constexpr int LSTRIDE = kSampleLdsStride;
  char* s_raw = sm.raw;
  uint32_t (*s_tex)[FF * LSTRIDE] = reinterpret_cast<uint32_t (*)[FF * LSTRIDE]>(s_raw);
  auto& s_bi = sm.bi;
  int32_t (*s_base)[64] = s_bi[0];
  int32_t (*s_irr)[64] = s_bi[1];        // (by0 << 16) | bx0: window anchor of the observations staged with clamped coordinates
  using IrrRec = SampleIrrRec;
  constexpr int kIrrCap = kSampleIrrCap;
  auto& s_q = sm.q;
  auto& s_icnt = sm.icnt;
  auto& s_win = sm.win;
  auto& s_red = sm.red;
  int32_t& s_fail = sm.fail;

  const int lane = tix & 63;
  const int wave = tix >> 6;
  int obs = bid * (WAVES * 64) + tix;
  bool active = obs < p.n_obs;
  if (tix == 0) s_fail = 0;

  // camera geometry tables -> LDS (the texel region is free until the staging phase)
  // (resident solve: the table of the point being sampled goes to the workgroup's PERSISTENT copy, which the next elimination reads)
  CamGeom* s_geom = RES ? geom_lds : reinterpret_cast<CamGeom*>(s_raw + sizeof(double) * 3 * WAVES * 64);
  double* s_bk = reinterpret_cast<double*>(reinterpret_cast<CamGeom*>(s_raw + sizeof(double) * 3 * WAVES * 64) + kMaxFrames);
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tl = p.dbg ? __builtin_amdgcn_s_memtime() : 0;
  const unsigned long long t_begin = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0;   // 100 MHz, device-wide
#define PBA_STK(k) do { if (p.dbg) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tk[k] += tn - tl; tl = tn; } } while (0)
  // fused form: the tile descriptor and the observation's indices do not depend on the tables -- requested first so
  // that their two dependent round trips overlap the table loads instead of following the barrier
  FusedIdx fi{make_int4(0, 0, 0, 0), 0, 0, 0, 0};
  if constexpr (RES) { fi.ti = rl.ti; fi.pt = rl.pt; fi.slot = rl.slot; fi.l0 = rl.l0; fi.cnt = rl.cnt; }
  else if (FUSED) fi = fused_prefetch_indices<WAVES * 64>(p, bid, tix);
  // (resident solve, step trips: p.geom == null -- the candidate table is in `geom_lds` already, polled there from the solve's tagged words,
  // and geom_prev / delta_c name LDS copies; its first linearisation stages the initial table from global memory like everybody else)
  if (!RES || p.geom) stage_geom<WAVES * 64, RES>(p.geom, s_geom, p.n_frames, tix);
  if (FUSED) fused_stage_step_table<WAVES * 64, false>(p, s_bk, tix);
  lds_barrier();
  PBA_STK(0);

  int pt = 0, slot = 0;
  double X[3] = {0.0, 0.0, 0.0};
  double bs_mcc = 0.0, bs_st2 = 0.0, bs_x2 = 0.0;
  if (FUSED) {
    // ---- phase 0: back-substitution for this workgroup's points (fused_backsub) ------------------------------
    if constexpr (RES) fused_backsub<WAVES * 64, ResLane<R>>(p, rays, fi, s_bk, reinterpret_cast<double*>(&s_tex[0][0]), pt, slot, obs, active, X, bs_mcc, bs_st2, bs_x2, tix, &rl);
    else fused_backsub<WAVES * 64>(p, rays, fi, s_bk, reinterpret_cast<double*>(&s_tex[0][0]), pt, slot, obs, active, X, bs_mcc, bs_st2, bs_x2, tix);
    if (PBA_STEP_SUMS_EARLY) fused_wave_step_sums(bs_mcc, bs_st2, bs_x2);
  } else if (active) {
    pt = p.obs_point[obs];
    slot = p.obs_slot[obs];
    X[0] = p.xyz[3 * (size_t)pt]; X[1] = p.xyz[3 * (size_t)pt + 1]; X[2] = p.xyz[3 * (size_t)pt + 2];
  }

  if (rays && active) {          // inverse-depth variant: parameters -> world point
    // a candidate with rho <= 0 (a step through the camera centre) mirrors the point behind the ray origin but stays
    // finite: flagged as an evaluation failure so that the trust-region loop rejects the step
    if (!(X[0] > 0.0)) atomicOr(&s_fail, 1);
    double Xw[3], qd[3];
    point_world(rays, pt, X, Xw, qd);
    X[0] = Xw[0]; X[1] = Xw[1]; X[2] = Xw[2];
  }
  PBA_STK(1);
  // ---- phase 1: geometry, one lane per observation (fp64) ------------------------------------------------
  double u = 0.0, v = 0.0;
  float xf[W], yf[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { xf[j] = 0.f; yf[j] = 0.f; }
  int bx = 0, by = 0;
  bool regular = false, reg_clamped = false;
  if (active) {
    double xw[3];
    transform_point(s_geom[slot], X, xw);
    project_point(xw, p.fx, p.fy, p.cx, p.cy, u, v);
    // photobundle.cc:715-717: v + T(y), u + T(x) in double, rounded to float in SampleWithDerivative (:117-118)
    bool reg = true;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      xf[j] = (float)(u + (double)(j - R));
      yf[j] = (float)(v + (double)(j - R));
    }
    bx = trunc_x86(xf[0]);
    by = trunc_x86(yf[0]);
    reg = (bx >= 0) && (by >= 0) && (FAST || bx >= 4 || by >= 4);   // (the 4 x 4 corner: see vlerp_u8_interior)
#pragma unroll
    for (int j = 1; j < W; ++j) reg = reg && (trunc_x86(xf[j]) == bx + j) && (trunc_x86(yf[j]) == by + j);
    const bool inside = (bx + W - 1 <= p.cols - 2) && (by + W - 1 <= p.rows - 2);
    regular = reg && inside;
    // Consecutive taps that hang over the RIGHT / BOTTOM border only: LinearInitAxis (sample_eigen.h:38-51) turns a tap
    // with ix > size - 2 into (size - 1, size - 1, d = 1), which is what the regular separable walk computes on a window
    // staged with clamped coordinates once the weights of those columns / rows are forced to 1 (both texels are the same
    // pixel, 1 a + 0 a is exact).  About half of the border observations; the rest (left / top: the truncation toward
    // zero maps two taps to pixel 0) keeps the per-tap pass.
    reg_clamped = !FAST && reg && !inside && bx < 65536 && by < 32768 && p.rows < 32768 && p.cols < 65536;
  }
  if (PBA_EXPERIMENT_SKIP_IRREGULAR == 2 && p.n_obs > 0) active = active && regular;   // timing experiment (WRONG results): paths compiled in, never taken
  // Irregular observations (patch over the image border, clamped taps: sample_eigen.h:38-51) whose taps all fall into
  // one F x F window of CLAMPED pixel coordinates anchored at the first tap (by0, bx0) are staged like the regular ones,
  // texel by texel with clamped addresses, and walked with per-tap indices out of LDS (kWindow: whole footprint resident,
  // i.e. patch radius <= 2).  Anything else -- float-rounding anomalies that stretch the window, images of 32k+ rows --
  // keeps the per-pixel path from global memory.  A wave runs the irregular code if ANY of its lanes needs it, so its
  // latency matters: ~240 of 6250 waves at BASELINE configs[1].
  constexpr bool kWindow = (RB == F);
  bool win_irr = false;
  int by0 = 0, bx0 = 0;
  if (reg_clamped) { by0 = by; bx0 = bx; }
  if (!FAST && active && !regular && !reg_clamped && p.rows < 32768 && p.cols < 65536) {
    int a1, a2, l1, l2; float dd;
    linear_init_axis(yf[0], p.rows, a1, a2, dd);
    linear_init_axis(yf[W - 1], p.rows, l1, l2, dd);
    by0 = a1;
    bool fits = (l2 - a1) <= W && l2 >= a1;
    linear_init_axis(xf[0], p.cols, a1, a2, dd);
    linear_init_axis(xf[W - 1], p.cols, l1, l2, dd);
    bx0 = a1;
    fits = fits && (l2 - a1) <= W && l2 >= a1;
    // monotone taps in between (NaN / wild coordinates fail here)
#pragma unroll
    for (int j = 1; j < W - 1; ++j) {
      int m1, m2;
      linear_init_axis(yf[j], p.rows, m1, m2, dd);
      fits = fits && m1 >= by0 && m2 <= by0 + W;
      linear_init_axis(xf[j], p.cols, m1, m2, dd);
      fits = fits && m1 >= bx0 && m2 <= bx0 + W;
    }
    win_irr = fits;
  }
  // >= 0: regular, linear texel index of the footprint origin;  -1: nothing to stage;  <= -2: windowed irregular, slot
  s_base[wave][lane] = (active && regular) ? (int32_t)(slot * (p.rows * p.cols) + by * p.cols + bx) : (((kWindow && win_irr) || reg_clamped) ? -2 - slot : -1);
  s_irr[wave][lane] = (by0 << 16) | bx0;
  const unsigned long long im_all = kWindow ? __ballot(win_irr) : 0ull;
  if (kWindow && lane == 0) s_icnt[wave] = __popcll(im_all);
  lds_barrier();
  // workgroup-uniform: how many windowed irregular observations the four waves hold together, and where this wave's start in the queue
  int q_off = 0, q_total = 0;
  if (kWindow) {
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { const int c = s_icnt[w]; q_off += (w < wave) ? c : 0; q_total += c; }
  }
  const bool share_irr = kWindow && q_total > 1 && q_total <= kIrrCap;
  int my_q = -1;
  if (share_irr && win_irr) {
    my_q = q_off + __popcll(im_all & ((1ull << lane) - 1ull));
    IrrRec rcd; rcd.u = u; rcd.v = v; rcd.wyx = (by0 << 16) | bx0; rcd.pt = pt; rcd.src = (wave << 8) | lane; rcd.pad = 0;
    s_q[my_q] = rcd;
  }
  PBA_STK(2);

  // ---- phases 2 + 3, per batch of RB footprint rows: cooperative staging global -> LDS, then the per-lane walk -----
  // Staging: NCH consecutive lanes fetch one footprint row of one observation as CW-texel vectors, so one load
  // instruction covers row r of OPI observations; the row offset r * cols is wave-uniform and the LDS destination of
  // texel (r, c) is a compile-time offset from a per-lane base, which leaves almost no address arithmetic per load.
  // All loads of a group batch are issued before the first LDS store so that the L2 latency is paid once per batch.
  // Walk: separable exact bilinear rule; the horizontal lerps of a footprint row are shared by the two pixel rows that
  // touch it.  Every wave stages its OWN 64 footprints and walks only those: no workgroup barrier, the waves drift apart.
  constexpr int CW = (F % 4 == 0) ? 4 : (F % 3 == 0 ? 3 : 2);
  constexpr int NCH = F / CW;
  constexpr int OPI = 64 / NCH;
  constexpr int NG = (64 + OPI - 1) / OPI;
  constexpr int GB = sample_stage_groups(NG, RB * CW);
  constexpr int NPL = JAC ? 3 : 1;
  double m11 = 0, m12 = 0, m22 = 0, b1 = 0, b2 = 0, cc = 0;
  const bool walk = active && (regular || reg_clamped);
  const float* p0 = p.desc + (size_t)pt * (W * W);
  // kLean (large patches): 1 - dx and dy are formed again where they are used instead of living in 33 registers
  constexpr bool kLean = (R >= 4);
  float dxs[W], dys[kLean ? 1 : W];
  double omdx[kLean ? 1 : W];
  double Hp[FAST ? 1 : NPL][W];
  float fHp[FAST ? NPL : 1][W];
  float fa11 = 0.f, fa12 = 0.f, fa22 = 0.f, fc1 = 0.f, fc2 = 0.f, fc0 = 0.f;
  const bool bf16 = FAST && p.prec == 2;
  auto rnd = [&](float v) {      // round-to-nearest-even to 8 significant bits (bf16), only in the sweep's third mode
    if (!bf16) return v;
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xffff0000u);
  };
  // columns / rows beyond these indices sit on the clamped border pixel (reg_clamped only)
  const int jx_clamp = reg_clamped ? p.cols - 2 - bx : W, jy_clamp = reg_clamped ? p.rows - 2 - by : W;
#pragma unroll
  for (int j = 0; j < W; ++j) {
    dxs[j] = (j > jx_clamp) ? 1.0f : __fsub_rn((float)(bx + j + 1), xf[j]);
    if (!kLean) {
      dys[j] = (j > jy_clamp) ? 1.0f : __fsub_rn((float)(by + j + 1), yf[j]);
      omdx[j] = __dsub_rn(1.0, (double)dxs[j]);
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int r0 = b * RB;
    const int nr = (F - r0 < RB) ? F - r0 : RB;
    {
      const int ch = lane % NCH, oi = lane / NCH;
      const char* fbytes = reinterpret_cast<const char*>(p.frames);
#pragma unroll 1
      for (int g0 = 0; g0 < NG; g0 += GB) {
        uint32_t tx[GB][RB][CW];
        int32_t bs[GB];
#pragma unroll
        for (int gg = 0; gg < GB; ++gg) {
          const int o = (g0 + gg) * OPI + oi;
          bs[gg] = (oi < OPI && o < 64) ? s_base[wave][o & 63] : -1;
          const uint32_t boff = ((uint32_t)bs[gg] + (uint32_t)(ch * CW)) * 4u;
#pragma unroll
          for (int rr = 0; rr < RB; ++rr) {
            if (rr >= nr) continue;
#pragma unroll
            for (int j = 0; j < CW; ++j) tx[gg][rr][j] = 0;
            if (bs[gg] >= 0) __builtin_memcpy(tx[gg][rr], fbytes + (boff + (uint32_t)((r0 + rr) * p.cols) * 4u), sizeof(uint32_t) * CW);
          }
          if (bs[gg] <= -2) {
            // clamped window (per-tap pass at R <= 2, right / bottom overhang at every radius): the same rows of the
            // F x F window, every texel at its clamped coordinates
            const int ir = s_irr[wave][o & 63];
            const int wy = ir >> 16, wx = ir & 0xffff;
            const uint32_t* fr = p.frames + (size_t)(-2 - bs[gg]) * p.rows * p.cols;
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
              if (rr >= nr) continue;
              const int row = min(wy + r0 + rr, p.rows - 1);
#pragma unroll
              for (int j = 0; j < CW; ++j) tx[gg][rr][j] = fr[(size_t)row * p.cols + min(wx + ch * CW + j, p.cols - 1)];
            }
          }
        }
#pragma unroll
        for (int gg = 0; gg < GB; ++gg) {
          const int o = (g0 + gg) * OPI + oi;
          if (bs[gg] != -1) {
            uint32_t* dst = &s_tex[wave][(ch * CW) * LSTRIDE + o];
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
              if (rr >= nr) continue;
#pragma unroll
              for (int j = 0; j < CW; ++j) dst[(rr * F + j) * LSTRIDE] = tx[gg][rr][j];
            }
          }
        }
      }
    }
    wave_lds_sync();
    if (b == 0) PBA_STK(3);
    if (walk) {
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        if (rr >= nr) continue;
        const int r = r0 + rr;
        uint32_t t[F];
#pragma unroll
        for (int c = 0; c < F; ++c) t[c] = s_tex[wave][(rr * F + c) * LSTRIDE + lane];
        // column by column: the horizontal lerps of column j are consumed against the previous row's and then replace
        // them, so only one row of lerps is ever live (the sums still run over j in ascending order)
        const int i = r - 1;
        if (FAST) {
          // ---- reduced-precision walk (opt-in, BASELINE configs[4] tolerance sweep; NOT bit-compatible with the
          //   reference): prec 1: fp32 interpolation and fp32 accumulation of M, b, c;  prec 2: additionally the
          //   residual and the gradients are rounded to bf16 before they enter the (fp32) accumulation.
          const float dy = kLean ? __fsub_rn((float)(by + (r >= 1 ? i : 0) + 1), (float)(v + (double)((r >= 1 ? i : 0) - R))) : dys[kLean ? 0 : (r >= 1 ? i : 0)], omdy = 1.0f - dy;
#pragma unroll
          for (int j = 0; j < W; ++j) {
            const float om = 1.0f - dxs[j];
            const float h0 = fmaf(dxs[j], tex_I(t[j]), om * tex_I(t[j + 1]));
            float h1 = 0.f, h2 = 0.f;
            if (JAC) {
              h1 = fmaf(dxs[j], tex_gx2(t[j]), om * tex_gx2(t[j + 1]));
              h2 = fmaf(dxs[j], tex_gy2(t[j]), om * tex_gy2(t[j + 1]));
            }
            if (r >= 1) {
              const float sI = fmaf(dy, fHp[0][j], omdy * h0);
              const float e = rnd(p0[i * W + j] - sI);
              fc0 = fmaf(e, e, fc0);
              if (JAC) {
                const float gx = rnd(fmaf(dy, fHp[NPL > 1 ? 1 : 0][j], omdy * h1));
                const float gy = rnd(fmaf(dy, fHp[NPL > 2 ? 2 : 0][j], omdy * h2));
                fa11 = fmaf(gx, gx, fa11); fa12 = fmaf(gx, gy, fa12); fa22 = fmaf(gy, gy, fa22);
                fc1 = fmaf(gx, e, fc1); fc2 = fmaf(gy, e, fc2);
              }
            }
            fHp[0][j] = h0;
            if (JAC) { fHp[NPL > 1 ? 1 : 0][j] = h1; fHp[NPL > 2 ? 2 : 0][j] = h2; }
          }
        } else {
          const float dy = kLean ? (((r >= 1 ? i : 0) > jy_clamp) ? 1.0f : __fsub_rn((float)(by + (r >= 1 ? i : 0) + 1), (float)(v + (double)((r >= 1 ? i : 0) - R)))) : dys[kLean ? 0 : (r >= 1 ? i : 0)];
          const float omdy = __fsub_rn(1.0f, dy);
          // PBA_WALK_ROWWISE: all horizontal lerps of the row first (independent work for the scheduler), then the
          // pixels; otherwise column by column (a third fewer live registers).  Same sums in the same order.
          double Hc[PBA_WALK_ROWWISE ? NPL : 1][W];
          if (PBA_WALK_ROWWISE) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
              const double om_j = kLean ? __dsub_rn(1.0, (double)dxs[j]) : omdx[kLean ? 0 : j];
              Hc[0][j] = hlerp_exact(dxs[j], om_j, tex_I(t[j]), tex_I(t[j + 1]));
              if (JAC) {
                Hc[PBA_WALK_ROWWISE && NPL > 1 ? 1 : 0][j] = hlerp_exact(dxs[j], om_j, tex_gx2(t[j]), tex_gx2(t[j + 1]));
                Hc[PBA_WALK_ROWWISE && NPL > 2 ? 2 : 0][j] = hlerp_exact(dxs[j], om_j, tex_gy2(t[j]), tex_gy2(t[j + 1]));
              }
            }
          }
#pragma unroll
          for (int j = 0; j < W; ++j) {
            double h0, h1 = 0.0, h2 = 0.0;
            if (PBA_WALK_ROWWISE) {
              h0 = Hc[0][j];
              if (JAC) { h1 = Hc[PBA_WALK_ROWWISE && NPL > 1 ? 1 : 0][j]; h2 = Hc[PBA_WALK_ROWWISE && NPL > 2 ? 2 : 0][j]; }
            } else {
              const double om_j = kLean ? __dsub_rn(1.0, (double)dxs[j]) : omdx[kLean ? 0 : j];
              h0 = hlerp_exact(dxs[j], om_j, tex_I(t[j]), tex_I(t[j + 1]));
              if (JAC) {
                h1 = hlerp_exact(dxs[j], om_j, tex_gx2(t[j]), tex_gx2(t[j + 1]));
                h2 = hlerp_exact(dxs[j], om_j, tex_gy2(t[j]), tex_gy2(t[j + 1]));
              }
            }
            if (r >= 1) {
              const float sI = vlerp_u8_interior(dy, omdy, Hp[0][j], h0);
              const double e = (double)(RES ? rl.desc[i * W + j] : p0[i * W + j]) - (double)sI;   // photobundle.cc:720 (i0 - i1)
              if (UNITW) {
                cc = fma(e, e, cc);
                if (JAC) {
                  // gradients stay in "2G" units here; the exact power-of-two scales are applied to the sums below
                  const double gx = (double)vlerp_u8_interior(dy, omdy, Hp[NPL > 1 ? 1 : 0][j], h1);
                  const double gy = (double)vlerp_u8_interior(dy, omdy, Hp[NPL > 2 ? 2 : 0][j], h2);
                  m11 = fma(gx, gx, m11); m12 = fma(gx, gy, m12); m22 = fma(gy, gy, m22);
                  b1 = fma(gx, e, b1); b2 = fma(gy, e, b2);
                }
              } else {
                const double w2 = p.w2[i * W + j];
                cc += w2 * e * e;
                if (JAC) {
                  const double gx = (double)vlerp_u8_interior(dy, omdy, Hp[NPL > 1 ? 1 : 0][j], h1);
                  const double gy = (double)vlerp_u8_interior(dy, omdy, Hp[NPL > 2 ? 2 : 0][j], h2);
                  const double wgx = w2 * gx, wgy = w2 * gy;
                  m11 += wgx * gx; m12 += wgx * gy; m22 += wgy * gy;
                  b1 += wgx * e; b2 += wgy * e;
                }
              }
            }
            if (!PBA_WALK_ROWWISE) {
              Hp[0][j] = h0;
              if (JAC) { Hp[NPL > 1 ? 1 : 0][j] = h1; Hp[NPL > 2 ? 2 : 0][j] = h2; }
            }
          }
          if (PBA_WALK_ROWWISE) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
              Hp[0][j] = Hc[0][j];
              if (JAC) { Hp[NPL > 1 ? 1 : 0][j] = Hc[PBA_WALK_ROWWISE && NPL > 1 ? 1 : 0][j]; Hp[NPL > 2 ? 2 : 0][j] = Hc[PBA_WALK_ROWWISE && NPL > 2 ? 2 : 0][j]; }
            }
          }
        }
      }
    }
    if (b + 1 < NB) wave_lds_sync();   // the next batch overwrites the rows just read
  }
  if (walk) {
    if (FAST) {
      cc = (double)fc0;
      if (JAC) { m11 = 0.25 * (double)fa11; m12 = 0.25 * (double)fa12; m22 = 0.25 * (double)fa22; b1 = 0.5 * (double)fc1; b2 = 0.5 * (double)fc2; }
    } else if (JAC) {
      m11 *= 0.25; m12 *= 0.25; m22 *= 0.25; b1 *= 0.5; b2 *= 0.5;   // (2G)^2 / 4, (2G) e / 2: exact
    }
  } else if (active && !win_irr && !PBA_EXPERIMENT_SKIP_IRREGULAR) {
    // rounding-irregular / wild observation (and every irregular one at patch radius > 2): per-pixel generic rule from global memory
    const uint32_t* frame = p.frames + (size_t)slot * p.rows * p.cols;
    // (pixel coordinates re-derived from (u, v) here, so that the xf / yf arrays are dead during the regular walk).
    // One patch row at a time with all of the row's 4 W texel loads in flight together: a wave that holds a single
    // irregular lane executes this path for everybody, so its latency (not its throughput) is what matters.
#pragma unroll 1
    for (int i = 0; i < W; ++i) {
      const float yfi = (float)(v + (double)(i - R));
      int y1, y2; float dy;
      linear_init_axis(yfi, p.rows, y1, y2, dy);
      const float omdy = __fsub_rn(1.0f, dy);
      uint32_t t11[W], t12[W], t21[W], t22[W];
      float dxj[W];
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const float xfj = (float)(u + (double)(j - R));
        int x1, x2;
        linear_init_axis(xfj, p.cols, x1, x2, dxj[j]);
        t11[j] = frame[(size_t)y1 * p.cols + x1]; t12[j] = frame[(size_t)y1 * p.cols + x2];
        t21[j] = frame[(size_t)y2 * p.cols + x1]; t22[j] = frame[(size_t)y2 * p.cols + x2];
      }
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const float dx = dxj[j];
        const double omdxj = __dsub_rn(1.0, (double)dx);
        const float sI = vlerp_exact(dy, omdy, hlerp_exact(dx, omdxj, tex_I(t11[j]), tex_I(t12[j])), hlerp_exact(dx, omdxj, tex_I(t21[j]), tex_I(t22[j])));
        const double e = (double)p0[i * W + j] - (double)sI;
        const double w2 = p.w2[i * W + j];
        cc += w2 * e * e;
        if (JAC) {
          // 2*G is blended and the exact power-of-two scale is applied at the end (as in sample_generic)
          const double gx = (double)(0.5f * vlerp_exact(dy, omdy, hlerp_exact(dx, omdxj, tex_gx2(t11[j]), tex_gx2(t12[j])), hlerp_exact(dx, omdxj, tex_gx2(t21[j]), tex_gx2(t22[j]))));
          const double gy = (double)(0.5f * vlerp_exact(dy, omdy, hlerp_exact(dx, omdxj, tex_gy2(t11[j]), tex_gy2(t12[j])), hlerp_exact(dx, omdxj, tex_gy2(t21[j]), tex_gy2(t22[j]))));
          const double wgx = w2 * gx, wgy = w2 * gy;
          m11 += wgx * gx; m12 += wgx * gy; m22 += wgy * gy;
          b1 += wgx * e; b2 += wgy * e;
        }
      }
    }
  }

  if constexpr (kWindow) {
    // Windowed irregular observations, one at a time with the WAVE on one patch: lane = pixel (W^2 <= 25 lanes busy), the
    // reference's per-tap rule (sample_eigen.h:38-51, :82-101) with the four texels of the pixel taken from the staged
    // window `tw` (texel-major, stride LSTRIDE) of the observation, then six fixed-order wave reductions.  A lane-serial walk of
    // the per-tap rule costs ~3x the regular walk and every lane of the wave waits for it; this costs ~200 instructions per
    // irregular observation.
    static_assert(W * W <= 64, "one lane per pixel");
    auto patch_sums = [&](double us, double vs, int wy, int wx, const uint32_t* tw, float dsc_px, double (&q6)[6]) {
      double q_cc = 0.0, q11 = 0.0, q12 = 0.0, q22 = 0.0, q1 = 0.0, q2 = 0.0;
      if (lane < W * W) {
        const int i = lane / W, j = lane - i * W;
        const float yfi = (float)(vs + (double)(i - R)), xfj = (float)(us + (double)(j - R));
        int y1, y2, x1, x2; float dy, dx;
        linear_init_axis(yfi, p.rows, y1, y2, dy);
        linear_init_axis(xfj, p.cols, x1, x2, dx);
        const float omdy = __fsub_rn(1.0f, dy);
        const double omdxj = __dsub_rn(1.0, (double)dx);
        const int o1 = (y1 - wy) * F - wx, o2 = (y2 - wy) * F - wx;
        const uint32_t t11 = tw[(o1 + x1) * LSTRIDE], t12 = tw[(o1 + x2) * LSTRIDE];
        const uint32_t t21 = tw[(o2 + x1) * LSTRIDE], t22 = tw[(o2 + x2) * LSTRIDE];
        const float sI = vlerp_exact(dy, omdy, hlerp_exact(dx, omdxj, tex_I(t11), tex_I(t12)), hlerp_exact(dx, omdxj, tex_I(t21), tex_I(t22)));
        const double e = (double)dsc_px - (double)sI;
        const double w2 = UNITW ? 1.0 : p.w2[lane];
        q_cc = w2 * e * e;
        if (JAC) {
          const double gx = (double)(0.5f * vlerp_exact(dy, omdy, hlerp_exact(dx, omdxj, tex_gx2(t11), tex_gx2(t12)), hlerp_exact(dx, omdxj, tex_gx2(t21), tex_gx2(t22))));
          const double gy = (double)(0.5f * vlerp_exact(dy, omdy, hlerp_exact(dx, omdxj, tex_gy2(t11), tex_gy2(t12)), hlerp_exact(dx, omdxj, tex_gy2(t21), tex_gy2(t22))));
          const double wgx = w2 * gx, wgy = w2 * gy;
          q11 = wgx * gx; q12 = wgx * gy; q22 = wgy * gy; q1 = wgx * e; q2 = wgy * e;
        }
      }
      q6[0] = q_cc; q6[1] = q11; q6[2] = q12; q6[3] = q22; q6[4] = q1; q6[5] = q2;
      if (JAC) wave_sum_n<6>(q6);
      else q6[0] = wave_sum(q6[0]);
    };
    if (share_irr) {
      // ---- the workgroup's queue, dealt round robin: wave w takes entries w, w + WAVES, ... (same sums, same order inside a patch:
      // the result does not depend on who computes it).  Two workgroup barriers, paid only by workgroups that hold such patches.
      lds_barrier();                                   // every wave's windows are staged, every record is written, s_base | s_irr are dead
      double* s_res = reinterpret_cast<double*>(&s_bi[0][0][0]);
      for (int q0 = wave; q0 < q_total; q0 += 4 * WAVES) {
        // up to four patches per trip: their descriptor values (lane = pixel) are requested together
        float dsc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = q0 + k * WAVES;
          dsc[k] = (q < q_total && lane < W * W) ? p.desc[(size_t)s_q[q < q_total ? q : 0].pt * (W * W) + lane] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = q0 + k * WAVES;
          if (q >= q_total) break;
          const IrrRec rcd = s_q[q];
          const int sw = rcd.src >> 8, sl = rcd.src & 0xff;
          double q6[6];
          patch_sums(rcd.u, rcd.v, rcd.wyx >> 16, rcd.wyx & 0xffff, &s_tex[sw][sl], dsc[k], q6);
          if (lane < 6) s_res[6 * q + lane] = (lane == 0) ? q6[0] : (lane == 1 ? q6[1] : (lane == 2 ? q6[2] : (lane == 3 ? q6[3] : (lane == 4 ? q6[4] : q6[5]))));
        }
      }
      lds_barrier();
      if (my_q >= 0) { cc = s_res[6 * my_q]; m11 = s_res[6 * my_q + 1]; m12 = s_res[6 * my_q + 2]; m22 = s_res[6 * my_q + 3]; b1 = s_res[6 * my_q + 4]; b2 = s_res[6 * my_q + 5]; }
    } else {
    unsigned long long im = im_all;
    while (im) {
      // up to four irregular observations per trip: their descriptor values (lane = pixel) are requested together, so
      // that one global round trip serves four patches
      int srcs[4]; float dsc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        srcs[k] = im ? __builtin_amdgcn_readfirstlane(__ffsll((long long)im) - 1) : -1;
        if (im) im &= im - 1;
        dsc[k] = 0.f;
        if (srcs[k] >= 0 && lane < W * W) dsc[k] = p.desc[(size_t)__builtin_amdgcn_readlane(pt, srcs[k] < 0 ? 0 : srcs[k]) * (W * W) + lane];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int src = srcs[k];
        if (src < 0) break;
        const double us = readlane_f64(u, src), vs = readlane_f64(v, src);
        const int wy = __builtin_amdgcn_readlane(by0, src), wx = __builtin_amdgcn_readlane(bx0, src);
        double q6[6];
        patch_sums(us, vs, wy, wx, &s_tex[wave][src], dsc[k], q6);
        if (lane == src) { cc = q6[0]; m11 = q6[1]; m12 = q6[2]; m22 = q6[3]; b1 = q6[4]; b2 = q6[5]; }
      }
    }
    }
  }
  if constexpr (!kWindow) {
    // The same per-tap pass at patch radius > 2, where only RB rows of the footprints are ever staged: the wave loads the
    // clamped F x F window of ONE irregular observation into its own LDS buffer (F^2 texels, F^2 / 64 loads per lane),
    // then lane = pixel over ceil(W^2 / 64) rounds and six wave sums.  (These observations used to take the per-pixel
    // path from global memory -- 11 dependent rounds of 44 loads at 11x11 -- with the whole workgroup waiting.)
    unsigned long long im = FAST ? 0ull : __ballot(win_irr);
    while (im) {
      const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)im) - 1);
      im &= im - 1;
      const double us = readlane_f64(u, src), vs = readlane_f64(v, src);
      const int wy = __builtin_amdgcn_readlane(by0, src), wx = __builtin_amdgcn_readlane(bx0, src);
      const int sl = __builtin_amdgcn_readlane(slot, src), pts = __builtin_amdgcn_readlane(pt, src);
      const uint32_t* fr = p.frames + (size_t)sl * p.rows * p.cols;
      constexpr int NPX = (W * W + 63) / 64;
      float dsc[NPX];
#pragma unroll
      for (int q = 0; q < NPX; ++q) dsc[q] = (q * 64 + lane < W * W) ? p.desc[(size_t)pts * (W * W) + q * 64 + lane] : 0.f;
      for (int t = lane; t < F * F; t += 64) {
        const int wr = t / F, wc = t - wr * F;
        s_win[wave][t] = fr[(size_t)min(wy + wr, p.rows - 1) * p.cols + min(wx + wc, p.cols - 1)];
      }
      wave_lds_sync();
      double q6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      static_assert(NPX <= 2, "descriptor select below");
#pragma unroll 1     // (code size: the kernel is ~90 KB at 11x11 as it is)
      for (int q = 0; q < NPX; ++q) {
        const int pix = q * 64 + lane;
        const float dsc_q = (NPX > 1 && q == 1) ? dsc[NPX > 1 ? 1 : 0] : dsc[0];
        if (pix < W * W) {
          const int i = pix / W, j = pix - i * W;
          const float yfi = (float)(vs + (double)(i - R)), xfj = (float)(us + (double)(j - R));
          int y1, y2, x1, x2; float dy, dx;
          linear_init_axis(yfi, p.rows, y1, y2, dy);
          linear_init_axis(xfj, p.cols, x1, x2, dx);
          const float omdy = __fsub_rn(1.0f, dy);
          const double omdxj = __dsub_rn(1.0, (double)dx);
          const uint32_t* tw = &s_win[wave][0];
          const int o1 = (y1 - wy) * F - wx, o2 = (y2 - wy) * F - wx;
          const uint32_t t11 = tw[o1 + x1], t12 = tw[o1 + x2], t21 = tw[o2 + x1], t22 = tw[o2 + x2];
          const float sI = vlerp_exact(dy, omdy, hlerp_exact(dx, omdxj, tex_I(t11), tex_I(t12)), hlerp_exact(dx, omdxj, tex_I(t21), tex_I(t22)));
          const double e = (double)dsc_q - (double)sI;
          const double w2 = UNITW ? 1.0 : p.w2[pix];
          q6[0] += w2 * e * e;
          if (JAC) {
            const double gx = (double)(0.5f * vlerp_exact(dy, omdy, hlerp_exact(dx, omdxj, tex_gx2(t11), tex_gx2(t12)), hlerp_exact(dx, omdxj, tex_gx2(t21), tex_gx2(t22))));
            const double gy = (double)(0.5f * vlerp_exact(dy, omdy, hlerp_exact(dx, omdxj, tex_gy2(t11), tex_gy2(t12)), hlerp_exact(dx, omdxj, tex_gy2(t21), tex_gy2(t22))));
            const double wgx = w2 * gx, wgy = w2 * gy;
            q6[1] += wgx * gx; q6[2] += wgx * gy; q6[3] += wgy * gy; q6[4] += wgx * e; q6[5] += wgy * e;
          }
        }
      }
      wave_sum_n<6>(q6);
      if (lane == src) { cc = q6[0]; m11 = q6[1]; m12 = q6[2]; m22 = q6[3]; b1 = q6[4]; b2 = q6[5]; }
      wave_lds_sync();      // the next observation overwrites the window
    }
  }
  PBA_STK(4);
  // ---- phase 4: loss (HuberLoss::Evaluate + Corrector with rho'' <= 0), record, block cost ---------------
  double cost_obs = 0.0;
  if (active) {
    double rho0 = cc, rho1 = 1.0;
    if (p.huber > 0.0 && cc > p.huber * p.huber) {
      const double r = sqrt(cc);
      rho0 = 2.0 * p.huber * r - p.huber * p.huber;
      rho1 = fmax(DBL_MIN, p.huber / r);
    }
    cost_obs = 0.5 * rho0;
    if (!isfinite(cc)) atomicOr(&s_fail, 1);
    if (JAC) {
      // streamed: the records are next read by another kernel (and, the L2 being per XCD, from HBM anyway); keeping
      // them out of the L2 as dirty lines also shortens the write-back at the end of the kernel
      const double rv[6] = {rho1 * m11, rho1 * m12, rho1 * m22, rho1 * b1, rho1 * b2, cost_obs};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        if constexpr (RES) rl.recc[k] = rv[k];      // (resident solve: the record never leaves the lane)
        else if (PBA_REC_SC1) store_agent(p.rec + k * p.rec_stride + obs, rv[k]);
        else __builtin_nontemporal_store(rv[k], p.rec + k * p.rec_stride + obs);
      }
    }
  }
  // deterministic block reductions: butterfly inside each wave, then the waves in order
  if (FUSED) {
    if (!PBA_STEP_SUMS_EARLY) fused_wave_step_sums(bs_mcc, bs_st2, bs_x2);
    fused_block_partials<WAVES>(p, bid, lane, wave, cost_obs, bs_mcc, bs_st2, bs_x2, s_red, s_fail, tix);
  } else {
    const double v = wave_sum(cost_obs);
    if (lane == 0) s_red[wave] = v;
    lds_barrier();
    double a = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) a += s_red[w];
    if (tix == 0) {
      p.block_cost[bid] = a;
      p.block_fail[bid] = s_fail;
    }
  }
  PBA_STK(5);
  tk[6] = t_begin; tk[7] = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
  tk[0] |= (unsigned long long)(__builtin_amdgcn_s_getreg(6164) & 15u) << 56;   // HW_REG_XCC_ID[3:0]
  if (p.dbg && tix == 0) for (int k = 0; k < 8; ++k) p.dbg[blockIdx.x * 8 + k] = tk[k];
#undef PBA_STK
  (void)n_blocks;
  if (FUSED && !RES) fused_finalize<WAVES>(p, lane, wave, &s_base[0][0], reinterpret_cast<double*>(&s_tex[0][0]), t_begin, tix);
}

// amdgpu_waves_per_eu(N, N): the register allocator / scheduler works for exactly N resident waves per SIMD (with only a
// lower bound it trades instruction-level parallelism for an occupancy the kernel does not profit from: measured).
template <int R, bool JAC, int WAVES, bool FUSED, bool UNITW, bool FAST>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu((R <= 2 ? (JAC ? PBA_SAMPLE_WAVES_PER_SIMD : 2) : (R >= 4 ? PBA_SAMPLE_WAVES_LARGE : 2)), (R <= 2 ? (JAC ? PBA_SAMPLE_WAVES_PER_SIMD : 2) : (R >= 4 ? PBA_SAMPLE_WAVES_LARGE : 2)))))
void k_sample(SampleParams p_in) {
  SampleParams p = p_in;
  if (!PBA_PHASE_TIMING) p.dbg = nullptr;
  if (FUSED && !fused_resolve_parity(p)) return;
  if (FUSED && p.lm_init_dst && blockIdx.x == 0 && threadIdx.x < sizeof(LmState) / 4)      // (consumed by the NEXT kernel of the stream)
    reinterpret_cast<unsigned*>(p.lm_init_dst)[threadIdx.x] = __hip_atomic_load(reinterpret_cast<const unsigned*>(p.lm_init_src) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __shared__ SampleSmem<R, WAVES> sm;
  ResLane<R> unused;      // (three-kernel path: nothing is resident)
  sample_wg<R, JAC, WAVES, FUSED, UNITW, FAST, false>(p, sm, unused, xcd_logical_block(blockIdx.x, gridDim.x), (int)gridDim.x, (int)threadIdx.x);
}

// =====================================================================================================
// multi-channel descriptors (reference photobundle.cc:229-245: IntensityAndGradient = 3 channels, BitPlanes = 8)
// =====================================================================================================
// Multi-channel frames are kept as VALUE planes only ([slot][channel][rows*cols] float, 4 B per texel): the gradient
// images of a channel (DescriptorFrame ctor, photobundle.cc:172-175 = imgproc.cc:27-95: 0.5 * central difference, zero on
// the one-pixel border) are formed from the neighbouring values where they are used -- the same float subtraction and
// exact halving, so the same bits -- instead of being stored next to the value.  r3: the {value, Gx, Gy, 0} float4 texels
// of round 2 made k_sample_mc HBM-bound (1.67 GB fetched per launch at C = 8, 3.1 TB/s; 477 MB of frames for 8 slots x 8
// channels, no reuse caught by the caches); planes of values are a quarter of that and fit the 256 MB MALL.
__device__ __forceinline__ void mc_texel(const float* __restrict__ plane, int rows, int cols, int y, int x, float& v, float& gx, float& gy) {
  const size_t i = (size_t)y * cols + x;
  v = plane[i];
  gx = 0.f; gy = 0.f;
  if (y >= 1 && y < rows - 1 && x >= 1 && x < cols - 1) {
    gx = 0.5f * __fsub_rn(plane[i + 1], plane[i - 1]);
    gy = 0.5f * __fsub_rn(plane[i + cols], plane[i - cols]);
  }
}

__global__ void k_unpack_channel(const float* __restrict__ plane, int rows, int cols, float* I, float* Gx, float* Gy) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= cols) return;
  const size_t i = (size_t)y * cols + x;
  float v, gx, gy;
  mc_texel(plane, rows, cols, y, x, v, gx, gy);
  I[i] = v; Gx[i] = gx; Gy[i] = gy;
}

// -----------------------------------------------------------------------------------------------------
// Device-side producers of the descriptor channels and of the image pyramid (r3).  Same arithmetic, in the same order,
// as photobundle_amd/host/imgproc.h / photobundle_pyramid.cc (the host restatement of reference src/imgproc.cc:109-245,
// photobundle.cc:225-248 and of the cv::GaussianBlur / cv::pyrDown calls inside them): integer paths are exact, the float
// path runs under `#pragma clang fp contract(off)` with plain operators (the __fmul_rn / __fadd_rn intrinsics are plain
// operators in inlined bodies to this compiler and WOULD be fused) so that it rounds like the FMA-free host build.  One thread per output pixel,
// taps straight from global memory (once per frame, 466k pixels: the L2 serves the overlap).
// -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101_dev(int i, int n) {     // BORDER_REFLECT_101, any offset
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; }
  return i;
}

// cv::GaussianBlur(src, dst, Size(3,3), sigma) on 8-bit images in OpenCV's fixed point: kernel cvRound(k * 256), the two
// passes in integers, (sum + 2^15) >> 16 (imgproc.h gaussianBlur3x3).  k0..k2 are the integer weights.
__global__ void k_blur3_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int cols, int k0, int k1, int k2) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= cols) return;
  const int xm = reflect101_dev(x - 1, cols), xp = reflect101_dev(x + 1, cols);
  int t[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const uint8_t* s = src + (size_t)reflect101_dev(y + d - 1, rows) * cols;
    t[d] = k0 * (int)s[xm] + k1 * (int)s[x] + k2 * (int)s[xp];
  }
  const int r = (k0 * t[0] + k1 * t[1] + k2 * t[2] + (1 << 15)) >> 16;
  dst[(size_t)y * cols + x] = (uint8_t)min(255, max(0, r));
}

// imgproc.cc:126-197: bit b set when the b-th 3x3 neighbour (row-major, centre skipped) >= centre; zero border
__global__ void k_census(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int cols) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= cols) return;
  const size_t i = (size_t)y * cols + x;
  unsigned v = 0;
  if (y >= 1 && y < rows - 1 && x >= 1 && x < cols - 1) {
    const int c = src[i];
    v = ((unsigned)(src[i - cols - 1] >= c) << 0) | ((unsigned)(src[i - cols] >= c) << 1) | ((unsigned)(src[i - cols + 1] >= c) << 2) |
        ((unsigned)(src[i - 1] >= c) << 3) | ((unsigned)(src[i + 1] >= c) << 4) | ((unsigned)(src[i + cols - 1] >= c) << 5) |
        ((unsigned)(src[i + cols] >= c) << 6) | ((unsigned)(src[i + cols + 1] >= c) << 7);
  }
  dst[i] = (uint8_t)v;
}

// imgproc.cc:199-245: the eight bit planes of the census image as float, each smoothed 5x5 (imgproc.h gaussianBlur5x5:
// k2 c + k1 (l1 + r1) + k0 (l2 + r2) along the rows, then the same down the columns, float, REFLECT_101).  One thread
// reads the 25 census bytes of its pixel once and produces all eight planes.  blur == 0: the planes themselves.
__global__ void k_bitplanes(const uint8_t* __restrict__ census, float* __restrict__ out, int rows, int cols, int blur,
                            float k0, float k1, float k2) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= cols) return;
  const size_t npix = (size_t)rows * cols, i = (size_t)y * cols + x;
  if (!blur) {
    const unsigned c = census[i];
#pragma unroll
    for (int b = 0; b < 8; ++b) out[(size_t)b * npix + i] = (float)((c >> b) & 1u);
    return;
  }
  int xs[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) xs[d] = reflect101_dev(x + d - 2, cols);
  unsigned c[5][5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const uint8_t* s = census + (size_t)reflect101_dev(y + r - 2, rows) * cols;
#pragma unroll
    for (int d = 0; d < 5; ++d) c[r][d] = s[xs[d]];
  }
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    float t[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const float s0 = (float)((c[r][0] >> b) & 1u), s1 = (float)((c[r][1] >> b) & 1u), s2 = (float)((c[r][2] >> b) & 1u);
      const float s3 = (float)((c[r][3] >> b) & 1u), s4 = (float)((c[r][4] >> b) & 1u);
      float v = s2 * k2;              // plain operators: the pragma above governs them (not the bodies of inlined intrinsics)
      v += (s1 + s3) * k1;
      v += (s0 + s4) * k0;
      t[r] = v;
    }
    float v = t[2] * k2;
    v += (t[1] + t[3]) * k1;
    v += (t[0] + t[4]) * k0;
    out[(size_t)b * npix + i] = v;
  }
}

// photobundle.cc:229-235 (IntensityAndGradient): channels I, Gx, Gy of the u8 frame as float (imgproc.cc:27-95: 0.5 *
// central difference, zero one-pixel border)
__global__ void k_channels_intensity_gradient(const uint8_t* __restrict__ img, float* __restrict__ out, int rows, int cols) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= cols) return;
  const size_t npix = (size_t)rows * cols, i = (size_t)y * cols + x;
  float gx = 0.f, gy = 0.f;
  if (y >= 1 && y < rows - 1 && x >= 1 && x < cols - 1) {
    gx = 0.5f * __fsub_rn((float)img[i + 1], (float)img[i - 1]);
    gy = 0.5f * __fsub_rn((float)img[i + cols], (float)img[i - cols]);
  }
  out[i] = (float)img[i];
  out[npix + i] = gx;
  out[2 * npix + i] = gy;
}

// cv::pyrDown on 8-bit images (photobundle_pyramid.cc pyrDownU8: [1 4 6 4 1] along the rows at the even columns, then
// down the columns at the even rows, integers, (sum + 128) >> 8, REFLECT_101, size ((rows+1)/2, (cols+1)/2)).  The
// source is a packed frame of the finer level's engine (intensity in the low byte) or a plain u8 image.
template <class TSrc>
__global__ void k_pyr_down(const TSrc* __restrict__ src, uint8_t* __restrict__ dst, int rows, int cols, int drows, int dcols) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= dcols) return;
  int xs[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) xs[d] = reflect101_dev(2 * x + d - 2, cols);
  const int w[5] = {1, 4, 6, 4, 1};
  int acc = 0;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const TSrc* s = src + (size_t)reflect101_dev(2 * y + r - 2, rows) * cols;
    int h = 0;
#pragma unroll
    for (int d = 0; d < 5; ++d) h += w[d] * (int)((unsigned)s[xs[d]] & 0xffu);
    acc += w[r] * h;
  }
  dst[(size_t)y * dcols + x] = (uint8_t)((acc + 128) >> 8);
}

// Generic tap of one channel, any position (clamped / irregular observations).
template <bool JAC>
__device__ __forceinline__ void sample_generic_mc(const float* __restrict__ plane, int rows, int cols, float yf, float xf,
                                                  float& sI, float& sgx, float& sgy) {
  int x1, x2, y1, y2; float dx, dy;
  linear_init_axis(yf, rows, y1, y2, dy);
  linear_init_axis(xf, cols, x1, x2, dx);
  float v[4], gx[4] = {0.f, 0.f, 0.f, 0.f}, gy[4] = {0.f, 0.f, 0.f, 0.f};
  if (JAC) {
    mc_texel(plane, rows, cols, y1, x1, v[0], gx[0], gy[0]);
    mc_texel(plane, rows, cols, y1, x2, v[1], gx[1], gy[1]);
    mc_texel(plane, rows, cols, y2, x1, v[2], gx[2], gy[2]);
    mc_texel(plane, rows, cols, y2, x2, v[3], gx[3], gy[3]);
  } else {
    v[0] = plane[(size_t)y1 * cols + x1]; v[1] = plane[(size_t)y1 * cols + x2];
    v[2] = plane[(size_t)y2 * cols + x1]; v[3] = plane[(size_t)y2 * cols + x2];
  }
  const double omdx = __dsub_rn(1.0, (double)dx);
  const float omdy = __fsub_rn(1.0f, dy);
  sI = vlerp_exact(dy, omdy, hlerp_exact(dx, omdx, v[0], v[1]), hlerp_exact(dx, omdx, v[2], v[3]));
  if (JAC) {
    sgx = vlerp_exact(dy, omdy, hlerp_exact(dx, omdx, gx[0], gx[1]), hlerp_exact(dx, omdx, gx[2], gx[3]));
    sgy = vlerp_exact(dy, omdy, hlerp_exact(dx, omdx, gy[0], gy[1]), hlerp_exact(dx, omdx, gy[2], gy[3]));
  }
}

__global__ void k_sample_probe_mc(const float* __restrict__ plane, int rows, int cols, int n, const float* __restrict__ y,
                                  const float* __restrict__ x, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sI, sgx, sgy;
  sample_generic_mc<true>(plane, rows, cols, y[i], x[i], sI, sgx, sgy);
  out[3 * i] = sI; out[3 * i + 1] = sgx; out[3 * i + 2] = sgy;
}

// The sampling pass over C channels: one lane per observation like k_sample; the channel loop is the outer (run time)
// loop, inside it the footprint rows of that channel stream through LDS in batches and the lane accumulates the SAME six
// sums across all channels in the reference's residual order (channel-major, photobundle.cc:708-722), because every
// pixel of every channel shares the projection Jacobian A: M = sum_k sum_pix w^2 g g^T etc.
#ifndef PBA_MC_WAVES_SMALL
#define PBA_MC_WAVES_SMALL 2
#endif
#define PBA_MC_WAVES(R) ((R) <= 2 ? PBA_MC_WAVES_SMALL : 2)
// footprint rows per staging batch: the whole footprint while its haloed window fits 64 floats per lane, two rows beyond
// (measured at C = 8, 5x5: the whole footprint in one batch at 2 waves/SIMD 460 us per launch; two batches of three rows at 3
// or 4 waves/SIMD -- 166 VGPRs, 42 KB of LDS -- 507 us: the second staging round costs more than the occupancy returns)
#ifndef PBA_MC_RB2
#define PBA_MC_RB2 6
#endif
constexpr int sample_mc_rows_per_batch(int R) { return R == 2 ? PBA_MC_RB2 : (((2 * R + 4) * (2 * R + 4) <= 64) ? 2 * R + 2 : 2); }

//   FUSED: like k_sample's fused form -- back-substitution of the step for the workgroup's whole points first, sampling at
//   the candidate it just formed, step finalisation (and, single rank, the trust-region decision) by the last workgroup.
//   UNITW: unit patch weights (MakePatchWeights without the Gaussian, the reference's default): w^2 = 1 is folded away.
template <int R, bool JAC, int WAVES, bool FUSED, bool UNITW>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(PBA_MC_WAVES(R), PBA_MC_WAVES(R)))) void k_sample_mc(SampleParams p_in, const float* __restrict__ frames_mc, int n_channels) {
  static_assert(!FUSED || (WAVES * 64) % 128 == 0, "fused tiles are 128 observations");
  SampleParams p = p_in;
  p.dbg = nullptr;
  if (FUSED && !fused_resolve_parity(p)) return;
  if (FUSED && p.lm_init_dst && blockIdx.x == 0 && threadIdx.x < sizeof(LmState) / 4)      // (as in k_sample)
    reinterpret_cast<unsigned*>(p.lm_init_dst)[threadIdx.x] = __hip_atomic_load(reinterpret_cast<const unsigned*>(p.lm_init_src) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  constexpr int W = 2 * R + 1, F = 2 * R + 2;
  constexpr int H = JAC ? 1 : 0;                    // halo: the gradients of a footprint texel need its four neighbours
  constexpr int FW = F + 2 * H;                     // staged columns
  constexpr int RB = sample_mc_rows_per_batch(R);   // footprint rows per batch
  constexpr int SR = RB + 2 * H;                    // staged rows per batch
  constexpr int NB = (F + RB - 1) / RB;
  constexpr int FF = SR * FW;
  constexpr int LSTRIDE = 65;
  constexpr int NPL = JAC ? 3 : 1;
  constexpr size_t kTexBytes = sizeof(float) * WAVES * FF * LSTRIDE;
  constexpr size_t kBsBytes = FUSED ? sizeof(double) * 3 * WAVES * 64 : 0;      // back-substitution scratch ahead of the tables
  constexpr size_t kPreBytes = kBsBytes + (FUSED ? 2 : 1) * kMaxFrames * sizeof(CamGeom);
  static_assert(!FUSED || kTexBytes >= sizeof(double) * 4 * WAVES * 64, "the finalisation reuses the texel region");
  __shared__ __attribute__((aligned(16))) char s_raw[kTexBytes > kPreBytes ? kTexBytes : kPreBytes];
  float (*s_tex)[FF * LSTRIDE] = reinterpret_cast<float (*)[FF * LSTRIDE]>(s_raw);
  __shared__ int32_t s_base[WAVES][64];
  __shared__ double s_red[4 * WAVES];
  __shared__ int32_t s_fail;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bid = blockIdx.x;
  int obs = blockIdx.x * (WAVES * 64) + threadIdx.x;
  bool active = obs < p.n_obs;
  if (threadIdx.x == 0) s_fail = 0;
  CamGeom* s_geom = reinterpret_cast<CamGeom*>(s_raw + kBsBytes);
  double* s_bk = reinterpret_cast<double*>(s_geom + kMaxFrames);
  FusedIdx fi{make_int4(0, 0, 0, 0), 0, 0, 0, 0};
  if (FUSED) fi = fused_prefetch_indices<WAVES * 64>(p, bid, (int)threadIdx.x);
  stage_geom<WAVES * 64>(p.geom, s_geom, p.n_frames, threadIdx.x);
  if (FUSED) fused_stage_step_table<WAVES * 64>(p, s_bk, (int)threadIdx.x);
  lds_barrier();

  int pt = 0, slot = 0;
  double prm[3] = {0.0, 0.0, 0.0};
  double bs_mcc = 0.0, bs_st2 = 0.0, bs_x2 = 0.0;
  if (FUSED) {
    fused_backsub<WAVES * 64>(p, p.rays, fi, s_bk, reinterpret_cast<double*>(s_raw), pt, slot, obs, active, prm, bs_mcc, bs_st2, bs_x2, (int)threadIdx.x);
  } else if (active) {
    pt = p.obs_point[obs];
    slot = p.obs_slot[obs];
    prm[0] = p.xyz[3 * (size_t)pt]; prm[1] = p.xyz[3 * (size_t)pt + 1]; prm[2] = p.xyz[3 * (size_t)pt + 2];
  }
  double u = 0.0, v = 0.0;
  int bx = 0, by = 0;
  bool regular = false;
  float dxs[W], dys[W];
  double omdx[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { dxs[j] = 1.f; dys[j] = 1.f; omdx[j] = 0.0; }
  const size_t npix = (size_t)p.rows * p.cols;
  if (active) {
    if (p.rays && !(prm[0] > 0.0)) atomicOr(&s_fail, 1);     // inverse-depth variant: rho <= 0 is an evaluation failure (k_sample)
    double X[3], qd[3];
    point_world(p.rays, pt, prm, X, qd);
    double xw[3];
    transform_point(s_geom[slot], X, xw);
    project_point(xw, p.fx, p.fy, p.cx, p.cy, u, v);
    float xf[W], yf[W];
#pragma unroll
    for (int j = 0; j < W; ++j) { xf[j] = (float)(u + (double)(j - R)); yf[j] = (float)(v + (double)(j - R)); }
    bx = trunc_x86(xf[0]);
    by = trunc_x86(yf[0]);
    // regular: consecutive taps whose whole footprint is INTERIOR (the gradients of its texels come from a one-pixel halo
    // of values; the image border, where they are zero by definition, goes to the per-tap path)
    bool reg = (bx >= 1) && (bx + F <= p.cols - 1) && (by >= 1) && (by + F <= p.rows - 1);
#pragma unroll
    for (int j = 1; j < W; ++j) reg = reg && (trunc_x86(xf[j]) == bx + j) && (trunc_x86(yf[j]) == by + j);
    regular = reg;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      dxs[j] = __fsub_rn((float)(bx + j + 1), xf[j]);
      dys[j] = __fsub_rn((float)(by + j + 1), yf[j]);
      omdx[j] = __dsub_rn(1.0, (double)dxs[j]);
    }
  }
  // index of the first staged value (footprint origin minus the halo) in channel 0 of the observation's frame (-1: not staged)
  s_base[wave][lane] = (active && regular) ? (int32_t)((size_t)slot * n_channels * npix + (size_t)(by - H) * p.cols + (bx - H)) : -1;
  lds_barrier();      // the camera table is dead from here on: its LDS is reused by the texel batches

  const bool walk = active && regular;
  double m11 = 0, m12 = 0, m22 = 0, b1 = 0, b2 = 0, cc = 0;
  constexpr int OPI = 64 / FW;                      // observations per staging pass: lane = (observation, staged column)
  constexpr int NG = (64 + OPI - 1) / OPI;
  const int oi = lane / FW, tc = lane - oi * FW;
#pragma unroll 1
  for (int k = 0; k < n_channels; ++k) {
    const float* p0 = p.desc + ((size_t)pt * n_channels + k) * (W * W);
    double Hp[NPL][W];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int j = 0; j < W; ++j) Hp[pl][j] = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int r0 = b * RB;
      const int nr = (F - r0 < RB) ? F - r0 : RB;
      wave_lds_sync();                              // the previous batch (or channel) has been walked
#pragma unroll 1
      for (int g = 0; g < NG; ++g) {
        const int o = g * OPI + oi;
        const int32_t bs = (oi < OPI && o < 64) ? s_base[wave][o & 63] : -1;
        if (bs >= 0) {
          const float* src = frames_mc + (size_t)bs + (size_t)k * npix + tc;
          float tv[SR];
#pragma unroll
          for (int sr = 0; sr < SR; ++sr) if (sr < nr + 2 * H) tv[sr] = src[(size_t)(r0 + sr) * p.cols];     // all loads of the pass in flight
#pragma unroll
          for (int sr = 0; sr < SR; ++sr) if (sr < nr + 2 * H) s_tex[wave][(sr * FW + tc) * LSTRIDE + o] = tv[sr];
        }
      }
      wave_lds_sync();
      if (walk) {
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
          if (rr >= nr) continue;
          const int r = r0 + rr, i = r - 1;
          const float dy = dys[r >= 1 ? i : 0];
          const float omdy = __fsub_rn(1.0f, dy);
          __builtin_amdgcn_sched_barrier(0);      // one footprint row at a time: hoisting the LDS reads of later rows costs registers (spills at R = 1, 3, 5)
          // values of the footprint row (with its left / right halo) and, for the gradients, of the rows above and below
          float vm[FW], t[NPL][F];
#pragma unroll
          for (int c = 0; c < FW; ++c) vm[c] = s_tex[wave][((rr + H) * FW + c) * LSTRIDE + lane];
#pragma unroll
          for (int c = 0; c < F; ++c) {
            t[0][c] = vm[c + H];
            if (JAC) {
              const float up = s_tex[wave][(rr * FW + c + H) * LSTRIDE + lane], dn = s_tex[wave][((rr + 2 * H) * FW + c + H) * LSTRIDE + lane];
              t[NPL > 1 ? 1 : 0][c] = 0.5f * __fsub_rn(vm[c + 2 * H], vm[c]);       // imgproc.cc:27-95 on the channel image
              t[NPL > 2 ? 2 : 0][c] = 0.5f * __fsub_rn(dn, up);
            }
          }
#pragma unroll
          for (int j = 0; j < W; ++j) {
            const double h0 = hlerp_exact(dxs[j], omdx[j], t[0][j], t[0][j + 1]);
            double h1 = 0.0, h2 = 0.0;
            if (JAC) {
              h1 = hlerp_exact(dxs[j], omdx[j], t[NPL > 1 ? 1 : 0][j], t[NPL > 1 ? 1 : 0][j + 1]);
              h2 = hlerp_exact(dxs[j], omdx[j], t[NPL > 2 ? 2 : 0][j], t[NPL > 2 ? 2 : 0][j + 1]);
            }
            if (r >= 1) {
              const float sI = vlerp_exact(dy, omdy, Hp[0][j], h0);
              const double e = (double)p0[i * W + j] - (double)sI;
              const double w2 = UNITW ? 1.0 : p.w2[i * W + j];
              cc += w2 * e * e;
              if (JAC) {
                const double gx = (double)vlerp_exact(dy, omdy, Hp[NPL > 1 ? 1 : 0][j], h1);
                const double gy = (double)vlerp_exact(dy, omdy, Hp[NPL > 2 ? 2 : 0][j], h2);
                const double wgx = w2 * gx, wgy = w2 * gy;
                m11 += wgx * gx; m12 += wgx * gy; m22 += wgy * gy;
                b1 += wgx * e; b2 += wgy * e;
              }
            }
            Hp[0][j] = h0;
            if (JAC) { Hp[NPL > 1 ? 1 : 0][j] = h1; Hp[NPL > 2 ? 2 : 0][j] = h2; }
          }
        }
      }
    }
    if (active && !regular) {
      const float* frame = frames_mc + ((size_t)slot * n_channels + k) * npix;
#pragma unroll 1
      for (int i = 0; i < W; ++i) {
        const float yfi = (float)(v + (double)(i - R));
#pragma unroll 1
        for (int j = 0; j < W; ++j) {
          const float xfj = (float)(u + (double)(j - R));
          float sI, sgx = 0.f, sgy = 0.f;
          sample_generic_mc<JAC>(frame, p.rows, p.cols, yfi, xfj, sI, sgx, sgy);
          const double e = (double)p0[i * W + j] - (double)sI;
          const double w2 = p.w2[i * W + j];
          cc += w2 * e * e;
          if (JAC) {
            const double gx = (double)sgx, gy = (double)sgy;
            const double wgx = w2 * gx, wgy = w2 * gy;
            m11 += wgx * gx; m12 += wgx * gy; m22 += wgy * gy;
            b1 += wgx * e; b2 += wgy * e;
          }
        }
      }
    }
  }

  // loss (HuberLoss::Evaluate + Corrector with rho'' <= 0) over the WHOLE block (all channels), record, block cost
  double cost_obs = 0.0;
  if (active) {
    double rho0 = cc, rho1 = 1.0;
    if (p.huber > 0.0 && cc > p.huber * p.huber) {
      const double r = sqrt(cc);
      rho0 = 2.0 * p.huber * r - p.huber * p.huber;
      rho1 = fmax(DBL_MIN, p.huber / r);
    }
    cost_obs = 0.5 * rho0;
    if (!isfinite(cc)) atomicOr(&s_fail, 1);
    if (JAC) {
      p.rec[0 * p.rec_stride + obs] = rho1 * m11;
      p.rec[1 * p.rec_stride + obs] = rho1 * m12;
      p.rec[2 * p.rec_stride + obs] = rho1 * m22;
      p.rec[3 * p.rec_stride + obs] = rho1 * b1;
      p.rec[4 * p.rec_stride + obs] = rho1 * b2;
      p.rec[5 * p.rec_stride + obs] = cost_obs;
    }
  }
  if (FUSED) {
    lds_barrier();        // every wave is done with the texel region / s_base before the finalisation reuses them
    fused_wave_step_sums(bs_mcc, bs_st2, bs_x2);
    fused_block_partials<WAVES>(p, bid, lane, wave, cost_obs, bs_mcc, bs_st2, bs_x2, s_red, s_fail, (int)threadIdx.x);
    fused_finalize<WAVES>(p, lane, wave, &s_base[0][0], reinterpret_cast<double*>(s_raw), 0ull, (int)threadIdx.x);
  } else {
    const double ws = wave_sum(cost_obs);
    if (lane == 0) s_red[wave] = ws;
    lds_barrier();
    if (threadIdx.x == 0) {
      double a = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) a += s_red[w];
      p.block_cost[blockIdx.x] = a;
      p.block_fail[blockIdx.x] = s_fail;
    }
  }
}

// =====================================================================================================
// Schur elimination of the points (SchurEliminator::Eliminate restated for the device)
//   tiles of <= 128 observations holding whole points; one lane per observation for the per-block algebra,
//   then every thread owns one full 6x6 block (camera pair a <= b) of the reduced matrix for a subset of the
//   tile's points and accumulates it in registers across all tiles of the (persistent) workgroup.  No atomics:
//   the order of every floating-point sum is fixed by the tile / grid decomposition, so runs are reproducible.
// =====================================================================================================
constexpr int kTile = 128;              // observations (= threads) per tile
constexpr int kObsStride = 37;          // doubles per observation in LDS (36 + 1 pad)

constexpr size_t kSchurSmemBytes = sizeof(double) * (kTile * kObsStride + kTile) + kTile * kMaxFrames;

struct SchurParams {
  const double* xyz;
  const double* rays;           // inverse-depth variant (point_world), else null
  const CamGeom* geom;
  const double* rec;            // SoA [6][rec_stride]
  const int32_t* obs_point;
  const uint8_t* obs_slot;
  const int4* tile_info;        // [n_tiles] {first observation, observations, first point, points} (whole points)
  const int2* lane_rec;         // [n_tiles][kTile] per tile lane: {point, slot | first lane of the observation's point << 8 | number of
                                // observations of that point << 16}; independent of the tile descriptor (requested with it)
  double* sp;                   // [n_points][3] Jacobi scale of the point columns (written when init_scale)
  double* ptrec;                // [n_points][12]: P (6, sym packed 00 01 02 11 12 22), g_p (3), D_p^2 (3)
  double* partial;              // [gridDim.x][part_stride]
  int64_t rec_stride;
  int32_t n_tiles, n_frames;
  int32_t n_free;               // free cameras
  int32_t n_pairs;              // n_free (n_free + 1) / 2
  int32_t part_stride;          // 36 n_pairs + 3 * 6 n_free + 3
  int32_t init_scale;
  int32_t jacobi;
  double fx, fy;
  double radius, inv_radius, min_diag, max_diag;
  unsigned long long* dbg;     // optional [gridDim.x][8] per-phase cycle sums of thread 0 (diagnostics)
  // asynchronous driver: parity / radius resolved from the device state
  const LmState* lm;
  int32_t enq_cur;
  int32_t final_pass;          // run although the solve has terminated (gradient norms of the final point)
  const double* xyz_alt; const CamGeom* geom_alt; const double* rec_alt;
  // deferred publication of the PREVIOUS step's outcome (null: nothing to publish): host-mapped stores complete over
  // PCIe, and a kernel does not retire before its stores have, so they are issued at the start of this long kernel
  // rather than at the end of the sampling kernel that produced the outcome (measured: 6 us per iteration)
  LmState* pub_state; const double* pub_scal; double* pub_host_scal;
  unsigned long long* pub_host_seq; unsigned long long pub_seq;
  unsigned long long* stamp;   // null, or the device time-stamp block (kStamp*)
};

__device__ __forceinline__ int sym6(int i, int j) {  // packed upper triangle of a symmetric 6x6 (21 entries)
  const int a = i < j ? i : j, b = i < j ? j : i;
  return a * 6 - (a * (a - 1)) / 2 + (b - a);
}

// Packed / partial layout (doubles), n = 6 * n_free:
//   [0, 36 n_pairs)           T: pair (a <= b, enumerated row by row) -> row-major 6x6 block (a, b)
//   [.., +n) rhs   [.., +n) g_c   [.., +n) diag(U)
//   partial only: +0 gmax_pts, +1 gnorm2_pts, +2 schur_fail
// The body of k_schur for the 128 threads `tid` of one tile owner.
//   three-kernel path (RL = const void): persistent loop over the tiles part, part + n_parts, ...; everything through global memory.
//   resident solve (RL = ResLane<R>, pba_resident.h): ONE tile per call whose indices, point, records and Jacobi scale come from the
//   lane's registers and whose P | g_p | D^2 go back there; the camera table is already in LDS (`geom_res`); `has_tile` = false for the
//   idle half of the last workgroup (it only keeps the barriers company).
template <class RL>
__device__ __forceinline__ void schur_body(SchurParams& p, char* smem, const int tid, const int part, const int n_parts, RL* rl,
                                           const CamGeom* geom_res, const bool has_tile) {
  constexpr bool RES = !std::is_void<RL>::value;
  double* s_obs = reinterpret_cast<double*>(smem);                       // [kTile][kObsStride]
  // V_l (6) g_l (3) per lane live INSIDE the s_obs region (after the staged camera table, before W | Y are
  // written): 40 KB per workgroup => 4 workgroups per CU instead of 3
  static_assert(1024 + kTile * 9 <= kTile * kObsStride && kMaxFrames * sizeof(CamGeom) <= 1024 * sizeof(double), "aliasing fits");
  double* s_vg = s_obs + 1024;                                           // [kTile][9]
  double* s_tot = s_obs + 1024 + kTile * 9;                              // [<= kTile points][9] point totals
  uint16_t* s_ptl = reinterpret_cast<uint16_t*>(s_obs + 1024 + 2 * kTile * 9);   // [<= kTile points] first lane | lanes << 8
  static_assert(1024 + 2 * kTile * 9 + kTile / 4 <= kTile * kObsStride, "point totals fit before W | Y");
  double* s_red = s_obs + kTile * kObsStride;                            // [kTile]
  int8_t* s_lane_of = reinterpret_cast<int8_t*>(s_red + kTile);          // [kMaxFrames (FREE index)][kTile points]: a camera's lanes are contiguous bytes

  const int nf = p.n_free;
  const int n_groups = kTile / p.n_pairs;          // >= 1 (n_pairs <= 136 is clamped by kMaxFrames = 16 -> 120/136)
  const int grp = tid / p.n_pairs;
  const int pair = tid - grp * p.n_pairs;
  const bool owner = grp < n_groups;
  int pa = 0, pb = 0;
  {
    int a = 0, rem = pair;
    while (rem >= nf - a) { rem -= nf - a; ++a; }   // pairs enumerated row by row: (a, a..nf-1)
    pa = a; pb = a + rem;
  }
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  // camera-side sums: entry e = tid + kTile u  <->  (free camera e / 33, value e % 33) of [U_a (21, packed upper
  // triangle) | r_a (6) | g_c,a (6)]: every thread sums its (<= kCamAcc) entries over the tile's points
  constexpr int kCamVals = 33;
  constexpr int kCamAcc = (kCamVals * (kMaxFrames - 1) + kTile - 1) / kTile;
  double acc_cam[kCamAcc];
#pragma unroll
  for (int u = 0; u < kCamAcc; ++u) acc_cam[u] = 0.0;
  double gmax = 0.0, gn2 = 0.0;
  int fail = 0;
  // the camera table is the same for every tile but its LDS copy is overwritten by W | Y: kept in registers (one global
  // read per kernel) and re-staged from there
  static_assert(sizeof(CamGeom) % 8 == 0, "CamGeom is copied as 8-byte words");
  constexpr int kGeomRegs = 4;       // covers 10 frames; the words beyond come from global memory (L2) per tile
  const int n_geom_words = p.n_frames * (int)(sizeof(CamGeom) / 8);
  unsigned long long greg[kGeomRegs];
#pragma unroll
  for (int u = 0; u < kGeomRegs; ++u) {
    const int k = tid + u * kTile;
    greg[u] = (!RES && k < n_geom_words) ? reinterpret_cast<const unsigned long long*>(p.geom)[k] : 0ull;
  }

  if (tid < kCamVals) s_red[tid] = 0.0;   // "row 128" of the camera-side sums: zeros (s_red is otherwise unused until the epilogue)
  static_assert(kCamVals <= kTile, "zero row fits s_red");
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = p.dbg ? __builtin_amdgcn_s_memtime() : 0;
  const unsigned long long t_rt0 = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
#define PBA_TICK(k) do { if (p.dbg) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tph[k] += tn - tlast; tlast = tn; } } while (0)
  int4 ti_next = make_int4(0, 0, 0, 0);
  // the observation's indices are requested one tile ahead (at the start of the pair-block phase) so that the
  // per-observation phase starts with its second round trip (point, Jacobi scale, record) instead of its first
  int nx_pt = 0, nx_slot = 0, nx_l0 = 0, nx_cnt = 0;
  double nx_x[3] = {0.0, 0.0, 0.0};
  if constexpr (RES) {
    if (has_tile) { ti_next = rl->ti; nx_pt = rl->pt; nx_slot = rl->slot; nx_l0 = rl->l0; nx_cnt = rl->cnt; nx_x[0] = rl->X[0]; nx_x[1] = rl->X[1]; nx_x[2] = rl->X[2]; }
  } else {
    ti_next = p.tile_info[min(part, p.n_tiles - 1)];
    {
      const int2 r = p.lane_rec[(size_t)min(part, p.n_tiles - 1) * kTile + tid];     // with the descriptor, not after it
      nx_pt = r.x; nx_slot = r.y & 0xff; nx_l0 = (r.y >> 8) & 0xff; nx_cnt = (r.y >> 16) & 0xff;
    }
    if (tid < ti_next.y) {
      nx_x[0] = p.xyz[3 * (size_t)nx_pt]; nx_x[1] = p.xyz[3 * (size_t)nx_pt + 1]; nx_x[2] = p.xyz[3 * (size_t)nx_pt + 2];
    }
  }
  // (resident solve: exactly one trip, the lane's own tile)
  for (int tile = part; RES ? (tile == part) : (tile < p.n_tiles); tile += n_parts) {
    const int4 ti = ti_next;
    if constexpr (!RES) ti_next = p.tile_info[min(tile + n_parts, p.n_tiles - 1)];   // prefetch the next tile's descriptor
    const int cur_pt = nx_pt, cur_slot = nx_slot, cur_l0 = nx_l0, cur_cnt = nx_cnt;
    const double cur_x[3] = {nx_x[0], nx_x[1], nx_x[2]};
    const int o0 = ti.x, n_here = ti.y, pt0 = ti.z, n_pts = ti.w;
    const bool active = tid < n_here;
    const int obs = o0 + tid;

    for (int k = tid; k < kMaxFrames * kTile / 4; k += kTile) reinterpret_cast<uint32_t*>(s_lane_of)[k] = 0x80808080u;   // -128: no observation
    // P1 only; P2 overwrites the region with W | Y  (resident solve: the workgroup's persistent copy of the table, staged once per step)
    const CamGeom* s_geom = RES ? geom_res : reinterpret_cast<const CamGeom*>(s_obs);
    if constexpr (!RES) {
#pragma unroll
    for (int u = 0; u < kGeomRegs; ++u) {
      const int k = tid + u * kTile;
      if (k < n_geom_words) reinterpret_cast<unsigned long long*>(s_obs)[k] = greg[u];
    }
    }
    if constexpr (!RES) {
      // the words beyond the register copy (windows of 11+ frames), all in flight together
      constexpr int kTail = (kMaxFrames * (int)(sizeof(CamGeom) / 8) + kTile - 1) / kTile - kGeomRegs;
      unsigned long long gt[kTail];
#pragma unroll
      for (int u = 0; u < kTail; ++u) {
        const int k = tid + (kGeomRegs + u) * kTile;
        gt[u] = (k < n_geom_words) ? reinterpret_cast<const unsigned long long*>(p.geom)[k] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < kTail; ++u) {
        const int k = tid + (kGeomRegs + u) * kTile;
        if (k < n_geom_words) reinterpret_cast<unsigned long long*>(s_obs)[k] = gt[u];
      }
    }
    lds_barrier();
    PBA_TICK(0);

    // ---- P1: per observation geometry and point-side contributions -----------------------------------
    int pt = 0, fa = -1, l0 = 0, l1 = 0;
    double Ac[2][6], Ap[2][3], M[3] = {0, 0, 0}, b[2] = {0, 0};
    double MAp[2][3];
    double s_pt[3] = {1.0, 1.0, 1.0};
    if (active) {
      pt = cur_pt;
      const int slot = cur_slot;
      l0 = cur_l0; l1 = l0 + cur_cnt;
      if constexpr (RES) { if (!p.init_scale) { s_pt[0] = rl->sp[0]; s_pt[1] = rl->sp[1]; s_pt[2] = rl->sp[2]; } }
      else if (!p.init_scale) { s_pt[0] = p.sp[3 * (size_t)pt]; s_pt[1] = p.sp[3 * (size_t)pt + 1]; s_pt[2] = p.sp[3 * (size_t)pt + 2]; }
      const CamGeom& g = s_geom[slot];
      fa = g.free_index;
      const double prm[3] = {cur_x[0], cur_x[1], cur_x[2]};
      double X[3], qd[3];
      point_world(p.rays, pt, prm, X, qd);
      double xw[3];
      transform_point(g, X, xw);
      projection_jacobians(g, X, xw, p.fx, p.fy, Ac, Ap);
      point_jacobian(p.rays, qd, Ap);
      if constexpr (RES) { M[0] = rl->rec[0]; M[1] = rl->rec[1]; M[2] = rl->rec[2]; b[0] = rl->rec[3]; b[1] = rl->rec[4]; }
      else {
      M[0] = p.rec[0 * p.rec_stride + obs]; M[1] = p.rec[1 * p.rec_stride + obs]; M[2] = p.rec[2 * p.rec_stride + obs];
      b[0] = p.rec[3 * p.rec_stride + obs]; b[1] = p.rec[4 * p.rec_stride + obs];
      }
      // MAp = M Ap (2x3);  V_l = Ap^T M Ap;  g_l = -Ap^T b   (J = -w g A  =>  J^T r = -A^T b)
#pragma unroll
      for (int k = 0; k < 3; ++k) { MAp[0][k] = M[0] * Ap[0][k] + M[1] * Ap[1][k]; MAp[1][k] = M[1] * Ap[0][k] + M[2] * Ap[1][k]; }
      double* vg = s_vg + tid * 9;
      vg[0] = Ap[0][0] * MAp[0][0] + Ap[1][0] * MAp[1][0];
      vg[1] = Ap[0][0] * MAp[0][1] + Ap[1][0] * MAp[1][1];
      vg[2] = Ap[0][0] * MAp[0][2] + Ap[1][0] * MAp[1][2];
      vg[3] = Ap[0][1] * MAp[0][1] + Ap[1][1] * MAp[1][1];
      vg[4] = Ap[0][1] * MAp[0][2] + Ap[1][1] * MAp[1][2];
      vg[5] = Ap[0][2] * MAp[0][2] + Ap[1][2] * MAp[1][2];
#pragma unroll
      for (int k = 0; k < 3; ++k) vg[6 + k] = -(Ap[0][k] * b[0] + Ap[1][k] * b[1]);
      if (fa >= 0) s_lane_of[fa * kTile + (pt - pt0)] = (int8_t)tid;
      if (tid == l0) s_ptl[pt - pt0] = (uint16_t)(l0 | ((l1 - l0) << 8));
    }
    lds_barrier();
    PBA_TICK(1);

    // ---- P2: point totals, damping, effective inverse, per-observation Schur factors -------------------
    double V[6] = {0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0};
    // point totals, transposed: thread k <-> (point k / 9, component k % 9) adds that component over the point's
    // observations in lane order (<= kMaxFrames independent LDS reads), then every lane picks up its point's nine sums
    for (int k = tid; k < 9 * n_pts; k += kTile) {
      const int q = k / 9, c = k - 9 * q;
      const unsigned lc = s_ptl[q];
      const int ql0 = (int)(lc & 0xffu), qcnt = (int)(lc >> 8);
      const double* src = s_vg + ql0 * 9 + c;
      double a = 0.0;
      double x[8];
#pragma unroll
      for (int l = 0; l < 8; ++l) x[l] = src[9 * (l < qcnt ? l : qcnt - 1)];
#pragma unroll
      for (int l = 0; l < 8; ++l) a += (l < qcnt) ? x[l] : 0.0;
      if (qcnt > 8) {
#pragma unroll
        for (int l = 8; l < kMaxFrames; ++l) x[l - 8] = src[9 * (l < qcnt ? l : qcnt - 1)];
#pragma unroll
        for (int l = 8; l < kMaxFrames; ++l) a += (l < qcnt) ? x[l - 8] : 0.0;
      }
      s_tot[k] = a;
    }
    lds_barrier();
    if (active) {
      const double* tot = s_tot + (pt - pt0) * 9;
#pragma unroll
      for (int k = 0; k < 6; ++k) V[k] = tot[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) gp[k] = tot[6 + k];
    }
    lds_barrier();     // every lane has its point totals: the region may now be overwritten with W | Y
    PBA_TICK(2);
    // gradient-only pass (the last accepted point of a solve that stops at the iteration limit): only g_p and the
    // camera sums of g_c,l are consumed, so the point-block inverse, U_l, r_l, W | Y and the pair blocks are skipped
    const bool grad_only = p.final_pass != 0 && !p.init_scale;
    double Pm[6] = {0, 0, 0, 0, 0, 0};
    if (active && grad_only) {
      if (tid == l0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { gmax = fmax(gmax, fabs(gp[k])); gn2 += gp[k] * gp[k]; }
      }
    } else if (active) {
      const bool head = (tid == l0);
      double s[3];
      const double vd[3] = {V[0], V[3], V[5]};
      if (p.init_scale) {
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] = p.jacobi ? 1.0 / (1.0 + sqrt(vd[k])) : 1.0;
        if constexpr (RES) { rl->sp[0] = s[0]; rl->sp[1] = s[1]; rl->sp[2] = s[2]; }      // (every lane of the point keeps its own copy)
        else if (head) { p.sp[3 * (size_t)pt] = s[0]; p.sp[3 * (size_t)pt + 1] = s[1]; p.sp[3 * (size_t)pt + 2] = s[2]; }
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] = s_pt[k];
      }
      double D2[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) D2[k] = fmin(fmax(s[k] * s[k] * vd[k], p.min_diag), p.max_diag) * p.inv_radius;
      // Vs = s V s + D^2 (LevenbergMarquardtStrategy diagonal on the Jacobi-scaled block), Cholesky inverse,
      // P = s Vs^-1 s  (the point block's inverse mapped back to unscaled coordinates)
      const double a00 = s[0] * s[0] * V[0] + D2[0], a01 = s[0] * s[1] * V[1], a02 = s[0] * s[2] * V[2];
      const double a11 = s[1] * s[1] * V[3] + D2[1], a12 = s[1] * s[2] * V[4], a22 = s[2] * s[2] * V[5] + D2[2];
      bool pd = a00 > 0.0;
      const double i00 = fast_rsqrt(a00);
      const double l10 = a01 * i00, l20 = a02 * i00;
      const double d1 = a11 - l10 * l10;
      pd = pd && d1 > 0.0;
      const double i11 = fast_rsqrt(d1);
      const double l21 = (a12 - l20 * l10) * i11;
      const double d2 = a22 - l20 * l20 - l21 * l21;
      pd = pd && d2 > 0.0;
      const double i22 = fast_rsqrt(d2);
      if (pd) {
        const double i10 = -l10 * i00 * i11;
        const double i21 = -l21 * i11 * i22;
        const double i20 = -(l20 * i00 + l21 * i10) * i22;
        const double v00 = i00 * i00 + i10 * i10 + i20 * i20;
        const double v01 = i10 * i11 + i20 * i21;
        const double v02 = i20 * i22;
        const double v11 = i11 * i11 + i21 * i21;
        const double v12 = i21 * i22;
        const double v22 = i22 * i22;
        Pm[0] = s[0] * s[0] * v00; Pm[1] = s[0] * s[1] * v01; Pm[2] = s[0] * s[2] * v02;
        Pm[3] = s[1] * s[1] * v11; Pm[4] = s[1] * s[2] * v12; Pm[5] = s[2] * s[2] * v22;
      } else if (head) {
        fail = 1;
      }
      if constexpr (RES) {
#pragma unroll
        for (int k = 0; k < 6; ++k) rl->pr[k] = Pm[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { rl->pr[6 + k] = gp[k]; rl->pr[9 + k] = D2[k]; }
      }
      if (head) {
        if constexpr (!RES) {
        double* pr = p.ptrec + 12 * (size_t)pt;
#pragma unroll
        for (int k = 0; k < 6; ++k) pr[k] = Pm[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { pr[6 + k] = gp[k]; pr[9 + k] = D2[k]; }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { gmax = fmax(gmax, fabs(gp[k])); gn2 += gp[k] * gp[k]; }
      }
    }
    // camera-side record [U_l | r_l | g_c,l] first; W_l = Ac^T M Ap (6x3) is only formed row by row here (r_l = g_c,l -
    // W_l (P g_p)) and again, together with Y_l = W_l P, when the pair goes to LDS after the camera-side sums: Ac, M Ap
    // and P are what stays in registers in between, and nothing of it survives into the pair-block phase, where the
    // register pressure peaks
    if (active && fa >= 0 && grad_only) {
      double* so = s_obs + tid * kObsStride;
#pragma unroll
      for (int j = 0; j < 6; ++j) so[27 + j] = -(Ac[0][j] * b[0] + Ac[1][j] * b[1]);
    } else if (active && fa >= 0) {
      double* so = s_obs + tid * kObsStride;
      const double Pg[3] = {Pm[0] * gp[0] + Pm[1] * gp[1] + Pm[2] * gp[2], Pm[1] * gp[0] + Pm[3] * gp[1] + Pm[4] * gp[2],
                            Pm[2] * gp[0] + Pm[4] * gp[1] + Pm[5] * gp[2]};
      // r_l = g_c,l - W_l (P g_p) = -Ac^T (b + (M Ap)(P g_p)): the 2-vector first, then one 2-term product per row
      // (forming the rows of W_l for it cost 36 more operations per observation)
      const double bq0 = b[0] + (MAp[0][0] * Pg[0] + MAp[0][1] * Pg[1] + MAp[0][2] * Pg[2]);
      const double bq1 = b[1] + (MAp[1][0] * Pg[0] + MAp[1][1] * Pg[1] + MAp[1][2] * Pg[2]);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        so[27 + j] = -(Ac[0][j] * b[0] + Ac[1][j] * b[1]);           // g_c,l = -Ac^T b
        so[21 + j] = -(Ac[0][j] * bq0 + Ac[1][j] * bq1);
      }
      // U_l = Ac^T M Ac (packed upper triangle)
      double MAc[2][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) { MAc[0][j] = M[0] * Ac[0][j] + M[1] * Ac[1][j]; MAc[1][j] = M[1] * Ac[0][j] + M[2] * Ac[1][j]; }
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) so[sym6(i, j)] = Ac[0][i] * MAc[0][j] + Ac[1][i] * MAc[1][j];
    }
    lds_barrier();
    PBA_TICK(3);
    if (!RES && tile + n_parts < p.n_tiles) {      // next tile's indices: two phases ahead of its coordinates
      const int2 r = p.lane_rec[(size_t)(tile + n_parts) * kTile + tid];
      nx_pt = r.x; nx_slot = r.y & 0xff; nx_l0 = (r.y >> 8) & 0xff; nx_cnt = (r.y >> 16) & 0xff;
    }

    // ---- P3b: camera-side sums of [U_l | r_l | g_c,l] by camera -------------------------------------------------
#pragma unroll
    for (int u = 0; u < kCamAcc; ++u) {
      const int e = tid + u * kTile;
      if (e < kCamVals * nf) {
        const int a = e / kCamVals, v = e - a * kCamVals;
        if (grad_only && v < 27) continue;
        for (int q0 = 0; q0 < n_pts; q0 += 16) {
          // 16 lane indices of camera a in one 128-bit read, then independent loads, then the adds in point order
          const int4 l4 = *reinterpret_cast<const int4*>(s_lane_of + a * kTile + q0);
          const int lw[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (q0 + 8 * h >= n_pts) continue;      // (uniform) nothing but sentinels in this half: 16-frame tiles hold 8 points
            // a camera without an observation of the point (byte 0x80 = lane 128, also every entry beyond n_pts after
            // the per-tile reset) reads row 128 = the zeroed head of s_red: no select on the address or on the loaded
            // value, and the eight reads stay in flight together (with a select on the value the compiler waited for
            // each read in turn)
            int off[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) off[k] = (int)((lw[2 * h + (k >> 2)] >> (8 * (k & 3))) & 0xff) * kObsStride + v;
            double x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = s_obs[off[k]];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc_cam[u] += x[k];
          }
        }
      }
    }
    lds_barrier();
    PBA_TICK(4);
    if (!RES && tile + n_parts < p.n_tiles && tid < ti_next.y) {      // next tile's point coordinates: one phase ahead
      nx_x[0] = p.xyz[3 * (size_t)nx_pt]; nx_x[1] = p.xyz[3 * (size_t)nx_pt + 1]; nx_x[2] = p.xyz[3 * (size_t)nx_pt + 2];
    }
    if (grad_only) continue;      // (uniform) the barrier above already separates this tile's reads from the next tile's stores
    if (active && fa >= 0) {
      // r4: the pair blocks are formed from the RANK-2 factors of W_l = Ac^T (M Ap) and Y_l = W_l P instead of from W | Y
      // themselves:  Y_a W_b^T = Ac_a^T (Q_a (M Ap)_b^T) Ac_b  with  Q = (M Ap) P  (2 x 3).  Per observation 22 doubles go to
      // LDS (Ac without the structural zeros of its translation part: rotation 2 x 3 | ju0 ju2 jv1 jv2; M Ap; Q) for 18 FMAs --
      // it was 36 values and 90 FMAs -- and a pair block costs 92 FMAs + 32 LDS reads instead of 108 + 36 (below).
      double* so = s_obs + tid * kObsStride;
#pragma unroll
      for (int k = 0; k < 3; ++k) { so[k] = Ac[0][k]; so[3 + k] = Ac[1][k]; }
      so[6] = Ac[0][3]; so[7] = Ac[0][5]; so[8] = Ac[1][4]; so[9] = Ac[1][5];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        so[10 + 3 * r] = MAp[r][0]; so[11 + 3 * r] = MAp[r][1]; so[12 + 3 * r] = MAp[r][2];
        so[16 + 3 * r] = MAp[r][0] * Pm[0] + MAp[r][1] * Pm[1] + MAp[r][2] * Pm[2];
        so[17 + 3 * r] = MAp[r][0] * Pm[1] + MAp[r][1] * Pm[3] + MAp[r][2] * Pm[4];
        so[18 + 3 * r] = MAp[r][0] * Pm[2] + MAp[r][1] * Pm[4] + MAp[r][2] * Pm[5];
      }
    }
    lds_barrier();
    PBA_TICK(5);

    // ---- P3a: block owners: T(a, b) -= Y_la W_lb^T over this group's points ----------------------------------
    if (owner) {
      const int8_t* la_row = s_lane_of + pa * kTile;
      const int8_t* lb_row = s_lane_of + pb * kTile;
      for (int q = grp; q < n_pts; q += n_groups) {
        const int la = la_row[q], lb = lb_row[q];
        if (la < 0 || lb < 0) continue;
        const double* Fa = s_obs + la * kObsStride;
        const double* Fb = s_obs + lb * kObsStride;
        double aa[10], qa[6], ab[10], mb[6];
#pragma unroll
        for (int k = 0; k < 10; ++k) { aa[k] = Fa[k]; ab[k] = Fb[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { qa[k] = Fa[16 + k]; mb[k] = Fb[10 + k]; }
        // N = Q_a (M Ap)_b^T (2 x 2)
        double N[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int t = 0; t < 2; ++t) N[r][t] = fma(qa[3 * r + 2], mb[3 * t + 2], fma(qa[3 * r + 1], mb[3 * t + 1], qa[3 * r] * mb[3 * t]));
        // Z = N Ac_b (2 x 6); Ac = [rotation (2 x 3) | ju0 0 ju2 ; 0 jv1 jv2]
        double Z[2][6];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
          for (int j = 0; j < 3; ++j) Z[r][j] = fma(N[r][1], ab[3 + j], N[r][0] * ab[j]);
          Z[r][3] = N[r][0] * ab[6];
          Z[r][4] = N[r][1] * ab[8];
          Z[r][5] = fma(N[r][1], ab[9], N[r][0] * ab[7]);
        }
        // T(a, b) -= Ac_a^T Z
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
    PBA_TICK(6);
  }
  if (p.dbg) tph[7] = __builtin_amdgcn_s_memrealtime() - t_rt0;     // tile loop, 100 MHz
  if (p.dbg && tid == 0) for (int k = 0; k < 8; ++k) p.dbg[part * (RES ? 16 : 8) + k] = tph[k];
  unsigned long long tep[6] = {0, 0, 0, 0, 0, 0};
#define PBA_TEP(k) do { if (RES && p.dbg) tep[k] = __builtin_amdgcn_s_memtime(); } while (0)
  PBA_TEP(0);
#undef PBA_TICK

  // ---- combine the point groups (fixed order), then per-block partials -----------------------------------
  // Partial layout (r4): entry e of workgroup b at ((e >> 4) * gridDim.x + b) * 16 + (e & 15) -- 16-entry (128-byte) chunks, all
  // workgroups' copies of one chunk contiguous: the reduction workgroup of a chunk then reads ONE sequential region of
  // gridDim.x x 128 bytes (it used to gather 128-byte pieces 9 - 37 KB apart: 2.4 TB/s), a store here still fills whole lines.
  auto out_at = [&](int e) -> double* { return p.partial + ((size_t)(e >> 4) * n_parts + part) * 16 + (e & 15); };
  // resident solve: the partials are consumed by other workgroups of the SAME launch -- every store is a write-through one; the idle
  // half of the last workgroup stores nothing
  auto put = [&](double* dst, double v, bool sc1) {
    if (RES) { if (has_tile) store_agent(dst, v); }
    else if (sc1) store_agent(dst, v);
    else *dst = v;
  };
  // r6: with many point groups (n_pairs <= 21: windows of up to seven frames) EVERY thread combines -- entry e = k n_pairs + pair of the
  // (entry-major) pair-block part of the partial, groups added in group order, i.e. the sums the owners of group 0 used to form alone:
  // at a 5-frame window that was 10 threads walking 11 x 36 dependent LDS reads each, 13.9 k cycles = 6 us at the end of every
  // workgroup (profiles/r06/resident_phase_trace.txt), twice the tile loop itself.  Same adds in the same order, so the same bits.
  const int n_ent = 36 * p.n_pairs;
  // (six or more groups = windows of up to seven frames; with fewer, the owners' own 36 x (n_groups - 1) adds are cheaper than the detour of
  // every accumulator through LDS: measured at configs[1], four groups, k_schur 38.9 -> 43.1 us with the spread form)
  const bool spread = n_groups >= 6;
  if (spread) {
    static_assert(36 * kTile <= kTile * kObsStride, "every owner's block fits the observation region");
    if (owner) {
      double* dst = s_obs + (size_t)grp * n_ent + pair;
#pragma unroll
      for (int k = 0; k < 36; ++k) dst[k * p.n_pairs] = acc[k];
    }
    lds_barrier();
    // (rolled: <= 18 entries per thread, 3 at a 5-frame window; the sums land in the slot of group 0)
#pragma unroll 1
    for (int e = tid; e < n_ent; e += kTile) {
      double v = s_obs[e];
#pragma unroll 1
      for (int g0 = 1; g0 < n_groups; g0 += 8) {
        double x[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) x[l] = s_obs[(size_t)min(g0 + l, n_groups - 1) * n_ent + e];
#pragma unroll
        for (int l = 0; l < 8; ++l) if (g0 + l < n_groups) v += x[l];
      }
      s_obs[e] = v;
    }
  } else {
    if (owner && grp > 0) {
      double* dst = s_obs + ((grp - 1) * p.n_pairs + pair) * 36;
#pragma unroll
      for (int k = 0; k < 36; ++k) dst[k] = acc[k];
    }
    lds_barrier();
    if (owner && grp == 0) {
      for (int g = 1; g < n_groups; ++g) {
        const double* src = s_obs + ((g - 1) * p.n_pairs + pair) * 36;
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] += src[k];
      }
    }
  }
  lds_barrier();
  PBA_TEP(1);
  // camera-side sums -> LDS: the owners of the diagonal blocks add U_a, the vector entries go out directly
  const int n = 6 * nf;
  double* s_cam = spread ? s_obs + n_ent : s_obs;                          // [nf][kCamVals] (spread: in the dead slot of group 1, 36 n_pairs >= 33 n_free doubles)
#pragma unroll
  for (int u = 0; u < kCamAcc; ++u) {
    const int e = tid + u * kTile;
    if (e < kCamVals * nf) {
      s_cam[e] = acc_cam[u];
      const int a = e / kCamVals, v = e - a * kCamVals;
      if (v >= 27) put(out_at(36 * p.n_pairs + n + 6 * a + (v - 27)), acc_cam[u], false);            // g_c
      else if (v >= 21) put(out_at(36 * p.n_pairs + 6 * a + (v - 21)), acc_cam[u], false);          // rhs
      else {
        // packed upper triangle: entry v is a diagonal (i, i) iff v == sym6(i, i) = 6 i - i (i - 1) / 2
#pragma unroll
        for (int i = 0; i < 6; ++i) if (v == 6 * i - (i * (i - 1)) / 2) put(out_at(36 * p.n_pairs + 2 * n + 6 * a + i), acc_cam[u], false);   // diag(U)
      }
    }
  }
  lds_barrier();
  PBA_TEP(2);
  if (spread) {
#pragma unroll 1
    for (int e = tid; e < n_ent; e += kTile) {
      const int k = e / p.n_pairs, pe = e - k * p.n_pairs;
      int a = 0, rem = pe;
      while (rem >= nf - a) { rem -= nf - a; ++a; }      // pairs enumerated row by row: diagonal iff rem == 0
      double v = s_obs[e];
      if (rem == 0) { const int i = k / 6, j = k - 6 * i; v += s_cam[a * kCamVals + sym6(i, j)]; }
      put(out_at(PBA_PARTIAL_T ? e : pe * 36 + k), v, PBA_PARTIAL_SC1 != 0);
    }
  } else if (owner && grp == 0) {
    if (pa == pb) {
      const double* U = s_cam + pa * kCamVals;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[6 * i + j] += U[sym6(i, j)];
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) {
      put(out_at(PBA_PARTIAL_T ? k * p.n_pairs + pair : pair * 36 + k), acc[k], PBA_PARTIAL_SC1 != 0);
    }
  }
  lds_barrier();
  PBA_TEP(3);
  // block reductions of the point-gradient statistics (butterfly per wave, then the two waves in order)
  {
    const double g2 = wave_sum(gn2), gm = wave_max(gmax), fl = wave_max((double)fail);
    if ((tid & 63) == 0) { s_red[(tid >> 6) * 3] = g2; s_red[(tid >> 6) * 3 + 1] = gm; s_red[(tid >> 6) * 3 + 2] = fl; }
    lds_barrier();
    if (tid == 0) {
      double a = 0.0, m = 0.0, f = 0.0;
      for (int w = 0; w < kTile / 64; ++w) { a += s_red[3 * w]; m = fmax(m, s_red[3 * w + 1]); f = fmax(f, s_red[3 * w + 2]); }
      put(out_at(36 * p.n_pairs + 3 * n + 0), m, false);
      put(out_at(36 * p.n_pairs + 3 * n + 1), a, false);
      put(out_at(36 * p.n_pairs + 3 * n + 2), f, false);
      if (!RES && p.stamp && part < kStampSchurBlocks) p.stamp[kStampSchur0 + part] = __builtin_amdgcn_s_memrealtime();
    }
  }
  PBA_TEP(4);
  if (RES && p.dbg && tid == 0) for (int k = 0; k < 4; ++k) p.dbg[part * 16 + 8 + k] = tep[k + 1] - tep[k];
#undef PBA_TEP
}

__global__ __launch_bounds__(kTile, 2) void k_schur(SchurParams p_in) {
  SchurParams p = p_in;
  if (!PBA_PHASE_TIMING) p.dbg = nullptr;
  if (p.pub_host_seq && blockIdx.x == 0)
    lm_publish(p.lm, p.pub_state, p.pub_scal, p.pub_host_scal, p.pub_host_seq, p.pub_seq, threadIdx.x, kTile);
  if (p.lm) {
    if (p.lm->done && !p.final_pass) return;
    if (p.final_pass && !lm_final_pass_needed(p.lm)) return;
    if (p.lm->cur != p.enq_cur) { p.xyz = p.xyz_alt; p.geom = p.geom_alt; p.rec = p.rec_alt; }
    p.radius = p.lm->radius;
    p.inv_radius = 1.0 / p.radius;
  }
  __shared__ __attribute__((aligned(16))) char smem[kSchurSmemBytes];
  schur_body<const void>(p, smem, (int)threadIdx.x, (int)blockIdx.x, (int)gridDim.x, nullptr, nullptr, true);
}

}  // namespace pba

// reduction of the Schur partials + the reduced camera solve (k_reduce_final, k_reduce_solve, k_solve_blocked, k_solve_generic)
#include "pba_solve.h"

namespace pba {

// =====================================================================================================
// back-substitution (SchurEliminator::BackSubstitute): one lane per point
// =====================================================================================================
struct BacksubParams {
  const double* xyz;
  const double* rays;        // inverse-depth variant (point_world), else null
  double* xyz_cand;
  const CamGeom* geom;
  const double* rec;
  const int32_t* pt_begin;
  const uint8_t* obs_slot;
  const double* sp;
  const double* ptrec;
  const double* delta_c;
  double* block_out;     // [gridDim.x][3]: mcc, step^2, x^2
  const double* cams_cand;   // candidate cameras (k_solve) -> geometry for the candidate pass, by workgroup 0
  CamGeom* geom_cand;
  int64_t rec_stride;
  int32_t n_points, n_frames, fixed_slot;
  double fx, fy;
};

__global__ __launch_bounds__(256) void k_backsub(BacksubParams p) {
  __shared__ double s_red[3][256];
  const int tid = threadIdx.x;
  const int pt = blockIdx.x * 256 + tid;
  if (p.geom_cand && blockIdx.x == gridDim.x - 1 && tid >= 256 - p.n_frames) cam_geom_one(p.cams_cand, p.geom_cand, 255 - tid, p.fixed_slot);
  double mcc = 0.0, st2 = 0.0, x2 = 0.0;
  if (pt < p.n_points) {
    const double X[3] = {p.xyz[3 * (size_t)pt], p.xyz[3 * (size_t)pt + 1], p.xyz[3 * (size_t)pt + 2]};   // parameters
    double Xw[3], qd[3];
    point_world(p.rays, pt, X, Xw, qd);
    double acc[3] = {0, 0, 0};
    for (int o = p.pt_begin[pt]; o < p.pt_begin[pt + 1]; ++o) {
      const int slot = p.obs_slot[o];
      const CamGeom& g = p.geom[slot];
      if (g.free_index < 0) continue;
      double xw[3], Ac[2][6], Ap[2][3];
      transform_point(g, Xw, xw);
      projection_jacobians(g, Xw, xw, p.fx, p.fy, Ac, Ap);
      point_jacobian(p.rays, qd, Ap);
      const double* dc = p.delta_c + 6 * slot;
      double t0 = 0.0, t1 = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { t0 += Ac[0][k] * dc[k]; t1 += Ac[1][k] * dc[k]; }
      const double m0 = p.rec[0 * p.rec_stride + o], m1 = p.rec[1 * p.rec_stride + o], m2 = p.rec[2 * p.rec_stride + o];
      const double u0 = m0 * t0 + m1 * t1, u1 = m1 * t0 + m2 * t1;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[k] += Ap[0][k] * u0 + Ap[1][k] * u1;   // W_l^T delta_c
    }
    const double* pr = p.ptrec + 12 * (size_t)pt;
    const double q0 = pr[6] + acc[0], q1 = pr[7] + acc[1], q2 = pr[8] + acc[2];
    const double d0 = -(pr[0] * q0 + pr[1] * q1 + pr[2] * q2);
    const double d1 = -(pr[1] * q0 + pr[3] * q1 + pr[4] * q2);
    const double d2 = -(pr[2] * q0 + pr[4] * q1 + pr[5] * q2);
    const double d[3] = {d0, d1, d2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double s = p.sp[3 * (size_t)pt + k];
      const double yk = -d[k] / s;                       // step in Jacobi-scaled coordinates is -y
      mcc += 0.5 * yk * (s * pr[6 + k]) + 0.5 * pr[9 + k] * yk * yk;
      st2 += d[k] * d[k];
      x2 += X[k] * X[k];
      p.xyz_cand[3 * (size_t)pt + k] = X[k] + d[k];
    }
  }
  s_red[0][tid] = mcc; s_red[1][tid] = st2; s_red[2][tid] = x2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { s_red[0][tid] += s_red[0][tid + s]; s_red[1][tid] += s_red[1][tid + s]; s_red[2][tid] += s_red[2][tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { p.block_out[3 * blockIdx.x] = s_red[0][0]; p.block_out[3 * blockIdx.x + 1] = s_red[1][0]; p.block_out[3 * blockIdx.x + 2] = s_red[2][0]; }
}

// Fixed-order sums of the back-substitution and candidate-pass block partials into the scalar block; with
// `publish` the whole block is also written to host-mapped memory followed by a sequence number the host polls.
__global__ __launch_bounds__(256) void k_finalize_step(const double* __restrict__ bs_out, int n_bs_blocks,
                                                        const double* __restrict__ block_cost,
                                                        const int32_t* __restrict__ block_fail, int n_cost_blocks,
                                                        double* __restrict__ scal, double* host_scal,
                                                        unsigned long long* host_seq, unsigned long long seq) {
  __shared__ double s_red[4][256];
  __shared__ int s_f[256];
  const int tid = threadIdx.x;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0; int f = 0;
  for (int b = tid; b < n_bs_blocks; b += 256) { a0 += bs_out[3 * b]; a1 += bs_out[3 * b + 1]; a2 += bs_out[3 * b + 2]; }
  for (int b = tid; b < n_cost_blocks; b += 256) { a3 += block_cost[b]; f |= block_fail[b]; }
  s_red[0][tid] = a0; s_red[1][tid] = a1; s_red[2][tid] = a2; s_red[3][tid] = a3; s_f[tid] = f;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      for (int k = 0; k < 4; ++k) s_red[k][tid] += s_red[k][tid + s];
      s_f[tid] |= s_f[tid + s];
    }
    __syncthreads();
  }
  if (tid == 0) {
    scal[kMccPts] = s_red[0][0]; scal[kStep2Pts] = s_red[1][0]; scal[kX2Pts] = s_red[2][0];
    scal[kCandCost] = s_red[3][0]; scal[kEvalFailCand] = (double)s_f[0];
  }
  if (host_scal) {
    __syncthreads();
    publish_scal(scal, host_scal, tid, blockDim.x);
    publish_seq(host_seq, seq, tid);
  }
}

// End of an asynchronous solve: device iteration log -> host-mapped log, then state, scalars and sequence number.
__global__ void k_flush(const LmState* lm, LmState* host_state, const double* scal, double* host_scal,
                        const pba_iteration_summary* log, pba_iteration_summary* host_log, int max_log,
                        unsigned long long* host_seq, unsigned long long seq) {
  flush_to_host(lm, host_state, scal, host_scal, log, host_log, max_log, host_seq, seq, threadIdx.x, blockDim.x);
}

// Publishes the (already reduced) scalar block to host-mapped memory: used after the multi-rank all-reduces and
// for gradient-only steps.
__global__ void k_publish(double* __restrict__ scal, double* host_scal, unsigned long long* host_seq,
                          unsigned long long seq, const double* xchg, int world) {
  const int tid = threadIdx.x;
  if (xchg) {
    if (tid == 0) xchg_unpack(xchg, scal, world);
    __syncthreads();
  }
  publish_scal(scal, host_scal, tid, blockDim.x);
  publish_seq(host_seq, seq, tid);
}

}  // namespace pba
