// pba_device.h -- device-side data layout shared by the kernels and the engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pba.h"

namespace pba {

constexpr int kMaxFrames = 16;
constexpr int kMaxRadius = 5;

// Packed frame texel (one u32 per pixel), bit-exact for u8 frames:
//   bits  0..7   I                      (photobundle.cc:231: image.cast<float>() -> integers 0..255)
//   bits  8..17  2*Gx  two's complement (imgproc.cc:38: 0.5f * (I[x+1] - I[x-1]) -> multiples of 0.5 in +-127.5)
//   bits 18..27  2*Gy  two's complement (imgproc.cc:39)
// Border rows/columns carry zero gradients (imgproc.cc:34-35, 42-43, 79-80, 93-94).
__host__ __device__ inline uint32_t pack_texel(int I, int gx2, int gy2) {
  return (uint32_t)(I & 0xff) | ((uint32_t)(gx2 & 0x3ff) << 8) | ((uint32_t)(gy2 & 0x3ff) << 18);
}

// Per-camera geometry, recomputed whenever the cameras change (k_cam_geom).
//   value path   : the exact operation order of ceres::AngleAxisRotatePoint (call site photobundle.cc:700)
//   R[9]         : row-major d(xw)/d(point)   (Rodrigues matrix, or I + [w]x in the small-angle branch)
//   dR[3][9]     : row-major d(xw)/d(w_k) = dR[k] * point   (R [B_k]x, B = (w w^T + (R^T - I)[w]x) / theta^2;
//                  [e_k]x in the small-angle branch, i.e. the derivative of the code as written)
struct CamGeom {
  double aa[3];
  double t[3];
  double w[3];     // aa * (1/theta)
  double ct, st;   // cos(theta), sin(theta)
  double R[9];
  double dR[27];
  int32_t rodrigues;  // theta^2 > DBL_EPSILON
  int32_t is_free;    // 0 for the constant camera
  int32_t free_index; // index among free cameras, -1 if constant
  int32_t pad;
};

// Device scalar block (Engine::d_scal), grouped so that the multi-rank transports can reduce slices in place.
enum Scal {
  // --- group B, SUM over ranks after the cost pass -------------------------------------------------
  kCandCost = 0,    // candidate cost (local shard)
  kMccPts,          // point part of the model cost change
  kStep2Pts,        // sum delta_p^2
  kX2Pts,           // sum xyz^2 at the current point
  kSumBCount = 4,
  // --- group M, MAX over ranks ----------------------------------------------------------------------
  kGmaxPts = 8,     // max |g_p|
  kSchurFail,       // > 0: a damped point block was not PD
  kEvalFailLin,     // > 0: non-finite residual block in the Jacobian pass
  kEvalFailCand,    // > 0: non-finite residual block in the cost pass
  kMaxCount = 4,
  // --- replicated (identical on every rank, never reduced) --------------------------------------------
  kMccCams = 16,
  kStep2Cams,
  kX2Cams,
  kGmaxCams,
  kGnorm2Cams,
  kSolveOk,         // reduced-system Cholesky succeeded and the camera step is finite
  kCostLin,         // GLOBAL cost at the linearisation point (copied out of the reduced packed buffer)
  kGnorm2Pts,       // GLOBAL sum g_p^2
  kNumScal = 32
};

// ---- device-resident Levenberg-Marquardt state (asynchronous driver) -------------------------------------------
// The trust-region decisions of pba_lm.cpp (Ceres TrustRegionMinimizer / LevenbergMarquardtStrategy) evaluated by
// the LAST workgroup of the candidate pass, so that the host can enqueue iterations back to back without a round
// trip per step.  The host still owns the loop: it enqueues, watches `done`, and reads the iteration log.
enum LmTermination { kLmRunning = 0, kLmMaxIterations, kLmGradientTolerance, kLmMinRadius, kLmParameterTolerance,
                     kLmFunctionTolerance, kLmInvalidSteps, kLmEvalFailure };

struct LmState {
  double radius, decrease_factor, x_cost, minimum_cost, initial_cost;
  double last_value[2];        // termination detail (e.g. step norm ratio)
  int32_t cur;                 // parity of the current point
  int32_t iteration;           // iterations completed (log entries written = n_log)
  int32_t done;                // LmTermination
  int32_t num_invalid, num_successful, num_unsuccessful;
  int32_t pending_grad;        // log index still waiting for the gradient norms of its (accepted) point, -1 none
  int32_t n_log;
  int32_t first;               // 1 until iteration 0 has been logged
  int32_t pad;
  // options
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double max_radius, min_radius, min_relative_decrease;
  int32_t max_num_iterations, max_invalid;
  unsigned long long done_seq;   // sequence number of the step that terminated the solve (0 while running)
};

}  // namespace pba
