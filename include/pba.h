/*
 * pba.h -- C-ABI of the MI355X photometric bundle-adjustment engine (libpba_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path the engine replaces: the inner seam of
 * PhotometricBundleAdjustment::optimize(), reference src/photobundle.cc:784-829, i.e.
 *
 *     ceres::Problem problem;                                            (:784)
 *     problem.AddResidualBlock(DescriptorError::Create(...), loss, camera_ptr, xyz);   (:801-802)
 *     problem.SetParameterBlockConstant(first_camera);                   (:809-813)
 *     ceres::Solve(GetSolverOptions(...), &problem, &summary);           (:829)
 *
 * plus the per-frame plane producer feeding it (DescriptorFrame, :151-257 / imgproc.cc:27-95).
 * Everything is `extern "C"`, plain pointers and sizes, no C++/torch types.  Functions return 0 on
 * success or a negative pba_status; they never throw.  All host buffers are caller-owned and are
 * only read/written during the call; all device memory is engine-owned.  One host thread per handle.
 *
 * Data conventions (identical to what the reference hands to Ceres):
 *   cameras  : 6 doubles per window slot = angle-axis (3) + translation (3) of the WORLD->CAMERA
 *              transform (photobundle.cc:646-656 PoseToParams of the inverted pose, :774-778)
 *   points   : 3 doubles, world XYZ (photobundle.cc:795)
 *   desc     : C (2R+1)^2 doubles per point: the row-major patch of every channel, channel-major (photobundle.cc:466-479,
 *              :597-603); values are float casts of pixels, stored as fp32 on the device
 *   obs      : one residual block per (point, slot) entry, grouped by point (photobundle.cc:791-804)
 *   weights  : (2R+1)^2 doubles (photobundle.cc:617-644)
 */
#ifndef PBA_H
#define PBA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBA_MAX_FRAMES 16
#define PBA_MAX_RADIUS 5
#define PBA_MAX_CHANNELS 8

typedef struct pba_engine pba_engine;

typedef enum pba_status {
  PBA_OK = 0,
  PBA_ERR_INVALID = -1,     /* bad argument */
  PBA_ERR_HIP = -2,         /* HIP runtime error (see pba_last_error) */
  PBA_ERR_NO_DEVICE = -3,   /* no gfx950 device visible: there is NO CPU fallback */
  PBA_ERR_STATE = -4,       /* call order violated (e.g. solve before set_problem) */
  PBA_ERR_COMM = -5,        /* collective transport error */
  PBA_ERR_NUMERIC = -6      /* non-finite evaluation at the initial point */
} pba_status;

/* Replaces the constructor arguments of PhotometricBundleAdjustment (photobundle.h:148) that matter on the
 * hot path: Calibration (calibration.h:19-26), ImageSize (types.h:57-63), Options::patchRadius /
 * slidingWindowSize / robustThreshold (photobundle.h:26-85). */
typedef struct pba_config {
  int32_t rows, cols;        /* image size */
  int32_t max_frames;        /* window slots (Options::slidingWindowSize), <= PBA_MAX_FRAMES */
  int32_t radius;            /* Options::patchRadius, 1..PBA_MAX_RADIUS */
  double fx, fy, cx, cy;     /* pinhole intrinsics */
  double huber;              /* Options::robustThreshold; <= 0 disables the loss (photobundle.cc:797-798) */
  int32_t device;            /* HIP device ordinal */
  int32_t flags;             /* bit 0: keep a copy of the reduced system for pba_get_reduced_system (test hook);
                              * bits 1-2: sampler precision for the BASELINE configs[4] tolerance sweep: 0 = exact
                              * restatement of sample_eigen.h:82-101 (default, the only mode with reference parity),
                              * 1 = fp32 interpolation and accumulation, 2 = fp32 with bf16-rounded residual/gradient
                              * operands.  Modes 1-2 need unit patch weights. */
  int32_t channels;          /* descriptor channels C (Options::descriptorType, photobundle.cc:229-245): 0 or 1 =
                              * Intensity (frames arrive as u8 through pba_set_frame_u8); 2..PBA_MAX_CHANNELS = float
                              * channel images through pba_set_frame_channels_f32 (IntensityAndGradient: 3, BitPlanes: 8).
                              * Descriptors then hold C patches per point, channel-major (photobundle.cc:597-603). */
  int32_t reserved;
} pba_config;

/* ceres::Solver::Options as configured by GetSolverOptions (photobundle.cc:738-761) + the Ceres defaults
 * that shape the path.  pba_default_solver_options fills the reference's values. */
typedef struct pba_solver_options {
  int32_t max_num_iterations;            /* 500  (photobundle.cc:751) */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  double function_tolerance;             /* 1e-6 (photobundle.cc:756) */
  double gradient_tolerance;             /* 1e-6 (:757) */
  double parameter_tolerance;            /* 1e-6 (:758) */
  double initial_trust_region_radius;    /* 1e4  */
  double max_trust_region_radius;        /* 1e16 */
  double min_trust_region_radius;        /* 1e-32 */
  double min_relative_decrease;          /* 1e-3 */
  double min_lm_diagonal;                /* 1e-6 */
  double max_lm_diagonal;                /* 1e32 */
  int32_t jacobi_scaling;                /* 1 */
  int32_t verbose;                       /* minimizer_progress_to_stdout (photobundle.cc:750) */
} pba_solver_options;

/* ceres::IterationSummary: the 18 fields the reference serialises (ceres_cereal.h:13-30), same names. */
typedef struct pba_iteration_summary {
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_nonmonotonic;
  int32_t step_is_successful;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double gradient_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
  double eta;
  double step_size;
  int32_t line_search_function_evaluations;
  int32_t line_search_gradient_evaluations;
  int32_t line_search_iterations;
  int32_t linear_solver_iterations;
  double iteration_time_in_seconds;
  double step_solver_time_in_seconds;
  double cumulative_time_in_seconds;
  double model_cost_change;   /* extra (parity debugging) */
  double candidate_cost;      /* extra */
} pba_iteration_summary;

/* ceres::Solver::Summary subset consumed at photobundle.cc:867-874. */
typedef struct pba_solver_summary {
  double initial_cost, final_cost, fixed_cost;
  int32_t num_successful_steps, num_unsuccessful_steps;
  int32_t num_iterations;            /* entries written to the iterations array */
  int32_t num_residuals;             /* global (all ranks) */
  int32_t num_residual_blocks;       /* global (all ranks) */
  int32_t termination_type;          /* 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE */
  double total_time_in_seconds;
  int64_t num_jacobian_passes;       /* linearisations (residual + Jacobian) */
  int64_t num_cost_passes;           /* candidate (residual-only) evaluations */
  int64_t num_resolve_passes;        /* extra damped Schur solves after rejected / invalid steps */
  char message[256];
} pba_solver_summary;

/* Output of one trust-region step attempt (pba_step). */
typedef struct pba_step_info {
  double cost;               /* cost at the current linearisation point */
  double gradient_max_norm;  /* |J^T r|_inf at the current point (unscaled) */
  double gradient_norm;
  double model_cost_change;
  double step_norm;          /* |delta| unscaled */
  double x_norm;             /* |x| over the free cameras that have residual blocks and all points, at the current point */
  double candidate_cost;
  int32_t linear_solver_ok;  /* 0: LINEAR_SOLVER_FAILURE (non-PD block / non-finite step) */
  int32_t eval_ok;           /* 0: candidate evaluation non-finite */
} pba_step_info;

/* Device-side accounting for bench.py / rocprof cross-checks. */
typedef struct pba_counters {
  double linearize_ms;       /* sum of HIP-event durations of the Jacobian-pass kernel on the engine stream */
  double cost_ms;            /* ... of the cost-pass kernel */
  double schur_ms;           /* ... of the Schur elimination kernel */
  int64_t linearize_launches, cost_launches, schur_launches;
  int64_t n_obs, n_points;   /* local shard */
  double solve_ms;           /* ... of the reduction of the Schur partials + reduced camera solve (k_reduce_solve, or k_reduce_final
                                + k_solve_blocked at more than one rank, without the exchange between them) */
  double exchange_ms;        /* ... of the two per-step exchanges (all-reduce of the packed reduced system + of the step scalars):
                                RCCL collectives or the peer-exchange kernels; 0 at a single rank */
  int64_t solve_launches, exchange_launches;
} pba_counters;

const char* pba_status_string(int status);
const char* pba_last_error(const pba_engine* e);
void pba_default_solver_options(pba_solver_options* o);

/* ---- lifetime ------------------------------------------------------------------------------------------ */
/* Allocates the device state for cfg->max_frames frames of cfg->rows x cfg->cols, loads the library's code object and
 * runs one frame-sized host -> device transfer, so that the first pba_set_frame_* / pba_solve of a process see the
 * steady-state latency (the runtime's lazy set-up costs ~10 ms each otherwise).  ~30 ms. */
int pba_create(const pba_config* cfg, pba_engine** out);
void pba_destroy(pba_engine* e);

/* ---- frames: replaces DescriptorFrame::Create + ImageGradient::compute (photobundle.cc:225-248, :126-135) --
 * `image` is the dense row-major rows x cols u8 frame addFrame() receives (photobundle.h:160).  The engine
 * builds its device plane (I, Gx, Gy bit-exactly as imgproc.cc:27-95) and keeps it in window slot `slot`. */
int pba_set_frame_u8(pba_engine* e, int slot, const uint8_t* image);
/* Multi-channel descriptors (pba_config.channels = C > 1): `channels` holds the C float channel images DescriptorFrame::
 * Create produces (photobundle.cc:225-248), [C][rows*cols] row-major; the engine adds every channel's own gradient
 * images (DescriptorFrame ctor :172-175, imgproc.cc:27-95). */
int pba_set_frame_channels_f32(pba_engine* e, int slot, int32_t n_channels, const float* channels);
/* Debug/test readback of the device planes as float I, Gx, Gy (each rows*cols). */
int pba_get_frame_planes(pba_engine* e, int slot, float* I, float* Gx, float* Gy);
/* ... and of channel `channel` of a multi-channel slot: value, Gx, Gy (each rows*cols). */
int pba_get_frame_channel(pba_engine* e, int slot, int32_t channel, float* I, float* Gx, float* Gy);
/* Debug/test: the engine's sampler (SampleLinear, sample_eigen.h:56-102, in the reference's mixed float/double arithmetic)
 * at n float positions (y[i], x[i]) of channel `channel` (0 on a single-channel engine) of the frame in `slot`;
 * out3[3 i .. 3 i + 2] = value, Gx, Gy. */
int pba_sample_frame(pba_engine* e, int slot, int32_t channel, int32_t n, const float* y, const float* x, float* out3);

/* ---- device-side producers (round 3): the channel images and the pyramid without the host round trip --------------
 * pba_set_frame_descriptor_u8: DescriptorFrame::Create on the device (photobundle.cc:225-248).  `descriptor` selects
 * the channels built from the u8 frame: PBA_DESCRIPTOR_INTENSITY (= pba_set_frame_u8), _INTENSITY_AND_GRADIENT (I, Gx,
 * Gy as float: photobundle.cc:229-235, imgproc.cc:27-95; engine created with channels = 3), _BITPLANES (census of the
 * 3x3-smoothed frame, its eight bit planes smoothed 5x5: imgproc.cc:109-245 with sigma_ct / sigma_bp of imgproc.h:44-46,
 * <= 0 skips the smoothing; channels = 8).  Bit-identical to pba_set_frame_channels_f32 fed with the host's channel images
 * (photobundle_amd/host/imgproc.h), at 1/12 .. 1/32 of the upload. */
enum { PBA_DESCRIPTOR_INTENSITY = 0, PBA_DESCRIPTOR_INTENSITY_AND_GRADIENT = 1, PBA_DESCRIPTOR_BITPLANES = 2 };
int pba_set_frame_descriptor_u8(pba_engine* e, int slot, const uint8_t* image, int32_t descriptor, float sigma_ct, float sigma_bp);
/* The channel images of a multi-channel slot back on the host, [C][rows*cols] (the layout pba_set_frame_channels_f32
 * takes): a front-end that needs them (saliency over all channels, descriptor patches: photobundle.cc:213-221, :466-479)
 * reads the device-produced ones instead of running DescriptorFrame::Create on the CPU as well. */
int pba_get_frame_channels_f32(pba_engine* e, int slot, float* channels);
/* cv::pyrDown of the frame in slot `finer_slot` of engine `finer` into slot `slot` of `e` (photobundle_pyramid.cc:45-52:
 * the next pyramid level; `e` must have been created for ((rows+1)/2, (cols+1)/2) on the same device), device to device.
 * `image_out` (nullable, (rows+1)/2 * (cols+1)/2 bytes) receives the down-sampled frame for the host front-end; the
 * call is asynchronous when it is null.  The depth map never enters the engine (it only initialises points on the host,
 * photobundle.cc:540-560), so cv::resize of the depth (:54-56) stays with the caller. */
int pba_set_frame_pyr_down(pba_engine* e, int slot, pba_engine* finer, int finer_slot, uint8_t* image_out);

/* ---- device front-end (round 4): the per-frame work of PhotometricBundleAdjustment::addFrame on the frame that already sits in
 * the engine (reference src/photobundle.cc:505-603), so that neither the float planes nor the channel images travel back:
 *   pba_frontend_visibility   ZNCC test of the tracked points (ZnccPatch_<2, float>, :315-361, :512-542) on the u8 frame MOST
 *                             RECENTLY handed to pba_set_frame_u8 / pba_set_frame_descriptor_u8 (or produced by
 *                             pba_set_frame_pyr_down).  uv [n][2]: projections K (T_c X) of the points the caller already found
 *                             inside the border (:519-523), rc [n][2]: their rounded (row, column), patches [n][26]: the stored
 *                             zero-mean 5x5 patch and its norm.  hit[i] = score > min_score; the (2 mask_radius + 1)^2 block around
 *                             every hit leaves the engine's selection mask (:536-538), which the call first resets.  n may be 0.
 *                             PBA_ERR_STATE when the frame uploaded last left no u8 image behind (pba_set_frame_channels_f32, or no
 *                             upload at all): the ZNCC would otherwise score a stale image.
 *   pba_frontend_candidates   saliency map of the frame in `slot` (sum over the channels of |Ix| + |Iy|, :213-221), then every pixel
 *                             of [border, rows - border - 1) x [border, cols - border - 1) with min_depth <= depth <= max_depth
 *                             that is unmasked and a STRICT local maximum of the saliency over (2 nms_radius + 1)^2 (:555-573,
 *                             src/imgproc.h:176-212; nms_radius <= 0: every valid-depth pixel).  depth: rows * cols floats of the
 *                             caller (borrowed for the call).  *n_out = number of candidates, kept on the device in the row-major
 *                             order of the reference's scan; PBA_ERR_INVALID for nms_radius > border or a border that leaves no
 *                             interior (2 border >= rows or cols);
 *   pba_frontend_get_candidates   copies the first n of them out (the caller selects the maxNumPoints most salient, :578-585);
 *   pba_frontend_descriptors  ExtractPatch (:466-479, :597-603) at n integer pixels xy [n][2] = (x, y): desc [n][C][(2 R + 1)^2]
 *                             channel values as float (exact: the reference stores the same floats widened to double).
 * Same arithmetic, type by type and in the same order, as the host restatement in photobundle_amd/host/photobundle.cc: the
 * class produces byte-identical trajectories with either (PBA_HOST_FRONTEND=1 selects the host one). */
typedef struct pba_candidate { float saliency; int32_t x, y; } pba_candidate;
int pba_frontend_visibility(pba_engine* e, int32_t n, const double* uv, const int32_t* rc, const float* patches26, double min_score,
                            int32_t mask_radius, uint8_t* hit);
int pba_frontend_candidates(pba_engine* e, int32_t slot, const float* depth, double min_depth, double max_depth, int32_t nms_radius,
                            int32_t border, int32_t* n_out);
int pba_frontend_get_candidates(pba_engine* e, pba_candidate* out, int32_t n);
int pba_frontend_descriptors(pba_engine* e, int32_t slot, int32_t n, const int32_t* xy, float* desc);
/* Test hook: what pba_frontend_visibility computes for point i, out27[27 i ..] = the zero-mean patch of the new frame at uv (25),
 * its norm, the score against patches26[i] (-1 when the product of the norms is <= 1e-6).  No mask, no flags. */
int pba_frontend_zncc_probe(pba_engine* e, int32_t n, const double* uv, const float* patches26, float* out27);

/* ---- problem: replaces the AddResidualBlock loop (photobundle.cc:786-806) -------------------------------
 * Call order: pba_set_frame_u8 (every slot the observation list uses), pba_set_problem and pba_set_cameras may come in
 * any order, all three before pba_linearize / pba_solve; a pass that finds one of them missing, an observation whose
 * slot is >= n_frames, or a slot without an uploaded frame returns PBA_ERR_STATE (nothing is launched).
 * Limit: at most 15 FREE cameras (n_frames - 1 with a constant camera, i.e. 16 window slots need fixed_slot >= 0);
 * pba_set_cameras returns PBA_ERR_INVALID beyond that. */
int pba_set_problem(pba_engine* e, int32_t n_points, const double* xyz, const double* desc,
                    int32_t n_obs, const int32_t* obs_point, const int32_t* obs_slot, const double* weights);
/* cams6: [n_frames][6]; fixed_slot: SetParameterBlockConstant (photobundle.cc:809-813), -1 for none. */
int pba_set_cameras(pba_engine* e, const double* cams6, int32_t n_frames, int32_t fixed_slot);
/* Inverse-depth variant of the point parameterisation (named by the project's north star; the REFERENCE optimises free
 * world points, photobundle.cc:692/:795, so this mode has no reference counterpart and is outside every parity claim).
 * Call after pba_set_problem: point i then lives on the fixed world ray rays6[i] = {origin (3), direction (3)} with the
 * single free parameter rho[i] > 0, X_i = origin + direction / rho_i.  pba_get_state afterwards returns the parameters
 * (rho, 0, 0) in `xyz`; pba_get_points_world returns world coordinates in either mode.  pba_set_problem switches the
 * mode off again. */
int pba_set_inverse_depth(pba_engine* e, const double* rays6, const double* rho);
int pba_get_points_world(pba_engine* e, double* xyz);
/* Current (best) state; either pointer may be NULL. */
int pba_get_state(pba_engine* e, double* cams6, double* xyz);

/* ---- primitive passes used by the host LM driver (exposed for tests and alternative drivers) ------------- */
/* Jacobian pass at the current point: residuals, analytic per-patch structure tensors, loss correction. */
int pba_linearize(pba_engine* e, double* cost);
/* Damped Schur solve with trust-region `radius` from the stored linearisation, back-substitution, candidate
 * point and its cost (cost pass).  init_scale != 0 (iteration 0) also fixes the Jacobi scaling vector. */
int pba_step(pba_engine* e, double radius, int32_t init_scale, const pba_solver_options* o, pba_step_info* out);
/* Make the candidate the current point (the caller then calls pba_linearize). */
int pba_accept(pba_engine* e);
/* Test hooks: copies of the reduced (global) system of the LAST pba_step (after pba_solve: of the last FULL iteration -- the
 * gradient-only pass that ends a solve at the iteration limit leaves them alone): n = 6 * free cameras.
 * S[n*n] row-major = s_c (U - sum W P W^T) s_c + D_c^2, rhs[n], both in Jacobi-scaled space. */
int pba_get_reduced_system(pba_engine* e, double* S, double* rhs, int32_t* n);
/* Test hook: per-observation record of the last linearisation: [n_obs][6] = rho'*M11, M12, M22, rho'*b1, b2, rho/2. */
int pba_get_obs_records(pba_engine* e, double* rec6);

/* ---- the LM loop: replaces ceres::Solve (photobundle.cc:829) --------------------------------------------- */
int pba_solve(pba_engine* e, const pba_solver_options* o, pba_solver_summary* summary,
              pba_iteration_summary* iterations, int32_t max_iterations_out);

/* ---- multi-GPU: points are sharded across ranks, one all-reduce of the reduced camera system per solve --- */
/* RCCL transport (one process per GPU): rank 0 calls pba_comm_unique_id, broadcasts the 128 bytes out of band. */
int pba_comm_unique_id(void* id128);
int pba_comm_init_rccl(pba_engine* e, const void* id128, int32_t rank, int32_t world);
/* Host-staged transport: fn must sum `n` doubles in place across all ranks (op 0) or take the max (op 1). */
typedef int (*pba_allreduce_fn)(double* buf, int64_t n, int32_t op, void* ctx);
int pba_comm_init_callback(pba_engine* e, pba_allreduce_fn fn, void* ctx, int32_t rank, int32_t world);

/* Device-side exchange (optional, after pba_comm_init_rccl / pba_comm_init_callback, collective: every rank calls it):
 * the two per-step all-reduces become flag-and-slot reads of peer-mapped mailboxes (hipIpc*, fine-grained device memory)
 * inside small kernels on the engine's stream -- no collective launch per LM step, sums in rank order (bit-identical on all
 * ranks).  The base transport carries the IPC handles and stays the fallback: the call returns PBA_OK whether or not the
 * mapping succeeded (every rank takes the same decision); pba_comm_transport names what is in use ("rccl", "rccl+peer",
 * "callback", "callback+peer", "none").  Waits are bounded by PBA_WAIT_TIMEOUT_S -> PBA_ERR_COMM. */
int pba_comm_enable_peer_exchange(pba_engine* e);
const char* pba_comm_transport(const pba_engine* e);
/* Ranks the transport itself reports (ncclCommCount for RCCL, the callback's world otherwise; 1 without a transport). */
int pba_comm_rank_count(const pba_engine* e);

/* Per-kernel accounting.  pba_set_profiling mode 0: off.  1 (what pba_reset_counters switches on): HIP events around
 * every kernel of the host-stepped driver -- each bracket ends with an event record (a cache write-back the pipelined run
 * does not pay) and the exchanges are timed too.  2: device time stamps inside the ASYNCHRONOUS pipeline (single rank):
 * the three kernels of an LM iteration run back to back, the interval between two consecutive kernel ends is a kernel's
 * share of the iteration; nothing is added to the stream.  pba_get_counters reports whichever mode is on. */
int pba_set_profiling(pba_engine* e, int32_t mode);
int pba_get_counters(pba_engine* e, pba_counters* c);
int pba_reset_counters(pba_engine* e);

/* Which driver ran the LAST pba_solve: "resident" (the whole solve as ONE cooperative launch, every workgroup keeping its tiles'
 * state in registers across the iterations: windows that fit one resident round of workgroups -- <= 2 x 128 observations per CU --
 * on a single rank, patch radius <= 2, <= 8 free cameras; PBA_RESIDENT=0 switches it off), "pipelined" (three kernels per
 * iteration enqueued ahead of the device-side decisions), "host-stepped" (PBA_ASYNC=0, event profiling), "none".  The three take
 * the same decisions on the same numbers; "resident" and "pipelined" are bit-identical. */
const char* pba_solve_driver(const pba_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* PBA_H */
