/*
 * pba_oracle.h -- C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the
 * reference's photometric bundle-adjustment hot path
 * (reference src/photobundle.cc:669-736 DescriptorError, :738-761 solver
 * options, :764-829 problem assembly + ceres::Solve) together with the
 * behaviour of the Ceres Solver 1.x components the reference delegates to
 * (AutoDiffCostFunction / Jet, HuberLoss + Corrector, TrustRegionMinimizer,
 * LevenbergMarquardtStrategy, SchurEliminator, rotation.h).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; nothing under photobundle_amd/ (the product) links or calls it.
 *
 * PARITY UNPINNED: Ceres, Eigen, OpenCV and Boost are absent from this image
 * and are not vendored in the reference, so the reference cannot be compiled
 * and it ships no tests / golden vectors.  The oracle is pinned instead by
 * independent cross-checks (numpy finite differences, scipy rotations, scipy
 * least-squares minima; see tests/test_oracle_*.py).
 */
#ifndef PBA_ORACLE_H
#define PBA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One sliding-window problem at the inner seam (photobundle.cc:784-829). */
typedef struct oracle_problem {
  int32_t rows, cols;        /* image size                                   */
  double fx, fy, cx, cy;     /* pinhole K (calibration.h:34-38)              */
  int32_t n_frames;          /* window slots (cameras)                       */
  int32_t radius;            /* patch radius R, P = (2R+1)^2                 */
  int32_t n_points;
  int32_t n_obs;
  int32_t fixed_slot;        /* constant camera (photobundle.cc:809-813), -1 none */
  int32_t n_channels;        /* descriptor channels C (photobundle.cc:229-245); 0 means 1 */
  double huber;              /* HuberLoss(a) if > 0 (photobundle.cc:797-798) */
  const float* planes;       /* [n_frames][C][3][rows*cols]: per channel I, Gx, Gy */
  const double* desc;        /* [n_points][C][P] reference descriptors (double), channel-major */
  const double* weights;     /* [P] patch weights (photobundle.cc:617-644)   */
  const int32_t* obs_point;  /* [n_obs] point index, grouped by point        */
  const int32_t* obs_slot;   /* [n_obs] frame slot                           */
  double* cams;              /* [n_frames][6] angle-axis + t, world->camera  */
  double* xyz;               /* [n_points][3] world points                   */
} oracle_problem;

/* ceres::Solver::Options subset (photobundle.cc:738-761) + Ceres defaults. */
typedef struct oracle_options {
  int32_t max_num_iterations;      /* 500 */
  int32_t num_threads;             /* min(omp_max,4) in the reference */
  double function_tolerance;       /* 1e-6 */
  double gradient_tolerance;       /* 1e-6 */
  double parameter_tolerance;      /* 1e-6 */
  double initial_trust_region_radius; /* 1e4  */
  double max_trust_region_radius;     /* 1e16 */
  double min_trust_region_radius;     /* 1e-32 */
  double min_relative_decrease;       /* 1e-3 */
  double min_lm_diagonal;             /* 1e-6 */
  double max_lm_diagonal;             /* 1e32 */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  int32_t jacobi_scaling;          /* 1 */
  int32_t use_autodiff;            /* 1: 9-wide dual numbers (reference path); 0: analytic J (faster CPU line) */
  int32_t extended_precision;      /* 1: the REFEREE -- every accumulation / the linear algebra in x87 extended precision (long double) */
} oracle_options;

/* ceres::IterationSummary subset (ceres_cereal.h:13-30). */
typedef struct oracle_iteration {
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
  double model_cost_change;   /* extra: for parity debugging */
  double candidate_cost;      /* extra */
} oracle_iteration;

typedef struct oracle_summary {
  double initial_cost, final_cost, fixed_cost;
  int32_t num_successful_steps, num_unsuccessful_steps;
  int32_t num_iterations;          /* entries written to iterations[] */
  int32_t num_residuals;
  int32_t num_residual_blocks;
  int32_t termination_type;        /* 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE */
  double total_time_in_seconds;
  int64_t num_jacobian_passes;     /* residual+Jacobian evaluations */
  int64_t num_cost_passes;         /* residual-only evaluations */
  char message[256];
} oracle_summary;

void oracle_default_options(oracle_options* o);

/* imgproc.cc:27-95: u8 -> float cast (photobundle.cc:231) + central differences * 0.5, zero border. */
void oracle_planes_from_u8(const uint8_t* img, int rows, int cols, float* I, float* Gx, float* Gy);
void oracle_imgradient_f32(const float* img, int rows, int cols, float* Gx, float* Gy);

/* Multi-channel descriptors: DescriptorFrame::Create (photobundle.cc:225-248).  descriptor_type 0 Intensity (1 channel),
 * 1 IntensityAndGradient (3), 2 BitPlanes (8: census transform of the 3x3-smoothed image, imgproc.cc:126-245, every bit
 * plane smoothed 5x5).  The two cv::GaussianBlur calls are restated from OpenCV's documented behaviour (absent
 * dependency, version unpinned): fixed-point 8-bit kernel for the u8 image, symmetric float form for the planes. */
int oracle_num_channels(int descriptor_type);
void oracle_descriptor_channels(const uint8_t* img, int rows, int cols, int descriptor_type, float* channels /* [C][rows*cols] */);
/* [C][3][rows*cols]: every channel followed by ITS gradient images (DescriptorFrame ctor, photobundle.cc:172-175) */
void oracle_channel_planes(const float* channels, int n_channels, int rows, int cols, float* planes);
void oracle_census(const uint8_t* src, int rows, int cols, uint8_t* dst);
void oracle_gaussian_blur_u8_3x3(const uint8_t* src, int rows, int cols, double sigma, uint8_t* dst);
void oracle_gaussian_blur_f32_5x5(const float* src, int rows, int cols, double sigma, float* dst);

/* sample_eigen.h:56-102 with :34-52 index rule; (y, x) already rounded to float. out = {I, Gx, Gy}. */
void oracle_sample_linear(const float* I, const float* Gx, const float* Gy, int rows, int cols,
                          float y, float x, float out[3]);

/* Ceres rotation.h restatements (double). R is column-major 3x3 (ColumnMajorAdapter3x3, photobundle.cc:650,661). */
void oracle_angle_axis_rotate_point(const double aa[3], const double pt[3], double out[3]);
void oracle_angle_axis_to_rotation_matrix(const double aa[3], double R_colmajor[9]);
void oracle_rotation_matrix_to_angle_axis(const double R_colmajor[9], double aa[3]);

/* photobundle.cc:617-644 */
void oracle_make_patch_weights(int radius, int do_gaussian, double* w);

/* photobundle.cc:466-479 integer-pixel descriptor extraction from a float channel. */
void oracle_extract_patch(const float* I, int rows, int cols, int u, int v, int radius, double* dst);

/* One residual block (photobundle.cc:696-727), raw (no loss correction).
 * residuals[P]; jac_cam[P*6], jac_pt[P*3] row-major (may be NULL). use_autodiff selects dual numbers. */
void oracle_eval_block(const oracle_problem* p, int obs, int use_autodiff,
                       double* residuals, double* jac_cam, double* jac_pt);

/* Full evaluation at (p->cams, p->xyz): cost = 1/2 sum rho(|r|^2).
 * Optional outputs (NULL to skip):
 *   block_sqnorm[n_obs]                raw squared norm of each block
 *   grad_cams[n_frames*6], grad_pts[n_points*3]   J^T r with loss correction (fixed camera rows = 0)
 *   U[n_frames*36] row-major 6x6, V[n_points*9] row-major 3x3, W[n_obs*18] row-major 6x3  (J^T J blocks, corrected) */
void oracle_linearize(const oracle_problem* p, int use_autodiff, int num_threads, double* cost,
                      double* block_sqnorm, double* grad_cams, double* grad_pts,
                      double* U, double* V, double* W);

/* Test hook: per residual block o, out[72 o ..] = Jc^T Jc (36, row-major) | Jc^T Jp (18) | Jp^T Jp (9) | Jc^T r (6) |
 * Jp^T r (3) of the loss-corrected rows (the camera derivative is reported for the constant camera too). */
void oracle_block_products(const oracle_problem* p, int use_autodiff, int num_threads, double* out);

/* Ceres-faithful trust-region LM with Schur elimination of the points; updates p->cams / p->xyz in place. */
int oracle_solve(oracle_problem* p, const oracle_options* o, oracle_summary* s,
                 oracle_iteration* iterations, int max_iterations_out);

#ifdef __cplusplus
}
#endif
#endif
