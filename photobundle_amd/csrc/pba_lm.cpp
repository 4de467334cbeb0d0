// pba_lm.cpp -- host Levenberg-Marquardt driver: replaces ceres::Solve at reference src/photobundle.cc:829.
//
// Control flow = Ceres (>= 1.12) TrustRegionMinimizer + LevenbergMarquardtStrategy with the settings of
// GetSolverOptions (photobundle.cc:738-761) and the Ceres defaults listed in SURVEY.md 8c.  All heavy work is
// behind the C-ABI primitives (pba_linearize / pba_step / pba_accept); this file only decides.
//
// One deviation in scheduling (not in results): the gradient norms of a freshly accepted point come out of the
// same device pass that computes the NEXT trust-region step, so that step is computed speculatively right after
// an acceptance; if the iteration then terminates the solve (gradient tolerance / iteration limit) the
// speculative step is simply dropped (at the iteration limit only the gradient part is run).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "../../include/pba.h"

#include "pba_internal.h"
#include "pba_device.h"

namespace {
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}

static int summarize_device_solve(pba_engine* e, const pba_solver_options* o, pba_solver_summary* sum, pba_iteration_summary* its,
                                  int32_t max_out, double t_start, bool verbose, int64_t jac_passes);

// Resident variant (pba_resident.h): the whole solve is ONE cooperative launch -- every workgroup keeps its tiles' state in registers
// across the iterations, the serial workgroup takes the same decisions (lm_decide) -- and the host only waits for the flush.
static int solve_async(pba_engine* e, const pba_solver_options* o, pba_solver_summary* sum, pba_iteration_summary* its,
                       int32_t max_out, double t_start, bool verbose);

static int solve_resident(pba_engine* e, const pba_solver_options* o, pba_solver_summary* sum, pba_iteration_summary* its,
                          int32_t max_out, double t_start, bool verbose) {
  unsigned long long seq = 0;
  int rc = pba_internal_resident_launch(e, o, &seq);
  if (rc == PBA_INTERNAL_RESIDENT_REFUSED) return solve_async(e, o, sum, its, max_out, t_start, verbose);
  if (rc) return rc;
  if ((rc = pba_internal_async_wait(e, seq))) { pba_internal_resident_failed(e); return rc; }
  if ((rc = pba_internal_async_end(e))) return rc;
  const pba::LmState* st = static_cast<const pba::LmState*>(pba_internal_async_state(e));
  pba_internal_resident_done(e, st->iteration);
  pba_internal_resident_trace(e, st->iteration);
  // one Jacobian pass at the initial point + one (speculative) Jacobian pass per step taken
  return summarize_device_solve(e, o, sum, its, max_out, t_start, verbose, 1 + (int64_t)st->iteration);
}

// Asynchronous variant: the same trust-region rules are evaluated on the device by the last workgroup of every
// candidate pass (pba_kernels.h: lm_decide), so the host enqueues iterations back to back (at most kAhead in flight
// beyond the last one it has seen finish) instead of paying a launch + completion round trip per step.
static int solve_async(pba_engine* e, const pba_solver_options* o, pba_solver_summary* sum, pba_iteration_summary* its,
                       int32_t max_out, double t_start, bool verbose) {
  using pba::LmState;
  constexpr int kAhead = 3;
  static const bool tr = getenv("PBA_TRACE_SOLVE") != nullptr;
  double tt[8] = {0};
  tt[0] = now();
  int rc = pba_internal_async_begin(e, o);
  if (rc) return rc;
  tt[1] = now();
  const volatile LmState* st = static_cast<const volatile LmState*>(pba_internal_async_state(e));
  unsigned long long seq = 0, seqs[kAhead + 1] = {0};
  if ((rc = pba_internal_async_enqueue(e, 0, 0, o, &seq))) return rc;
  tt[2] = now();
  int enq = 0;
  unsigned long long last_seq = 0;
  // Multi-rank: every enqueued step carries collectives, so all ranks must enqueue the SAME number of steps although
  // each sees the termination flag at a different moment: they all stop kAhead steps after the terminating one.
  const bool multi = pba_internal_is_multi(e) != 0;
  while (enq < o->max_num_iterations) {
    // done and done_seq reach the host mirror as independent 32-bit stores: act on `done` in multi-rank mode only once
    // the (non-zero) sequence number of the terminating step is visible too
    {
      const unsigned long long ds = st->done_seq;
      if (st->done && (!multi || (ds != 0 && last_seq >= ds + kAhead))) break;
    }
    if ((rc = pba_internal_async_enqueue(e, 1, enq == 0 ? 1 : 0, o, &seq))) return rc;
    seqs[enq % (kAhead + 1)] = seq;
    last_seq = seq;
    ++enq;
    if (enq > kAhead) { if ((rc = pba_internal_async_wait(e, seqs[(enq - kAhead) % (kAhead + 1)]))) return rc; }
  }
  // Iteration limit (or max_num_iterations <= 0): the gradient norms of the final point may still be missing.  The pass
  // that computes them is enqueued unconditionally and gates itself on the device state (lm_final_pass_needed), which
  // saves a host round trip; then ONE flush brings the last outcome and the iteration log to the host mirror.
  tt[3] = now();
  bool flushed = false;
  if (enq >= o->max_num_iterations) {
    if ((rc = pba_internal_async_enqueue(e, 2, o->max_num_iterations <= 0 ? 1 : 0, o, &seq))) return rc;
    flushed = pba_internal_final_flushes(e) != 0;      // single rank: the final pass flushes in its own last workgroup
  }
  if (!flushed && (rc = pba_internal_async_enqueue(e, 3, 0, o, &seq))) return rc;
  tt[4] = now();
  if ((rc = pba_internal_async_wait(e, seq))) return rc;
  tt[5] = now();
  (void)last_seq;
  if ((rc = pba_internal_async_end(e))) return rc;
  tt[6] = now();
  if (tr) std::fprintf(stderr, "solve_async us: entry->begin %.1f, begin %.1f, enqueue0 %.1f, loop %.1f, final enqueues %.1f, final wait %.1f, end %.1f\n",
                       1e6 * (tt[0] - t_start), 1e6 * (tt[1] - tt[0]), 1e6 * (tt[2] - tt[1]), 1e6 * (tt[3] - tt[2]), 1e6 * (tt[4] - tt[3]), 1e6 * (tt[5] - tt[4]), 1e6 * (tt[6] - tt[5]));
  return summarize_device_solve(e, o, sum, its, max_out, t_start, verbose, -1);
}

// The host mirror of the device-side trust-region state and iteration log -> pba_solver_summary / iteration summaries.
// jac_passes < 0: the pass counters of the engine (asynchronous driver); else the count to report (resident solve).
static int summarize_device_solve(pba_engine* e, const pba_solver_options* o, pba_solver_summary* sum, pba_iteration_summary* its,
                                  int32_t max_out, double t_start, bool verbose, int64_t jac_passes) {
  using pba::LmState;
  LmState fin;
  std::memcpy(&fin, const_cast<const LmState*>(static_cast<const LmState*>(pba_internal_async_state(e))), sizeof(fin));
  const pba_iteration_summary* log = pba_internal_async_log(e);
  const double total = now() - t_start;
  const int n_log = fin.n_log;
  for (int i = 0; i < n_log && i < max_out && its; ++i) {
    its[i] = log[i];
    its[i].iteration_time_in_seconds = total / (n_log > 0 ? n_log : 1);
    its[i].cumulative_time_in_seconds = total * (i + 1) / (n_log > 0 ? n_log : 1);
    if (verbose)
      std::printf("%4d  cost % .6e  change % .3e  |grad| %.3e  |step| %.3e  rho % .3e  radius %.3e  %s\n", its[i].iteration, its[i].cost,
                  its[i].cost_change, its[i].gradient_max_norm, its[i].step_norm, its[i].relative_decrease, its[i].trust_region_radius,
                  its[i].step_is_successful ? "ok" : (its[i].step_is_valid ? "rejected" : "invalid"));
  }
  sum->initial_cost = fin.initial_cost;
  sum->final_cost = fin.minimum_cost;
  sum->num_successful_steps = fin.num_successful;
  sum->num_unsuccessful_steps = fin.num_unsuccessful;
  sum->num_iterations = n_log < max_out ? n_log : max_out;
  sum->num_resolve_passes = fin.num_unsuccessful;
  pba_internal_pass_counts(e, &sum->num_jacobian_passes, &sum->num_cost_passes);
  if (jac_passes >= 0) { sum->num_jacobian_passes = jac_passes; sum->num_cost_passes = 0; }
  switch (fin.done) {
    case pba::kLmGradientTolerance:
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Gradient tolerance reached. Gradient max norm: %e <= %e", fin.last_value[0], o->gradient_tolerance);
      break;
    case pba::kLmMinRadius:
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Minimum trust region radius reached. Trust region radius: %e <= %e", fin.radius, o->min_trust_region_radius);
      break;
    case pba::kLmParameterTolerance:
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Parameter tolerance reached. Relative step_norm: %e <= %e.", fin.last_value[0], o->parameter_tolerance);
      break;
    case pba::kLmFunctionTolerance:
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Function tolerance reached. |cost_change|/cost: %e <= %e", fin.last_value[0], o->function_tolerance);
      break;
    case pba::kLmInvalidSteps:
      sum->termination_type = 2;
      std::snprintf(sum->message, sizeof(sum->message), "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps: %d", o->max_num_consecutive_invalid_steps);
      break;
    case pba::kLmEvalFailure:
      sum->termination_type = 2;
      std::snprintf(sum->message, sizeof(sum->message), fin.n_log == 0 ? "Initial residual and Jacobian evaluation failed." : "Residual and Jacobian evaluation failed.");
      break;
    default:
      sum->termination_type = 1;
      std::snprintf(sum->message, sizeof(sum->message), "Maximum number of iterations reached. Number of iterations: %d.", fin.iteration);
      break;
  }
  sum->total_time_in_seconds = now() - t_start;
  return PBA_OK;
}

extern "C" int pba_solve(pba_engine* e, const pba_solver_options* o, pba_solver_summary* sum, pba_iteration_summary* its,
                         int32_t max_out) {
  if (!e || !o || !sum) return PBA_ERR_INVALID;
  { const int rc0 = pba_internal_ready(e); if (rc0) return rc0; }   // solve before set_problem / set_cameras: PBA_ERR_STATE
  const double t_start = now();
  std::memset(sum, 0, sizeof(*sum));
  sum->termination_type = 1;
  std::snprintf(sum->message, sizeof(sum->message), "Maximum number of iterations reached.");
  double blocks = (double)pba_internal_local_blocks(e);
  if (pba_internal_allreduce_host(e, &blocks, 1, 0)) return PBA_ERR_COMM;
  // ceres::Solver::Summary counts in `int` (the reference's numResiduals too): a window beyond that range is refused up front
  // instead of reporting a wrapped count (3.2 M blocks x 121 pixels x 8 channels would)
  if (blocks * pba_internal_patch_len(e) > 2147483647.0) {
    std::snprintf(sum->message, sizeof(sum->message), "%.0f residual blocks x %d residuals exceed the int32 range of the summary", blocks, pba_internal_patch_len(e));
    return PBA_ERR_INVALID;
  }
  sum->num_residual_blocks = (int32_t)blocks;
  sum->num_residuals = (int32_t)(blocks * pba_internal_patch_len(e));
  sum->fixed_cost = 0.0;   // every residual block has a free point
  const bool verbose = o->verbose && pba_internal_rank(e) == 0;
  if (pba_internal_async_capable(e, o)) {
    pba_internal_reset_pass_counts(e);
    if (pba_internal_resident_capable(e, o)) return solve_resident(e, o, sum, its, max_out, t_start, verbose);
    return solve_async(e, o, sum, its, max_out, t_start, verbose);
  }

  int n_it = 0;
  auto push = [&](const pba_iteration_summary& s) {
    if (its && n_it < max_out) its[n_it] = s;
    ++n_it;
    if (verbose)
      std::printf("%4d  cost % .6e  change % .3e  |grad| %.3e  |step| %.3e  rho % .3e  radius %.3e  %s\n", s.iteration, s.cost,
                  s.cost_change, s.gradient_max_norm, s.step_norm, s.relative_decrease, s.trust_region_radius,
                  s.step_is_successful ? "ok" : (s.step_is_valid ? "rejected" : "invalid"));
  };

  double radius = o->initial_trust_region_radius, decrease_factor = 2.0;
  int num_consecutive_invalid = 0;
  pba_step_info info;
  std::memset(&info, 0, sizeof(info));

  // ---- IterationZero -----------------------------------------------------------------------------------
  double t_iter = now();
  pba_internal_reset_pass_counts(e);
  pba_internal_set_speculate(e, 1);
  int rc = pba_linearize(e, nullptr);
  if (rc) return rc;
  bool last = o->max_num_iterations <= 0;
  rc = pba_internal_step(e, radius, 1, o, &info, last ? 1 : 0);
  if (rc == PBA_ERR_NUMERIC) {
    sum->termination_type = 2;
    std::snprintf(sum->message, sizeof(sum->message), "Initial residual and Jacobian evaluation failed.");
    sum->total_time_in_seconds = now() - t_start;
    return PBA_OK;
  }
  if (rc) return rc;
  bool info_valid = !last;
  double x_cost = info.cost;
  sum->initial_cost = x_cost;
  double minimum_cost = x_cost;

  pba_iteration_summary it;
  std::memset(&it, 0, sizeof(it));
  it.iteration = 0; it.eta = 1e-1;
  it.cost = x_cost; it.gradient_max_norm = info.gradient_max_norm; it.gradient_norm = info.gradient_norm;
  it.step_is_valid = 1; it.step_is_successful = 1;

  auto finalize = [&]() -> bool {
    // TrustRegionMinimizer::FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it.step_is_successful) {
      ++sum->num_successful_steps;
      if (x_cost < minimum_cost || it.iteration == 0) { minimum_cost = x_cost; it.step_is_nonmonotonic = 0; }
      else it.step_is_nonmonotonic = 1;
    } else {
      ++sum->num_unsuccessful_steps;
    }
    it.trust_region_radius = radius;
    const double t = now();
    it.iteration_time_in_seconds = t - t_iter;
    it.cumulative_time_in_seconds = t - t_start;
    push(it);
    if (it.iteration >= o->max_num_iterations) {
      sum->termination_type = 1;
      std::snprintf(sum->message, sizeof(sum->message), "Maximum number of iterations reached. Number of iterations: %d.", it.iteration);
      return false;
    }
    if (it.step_is_successful && it.gradient_max_norm <= o->gradient_tolerance) {
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Gradient tolerance reached. Gradient max norm: %e <= %e", it.gradient_max_norm, o->gradient_tolerance);
      return false;
    }
    if (radius <= o->min_trust_region_radius) {
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Minimum trust region radius reached. Trust region radius: %e <= %e", radius, o->min_trust_region_radius);
      return false;
    }
    return true;
  };
  auto step_rejected = [&]() { radius = radius / decrease_factor; decrease_factor *= 2.0; };   // LM::StepRejected

  while (finalize()) {
    t_iter = now();
    const int iteration = it.iteration + 1;
    const double prev_gmax = it.gradient_max_norm, prev_gnorm = it.gradient_norm;
    std::memset(&it, 0, sizeof(it));
    it.iteration = iteration; it.eta = 1e-1;
    it.gradient_max_norm = prev_gmax; it.gradient_norm = prev_gnorm;
    const double t_solve = now();
    if (!info_valid) {
      // re-solve with the new damping from the stored linearisation (rejected / invalid predecessor)
      rc = pba_internal_step(e, radius, 0, o, &info, 0);
      if (rc) return rc;
      sum->num_resolve_passes++;
      it.step_solver_time_in_seconds = now() - t_solve;
    }
    info_valid = false;
    it.linear_solver_iterations = 1;
    it.model_cost_change = info.model_cost_change;
    const bool step_is_valid = info.linear_solver_ok && info.model_cost_change > 0.0;
    if (!step_is_valid) {
      // HandleInvalidStep
      ++num_consecutive_invalid;
      if (num_consecutive_invalid >= o->max_num_consecutive_invalid_steps) {
        sum->termination_type = 2;
        std::snprintf(sum->message, sizeof(sum->message), "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps: %d", o->max_num_consecutive_invalid_steps);
        it.cost = x_cost; it.trust_region_radius = radius;
        push(it);
        break;
      }
      pba_internal_set_speculate(e, 0);
      step_rejected();
      it.cost = x_cost;
      continue;
    }
    it.step_is_valid = 1;
    num_consecutive_invalid = 0;
    const double candidate_cost = info.eval_ok ? info.candidate_cost : std::numeric_limits<double>::max();
    it.candidate_cost = candidate_cost;
    // ParameterToleranceReached
    it.step_norm = info.step_norm;
    const double step_size_tolerance = o->parameter_tolerance * (info.x_norm + o->parameter_tolerance);
    if (it.step_norm <= step_size_tolerance) {
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Parameter tolerance reached. Relative step_norm: %e <= %e.", it.step_norm / (info.x_norm + o->parameter_tolerance), o->parameter_tolerance);
      break;
    }
    // FunctionToleranceReached
    it.cost_change = x_cost - candidate_cost;
    if (std::fabs(it.cost_change) <= o->function_tolerance * x_cost) {
      sum->termination_type = 0;
      std::snprintf(sum->message, sizeof(sum->message), "Function tolerance reached. |cost_change|/cost: %e <= %e", std::fabs(it.cost_change) / x_cost, o->function_tolerance);
      break;
    }
    // IsStepSuccessful
    it.relative_decrease = it.cost_change / info.model_cost_change;
    if (it.relative_decrease > o->min_relative_decrease) {
      // HandleSuccessfulStep: x <- candidate, re-linearise, LM::StepAccepted
      pba_internal_set_speculate(e, 1);   // accepted: keep betting on acceptance
      if ((rc = pba_accept(e))) return rc;
      if ((rc = pba_linearize(e, nullptr))) return rc;   // no-op when the candidate pass was a Jacobian pass
      x_cost = candidate_cost;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));
      radius = std::min(o->max_trust_region_radius, radius);
      decrease_factor = 2.0;
      last = iteration >= o->max_num_iterations;
      rc = pba_internal_step(e, radius, 0, o, &info, last ? 1 : 0);
      if (rc == PBA_ERR_NUMERIC) {
        sum->termination_type = 2;
        std::snprintf(sum->message, sizeof(sum->message), "Residual and Jacobian evaluation failed.");
        break;
      }
      if (rc) return rc;
      info_valid = !last;
      it.step_is_successful = 1;
      it.cost = x_cost;
      it.gradient_max_norm = info.gradient_max_norm;
      it.gradient_norm = info.gradient_norm;
    } else {
      // HandleUnsuccessfulStep
      pba_internal_set_speculate(e, 0);   // rejected: the retry only needs the cost
      step_rejected();
      it.cost = candidate_cost;
    }
  }
  pba_internal_pass_counts(e, &sum->num_jacobian_passes, &sum->num_cost_passes);
  sum->final_cost = minimum_cost;
  sum->num_iterations = n_it < max_out ? n_it : max_out;
  sum->total_time_in_seconds = now() - t_start;
  return PBA_OK;
}
