"""Loader of the C-ABI engine library (photobundle_amd/libpba_hip.so, built from photobundle_amd/csrc by
`make -C photobundle_amd/csrc` / __graft_entry__.build()).  There is no CPU fallback: a missing library or a
missing GPU is an error the caller sees."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PBA_LIB: diagnostics only (e.g. the TIMING=1 build with per-phase cycle stamps)
LIB_PATH = os.environ.get("PBA_LIB") or os.path.join(_HERE, "libpba_hip.so")
_LIB = None


class EngineUnavailable(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("max_frames", C.c_int32), ("radius", C.c_int32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("huber", C.c_double), ("device", C.c_int32), ("flags", C.c_int32), ("channels", C.c_int32),
                ("reserved", C.c_int32)]


class SolverOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("max_num_consecutive_invalid_steps", C.c_int32),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double), ("jacobi_scaling", C.c_int32), ("verbose", C.c_int32)]


class IterationSummary(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_nonmonotonic", C.c_int32),
                ("step_is_successful", C.c_int32), ("cost", C.c_double), ("cost_change", C.c_double),
                ("gradient_max_norm", C.c_double), ("gradient_norm", C.c_double), ("step_norm", C.c_double),
                ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double), ("eta", C.c_double),
                ("step_size", C.c_double), ("line_search_function_evaluations", C.c_int32),
                ("line_search_gradient_evaluations", C.c_int32), ("line_search_iterations", C.c_int32),
                ("linear_solver_iterations", C.c_int32), ("iteration_time_in_seconds", C.c_double),
                ("step_solver_time_in_seconds", C.c_double), ("cumulative_time_in_seconds", C.c_double),
                ("model_cost_change", C.c_double), ("candidate_cost", C.c_double)]


class SolverSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("fixed_cost", C.c_double),
                ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
                ("num_iterations", C.c_int32), ("num_residuals", C.c_int32), ("num_residual_blocks", C.c_int32),
                ("termination_type", C.c_int32), ("total_time_in_seconds", C.c_double),
                ("num_jacobian_passes", C.c_int64), ("num_cost_passes", C.c_int64),
                ("num_resolve_passes", C.c_int64), ("message", C.c_char * 256)]


class StepInfo(C.Structure):
    _fields_ = [("cost", C.c_double), ("gradient_max_norm", C.c_double), ("gradient_norm", C.c_double),
                ("model_cost_change", C.c_double), ("step_norm", C.c_double), ("x_norm", C.c_double),
                ("candidate_cost", C.c_double), ("linear_solver_ok", C.c_int32), ("eval_ok", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("linearize_ms", C.c_double), ("cost_ms", C.c_double), ("schur_ms", C.c_double),
                ("linearize_launches", C.c_int64), ("cost_launches", C.c_int64), ("schur_launches", C.c_int64),
                ("n_obs", C.c_int64), ("n_points", C.c_int64), ("solve_ms", C.c_double), ("exchange_ms", C.c_double),
                ("solve_launches", C.c_int64), ("exchange_launches", C.c_int64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_int32, C.c_void_p)

# every symbol include/pba.h declares (tests check the library exports all of them)
SYMBOLS = [
    "pba_status_string", "pba_last_error", "pba_default_solver_options", "pba_create", "pba_destroy",
    "pba_set_frame_u8", "pba_set_frame_channels_f32", "pba_get_frame_planes", "pba_get_frame_channel", "pba_sample_frame", "pba_set_frame_descriptor_u8", "pba_get_frame_channels_f32", "pba_set_frame_pyr_down", "pba_set_problem", "pba_set_cameras", "pba_set_inverse_depth", "pba_get_points_world", "pba_get_state",
    "pba_linearize", "pba_step", "pba_accept", "pba_get_reduced_system", "pba_get_obs_records", "pba_solve",
    "pba_comm_unique_id", "pba_comm_init_rccl", "pba_comm_init_callback", "pba_comm_enable_peer_exchange", "pba_comm_transport", "pba_comm_rank_count",
    "pba_set_profiling", "pba_get_counters", "pba_reset_counters", "pba_solve_driver",
    "pba_frontend_visibility", "pba_frontend_candidates", "pba_frontend_get_candidates", "pba_frontend_descriptors", "pba_frontend_zncc_probe",
]


def lib():
    """Returns the loaded CDLL; raises EngineUnavailable (never falls back) when the library is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(
            "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or make -C photobundle_amd/csrc); the engine has no CPU fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401  -- load torch's HIP runtime first so one libamdhip64 serves the process
    except Exception:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.pba_status_string.restype = C.c_char_p
    L.pba_status_string.argtypes = [C.c_int]
    L.pba_last_error.restype = C.c_char_p
    L.pba_last_error.argtypes = [C.c_void_p]
    L.pba_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.pba_destroy.argtypes = [C.c_void_p]
    L.pba_destroy.restype = None
    L.pba_set_frame_u8.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.pba_set_frame_channels_f32.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_void_p]
    L.pba_get_frame_planes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pba_get_frame_channel.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pba_sample_frame.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pba_set_frame_descriptor_u8.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int32, C.c_float, C.c_float]
    L.pba_get_frame_channels_f32.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.pba_set_frame_pyr_down.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.pba_frontend_visibility.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p]
    L.pba_frontend_candidates.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    L.pba_frontend_get_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.pba_frontend_zncc_probe.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pba_frontend_descriptors.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.pba_set_problem.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pba_set_cameras.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    L.pba_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pba_set_inverse_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pba_get_points_world.argtypes = [C.c_void_p, C.c_void_p]
    L.pba_linearize.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.pba_step.argtypes = [C.c_void_p, C.c_double, C.c_int32, C.POINTER(SolverOptions), C.POINTER(StepInfo)]
    L.pba_accept.argtypes = [C.c_void_p]
    L.pba_get_reduced_system.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    L.pba_get_obs_records.argtypes = [C.c_void_p, C.c_void_p]
    L.pba_solve.argtypes = [C.c_void_p, C.POINTER(SolverOptions), C.POINTER(SolverSummary), C.c_void_p, C.c_int32]
    L.pba_comm_unique_id.argtypes = [C.c_void_p]
    L.pba_comm_init_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    L.pba_comm_init_callback.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int32, C.c_int32]
    L.pba_comm_enable_peer_exchange.argtypes = [C.c_void_p]
    L.pba_comm_transport.argtypes = [C.c_void_p]
    L.pba_comm_transport.restype = C.c_char_p
    L.pba_solve_driver.argtypes = [C.c_void_p]
    L.pba_solve_driver.restype = C.c_char_p
    L.pba_comm_rank_count.argtypes = [C.c_void_p]
    L.pba_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
    L.pba_reset_counters.argtypes = [C.c_void_p]
    L.pba_set_profiling.argtypes = [C.c_void_p, C.c_int32]
    _LIB = L
    return L
