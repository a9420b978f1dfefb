"""ctypes binding of include/lili_hip.h (liblili_hip.so) plus a thin host-side mirror of the reference's
matcher interface.  There is NO fallback: if the HIP library is missing or no gfx950 device is present
every entry point raises.  Nothing here imports oracle/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblili_hip.so")

OK = 0
KIND_SURF, KIND_EDGE = 0, 1
MASK_SURF, MASK_EDGE = 1, 2
VARIANT_LIVOX, VARIANT_ROT, VARIANT_FRONTEND = 0, 1, 2
LOSS_NONE, LOSS_CAUCHY, LOSS_HUBER = 0, 1, 2
MEM_HOST, MEM_DEVICE = 0, 1
MAX_SLOTS = 8
GRAM_DOUBLES = 72


class LiliError(RuntimeError):
    pass


class Cloud(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n", C.c_size_t), ("stride", C.c_size_t), ("aux_offset", C.c_int), ("mem", C.c_int)]


class FeatureOut(C.Structure):
    _fields_ = [("data", C.c_void_p), ("capacity", C.c_size_t), ("stride", C.c_size_t), ("mem", C.c_int), ("count", C.c_size_t)]


class RotParams(C.Structure):
    _fields_ = [("n_scans", C.c_int), ("ds_rate", C.c_int), ("ds_v", C.c_float), ("near_range", C.c_float)]


class LivoxParams(C.Structure):
    _fields_ = [("surf_thres", C.c_double), ("edge_thres", C.c_double), ("near_range", C.c_float)]


class FrontendOptions(C.Structure):
    _fields_ = [("leaf_query", C.c_float), ("leaf_map", C.c_float), ("width", C.c_int), ("n_iters", C.c_int), ("slot", C.c_int), ("want_timing", C.c_int), ("flags", C.c_int)]


FRAME_SELF_MAP, FRAME_PUSH_EMPTY, FRAME_EXTERNAL_MAP, FRAME_EDGES = 1, 2, 4, 8


class FrontendResult(C.Structure):
    _fields_ = [("t", C.c_double * 3), ("q", C.c_double * 4), ("gn_status", C.c_int), ("matched", C.c_int), ("n_edge", C.c_int32), ("n_surf", C.c_int32),
                ("n_query", C.c_int32), ("n_map_raw", C.c_int32), ("n_map", C.c_int32), ("stage_us", C.c_double * 8)]


LM_MAX_LOG = 32
LM_TERMINATION = {0: "max_iterations", 1: "gradient_tolerance", 2: "parameter_tolerance", 3: "function_tolerance", 4: "stalled", 5: "numerical_failure", 6: "min_radius"}


class BackendOptions(C.Structure):
    _fields_ = [("leaf_surf", C.c_float), ("leaf_edge", C.c_float), ("leaf_surf_map", C.c_float), ("leaf_edge_map", C.c_float), ("width", C.c_int), ("want_timing", C.c_int), ("join_slot", C.c_int)]


class BackendResult(C.Structure):
    _fields_ = [("n_map_raw", C.c_int32 * 2), ("n_map", C.c_int32 * 2), ("n_query", C.c_int32 * 2), ("associated", C.c_int32), ("stage_us", C.c_double * 8)]


class LmOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("reserved_", C.c_int32), ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double)]


class LmIteration(C.Structure):
    _fields_ = [("cost", C.c_double), ("new_cost", C.c_double), ("rho", C.c_double), ("radius", C.c_double), ("step_norm", C.c_double),
                ("accepted", C.c_int32), ("iteration", C.c_int32)]


class LmSummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("successful_steps", C.c_int32), ("termination", C.c_int32), ("n_logged", C.c_int32),
                ("n_surf", C.c_int32), ("n_edge", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_radius", C.c_double),
                ("it", LmIteration * LM_MAX_LOG)]

    def as_dict(self):
        return dict(iterations=self.iterations, successful_steps=self.successful_steps, termination=LM_TERMINATION.get(self.termination, self.termination),
                    n_surf=self.n_surf, n_edge=self.n_edge, initial_cost=self.initial_cost, final_cost=self.final_cost, final_radius=self.final_radius,
                    log=[dict(it=e.iteration, cost=e.cost, new_cost=e.new_cost, rho=e.rho, radius=e.radius, step=e.step_norm, accepted=bool(e.accepted))
                         for e in self.it[:self.n_logged]])


class S2MParams(C.Structure):
    _fields_ = [("variant", C.c_int), ("loss", C.c_int), ("loss_a", C.c_double), ("lidar_const", C.c_double),
                ("kd_max_radius", C.c_double), ("edge_gate", C.c_double), ("surf_dist_thres", C.c_double),
                ("reflect_thres", C.c_double), ("surf_weight_min", C.c_double), ("edge_dist_max", C.c_double),
                ("q_lb", C.c_double * 4), ("t_lb", C.c_double * 3), ("scale_surf_num", C.c_double),
                ("scale_edge_num", C.c_double)]


def make_params(variant="rot", **kw):
    """Matcher parameters of the reference configs:
    L/config/config_fr_iosb.yaml ('livox'), R/config/config_fr_iosb.yaml ('rot'),
    L/src/LidarOdometry.cpp:365,389,400,507 ('frontend')."""
    p = S2MParams()
    if variant == "livox":
        p.variant, p.loss, p.loss_a = VARIANT_LIVOX, LOSS_CAUCHY, 1.0
        p.lidar_const, p.kd_max_radius, p.edge_gate = 20.0, 1.0, 1.0
        p.surf_dist_thres, p.reflect_thres, p.surf_weight_min, p.edge_dist_max = 0.12, 15.0, 0.2, 0.0
        p.q_lb[:] = [0.0, 0.0, 0.0, 1.0]
        p.t_lb[:] = [-0.0265, 0.0202, 0.05309]
        p.scale_surf_num = p.scale_edge_num = 0.0
    elif variant == "rot":
        p.variant, p.loss, p.loss_a = VARIANT_ROT, LOSS_CAUCHY, 1.0
        p.lidar_const, p.kd_max_radius, p.edge_gate = 7.5, 1.0, 1.0
        p.surf_dist_thres, p.reflect_thres, p.surf_weight_min, p.edge_dist_max = 0.12, 0.0, 0.3, 0.1
        p.q_lb[:] = [0.7071, 0.0, 0.0, 0.7071]
        p.t_lb[:] = [-0.18, 0.0, -0.095]
        p.scale_surf_num, p.scale_edge_num = 1000.0, 200.0
    elif variant == "frontend":
        p.variant, p.loss, p.loss_a = VARIANT_FRONTEND, LOSS_HUBER, 0.1
        p.lidar_const, p.kd_max_radius, p.edge_gate = 1.0, 1.0, 1.0
        p.surf_dist_thres, p.reflect_thres, p.surf_weight_min, p.edge_dist_max = 0.06, 0.0, 0.4, 0.0
        p.q_lb[:] = [1.0, 0.0, 0.0, 0.0]
        p.t_lb[:] = [0.0, 0.0, 0.0]
        p.scale_surf_num = p.scale_edge_num = 0.0
    else:
        raise ValueError(variant)
    for k, v in kw.items():
        if k in ("q_lb", "t_lb"):
            getattr(p, k)[:] = list(v)
        else:
            setattr(p, k, v)
    return p


_lib = None

_SIGS = {
    "lili_abi_version": (C.c_int, []),
    "lili_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "lili_ctx_destroy": (None, [C.c_void_p]),
    "lili_last_error": (C.c_char_p, [C.c_void_p]),
    "lili_sync": (C.c_int, [C.c_void_p]),
    "lili_set_debug": (C.c_int, [C.c_void_p, C.c_int]),
    "lili_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "lili_extract_rot": (C.c_int, [C.c_void_p, C.POINTER(Cloud), C.c_void_p, C.c_void_p, C.POINTER(RotParams), C.POINTER(FeatureOut), C.POINTER(FeatureOut), C.POINTER(FeatureOut)]),
    "lili_extract_rot_debug": (C.c_int, [C.c_void_p] + [C.c_void_p] * 11),
    "lili_extract_livox": (C.c_int, [C.c_void_p, C.POINTER(Cloud), C.c_int, C.c_void_p, C.POINTER(LivoxParams), C.POINTER(FeatureOut), C.POINTER(FeatureOut), C.POINTER(FeatureOut)]),
    "lili_extract_livox_debug": (C.c_int, [C.c_void_p] * 6),
    "lili_extract_rot_device": (C.c_int, [C.c_void_p, C.POINTER(Cloud), C.POINTER(Cloud), C.POINTER(Cloud)]),
    "lili_extract_livox_device": (C.c_int, [C.c_void_p, C.POINTER(Cloud), C.POINTER(Cloud)]),
    "lili_frontend_frame": (C.c_int, [C.c_void_p, C.POINTER(Cloud), C.c_int, C.c_void_p, C.POINTER(LivoxParams), C.POINTER(S2MParams), C.POINTER(FrontendOptions), C.c_void_p, C.c_void_p,
                                      C.POINTER(FrontendResult)]),
    "lili_frontend_frame_rot": (C.c_int, [C.c_void_p, C.POINTER(Cloud), C.c_void_p, C.c_void_p, C.POINTER(RotParams), C.POINTER(S2MParams), C.POINTER(FrontendOptions), C.c_void_p, C.c_void_p,
                                          C.POINTER(FrontendResult)]),
    "lili_backend_keyframe_prepare": (C.c_int, [C.c_void_p, C.POINTER(Cloud), C.POINTER(Cloud), C.c_void_p, C.c_void_p, C.POINTER(Cloud), C.POINTER(Cloud), C.POINTER(C.c_int), C.c_int,
                                                C.c_void_p, C.c_void_p, C.POINTER(S2MParams), C.POINTER(BackendOptions), C.c_void_p, C.POINTER(BackendResult)]),
    "lili_frontend_reset": (C.c_int, [C.c_void_p]),
    "lili_frontend_flush": (C.c_int, [C.c_void_p, C.POINTER(S2MParams), C.POINTER(FrontendOptions), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lili_voxel_filter": (C.c_int, [C.c_void_p, C.POINTER(Cloud), C.c_float, C.POINTER(FeatureOut), C.c_void_p]),
    "lili_localmap_reset": (C.c_int, [C.c_void_p, C.c_int]),
    "lili_localmap_push": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Cloud), C.c_void_p, C.c_void_p, C.c_int]),
    "lili_localmap_commit": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_double, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "lili_localmap_get": (C.c_int, [C.c_void_p, C.POINTER(FeatureOut)]),
    "lili_localmap_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lili_voxel_filter_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lili_map_set": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Cloud), C.c_double]),
    "lili_map_density": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "lili_s2m_linearize_window": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lili_s2m_associate_window": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lili_map_set_begin": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Cloud), C.c_double]),
    "lili_map_set_end": (C.c_int, [C.c_void_p, C.c_int]),
    "lili_map_focus": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_double]),
    "lili_map_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "lili_map_build_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lili_s2m_set_queries": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Cloud)]),
    "lili_s2m_associate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(S2MParams), C.POINTER(C.c_int)]),
    "lili_s2m_linearize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(S2MParams), C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "lili_s2m_get_surf_records": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]),
    "lili_s2m_get_edge_records": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]),
    "lili_s2m_get_neighbors": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]),
    "lili_s2m_pose_set": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "lili_s2m_pose_get": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "lili_s2m_last_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lili_s2m_accumulate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(S2MParams), C.c_void_p]),
    "lili_s2m_associate_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(S2MParams)]),
    "lili_s2m_counts_export": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "lili_s2m_counts_import": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "lili_s2m_linearize_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(S2MParams), C.c_void_p]),
    "lili_s2m_gn_update": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "lili_s2m_iterate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(S2MParams), C.c_int]),
    "lili_s2m_iterate_inner": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(S2MParams), C.c_int, C.c_int]),
    "lili_s2m_iterate_window": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(S2MParams), C.c_int]),
    "lili_s2m_debug_times": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]),
    "lili_s2m_pose_copy": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "lili_s2m_iterate_restart": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(S2MParams), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "lili_s2m_iterate_sharded": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(S2MParams), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lili_lm_default_options": (None, [C.POINTER(LmOptions)]),
    "lili_s2m_solve_lm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(S2MParams), C.POINTER(LmOptions), C.POINTER(LmSummary)]),
    "lili_s2m_solve_lm_window": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(S2MParams), C.POINTER(LmOptions), C.POINTER(LmSummary)]),
    "lili_s2m_counts_window_sharded": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lili_s2m_linearize_window_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(S2MParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    "lili_s2m_linearize_window_sharded": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(S2MParams), C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lili_s2m_iterate_window_sharded": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(S2MParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lili_s2m_linearize_window_gather": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(S2MParams), C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lili_s2m_linearize_window_gather_at": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(S2MParams), C.POINTER(C.c_int), C.c_int,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lili_s2m_iterate_window_gather": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(S2MParams), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lili_host_alloc": (C.c_void_p, [C.c_size_t]),
    "lili_host_free": (None, [C.c_void_p]),
    "lili_p2p_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "lili_p2p_handle": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lili_p2p_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lili_p2p_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "lili_p2p_status": (C.c_int, [C.c_void_p]),
    "lili_p2p_set_timeout": (C.c_int, [C.c_void_p, C.c_double]),
    "lili_p2p_destroy": (None, [C.c_void_p]),
    "lili_livox_custom_to_cloud": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_int]),
    "lili_imu_reset": (None, [C.c_void_p]),
    "lili_imu_integrate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p]),
    "lili_marg_add_lidar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]),
    "lili_gn_step_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lili_gram_to_factor": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
}


def exported_symbols():
    """Names include/lili_hip.h declares (used by the CPU-side symbol test)."""
    return sorted(_SIGS)


def load_library():
    """Loads liblili_hip.so.  Raises LiliError if it has not been built — there is no other backend."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("LILI_HIP_LIBRARY", LIB_PATH)   # A/B builds of the SAME library (tools/); never another backend
    if not os.path.exists(path):
        raise LiliError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). The hot path has no CPU fallback.")
    # PyTorch wheels bundle their own libamdhip64 (same soname, different file name).  If this library pulled in the
    # system copy first, a later `import torch` would start a SECOND HIP runtime in the process and find no GPU.
    # Importing torch first makes both share one runtime.  (C/C++ hosts without torch are unaffected.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)      # AttributeError here = ABI mismatch between header and library
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class PinnedArray:
    """numpy view of page-locked host memory (lili_host_alloc): clouds passed from it are DMA'd, not staged.  Keep the object
    alive as long as the array is used; close() (or garbage collection) frees the memory."""

    def __init__(self, shape, dtype=np.float32):
        self.lib = load_library()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = self.lib.lili_host_alloc(max(n, 1))
        if not self.ptr:
            raise LiliError("lili_host_alloc failed")
        buf = (C.c_ubyte * max(n, 1)).from_address(self.ptr)
        buf._lili_owner = _PinnedBlock(self.lib, self.ptr)      # the ctypes buffer is the base object of every numpy view: the block lives as long as any view
        self._block = buf._lili_owner
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        """Drops this object's reference; the memory is released once no numpy view of it is left (never under a live view)."""
        self.array = None
        self._block = None
        self.ptr = None

    __del__ = close


class _PinnedBlock:
    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):
        if self.ptr:
            try:
                self.lib.lili_host_free(C.c_void_p(self.ptr))
            except Exception:   # noqa: BLE001  (interpreter shutdown)
                pass
            self.ptr = None


def _f64(a, n):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    if a.size != n:
        raise ValueError(f"expected {n} doubles")
    return a


def cloud_from_numpy(arr, aux_col=None):
    """Describes a contiguous float32 (n, k>=3) host array as a lili_cloud (stride = 4k bytes)."""
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    if arr.ndim != 2 or arr.shape[1] < 3:
        raise ValueError("cloud array must be (n, >=3) float32")
    c = Cloud(arr.ctypes.data if arr.size else None, arr.shape[0], arr.shape[1] * 4,
              -1 if aux_col is None else int(aux_col) * 4, MEM_HOST)
    c._keep = arr
    return c


def cloud_from_device(ptr, n, stride, aux_offset=-1):
    return Cloud(ptr, n, stride, aux_offset, MEM_DEVICE)


class Context:
    """One HIP stream + device buffers (lili_ctx)."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.lili_ctx_create(C.byref(h), int(device), C.c_void_p(stream) if stream else None)
        if rc != OK:
            raise LiliError(f"lili_ctx_create failed ({rc}): no usable gfx950 device — the hot path has no CPU fallback")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.lili_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc != OK:
            raise LiliError(f"lili error {rc}: {self.lib.lili_last_error(self.h).decode()}")

    def sync(self):
        self._chk(self.lib.lili_sync(self.h))

    def set_option(self, name, value):
        self._chk(self.lib.lili_set_option(self.h, name.encode(), int(value)))

    def set_debug(self, keep_neighbors=True):
        self._chk(self.lib.lili_set_debug(self.h, int(bool(keep_neighbors))))


class ScanToMapMatcher:
    """Host-side mirror of the reference's matcher calls (names follow BackendFusion.cpp):

        setInputCloud            -> set_input_cloud            (L/src/BackendFusion.cpp:839-840)
        findCorrespondingSurfFeatures   -> find_corresponding_surf_features     (L:1601)
        findCorrespondingCornerFeatures -> find_corresponding_corner_features   (L:1531)
        N x CostFunction::Evaluate + JtJ -> linearize                            (LidarKeyframeFactor.h, MarginalizationFactor.cpp:3-71)
    """

    def __init__(self, ctx, params):
        self.ctx = ctx
        self.lib = ctx.lib
        self.params = params

    # -- map ------------------------------------------------------------------------------------
    def set_input_cloud(self, kind, cloud, max_sq_radius=None):
        if not isinstance(cloud, Cloud):
            cloud = cloud_from_numpy(cloud, aux_col=3 if (np.ndim(cloud) == 2 and np.shape(cloud)[1] > 3) else None)
        if max_sq_radius is None:
            max_sq_radius = self.params.kd_max_radius if kind == KIND_SURF else self.params.edge_gate
        self.ctx._chk(self.lib.lili_map_set(self.ctx.h, kind, C.byref(cloud), float(max_sq_radius)))

    def set_input_cloud_begin(self, kind, cloud, max_sq_radius=None):
        """Starts building the NEXT index of `kind` on a side stream (work already enqueued keeps the current one); finish with
        set_input_cloud_end.  The cloud's memory must stay valid until then (a reference is kept here)."""
        if not isinstance(cloud, Cloud):
            cloud = cloud_from_numpy(cloud, aux_col=3 if (np.ndim(cloud) == 2 and np.shape(cloud)[1] > 3) else None)
        if max_sq_radius is None:
            max_sq_radius = self.params.kd_max_radius if kind == KIND_SURF else self.params.edge_gate
        self._pending_cloud = getattr(self, "_pending_cloud", {})
        self._pending_cloud[kind] = cloud
        self.ctx._chk(self.lib.lili_map_set_begin(self.ctx.h, kind, C.byref(cloud), float(max_sq_radius)))

    def set_input_cloud_end(self, kind):
        self.ctx._chk(self.lib.lili_map_set_end(self.ctx.h, kind))
        getattr(self, "_pending_cloud", {}).pop(kind, None)

    def map_info(self, kind):
        n, nc, ce = C.c_int64(), C.c_int64(), C.c_double()
        self.ctx._chk(self.lib.lili_map_info(self.ctx.h, kind, C.byref(n), C.byref(nc), C.byref(ce)))
        return n.value, nc.value, ce.value

    def map_build_stats(self):
        """(index builds started from a guessed box, those repeated with the true box, builds repeated with the three-kernel scan)."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.ctx._chk(self.lib.lili_map_build_stats(self.ctx.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def map_focus(self, center=None, radius=0.0):
        """Hint for the following set_input_cloud calls: build the super-row copy only within `radius` of `center` (None / 0: everywhere)."""
        if center is None or not radius > 0:
            self.ctx._chk(self.lib.lili_map_focus(self.ctx.h, None, 0.0))
        else:
            c = (C.c_double * 3)(*[float(v) for v in center])
            self.ctx._chk(self.lib.lili_map_focus(self.ctx.h, c, float(radius)))

    def map_density(self, kind):
        """(mean points per gate-sized cell, fine cell edge or 0, squared radius covered by the fine index or 0)."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.ctx._chk(self.lib.lili_map_density(self.ctx.h, kind, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # -- queries --------------------------------------------------------------------------------
    def set_queries(self, slot, kind, cloud):
        if not isinstance(cloud, Cloud):
            cloud = cloud_from_numpy(cloud, aux_col=3 if (np.ndim(cloud) == 2 and np.shape(cloud)[1] > 3) else None)
        self.ctx._chk(self.lib.lili_s2m_set_queries(self.ctx.h, slot, kind, C.byref(cloud)))

    def _associate(self, slot, kind, q, t, want_count):
        t, q = _f64(t, 3), _f64(q, 4)
        n = C.c_int(0)
        self.ctx._chk(self.lib.lili_s2m_associate(self.ctx.h, slot, kind, _ptr(t), _ptr(q), C.byref(self.params),
                                                  C.byref(n) if want_count else None))
        return n.value if want_count else None

    def find_corresponding_surf_features(self, slot, q, t, want_count=True):
        return self._associate(slot, KIND_SURF, q, t, want_count)

    def find_corresponding_corner_features(self, slot, q, t, want_count=True):
        return self._associate(slot, KIND_EDGE, q, t, want_count)

    def linearize(self, slot, t, q, kind_mask=MASK_SURF | MASK_EDGE):
        t, q = _f64(t, 3), _f64(q, 4)
        gram = np.zeros(64)
        cost = C.c_double(0)
        counts = np.zeros(2, np.int32)
        self.ctx._chk(self.lib.lili_s2m_linearize(self.ctx.h, slot, kind_mask, _ptr(t), _ptr(q), C.byref(self.params),
                                                  _ptr(gram), C.byref(cost), _ptr(counts)))
        return gram.reshape(8, 8), cost.value, counts

    # -- copy-outs ------------------------------------------------------------------------------
    def surf_records(self, slot, capacity):
        qi = np.zeros(capacity, np.int32)
        cp = np.zeros((capacity, 3), np.float32)
        nn = np.zeros((capacity, 3), np.float32)
        d = np.zeros(capacity, np.float32)
        sc = np.zeros(capacity, np.float64)
        n = C.c_size_t(0)
        self.ctx._chk(self.lib.lili_s2m_get_surf_records(self.ctx.h, slot, capacity, _ptr(qi), _ptr(cp), _ptr(nn), _ptr(d), _ptr(sc), C.byref(n)))
        k = min(n.value, capacity)
        return dict(count=n.value, query_index=qi[:k], cp=cp[:k], n=nn[:k], d=d[:k], score=sc[:k])

    def edge_records(self, slot, capacity):
        qi = np.zeros(capacity, np.int32)
        cp = np.zeros((capacity, 3), np.float32)
        a = np.zeros((capacity, 3), np.float32)
        b = np.zeros((capacity, 3), np.float32)
        s = np.zeros(capacity, np.float32)
        n = C.c_size_t(0)
        self.ctx._chk(self.lib.lili_s2m_get_edge_records(self.ctx.h, slot, capacity, _ptr(qi), _ptr(cp), _ptr(a), _ptr(b), _ptr(s), C.byref(n)))
        k = min(n.value, capacity)
        return dict(count=n.value, query_index=qi[:k], cp=cp[:k], a=a[:k], b=b[:k], s=s[:k])

    def neighbors(self, slot, kind, n_q):
        idx = np.zeros((n_q, 5), np.int32)
        d2 = np.zeros((n_q, 5), np.float32)
        self.ctx._chk(self.lib.lili_s2m_get_neighbors(self.ctx.h, slot, kind, n_q, _ptr(idx), _ptr(d2)))
        return idx, d2

    # -- device-resident iterations -------------------------------------------------------------
    def pose_set(self, slot, t, q):
        t, q = _f64(t, 3), _f64(q, 4)
        self.ctx._chk(self.lib.lili_s2m_pose_set(self.ctx.h, slot, _ptr(t), _ptr(q)))

    def pose_get(self, slot):
        t, q = np.zeros(3), np.zeros(4)
        st = C.c_int(0)
        self.ctx._chk(self.lib.lili_s2m_pose_get(self.ctx.h, slot, _ptr(t), _ptr(q), C.byref(st)))
        return t, q, st.value

    def last_step(self, slot):
        """(delta[6] = dt, rotation vector; number of GN updates since pose_set; status of the last one)."""
        d = np.zeros(6)
        n, st = C.c_int(0), C.c_int(0)
        self.ctx._chk(self.lib.lili_s2m_last_step(self.ctx.h, slot, _ptr(d), C.byref(n), C.byref(st)))
        return d, n.value, st.value

    def accumulate(self, slot, d_gram_ptr, kind_mask=MASK_SURF):
        self.ctx._chk(self.lib.lili_s2m_accumulate(self.ctx.h, slot, kind_mask, C.byref(self.params), C.c_void_p(d_gram_ptr)))

    def associate_dev(self, slot, kind_mask=MASK_SURF):
        self.ctx._chk(self.lib.lili_s2m_associate_dev(self.ctx.h, slot, kind_mask, C.byref(self.params)))

    def counts_export(self, slot, d_counts_ptr):
        self.ctx._chk(self.lib.lili_s2m_counts_export(self.ctx.h, slot, C.c_void_p(d_counts_ptr)))

    def counts_import(self, slot, d_counts_ptr):
        self.ctx._chk(self.lib.lili_s2m_counts_import(self.ctx.h, slot, C.c_void_p(d_counts_ptr)))

    def linearize_dev(self, slot, d_gram_ptr, kind_mask=MASK_SURF):
        self.ctx._chk(self.lib.lili_s2m_linearize_dev(self.ctx.h, slot, kind_mask, C.byref(self.params), C.c_void_p(d_gram_ptr)))

    def gn_update(self, slot, d_gram_ptr):
        self.ctx._chk(self.lib.lili_s2m_gn_update(self.ctx.h, slot, C.c_void_p(d_gram_ptr)))

    def iterate_sharded(self, slot, n_iters, d_counts_ptr, d_gram_ptr, allreduce_fn=None, comm=None, restart_every=0, restart_slot=1,
                        kind_mask=MASK_SURF):
        """n_iters staged iterations with the two collectives enqueued from C (lili_s2m_iterate_sharded).  allreduce_fn: address
        of an ncclAllReduce-compatible function (int), comm: its communicator handle (int); None = single rank."""
        self.ctx._chk(self.lib.lili_s2m_iterate_sharded(self.ctx.h, slot, kind_mask, C.byref(self.params), int(n_iters), int(restart_every),
                                                        int(restart_slot), C.c_void_p(allreduce_fn), C.c_void_p(comm),
                                                        C.c_void_p(d_counts_ptr), C.c_void_p(d_gram_ptr)))

    def counts_window_sharded(self, slots, d_counts_ptr, allreduce_fn=None, comm=None, kind_mask=MASK_SURF | MASK_EDGE):
        arr = (C.c_int * len(slots))(*slots)
        self.ctx._chk(self.lib.lili_s2m_counts_window_sharded(self.ctx.h, arr, len(slots), kind_mask, C.c_void_p(allreduce_fn), C.c_void_p(comm), C.c_void_p(d_counts_ptr)))

    def linearize_window_sharded(self, slots, ts, qs, d_gram_ptr, allreduce_fn=None, comm=None, kind_mask=MASK_SURF | MASK_EDGE):
        """One evaluation of the joint window with the queries of every slot sharded over the ranks: [(gram 8x8, cost, counts)] per slot,
        the same bits on every rank."""
        n = len(slots)
        arr = (C.c_int * n)(*slots)
        t = np.ascontiguousarray(np.asarray(ts, np.float64).reshape(n, 3)); q = np.ascontiguousarray(np.asarray(qs, np.float64).reshape(n, 4))
        G = np.zeros((n, 64)); cost = np.zeros(n); cnt = np.zeros((n, 2), np.int32)
        self.ctx._chk(self.lib.lili_s2m_linearize_window_sharded(self.ctx.h, arr, n, kind_mask, _ptr(t), _ptr(q), C.byref(self.params), C.c_void_p(allreduce_fn),
                                                                 C.c_void_p(comm), C.c_void_p(d_gram_ptr), _ptr(G), _ptr(cost), _ptr(cnt)))
        return [(G[i].reshape(8, 8), float(cost[i]), cnt[i].copy()) for i in range(n)]

    def iterate_window_sharded(self, slots, n_iters, d_counts_ptr, d_gram_ptr, allreduce_fn=None, comm=None, kind_mask=MASK_SURF | MASK_EDGE):
        arr = (C.c_int * len(slots))(*slots)
        self.ctx._chk(self.lib.lili_s2m_iterate_window_sharded(self.ctx.h, arr, len(slots), kind_mask, C.byref(self.params), int(n_iters), C.c_void_p(allreduce_fn),
                                                               C.c_void_p(comm), C.c_void_p(d_counts_ptr), C.c_void_p(d_gram_ptr)))

    def linearize_window_gather(self, slots, owner, rank, d_gram_ptr, allreduce_fn=None, comm=None, kind_mask=MASK_SURF | MASK_EDGE):
        """Slot-per-rank window: the records of one evaluation at the slots' device poses, gathered into d_gram (n x GRAM_DOUBLES, device)."""
        arr, own = (C.c_int * len(slots))(*slots), (C.c_int * len(slots))(*owner)
        self.ctx._chk(self.lib.lili_s2m_linearize_window_gather(self.ctx.h, arr, len(slots), kind_mask, C.byref(self.params), own, int(rank), allreduce_fn, comm, d_gram_ptr))

    def linearize_window_gather_at(self, slots, ts, qs, owner, rank, d_gram_ptr, allreduce_fn=None, comm=None, kind_mask=MASK_SURF | MASK_EDGE):
        """One solver evaluation of the slot-per-rank window at the body poses (ts[k], qs[k]): [(gram 8x8, cost, counts)] per slot, the same bits on every rank."""
        n = len(slots)
        arr, own = (C.c_int * n)(*slots), (C.c_int * n)(*owner)
        t = np.ascontiguousarray(np.asarray(ts, np.float64).reshape(n, 3)); q = np.ascontiguousarray(np.asarray(qs, np.float64).reshape(n, 4))
        G = np.zeros((n, 64)); cost = np.zeros(n); cnt = np.zeros((n, 2), np.int32)
        self.ctx._chk(self.lib.lili_s2m_linearize_window_gather_at(self.ctx.h, arr, n, kind_mask, _ptr(t), _ptr(q), C.byref(self.params), own, int(rank), allreduce_fn, comm,
                                                                   d_gram_ptr, _ptr(G), _ptr(cost), _ptr(cnt)))
        return [(G[i].reshape(8, 8), float(cost[i]), cnt[i].copy()) for i in range(n)]

    def iterate_window_gather(self, slots, n_iters, owner, rank, d_gram_ptr, allreduce_fn=None, comm=None, kind_mask=MASK_SURF | MASK_EDGE):
        arr, own = (C.c_int * len(slots))(*slots), (C.c_int * len(slots))(*owner)
        self.ctx._chk(self.lib.lili_s2m_iterate_window_gather(self.ctx.h, arr, len(slots), kind_mask, C.byref(self.params), int(n_iters), own, int(rank), allreduce_fn, comm,
                                                              d_gram_ptr))

    def iterate_window(self, slots, n_iters, kind_mask=MASK_SURF):
        arr = (C.c_int * len(slots))(*[int(s) for s in slots])
        self.ctx._chk(self.lib.lili_s2m_iterate_window(self.ctx.h, arr, len(slots), kind_mask, C.byref(self.params), int(n_iters)))

    def debug_times(self, slot):
        out = (C.c_longlong * 16)()
        self.ctx._chk(self.lib.lili_s2m_debug_times(self.ctx.h, slot, out))
        return list(out)

    def pose_copy(self, dst_slot, src_slot):
        self.ctx._chk(self.lib.lili_s2m_pose_copy(self.ctx.h, dst_slot, src_slot))

    def iterate_restart(self, slot, n_iters, restart_every, restart_slot, kind_mask=MASK_SURF, time_association=False):
        """Returns the summed association-kernel time in ms when time_association (blocking), else None (async)."""
        ms = C.c_float(0)
        self.ctx._chk(self.lib.lili_s2m_iterate_restart(self.ctx.h, slot, kind_mask, C.byref(self.params), int(n_iters), int(restart_every),
                                                        int(restart_slot), C.byref(ms) if time_association else None))
        return ms.value if time_association else None

    def associate_window(self, slots, ts_assoc, qs_assoc, kind_mask=MASK_SURF | MASK_EDGE):
        """findCorresponding{Surf,Corner}Features of all keyframes of the window at the association poses; returns [(n_surf, n_edge)] per slot."""
        n = len(slots)
        arr = (C.c_int * n)(*[int(s) for s in slots])
        t = np.ascontiguousarray(np.asarray(ts_assoc, np.float64).reshape(n, 3))
        q = np.ascontiguousarray(np.asarray(qs_assoc, np.float64).reshape(n, 4))
        counts = np.zeros((n, 2), np.int32)
        self.ctx._chk(self.lib.lili_s2m_associate_window(self.ctx.h, arr, n, int(kind_mask), _ptr(t), _ptr(q), C.byref(self.params), _ptr(counts)))
        return [(int(counts[k, 0]), int(counts[k, 1])) for k in range(n)]

    def linearize_window(self, slots, ts, qs, kind_mask=MASK_SURF | MASK_EDGE):
        """One evaluation of the joint window: (Gram 8x8, cost, counts) per slot at the body poses (ts[k], qs[k]); one synchronisation for all."""
        n = len(slots)
        arr = (C.c_int * n)(*[int(s) for s in slots])
        t = np.ascontiguousarray(np.asarray(ts, np.float64).reshape(n, 3))
        q = np.ascontiguousarray(np.asarray(qs, np.float64).reshape(n, 4))
        G = np.zeros((n, 64)); cost = np.zeros(n); counts = np.zeros((n, 2), np.int32)
        self.ctx._chk(self.lib.lili_s2m_linearize_window(self.ctx.h, arr, n, int(kind_mask), _ptr(t), _ptr(q), C.byref(self.params), _ptr(G), _ptr(cost), _ptr(counts)))
        return [(G[k].reshape(8, 8).copy(), float(cost[k]), (int(counts[k, 0]), int(counts[k, 1]))) for k in range(n)]

    def lm_options(self, **kw):
        """Ceres 2.0 defaults with max_iterations = 15 (what the reference's ceres::Solve runs with, L/src/BackendFusion.cpp:984-992), overridden by kw."""
        o = LmOptions()
        self.lib.lili_lm_default_options(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def solve_lm(self, slot, kind_mask=MASK_SURF | MASK_EDGE, options=None, want_summary=True):
        """Levenberg-Marquardt on the records of the last association, one persistent launch (ceres::Solve's loop for the lidar blocks of one
        keyframe); returns the summary as a dict (blocking) or None (asynchronous)."""
        s = LmSummary() if want_summary else None
        self.ctx._chk(self.lib.lili_s2m_solve_lm(self.ctx.h, slot, kind_mask, C.byref(self.params), C.byref(options) if options is not None else None,
                                                     C.byref(s) if s is not None else None))
        return s.as_dict() if s is not None else None

    def solve_lm_window(self, slots, kind_mask=MASK_SURF | MASK_EDGE, options=None, want_summary=True):
        arr = (C.c_int * len(slots))(*slots)
        s = (LmSummary * len(slots))() if want_summary else None
        self.ctx._chk(self.lib.lili_s2m_solve_lm_window(self.ctx.h, arr, len(slots), kind_mask, C.byref(self.params),
                                                            C.byref(options) if options is not None else None, s))
        return [x.as_dict() for x in s] if s is not None else None

    def iterate(self, slot, n_iters, kind_mask=MASK_SURF):
        self.ctx._chk(self.lib.lili_s2m_iterate(self.ctx.h, slot, kind_mask, C.byref(self.params), int(n_iters)))

    def iterate_inner(self, slot, n_iters, kind_mask=MASK_SURF, want_cost=True):
        """n_iters x (linearise + reduce + GN update) on the records of the last association (ceres::Solve's inner loop)."""
        self.ctx._chk(self.lib.lili_s2m_iterate_inner(self.ctx.h, slot, kind_mask, C.byref(self.params), int(n_iters), 1 if want_cost else 0))


def extract_rot_device(ctx):
    """(full, edge, surf) device clouds of the last RotExtractor.extract on this context."""
    f, e, s = Cloud(), Cloud(), Cloud()
    ctx._chk(ctx.lib.lili_extract_rot_device(ctx.h, C.byref(f), C.byref(e), C.byref(s)))
    return f, e, s


def extract_livox_device(ctx):
    e, s = Cloud(), Cloud()
    ctx._chk(ctx.lib.lili_extract_livox_device(ctx.h, C.byref(e), C.byref(s)))
    return e, s


def voxel_filter_device(ctx, cloud, leaf, d_out_ptr, capacity):
    """VoxelGrid of a device cloud into a caller-owned device float4 buffer; returns the output device cloud."""
    fo = FeatureOut(d_out_ptr, capacity, 16, MEM_DEVICE, 0)
    ctx._chk(ctx.lib.lili_voxel_filter(ctx.h, C.byref(cloud), float(leaf), C.byref(fo), None))
    return Cloud(d_out_ptr, min(fo.count, capacity), 16, 12, MEM_DEVICE)


def voxel_filter(ctx, pts_xyza, leaf):
    """pcl::VoxelGrid on (n,4) float32 rows (x, y, z, aux).  Returns (centroids (m,4), counts (m,))."""
    pts = np.ascontiguousarray(pts_xyza, dtype=np.float32)
    cloud = cloud_from_numpy(pts, aux_col=3)
    cap = max(pts.shape[0], 1)
    out = np.zeros((cap, 4), np.float32)
    cnt = np.zeros(cap, np.int32)
    fo = FeatureOut(out.ctypes.data, cap, 16, MEM_HOST, 0)
    ctx._chk(ctx.lib.lili_voxel_filter(ctx.h, C.byref(cloud), float(leaf), C.byref(fo), _ptr(cnt)))
    return out[:fo.count], cnt[:fo.count]


def voxel_filter_stats(ctx):
    """(filters served with guessed key bits — no host round trip for the bounding box —, guesses that did not hold) of this context so far."""
    a, b = C.c_int32(0), C.c_int32(0)
    ctx._chk(ctx.lib.lili_voxel_filter_stats(ctx.h, C.byref(a), C.byref(b)))
    return a.value, b.value


class LocalMap:
    """Keyframe ring buffer + VoxelGrid + map index on the device (buildLocalMapWithLandMark / downSampleCloud /
    setInputCloud, L/src/BackendFusion.cpp:1387-1528, 839-840)."""

    def __init__(self, ctx, kind, width, leaf, max_sq_radius=1.0):
        self.ctx, self.kind, self.width, self.leaf, self.max_sq_radius = ctx, kind, int(width), float(leaf), float(max_sq_radius)
        ctx._chk(ctx.lib.lili_localmap_reset(ctx.h, kind))

    def push(self, feats_xyza, t, q):
        pts = np.ascontiguousarray(feats_xyza, dtype=np.float32)
        cloud = cloud_from_numpy(pts, aux_col=3 if pts.shape[1] > 3 else None)
        t, q = _f64(t, 3), _f64(q, 4)
        self.ctx._chk(self.ctx.lib.lili_localmap_push(self.ctx.h, self.kind, C.byref(cloud), _ptr(t), _ptr(q), self.width))

    def get(self, capacity):
        """The down-sampled local map of the last commit, (n, 4) float32 rows in map order."""
        out = np.zeros((max(int(capacity), 1), 4), np.float32)
        fo = FeatureOut(out.ctypes.data, out.shape[0], 16, MEM_HOST, 0)
        self.ctx._chk(self.ctx.lib.lili_localmap_get(self.ctx.h, C.byref(fo)))
        return out[:min(fo.count, out.shape[0])]

    def stats(self):
        """(incremental commits, full rebuilds) of this context so far."""
        a, b = C.c_int32(0), C.c_int32(0)
        self.ctx._chk(self.ctx.lib.lili_localmap_stats(self.ctx.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def commit(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self.ctx._chk(self.ctx.lib.lili_localmap_commit(self.ctx.h, self.kind, self.leaf, self.max_sq_radius, C.byref(a), C.byref(b)))
        return a.value, b.value


class RotExtractor:
    """Host-side mirror of LiLi-OM-ROT's Preprocessing::cloudHandler (R/src/Preprocessing.cpp:248-535)."""

    def __init__(self, ctx, n_scans=64, ds_rate=4, ds_v=0.6, near_range=3.0):
        self.ctx = ctx
        self.lib = ctx.lib
        self.params = RotParams(n_scans, ds_rate, ds_v, near_range)

    def extract_device(self, d_ptr, n, q_imu=(1.0, 0, 0, 0), q_lb=(1.0, 0, 0, 0)):
        """The scan is already in HBM as float4 rows (x, y, z, intensity) at device address d_ptr; the features stay in HBM
        (extract_rot_device() hands them on as device clouds).  Returns (n_full, n_edge, n_surf)."""
        cloud = Cloud(d_ptr, n, 16, 12, MEM_DEVICE)
        qi, ql = _f64(q_imu, 4), _f64(q_lb, 4)
        outs = [FeatureOut(None, 0, 16, MEM_DEVICE, 0) for _ in range(3)]
        self.ctx._chk(self.lib.lili_extract_rot(self.ctx.h, C.byref(cloud), _ptr(qi), _ptr(ql), C.byref(self.params),
                                                C.byref(outs[0]), C.byref(outs[1]), C.byref(outs[2])))
        return outs[0].count, outs[1].count, outs[2].count

    def _out_buffers(self, cap, reuse):
        """Three (cap, 4) float32 result buffers.  reuse: page-locked buffers owned by the extractor (DMA'd into, no per-call
        allocation; the returned arrays are VIEWS that the next call overwrites — what a ROS node's callback would do with one
        message in flight)."""
        if not reuse:
            return [np.zeros((cap, 4), np.float32) for _ in range(3)]
        if getattr(self, "_pin_cap", 0) < cap:
            # the old blocks are RETIRED, not freed: arrays handed out by earlier calls are views of them (freeing them here
            # left those views pointing at released page-locked memory); they go when the extractor does
            self._retired = getattr(self, "_retired", []) + list(getattr(self, "_pins", []))
            self._pins = [PinnedArray((cap, 4), np.float32) for _ in range(3)]
            self._pin_cap = cap
        return [p.array[:cap] for p in self._pins]

    def extract(self, pts_xyzi, q_imu=(1.0, 0, 0, 0), q_lb=(1.0, 0, 0, 0), debug=False, reuse=False):
        pts = np.ascontiguousarray(pts_xyzi, dtype=np.float32)
        n = pts.shape[0]
        cloud = cloud_from_numpy(pts, aux_col=3)
        cap = max(n, 1)
        bufs = self._out_buffers(cap, reuse)
        outs = [FeatureOut(b.ctypes.data, cap, 16, MEM_HOST, 0) for b in bufs]
        qi, ql = _f64(q_imu, 4), _f64(q_lb, 4)
        self.ctx._chk(self.lib.lili_extract_rot(self.ctx.h, C.byref(cloud), _ptr(qi), _ptr(ql), C.byref(self.params),
                                                C.byref(outs[0]), C.byref(outs[1]), C.byref(outs[2])))
        res = dict(full=bufs[0][:outs[0].count], edge=bufs[1][:outs[1].count], surf=bufs[2][:outs[2].count])
        if debug:
            counts = np.zeros(8, np.int32)
            rs, re_ = np.zeros(64, np.int32), np.zeros(64, np.int32)
            self.ctx._chk(self.lib.lili_extract_rot_debug(self.ctx.h, _ptr(counts), _ptr(rs), _ptr(re_), *([None] * 8)))
            nf, ne, ns, nfl, nlf, nsu = [int(v) for v in counts[:6]]
            arr = dict(full_src=np.zeros(nf, np.int32), curvature=np.zeros(nf, np.float32), label=np.zeros(nf, np.int32),
                       edge_idx=np.zeros(ne, np.int32), sharp_idx=np.zeros(ns, np.int32), flat_idx=np.zeros(nfl, np.int32),
                       lessflat_idx=np.zeros(nlf, np.int32), surf_cnt=np.zeros(nsu, np.int32))
            self.ctx._chk(self.lib.lili_extract_rot_debug(self.ctx.h, _ptr(counts), _ptr(rs), _ptr(re_), _ptr(arr["full_src"]),
                                                          _ptr(arr["curvature"]), _ptr(arr["label"]), _ptr(arr["edge_idx"]),
                                                          _ptr(arr["sharp_idx"]), _ptr(arr["flat_idx"]), _ptr(arr["lessflat_idx"]),
                                                          _ptr(arr["surf_cnt"])))
            res.update(arr)
            res.update(ring_start=rs[:self.params.n_scans], ring_end=re_[:self.params.n_scans], half_idx=int(counts[6]))
        return res


class LivoxExtractor:
    """Host-side mirror of LiLi-OM's Preprocessing::cloudHandler (L/src/Preprocessing.cpp:194-408)."""

    def __init__(self, ctx, surf_thres=0.28, edge_thres=4.0, near_range=0.1):
        self.ctx = ctx
        self.lib = ctx.lib
        self.params = LivoxParams(surf_thres, edge_thres, near_range)

    def extract(self, pts5, q_imu=(1.0, 0, 0, 0), debug=False, pcl_layout=False, reuse=False):
        """pts5: (n,5) float32 = x, y, z, intensity, curvature.  reuse: results land in page-locked buffers owned by the extractor
        (views that the next call overwrites), see RotExtractor._out_buffers."""
        pts = np.ascontiguousarray(pts5, dtype=np.float32)
        n = pts.shape[0]
        cloud = Cloud(pts.ctypes.data if n else None, n, 20, 12, MEM_HOST)
        cap = max(n, 24000)
        w = 12 if pcl_layout else 8
        if reuse:
            if getattr(self, "_pin_key", None) != (cap, w):
                for p_ in getattr(self, "_pins", []):
                    p_.close()
                self._pins = [PinnedArray((cap, w), np.float32) for _ in range(3)]
                self._pin_key = (cap, w)
            bufs = [p_.array for p_ in self._pins]
        else:
            bufs = [np.zeros((cap, w), np.float32) for _ in range(3)]
        outs = [FeatureOut(b.ctypes.data, cap, 4 * w, MEM_HOST, 0) for b in bufs]
        qi = _f64(q_imu, 4)
        self.ctx._chk(self.lib.lili_extract_livox(self.ctx.h, C.byref(cloud), 16, _ptr(qi), C.byref(self.params),
                                                  C.byref(outs[0]), C.byref(outs[1]), C.byref(outs[2])))
        res = dict(cutted=bufs[0][:outs[0].count], edge=bufs[1][:outs[1].count], surf=bufs[2][:outs[2].count])
        if debug:
            counts = np.zeros(3, np.int32)
            self.ctx._chk(self.lib.lili_extract_livox_debug(self.ctx.h, _ptr(counts), None, None, None, None))
            arr = dict(cut_src=np.zeros(counts[0], np.int32), cell_src=np.zeros(24000, np.int32),
                       edge_cell=np.zeros(counts[1], np.int32), surf_cell=np.zeros(counts[2], np.int32))
            self.ctx._chk(self.lib.lili_extract_livox_debug(self.ctx.h, _ptr(counts), _ptr(arr["cut_src"]), _ptr(arr["cell_src"]),
                                                            _ptr(arr["edge_cell"]), _ptr(arr["surf_cell"])))
            arr["cell_src"] = arr["cell_src"].reshape(6, 4000)
            res.update(arr)
        return res


class FrontendOdometry:
    """Host-side mirror of LidarOdometry::run (L/src/LidarOdometry.cpp:652-707) behind the extraction node (L/src/Preprocessing.cpp:219-401): one
    lili_frontend_frame call per Livox scan — extraction, down_size_filter_surf, the outer iterations against the local map of the last `width`
    frames, the ring push at the pose found and the next frame's local map, all device-resident.  The frame counter reproduces the node's start-up:
    frame 0 is only stored (system not initialised, L:659-663; its surf_frames entry is empty), frame 1 is matched against its own surf features with
    8 iterations (L:283-289, 501-502), later frames against the ring with `scan_match_cnt`.  The caller keeps poseInitialization's constant-velocity
    prediction (L:415-441)."""

    def __init__(self, ctx, params=None, surf_thres=0.28, edge_thres=4.0, near_range=0.1, leaf_query=0.4, leaf_map=0.4, width=20, slot=0,
                 scan_match_cnt=6, first_match_cnt=8, reference_startup=True):
        """reference_startup = False: the replay tools' simpler start (frame 0 enters the ring with its features at the given pose, frame 1 is matched
        against it with `first_match_cnt` iterations) — same C call, other flags."""
        self.reference_startup = bool(reference_startup)
        self.ctx, self.lib = ctx, ctx.lib
        self.params = params if params is not None else make_params("frontend")
        self.livox = LivoxParams(surf_thres, edge_thres, near_range)
        self.opt = FrontendOptions(leaf_query, leaf_map, width, scan_match_cnt, slot, 0, 0)
        self.res = FrontendResult()
        self.scan_match_cnt, self.first_match_cnt = int(scan_match_cnt), int(first_match_cnt)
        self.n_frames = 0

    def reset(self):
        self.ctx._chk(self.lib.lili_frontend_reset(self.ctx.h))
        self.n_frames = 0

    def flush(self):
        """Builds the local map of everything pushed so far (the frame call leaves that to the next frame); returns (ring points, map points)."""
        a, b = C.c_int32(0), C.c_int32(0)
        self.ctx._chk(self.lib.lili_frontend_flush(self.ctx.h, C.byref(self.params), C.byref(self.opt), C.byref(a), C.byref(b)))
        return a.value, b.value

    def frame(self, scan, t_pred, q_pred, q_imu=(1.0, 0, 0, 0), timing=False):
        """scan: (n,5) float32 rows x, y, z, intensity, curvature (host; page-locked memory is read in place) or a Cloud.  Returns (t, q, info)."""
        if isinstance(scan, Cloud):
            cloud, curv_off = scan, 16
        else:
            n = scan.shape[0]
            cloud, curv_off = Cloud(scan.ctypes.data if n else None, n, 20, 12, MEM_HOST), 16
        k = self.n_frames
        self.opt.n_iters = 0 if k == 0 else (self.first_match_cnt if k == 1 else self.scan_match_cnt)
        self.opt.flags = (FRAME_PUSH_EMPTY if k == 0 else (FRAME_SELF_MAP if k == 1 else 0)) if self.reference_startup else 0
        self.opt.want_timing = 1 if timing else 0
        qi, tp, qp = _f64(q_imu, 4), _f64(t_pred, 3), _f64(q_pred, 4)
        r = self.res
        self.ctx._chk(self.lib.lili_frontend_frame(self.ctx.h, C.byref(cloud), curv_off, _ptr(qi), C.byref(self.livox), C.byref(self.params), C.byref(self.opt),
                                                   _ptr(tp), _ptr(qp), C.byref(r)))
        self.n_frames += 1
        info = dict(gn_status=r.gn_status, matched=bool(r.matched), n_edge=r.n_edge, n_surf=r.n_surf, n_query=r.n_query, n_map_raw=r.n_map_raw, n_map=r.n_map)
        if timing:
            info["stage_us"] = [r.stage_us[j] for j in range(4)]
        t, q = np.array(r.t[:], np.float64), np.array(r.q[:], np.float64)
        if q[0] < 0:
            q = -q                    # unifyQuaternion (L:538-548)
        return t, q, info


class RotFrontendOdometry(FrontendOdometry):
    """The same node behind the LOAM-style extractor of LiLi-OM-ROT (R/src/Preprocessing.cpp:248-535 + R/src/LidarOdometry.cpp:638-693): one lili_frontend_frame_rot call
    per spinning-LiDAR scan.  external_map = True: the call for a caller that keeps its own maps (the back end's matcher on one scan, BASELINE configs[0]) — the scan is
    matched against the indices set with ScanToMapMatcher.set_input_cloud, nothing joins the ring; edges = True: the edge features are queries too; leaf_query = 0: the
    surf features themselves are the queries."""

    def __init__(self, ctx, params=None, n_scans=64, ds_rate=4, ds_v=0.6, near_range=3.0, q_lb=(1.0, 0, 0, 0), leaf_query=0.4, leaf_map=0.4, width=20, slot=0,
                 scan_match_cnt=6, first_match_cnt=8, reference_startup=True, external_map=False, edges=False):
        super().__init__(ctx, params, leaf_query=leaf_query, leaf_map=leaf_map, width=width, slot=slot, scan_match_cnt=scan_match_cnt, first_match_cnt=first_match_cnt,
                         reference_startup=reference_startup)
        self.rot = RotParams(n_scans, ds_rate, ds_v, near_range)
        self.q_lb = _f64(q_lb, 4)
        self.base_flags = (FRAME_EXTERNAL_MAP if external_map else 0) | (FRAME_EDGES if edges else 0)

    def frame(self, scan, t_pred, q_pred, q_imu=(1.0, 0, 0, 0), timing=False):
        """scan: (n,4) float32 rows x, y, z, intensity (host) or a Cloud (e.g. a device float4 array).  Returns (t, q, info)."""
        if isinstance(scan, Cloud):
            cloud = scan
        else:
            n = scan.shape[0]
            cloud = Cloud(scan.ctypes.data if n else None, n, 16, 12, MEM_HOST)
        k = self.n_frames
        if self.base_flags & FRAME_EXTERNAL_MAP:
            self.opt.n_iters, self.opt.flags = self.scan_match_cnt, self.base_flags
        else:
            self.opt.n_iters = 0 if k == 0 else (self.first_match_cnt if k == 1 else self.scan_match_cnt)
            self.opt.flags = (FRAME_PUSH_EMPTY if k == 0 else (FRAME_SELF_MAP if k == 1 else 0)) if self.reference_startup else 0
        self.opt.want_timing = 1 if timing else 0
        qi, tp, qp = _f64(q_imu, 4), _f64(t_pred, 3), _f64(q_pred, 4)
        r = self.res
        self.ctx._chk(self.lib.lili_frontend_frame_rot(self.ctx.h, C.byref(cloud), _ptr(qi), _ptr(self.q_lb), C.byref(self.rot), C.byref(self.params), C.byref(self.opt),
                                                       _ptr(tp), _ptr(qp), C.byref(r)))
        self.n_frames += 1
        info = dict(gn_status=r.gn_status, matched=bool(r.matched), n_edge=r.n_edge, n_surf=r.n_surf, n_query=r.n_query, n_map_raw=r.n_map_raw, n_map=r.n_map)
        if timing:
            info["stage_us"] = [r.stage_us[j] for j in range(4)]
        t, q = np.array(r.t[:], np.float64), np.array(r.q[:], np.float64)
        if q[0] < 0:
            q = -q                    # unifyQuaternion (R/src/LidarOdometry.cpp:524-534)
        return t, q, info


class BackendKeyframes:
    """Host-side mirror of what BackendFusion does per keyframe before ceres::Solve (L/src/BackendFusion.cpp:830-980, 1387-1528): ONE lili_backend_keyframe_prepare call —
    the keyframe of the previous solve joins both local-map rings, both maps are voxel-filtered and indexed, the new keyframe's features are down-sampled into the newest
    window slot, every keyframe of the window is associated.  Everything stays in HBM."""

    def __init__(self, ctx, params, leaf_surf=0.4, leaf_edge=0.2, width=40):
        self.ctx, self.lib, self.params = ctx, ctx.lib, params
        self.opt = BackendOptions(leaf_surf, leaf_edge, leaf_surf, leaf_edge, width, 0, -1)
        self.res = BackendResult()

    @staticmethod
    def _cloud(a):
        return a if isinstance(a, Cloud) else cloud_from_numpy(a, aux_col=3 if (np.ndim(a) == 2 and np.shape(a)[1] > 3) else None)

    def prepare(self, join, new_surf, new_edge, slots, ts_assoc, qs_assoc, timing=False):
        """join: None, (surf features, edge features, t, q) of the keyframe that joins the local map at its LiDAR pose, or (slot, t, q): the down-sampled features an earlier
        call left in that matcher slot (device to device); new_surf / new_edge: the new keyframe's features
        (numpy rows or Clouds); slots: the window, oldest first (the new keyframe takes slots[-1]).  Returns ([(n_surf, n_edge)] per slot, info)."""
        n = len(slots)
        arr = (C.c_int * n)(*[int(v) for v in slots])
        t = np.ascontiguousarray(np.asarray(ts_assoc, np.float64).reshape(n, 3)); q = np.ascontiguousarray(np.asarray(qs_assoc, np.float64).reshape(n, 4))
        counts = np.zeros((n, 2), np.int32)
        keep = [self._cloud(new_surf), self._cloud(new_edge)]
        js = je = tj = qj = None
        self.opt.join_slot = -1
        if join is not None and len(join) == 3:
            self.opt.join_slot = int(join[0])
            tj, qj = _f64(join[1], 3), _f64(join[2], 4)
        elif join is not None:
            keep += [self._cloud(join[0]), self._cloud(join[1])]
            js, je = C.byref(keep[2]), C.byref(keep[3])
            tj, qj = _f64(join[2], 3), _f64(join[3], 4)
        self.opt.want_timing = 1 if timing else 0
        r = self.res
        self.ctx._chk(self.lib.lili_backend_keyframe_prepare(self.ctx.h, js, je, _ptr(tj) if tj is not None else None, _ptr(qj) if qj is not None else None,
                                                             C.byref(keep[0]), C.byref(keep[1]), arr, n, _ptr(t), _ptr(q), C.byref(self.params), C.byref(self.opt), _ptr(counts),
                                                             C.byref(r)))
        info = dict(n_map_raw=tuple(r.n_map_raw), n_map=tuple(r.n_map), n_query=tuple(r.n_query), associated=bool(r.associated))
        if timing:
            info["stage_us"] = [r.stage_us[j] for j in range(3)]
        return [(int(counts[k, 0]), int(counts[k, 1])) for k in range(n)], info


def gn_step_host(gram, t, q):
    lib = load_library()
    g = _f64(gram, 64)
    t = _f64(t, 3).copy()
    q = _f64(q, 4).copy()
    d = np.zeros(6)
    st = lib.lili_gn_step_host(_ptr(g), _ptr(t), _ptr(q), _ptr(d))
    return st, t, q, d


def gram_to_factor(gram, cost):
    """(residuals[9], jacobian[9,7]) of the ceres adapter block (include/lili_ceres_adapter.h)."""
    lib = load_library()
    g = _f64(gram, 64)
    res = np.zeros(9)
    jac = np.zeros(63)
    rc = lib.lili_gram_to_factor(_ptr(g), float(cost), _ptr(res), _ptr(jac))
    if rc != OK:
        raise LiliError(f"lili_gram_to_factor failed ({rc})")
    return res, jac.reshape(9, 7)


def assoc_transform(t, q, params):
    """(Q2, T2) = (Q * q_lb^-1, T - Q2 * t_lb): L/src/BackendFusion.cpp:929-930 (host-side pose algebra)."""
    q = np.asarray(q, np.float64)
    qlb = np.array(list(params.q_lb))
    tlb = np.array(list(params.t_lb))
    n2 = float((qlb * qlb).sum())
    qi = np.array([qlb[0], -qlb[1], -qlb[2], -qlb[3]]) / n2
    a, b = q, qi
    Q2 = np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                   a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                   a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                   a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])
    u = Q2[1:4]
    uv = 2 * np.cross(u, tlb)
    rot = tlb + Q2[0] * uv + np.cross(u, uv)
    return Q2, np.asarray(t, np.float64) - rot


def keyframe_map_pose(t_po, q_po, t_bl, q_bl):
    """Pose a stored keyframe's features are moved by when they enter the local map: (q_po * q_bl, q_po * t_bl + t_po),
    L/src/BackendFusion.cpp:1425-1426, 1462-1463 (host-side pose algebra, f64).  Returns (t, q) as LocalMap.push takes them."""
    a, b = np.asarray(q_po, np.float64), np.asarray(q_bl, np.float64)
    q = np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                  a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                  a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                  a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])
    u, v = a[1:4], np.asarray(t_bl, np.float64)
    uv = np.cross(u, v)
    uv = uv + uv
    return (v + uv * a[0]) + np.cross(u, uv) + np.asarray(t_po, np.float64), q


def body_pose_from_lidar(t_lidar, q_lidar, params):
    """Body pose (T, Q) whose association transform (Q*q_lb^-1, T - Q2*t_lb) equals the given LiDAR pose
    (up to the non-unit norm of the configured q_lb, which the reference does not normalise either)."""
    qlb = np.array(list(params.q_lb))
    qn = qlb / np.linalg.norm(qlb)
    a, b = np.asarray(q_lidar, np.float64), qn
    Q = np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                  a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                  a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                  a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])
    _, T2 = assoc_transform([0.0, 0.0, 0.0], Q, params)
    return np.asarray(t_lidar, np.float64) - T2, Q


# ------------------------------------------------------------------------------------------------
# callers / data formats either side of the path
# ------------------------------------------------------------------------------------------------
CUSTOM_POINT = np.dtype([("offset_time", "<u4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"),
                         ("reflectivity", "u1"), ("tag", "u1"), ("line", "u1")])   # livox_ros_driver/CustomPoint, 19 B


def livox_custom_to_cloud(ctx, points):
    """Livox CustomMsg points (CUSTOM_POINT array or raw bytes, 19 B each) -> (n, 12) float32 PointXYZINormal rows
    (L/src/FormatConvert.cpp:11-35), converted on the device."""
    pts = np.ascontiguousarray(np.asarray(points, dtype=CUSTOM_POINT))
    n = pts.shape[0]
    out = np.zeros((n, 12), np.float32)
    ctx._chk(ctx.lib.lili_livox_custom_to_cloud(ctx.h, pts.ctypes.data if n else None, n, CUSTOM_POINT.itemsize, MEM_HOST,
                                                out.ctypes.data if n else None, MEM_HOST))
    return out


class ImuState(C.Structure):
    _fields_ = [("idx", C.c_int64), ("t_cur", C.c_double), ("gyr0", C.c_double * 3), ("first", C.c_int32), ("reserved", C.c_int32)]


class ImuIntegrator:
    """Host-side gyro integration over one scan (L/src/Preprocessing.cpp:129-171); returns q_imu (w,x,y,z)."""

    def __init__(self):
        self.lib = load_library()
        self.state = ImuState()
        self.lib.lili_imu_reset(C.byref(self.state))

    def integrate(self, stamps, gyr, t_scan_next):
        st = np.ascontiguousarray(stamps, dtype=np.float64)
        g = np.ascontiguousarray(gyr, dtype=np.float64).reshape(-1, 3)
        q = np.zeros(4, np.float64)
        rc = self.lib.lili_imu_integrate(C.byref(self.state), st.ctypes.data if st.size else None, g.ctypes.data if g.size else None,
                                         st.shape[0], float(t_scan_next), q.ctypes.data)
        if rc != 0:
            raise LiliError(f"lili_imu_integrate failed ({rc})")
        return q


def marg_add_lidar(gram, A, b, idx_t, idx_q):
    """Adds the lidar blocks of one keyframe's Gram record into MarginalizationInfo's dense A (pos x pos, C order), b."""
    lib = load_library()
    G = np.ascontiguousarray(gram, dtype=np.float64).reshape(64)
    assert A.flags.c_contiguous and A.dtype == np.float64 and b.dtype == np.float64 and A.shape[0] == A.shape[1] == b.shape[0]
    rc = lib.lili_marg_add_lidar(G.ctypes.data, A.ctypes.data, A.shape[1], b.ctypes.data, A.shape[0], int(idx_t), int(idx_q))
    if rc != 0:
        raise LiliError(f"lili_marg_add_lidar failed ({rc})")
    return A, b
