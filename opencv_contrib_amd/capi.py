"""ctypes binding of libmiflow.so (the C-ABI in include/miflow/c_api.h).

This is the only way Python reaches the product path; there is NO CPU fallback: if the HIP
library is missing or no device is visible, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIFLOW_LIB selects an A/B build variant of the same sources (tuning experiments); default libmiflow.so
LIB_PATH = os.path.join(_HERE, os.environ.get("MIFLOW_LIB", "libmiflow.so"))

MI_8UC1, MI_32SC1, MI_32FC1, MI_32FC2, MI_32SC4 = 0, 4, 5, 13, 28
MI_SEM_CPU_REF, MI_SEM_CUDA_COMPAT = 0, 1

STATUS = {0: "MI_OK", -1: "MI_ERR_BAD_ARG", -2: "MI_ERR_BAD_TYPE", -3: "MI_ERR_BAD_SIZE", -4: "MI_ERR_HIP",
          -5: "MI_ERR_OOM", -6: "MI_ERR_NOT_IMPL", -7: "MI_ERR_NO_DEVICE"}


class MiError(RuntimeError):
    """Raised for any non-zero mi_status (the C++ shim maps the same codes to cv::Exception)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class Mat(C.Structure):
    _fields_ = [("data", C.c_void_p), ("step", C.c_size_t), ("rows", C.c_int), ("cols", C.c_int), ("type", C.c_int)]


class TVL1Params(C.Structure):
    _fields_ = [("tau", C.c_double), ("lambda_", C.c_double), ("theta", C.c_double), ("epsilon", C.c_double),
                ("scale_step", C.c_double), ("gamma", C.c_double), ("nscales", C.c_int), ("warps", C.c_int),
                ("iterations", C.c_int), ("use_initial_flow", C.c_int), ("inner_iterations", C.c_int),
                ("median_filtering", C.c_int), ("semantics", C.c_int), ("exact_math", C.c_int),
                ("time_block", C.c_int), ("lanes", C.c_int), ("stop_slack", C.c_int), ("host_feedback", C.c_int)]


class SURFParams(C.Structure):
    _fields_ = [("hessian_threshold", C.c_double), ("n_octaves", C.c_int), ("n_octave_layers", C.c_int), ("extended", C.c_int),
                ("keypoints_ratio", C.c_float), ("upright", C.c_int)]


class FarnebackParams(C.Structure):
    _fields_ = [("num_levels", C.c_int), ("pyr_scale", C.c_double), ("fast_pyramids", C.c_int), ("win_size", C.c_int),
                ("num_iters", C.c_int), ("poly_n", C.c_int), ("poly_sigma", C.c_double), ("flags", C.c_int)]


class DensePyrLKParams(C.Structure):
    _fields_ = [("win_width", C.c_int), ("win_height", C.c_int), ("max_level", C.c_int), ("iters", C.c_int), ("use_initial_flow", C.c_int)]


class SparsePyrLKParams(C.Structure):
    _fields_ = [("win_width", C.c_int), ("win_height", C.c_int), ("max_level", C.c_int), ("iters", C.c_int), ("use_initial_flow", C.c_int)]


class StereoSGMParams(C.Structure):
    _fields_ = [("min_disparity", C.c_int), ("num_disparities", C.c_int), ("P1", C.c_int), ("P2", C.c_int),
                ("uniqueness_ratio", C.c_int), ("mode", C.c_int), ("emulate_cuda_quirks", C.c_int)]


class DispBilateralParams(C.Structure):
    _fields_ = [("ndisp", C.c_int), ("radius", C.c_int), ("iters", C.c_int), ("edge_threshold", C.c_float),
                ("max_disc_threshold", C.c_float), ("sigma_range", C.c_float)]


class StereoBMParams(C.Structure):
    _fields_ = [("num_disparities", C.c_int), ("block_size", C.c_int), ("prefilter_type", C.c_int),
                ("prefilter_cap", C.c_int), ("prefilter_size", C.c_int), ("texture_threshold", C.c_float),
                ("uniqueness_ratio", C.c_int), ("emulate_cuda_edge", C.c_int)]


_lib = None


def lib():
    """Loads libmiflow.so.  torch (if it is going to be used) must be imported first so both
    share one HIP runtime (same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m opencv_contrib_amd.build` "
                          "(there is no CPU fallback for the miflow product path)")
    try:
        import torch  # noqa: F401  (pins the HIP runtime the tensors live in)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
    PM = C.POINTER(Mat)
    sig = {
        "mi_last_error": (C.c_char_p, []),
        "mi_version": (C.c_char_p, []),
        "mi_device_count": (i, []),
        "mi_set_device": (i, [i]),
        "mi_get_device": (i, [C.POINTER(i)]),
        "mi_malloc": (i, [C.POINTER(vp), C.c_size_t]),
        "mi_malloc_pitch": (i, [C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, i]),
        "mi_free": (i, [vp]),
        "mi_memcpy_h2d": (i, [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, i, vp]),
        "mi_memcpy_d2h": (i, [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, i, vp]),
        "mi_memset": (i, [vp, C.c_size_t, i, C.c_size_t, i, vp]),
        "mi_stream_create": (i, [C.POINTER(vp)]),
        "mi_stream_destroy": (i, [vp]),
        "mi_stream_synchronize": (i, [vp]),
        "mi_release_cached_memory": (i, []),
        "mi_tvl1_default_params": (None, [C.POINTER(TVL1Params)]),
        "mi_tvl1_create": (i, [C.POINTER(TVL1Params), C.POINTER(vp)]),
        "mi_tvl1_set_params": (i, [vp, C.POINTER(TVL1Params)]),
        "mi_tvl1_get_params": (i, [vp, C.POINTER(TVL1Params)]),
        "mi_tvl1_calc": (i, [vp, PM, PM, PM, vp]),
        "mi_tvl1_calc_batch": (i, [vp, i, PM, PM, PM, vp]),
        "mi_tvl1_last_iterations": (i, [vp, i, C.POINTER(i), C.POINTER(i), i, vp]),
        "mi_tvl1_set_profiling": (i, [vp, i]),
        "mi_tvl1_query_plan": (i, [i, i, i, i, C.POINTER(i), C.POINTER(i)]),
        "mi_tvl1_get_profile": (i, [vp, C.POINTER(d), C.POINTER(C.c_longlong), C.POINTER(d)]),
        "mi_tvl1_get_profile_kind": (i, [vp, i, C.POINTER(d), C.POINTER(C.c_longlong), C.POINTER(d)]),
        "mi_tvl1_get_profile_level": (i, [vp, i, i, C.POINTER(d), C.POINTER(C.c_longlong), C.POINTER(d)]),
        "mi_tvl1_destroy": (None, [vp]),
        "mi_tvl1_multi_create": (i, [C.POINTER(TVL1Params), i, C.POINTER(i), C.POINTER(vp)]),
        "mi_tvl1_multi_device_count": (i, [vp]),
        "mi_tvl1_multi_set_chunk": (i, [vp, i]),
        "mi_tvl1_multi_calc_batch": (i, [vp, i, PM, PM, PM]),
        "mi_tvl1_multi_transport": (i, [vp, C.POINTER(i), C.POINTER(i)]),
        "mi_tvl1_multi_transport_why": (C.c_char_p, []),
        "mi_tvl1_multi_destroy": (None, [vp]),
        "mi_tvl1_centered_gradient": (i, [PM, PM, PM, vp]),
        "mi_tvl1_warp_backward": (i, [i] + [PM] * 11),
        "mi_tvl1_iterate": (i, [i, i, i, PM, PM, PM, PM, PM, PM, PM, PM, f, f, f, C.POINTER(d), vp]),
        "mi_resize_linear": (i, [i, PM, PM, d, d, i, f, vp]),
        "miflow_selftest_lane_shift": (i, [C.POINTER(i)]),
        "miflow_selftest_rccl_self_copy": (i, [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(i)]),
        "miflow_selftest_jw_fault": (i, [C.POINTER(i)]),
        "miflow_selftest_farneback_poison": (i, [vp, vp]),
        "miflow_selftest_tvl1_slots": (i, [vp, i, C.POINTER(i), i, vp]),
        "mi_stereobm_default_params": (None, [C.POINTER(StereoBMParams)]),
        "mi_stereobm_create": (i, [C.POINTER(StereoBMParams), C.POINTER(vp)]),
        "mi_stereobm_set_params": (i, [vp, C.POINTER(StereoBMParams)]),
        "mi_stereobm_get_params": (i, [vp, C.POINTER(StereoBMParams)]),
        "mi_stereobm_compute": (i, [vp, PM, PM, PM, vp]),
        "mi_stereobm_compute_batch": (i, [vp, i, PM, PM, PM, vp]),
        "mi_stereobm_destroy": (None, [vp]),
        "mi_stereobm_prefilter_xsobel": (i, [PM, PM, i, vp]),
        "mi_stereobm_prefilter_norm": (i, [PM, PM, i, i, vp]),
        "mi_stereobm_block_match": (i, [PM, PM, PM, PM, i, i, i, i, vp]),
        "mi_stereobm_textureness": (i, [PM, PM, i, f, vp]),
        "miflow_selftest_wave_min": (i, [C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
        "miflow_selftest_tmax16": (i, [C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
        "mi_farneback_default_params": (None, [C.POINTER(FarnebackParams)]),
        "mi_farneback_create": (i, [C.POINTER(FarnebackParams), C.POINTER(vp)]),
        "mi_farneback_set_params": (i, [vp, C.POINTER(FarnebackParams)]),
        "mi_farneback_get_params": (i, [vp, C.POINTER(FarnebackParams)]),
        "mi_farneback_calc": (i, [vp, PM, PM, PM, vp]),
        "mi_farneback_calc_batch": (i, [vp, i, PM, PM, PM, vp]),
        "mi_farneback_destroy": (None, [vp]),
        "mi_farneback_poly_exp": (i, [PM, PM, i, d, vp]),
        "mi_farneback_update_matrices": (i, [PM, PM, PM, PM, PM, vp]),
        "mi_farneback_blur5": (i, [PM, PM, i, i, vp]),
        "mi_farneback_update_flow": (i, [PM, PM, PM, vp]),
        "mi_farneback_iterate": (i, [PM, PM, PM, PM, PM, PM, i, i, i, vp]),
        "mi_farneback_gaussian_blur": (i, [PM, PM, i, d, i, vp]),
        "mi_pyr_down": (i, [PM, PM, vp]),
        "mi_surf_default_params": (None, [C.POINTER(SURFParams)]),
        "mi_surf_create": (i, [C.POINTER(SURFParams), C.POINTER(vp)]),
        "mi_surf_set_params": (i, [vp, C.POINTER(SURFParams)]),
        "mi_surf_get_params": (i, [vp, C.POINTER(SURFParams)]),
        "mi_surf_descriptor_size": (i, [vp]),
        "mi_surf_max_features": (i, [vp, i, i, C.POINTER(i)]),
        "mi_surf_detect": (i, [vp, PM, PM, PM, C.POINTER(i), vp]),
        "mi_surf_detect_and_compute": (i, [vp, PM, PM, PM, PM, C.POINTER(i), vp]),
        "mi_surf_detect_batch": (i, [vp, i, PM, PM, PM, C.POINTER(i), vp]),
        "mi_surf_compute_orientation": (i, [vp, PM, PM, i, vp]),
        "mi_surf_compute_descriptors": (i, [vp, PM, PM, i, PM, vp]),
        "mi_surf_release_memory": (None, [vp]),
        "mi_surf_destroy": (None, [vp]),
        "mi_surf_integral": (i, [vp, PM, i, PM, vp]),
        "mi_surf_det_trace": (i, [vp, PM, i, i, PM, PM, vp]),
        "miflow_selftest_wave_scan": (i, [C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
        "mi_densepyrlk_default_params": (None, [C.POINTER(DensePyrLKParams)]),
        "mi_densepyrlk_create": (i, [C.POINTER(DensePyrLKParams), C.POINTER(vp)]),
        "mi_densepyrlk_set_params": (i, [vp, C.POINTER(DensePyrLKParams)]),
        "mi_densepyrlk_get_params": (i, [vp, C.POINTER(DensePyrLKParams)]),
        "mi_densepyrlk_calc": (i, [vp, PM, PM, PM, vp]),
        "mi_densepyrlk_destroy": (None, [vp]),
        "mi_stereosgm_default_params": (None, [C.POINTER(StereoSGMParams)]),
        "mi_stereosgm_create": (i, [C.POINTER(StereoSGMParams), C.POINTER(vp)]),
        "mi_stereosgm_set_params": (i, [vp, C.POINTER(StereoSGMParams)]),
        "mi_stereosgm_get_params": (i, [vp, C.POINTER(StereoSGMParams)]),
        "mi_stereosgm_compute": (i, [vp, PM, PM, PM, vp]),
        "mi_stereosgm_destroy": (None, [vp]),
        "mi_sgm_census": (i, [PM, PM, vp]),
        "mi_sgm_aggregate_path": (i, [PM, PM, PM, i, i, i, i, i, i, vp]),
        "mi_sgm_winner_takes_all": (i, [PM, PM, PM, i, i, C.c_float, i, vp]),
        "mi_disp_bilateral_default_params": (None, [C.POINTER(DispBilateralParams)]),
        "mi_disp_bilateral_create": (i, [C.POINTER(DispBilateralParams), C.POINTER(vp)]),
        "mi_disp_bilateral_set_params": (i, [vp, C.POINTER(DispBilateralParams)]),
        "mi_disp_bilateral_get_params": (i, [vp, C.POINTER(DispBilateralParams)]),
        "mi_disp_bilateral_apply": (i, [vp, PM, PM, PM, vp]),
        "mi_disp_bilateral_destroy": (None, [vp]),
        "mi_sparsepyrlk_default_params": (None, [C.POINTER(SparsePyrLKParams)]),
        "mi_sparsepyrlk_create": (i, [C.POINTER(SparsePyrLKParams), C.POINTER(vp)]),
        "mi_sparsepyrlk_set_params": (i, [vp, C.POINTER(SparsePyrLKParams)]),
        "mi_sparsepyrlk_get_params": (i, [vp, C.POINTER(SparsePyrLKParams)]),
        "mi_sparsepyrlk_calc": (i, [vp, PM, PM, PM, PM, PM, PM, vp]),
        "mi_sparsepyrlk_destroy": (None, [vp]),
        "mi_surfcpu_orientation": (i, [PM, PM, i, i, vp]),
        "mi_surfcpu_descriptors": (i, [PM, PM, i, i, i, PM, vp]),
        "mi_bf_create": (i, [i, C.POINTER(vp)]),
        "mi_bf_destroy": (None, [vp]),
        "mi_bf_match": (i, [vp, PM, PM, PM, PM, PM, vp]),
        "mi_bf_knn_match2": (i, [vp, PM, PM, PM, PM, PM, vp]),
        "mi_bf_knn_match": (i, [vp, PM, PM, PM, i, i, PM, PM, PM, vp]),
        "mi_bf_radius_match": (i, [vp, PM, PM, PM, i, C.c_float, PM, PM, PM, PM, vp]),
        "mi_superres_to_gray8": (i, [PM, PM, vp]),
        "mi_split_flow": (i, [PM, PM, PM, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def declared_symbols():
    """Every MI_API function include/miflow/c_api.h declares (parsed from the header)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "miflow", "c_api.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"MI_API\s+[\w\s\*]+?\b(mi_\w+)\s*\(", txt)))


def check(rc: int):
    if rc != 0:
        raise MiError(rc, lib().mi_last_error().decode())


_TORCH_TYPES = None


def mat_from_tensor(t) -> Mat:
    """torch CUDA tensor -> mi_mat.  (H,W) uint8/float32/int32 -> 8UC1/32FC1/32SC1; (H,W,2) float32 -> 32FC2;
    (H,W,4) int32 -> 32SC4.  Rows may be pitched (stride(0) arbitrary); inner dims must be dense."""
    import torch
    if not t.is_cuda:
        raise MiError(-1, "tensor must live on the GPU (no CPU fallback)")
    if t.dim() == 2:
        cn = 1
        if t.stride(1) != 1:
            raise MiError(-1, "rows must be dense")
    elif t.dim() == 3:
        cn = t.shape[2]
        if t.stride(2) != 1 or t.stride(1) != cn:
            raise MiError(-1, "channels must be interleaved and dense")
    else:
        raise MiError(-3, "expected a 2-D or 3-D tensor")
    key = (t.dtype, cn)
    types = {(torch.uint8, 1): MI_8UC1, (torch.float32, 1): MI_32FC1, (torch.float32, 2): MI_32FC2,
             (torch.int32, 1): MI_32SC1, (torch.int32, 4): MI_32SC4, (torch.int32, 2): 12,
             # frames accepted by the superres adapters only (mi_superres_to_gray8)
             (torch.int16, 1): 3,   # CV_16SC1 disparity maps (DisparityBilateralFilter)
             (torch.uint8, 3): 16, (torch.uint8, 4): 24, (torch.uint16, 1): 2, (torch.uint16, 3): 18, (torch.uint16, 4): 26,
             (torch.float32, 3): 21, (torch.float32, 4): 29}
    if key not in types:
        raise MiError(-2, f"unsupported dtype/channels {key}")
    return Mat(t.data_ptr(), t.stride(0) * t.element_size(), t.shape[0], t.shape[1], types[key])


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
