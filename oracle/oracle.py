"""ctypes front-end of oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product (opencv_contrib_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


class TVL1Params(C.Structure):
    _fields_ = [("tau", C.c_double), ("lambda_", C.c_double), ("theta", C.c_double),
                ("epsilon", C.c_double), ("scale_step", C.c_double), ("gamma", C.c_double),
                ("nscales", C.c_int), ("warps", C.c_int), ("inner_iterations", C.c_int),
                ("outer_iterations", C.c_int), ("median_filtering", C.c_int),
                ("use_initial_flow", C.c_int), ("semantics", C.c_int)]


class TVL1Stats(C.Structure):
    _fields_ = [("nscales_used", C.c_int), ("level_w", C.c_int * 32), ("level_h", C.c_int * 32),
                ("iters", (C.c_int * 64) * 32)]


class SBMParams(C.Structure):
    _fields_ = [("num_disparities", C.c_int), ("block_size", C.c_int), ("prefilter_type", C.c_int),
                ("prefilter_cap", C.c_int), ("prefilter_size", C.c_int), ("texture_threshold", C.c_float),
                ("uniqueness_ratio", C.c_int), ("emulate_edge", C.c_int)]


class SURFParams(C.Structure):
    _fields_ = [("hessian_threshold", C.c_double), ("n_octaves", C.c_int), ("n_octave_layers", C.c_int), ("extended", C.c_int),
                ("keypoints_ratio", C.c_float), ("upright", C.c_int)]


class FBParams(C.Structure):
    _fields_ = [("num_levels", C.c_int), ("pyr_scale", C.c_double), ("fast_pyramids", C.c_int), ("win_size", C.c_int),
                ("num_iters", C.c_int), ("poly_n", C.c_int), ("poly_sigma", C.c_double), ("flags", C.c_int)]


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        try:
            build()
        except Exception:
            if not os.path.exists(_LIB_PATH):
                raise
        _lib = C.CDLL(_LIB_PATH)
        L = _lib
        L.orc_scaled_dim.restype = C.c_int
        L.orc_scaled_dim.argtypes = [C.c_int, C.c_double]
        L.orc_resize_linear_cv.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_resize_linear_cuda.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_float, C.c_float]
        L.orc_remap_cubic_cv.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_int, C.c_int]
        L.orc_median_blur.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int]
        L.orc_cubic_table.restype = C.POINTER(C.c_float)
        L.orc_tvl1_default_params.argtypes = [C.POINTER(TVL1Params)]
        L.orc_tvl1_calc.restype = C.c_int
        L.orc_tvl1_calc.argtypes = [C.POINTER(TVL1Params), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_long, _f32p, C.POINTER(TVL1Stats)]
        L.orc_tvl1_centered_gradient.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p]
        L.orc_tvl1_warp.argtypes = [C.c_int] + [_f32p] * 6 + [C.c_int, C.c_int] + [_f32p] * 5
        L.orc_tvl1_proc_one_scale.argtypes = [C.POINTER(TVL1Params), _f32p, _f32p, _f32p, _f32p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p]
        L.orc_tvl1_iteration.restype = C.c_float
        L.orc_tvl1_iteration.argtypes = [C.c_int] + [_f32p] * 4 + [C.c_void_p] * 9 + [C.c_int, C.c_int] + [C.c_float] * 4
        _bind_stereobm(L)
        _bind_farneback(L)
        _bind_surf(L)
    return _lib


def _bind_stereobm(L):
    L.orc_sbm_default_params.argtypes = [C.POINTER(SBMParams)]
    L.orc_sbm_prefilter_xsobel.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
    L.orc_sbm_prefilter_norm.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_sbm_block_match.restype = C.c_int
    L.orc_sbm_block_match.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_void_p]
    L.orc_sbm_textureness.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_float, _u8p]
    L.orc_sbm_compute.restype = C.c_int
    L.orc_sbm_compute.argtypes = [C.POINTER(SBMParams), _u8p, _u8p, C.c_int, C.c_int, _u8p]


def _bind_surf(L):
    i, f = C.c_int, C.c_float
    u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
    i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    L.orc_surf_default_params.argtypes = [C.POINTER(SURFParams)]
    L.orc_surf_calc_size.restype = i
    L.orc_surf_calc_size.argtypes = [i, i]
    L.orc_surf_tables.argtypes = [_f32p] * 4
    L.orc_surf_integral.argtypes = [_u8p, i, i, u32p]
    L.orc_surf_det_trace.argtypes = [u32p, i, i, i, i, _f32p, _f32p]
    L.orc_surf_find_maxima.restype = i
    L.orc_surf_find_maxima.argtypes = [_f32p, _f32p, C.c_void_p, i, i, i, i, f, i, i32p]
    L.orc_surf_interpolate.restype = i
    L.orc_surf_interpolate.argtypes = [_f32p, i, i, i, i32p, _f32p]
    L.orc_surf_orientation.restype = f
    L.orc_surf_orientation.argtypes = [u32p, i, i, f, f, f, _f32p, _f32p, _f32p]
    L.orc_surf_descriptor.argtypes = [_u8p, i, i, f, f, f, f, i, _f32p, _f32p]
    L.orc_surf_detect_describe.restype = i
    L.orc_surf_detect_describe.argtypes = [C.POINTER(SURFParams), _u8p, C.c_void_p, i, i, _f32p, i, C.c_void_p, i]


def _bind_farneback(L):
    i, d = C.c_int, C.c_double
    L.orc_fb_default_params.argtypes = [C.POINTER(FBParams)]
    L.orc_fb_gaussian_kernel.argtypes = [i, d, _f32p]
    L.orc_fb_prepare_gaussian.restype = i
    L.orc_fb_prepare_gaussian.argtypes = [i, d, _f32p, _f32p, _f32p, _f32p]
    L.orc_fb_gaussian_blur.argtypes = [_f32p, _f32p, i, i, i, C.c_void_p, i]
    L.orc_fb_poly_exp.argtypes = [_f32p, i, i, i, _f32p, _f32p, _f32p, _f32p, _f32p]
    L.orc_fb_update_matrices.argtypes = [_f32p] * 5 + [i, i]
    L.orc_fb_update_flow.argtypes = [_f32p, _f32p, _f32p, i, i]
    L.orc_fb_blur5.argtypes = [_f32p, _f32p, i, i, i, C.c_void_p]
    L.orc_fb_pyr_down.argtypes = [_f32p, i, i, _f32p, i, i]
    L.orc_fb_calc.restype = i
    L.orc_fb_calc.argtypes = [C.POINTER(FBParams), C.c_void_p, C.c_void_p, i, i, i, _f32p]


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


# ---------------------------------------------------------------- image primitives
def scaled_dim(n: int, f: float) -> int:
    return lib().orc_scaled_dim(n, f)


def resize_linear_cv(src, dsize=None, fx=0.0, fy=0.0):
    """cv::resize(src, dst, dsize, fx, fy, INTER_LINEAR) for CV_32FC1."""
    src = _c(src)
    sh, sw = src.shape
    if dsize is None:
        dw, dh = scaled_dim(sw, fx), scaled_dim(sh, fy)
        isx, isy = fx, fy
    else:
        dw, dh = dsize
        isx, isy = dw / sw, dh / sh
    dst = np.empty((dh, dw), np.float32)
    lib().orc_resize_linear_cv(src, sw, sh, dst, dw, dh, 1.0 / isx, 1.0 / isy)
    return dst


def resize_linear_cuda(src, dsize=None, fx=0.0, fy=0.0):
    """cv::cuda::resize(..., INTER_LINEAR) for CV_32FC1."""
    src = _c(src)
    sh, sw = src.shape
    if dsize is None:
        dw, dh = scaled_dim(sw, fx), scaled_dim(sh, fy)
    else:
        dw, dh = dsize
        fx, fy = dw / sw, dh / sh
    dst = np.empty((dh, dw), np.float32)
    if (dw, dh) == (sw, sh):
        return src.copy()
    lib().orc_resize_linear_cuda(src, sw, sh, dst, dw, dh, np.float32(1.0 / fx), np.float32(1.0 / fy))
    return dst


def remap_cubic_cv(src, mapx, mapy):
    src, mapx, mapy = _c(src), _c(mapx), _c(mapy)
    sh, sw = src.shape
    dh, dw = mapx.shape
    dst = np.empty((dh, dw), np.float32)
    lib().orc_remap_cubic_cv(src, sw, sh, mapx, mapy, dst, dw, dh)
    return dst


def median_blur(src, ksize=5):
    src = _c(src)
    h, w = src.shape
    dst = np.empty_like(src)
    lib().orc_median_blur(src, dst, w, h, ksize)
    return dst


def cubic_table():
    p = lib().orc_cubic_table()
    return np.ctypeslib.as_array(p, shape=(32, 4)).copy()


# ---------------------------------------------------------------- TV-L1
def tvl1_params(**kw) -> TVL1Params:
    """CPU class defaults (optflow/src/tvl1flow.cpp:386-400), overridden by kw.
    `iterations=N` is the cv::cuda spelling: inner=1, outer=N, median=1
    (equivalence used by cudaoptflow/perf/perf_optflow.cpp:319-323)."""
    p = TVL1Params()
    lib().orc_tvl1_default_params(C.byref(p))
    if "iterations" in kw:
        p.inner_iterations, p.outer_iterations, p.median_filtering = 1, int(kw.pop("iterations")), 1
    for k, v in kw.items():
        k = "lambda_" if k in ("lambda", "lambda_") else k
        if not hasattr(p, k):
            raise TypeError(f"unknown TV-L1 parameter {k}")
        setattr(p, k, v)
    return p


def tvl1_calc(I0, I1, params: TVL1Params | None = None, init_flow=None, return_stats=False):
    p = params or tvl1_params()
    I0 = np.ascontiguousarray(I0)
    I1 = np.ascontiguousarray(I1)
    if I0.dtype == np.uint8:
        typ = 0
    elif I0.dtype == np.float32:
        typ = 1
    else:
        raise ValueError("I0 must be uint8 or float32")  # CV_Assert, optflow tvl1flow.cpp:417
    if I0.shape != I1.shape or I0.dtype != I1.dtype:
        raise ValueError("I0/I1 size or type mismatch")  # :418-419
    h, w = I0.shape
    flow = np.zeros((h, w, 2), np.float32)
    if p.use_initial_flow:
        if init_flow is None or init_flow.shape != (h, w, 2):
            raise ValueError("initial flow of the frame size required")  # :420
        flow[...] = init_flow
    st = TVL1Stats()
    rc = lib().orc_tvl1_calc(C.byref(p), I0.ctypes.data, I1.ctypes.data, typ, w, h,
                             I0.strides[0], flow.reshape(-1), C.byref(st))
    if rc != 0:
        raise ValueError(f"orc_tvl1_calc failed: {rc}")
    if return_stats:
        ns = st.nscales_used
        stats = {"nscales": ns,
                 "levels": [(st.level_w[s], st.level_h[s]) for s in range(ns)],
                 "iters": [[st.iters[s][w_] for w_ in range(p.warps)] for s in range(ns)]}
        return flow, stats
    return flow


def tvl1_centered_gradient(src):
    src = _c(src)
    h, w = src.shape
    dx, dy = np.empty_like(src), np.empty_like(src)
    lib().orc_tvl1_centered_gradient(src, w, h, dx, dy)
    return dx, dy


def tvl1_warp(semantics, I0, I1, I1x, I1y, u1, u2):
    I0, I1, I1x, I1y, u1, u2 = map(_c, (I0, I1, I1x, I1y, u1, u2))
    h, w = I0.shape
    outs = [np.empty_like(I0) for _ in range(5)]
    lib().orc_tvl1_warp(semantics, I0, I1, I1x, I1y, u1, u2, w, h, *outs)
    return tuple(outs)  # I1w, I1wx, I1wy, grad, rho_c


def tvl1_iteration(semantics, I1wx, I1wy, grad, rho_c, u1, u2, p11, p12, p21, p22, l_t, theta, taut,
                   gamma=0.0, u3=None, p31=None, p32=None):
    """In place on copies; returns (error, u1, u2, p11, p12, p21, p22[, u3, p31, p32])."""
    st = [_c(a).copy() for a in (I1wx, I1wy, grad, rho_c)]
    dyn = [_c(a).copy() for a in (u1, u2)]
    u3c = _c(u3).copy() if u3 is not None else None
    ps = [_c(a).copy() for a in (p11, p12, p21, p22)]
    p3 = [_c(a).copy() if a is not None else None for a in (p31, p32)]
    h, w = st[0].shape
    ptr = lambda a: a.ctypes.data if a is not None else None
    e = lib().orc_tvl1_iteration(semantics, *st, ptr(dyn[0]), ptr(dyn[1]), ptr(u3c), ptr(ps[0]), ptr(ps[1]),
                                 ptr(ps[2]), ptr(ps[3]), ptr(p3[0]), ptr(p3[1]), w, h,
                                 l_t, theta, taut, gamma)
    out = (float(e), dyn[0], dyn[1], *ps)
    if u3 is not None:
        out = out + (u3c, p3[0], p3[1])
    return out


def tvl1_proc_one_scale(I0, I1, u1, u2, params: TVL1Params):
    """One pyramid level (gamma == 0): returns (u1, u2, iterations executed per warp)."""
    I0, I1 = _c(I0), _c(I1)
    u1, u2 = _c(u1).copy(), _c(u2).copy()
    h, w = I0.shape
    iters = np.zeros(64, np.int32)
    lib().orc_tvl1_proc_one_scale(C.byref(params), I0, I1, u1, u2, None, w, h, iters.ctypes.data)
    return u1, u2, iters[:params.warps].copy()


# ---------------------------------------------------------------- StereoBM (cv::cuda::StereoBM semantics)
def sbm_params(**kw) -> SBMParams:
    """createStereoBM defaults (cudastereo.hpp:90, stereobm.cpp:129-132), overridden by kw."""
    p = SBMParams()
    lib().orc_sbm_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown StereoBM parameter {k}")
        setattr(p, k, v)
    return p


def _u8(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8 or a.ndim != 2:
        raise ValueError("CV_8UC1 image required")  # CV_Assert, stereobm.cpp:151
    return a


def sbm_prefilter_xsobel(img, cap=31):
    img = _u8(img)
    out = np.empty_like(img)
    lib().orc_sbm_prefilter_xsobel(img, out, img.shape[0], img.shape[1], cap)
    return out


def sbm_prefilter_norm(img, cap=31, winsize=9):
    img = _u8(img)
    out = np.empty_like(img)
    lib().orc_sbm_prefilter_norm(img, out, img.shape[0], img.shape[1], cap, winsize)
    return out


def sbm_block_match(left, right, ndisp=64, winsz=19, uniqueness_ratio=0, emulate_edge=True, return_ssd=False):
    left, right = _u8(left), _u8(right)
    if left.shape != right.shape:
        raise ValueError("left/right size mismatch")  # stereobm.cpp:152
    disp = np.empty_like(left)
    ssd = np.empty(left.shape, np.uint32) if return_ssd else None
    rc = lib().orc_sbm_block_match(left, right, left.shape[0], left.shape[1], ndisp, winsz, uniqueness_ratio,
                                   int(emulate_edge), disp, ssd.ctypes.data if return_ssd else None)
    if rc != 0:
        raise ValueError(f"orc_sbm_block_match failed: {rc}")
    return (disp, ssd) if return_ssd else disp


def sbm_textureness(img, disp, winsz=19, avg_threshold=3.0):
    img = _u8(img)
    out = _u8(disp).copy()
    lib().orc_sbm_textureness(img, img.shape[0], img.shape[1], winsz, avg_threshold, out)
    return out


def sbm_compute(left, right, params: SBMParams | None = None):
    p = params or sbm_params()
    left, right = _u8(left), _u8(right)
    if left.shape != right.shape:
        raise ValueError("left/right size mismatch")
    disp = np.empty_like(left)
    rc = lib().orc_sbm_compute(C.byref(p), left, right, left.shape[0], left.shape[1], disp)
    if rc != 0:
        raise ValueError(f"orc_sbm_compute failed: {rc}")
    return disp


# ---------------------------------------------------------------- Farneback (cv::cuda::FarnebackOpticalFlow semantics)
OPTFLOW_USE_INITIAL_FLOW, OPTFLOW_FARNEBACK_GAUSSIAN = 4, 256
BORDER_REPLICATE, BORDER_REFLECT101 = 1, 4


def fb_params(**kw) -> FBParams:
    """FarnebackOpticalFlow::create defaults (cudaoptflow.hpp:285-293), overridden by kw."""
    p = FBParams()
    lib().orc_fb_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown Farneback parameter {k}")
        setattr(p, k, v)
    return p


def fb_gaussian_kernel(n, sigma):
    k = np.empty(n, np.float32)
    lib().orc_fb_gaussian_kernel(n, float(sigma), k)
    return k


def fb_prepare_gaussian(n, sigma):
    g, xg, xxg, ig = (np.zeros(8, np.float32) for _ in range(3)), None, None, np.zeros(4, np.float32)
    g, xg, xxg = np.zeros(8, np.float32), np.zeros(8, np.float32), np.zeros(8, np.float32)
    rc = lib().orc_fb_prepare_gaussian(n, float(sigma), g, xg, xxg, ig)
    if rc:
        raise ValueError("prepareGaussian failed")
    return g[: n + 1], xg[: n + 1], xxg[: n + 1], ig


def fb_gaussian_blur(src, ksize, sigma, border=BORDER_REFLECT101):
    src = _c(src)
    k = fb_gaussian_kernel(ksize, sigma)
    half = np.ascontiguousarray(k[ksize // 2:])
    dst = np.empty_like(src)
    lib().orc_fb_gaussian_blur(src, dst, src.shape[1], src.shape[0], ksize // 2, half.ctypes.data, border)
    return dst


def fb_poly_exp(src, poly_n=5, poly_sigma=1.1):
    src = _c(src)
    h, w = src.shape
    g, xg, xxg, ig = fb_prepare_gaussian(poly_n, poly_sigma)
    pad = lambda a: np.ascontiguousarray(np.concatenate([a, np.zeros(8 - len(a), np.float32)]))
    dst = np.empty((5 * h, w), np.float32)
    lib().orc_fb_poly_exp(src, w, h, poly_n, pad(g), pad(xg), pad(xxg), ig, dst)
    return dst


def fb_update_matrices(flowx, flowy, R0, R1):
    flowx, flowy, R0, R1 = map(_c, (flowx, flowy, R0, R1))
    h, w = flowx.shape
    M = np.empty((5 * h, w), np.float32)
    lib().orc_fb_update_matrices(flowx, flowy, R0, R1, M, w, h)
    return M


def fb_update_flow(M):
    M = _c(M)
    h, w = M.shape[0] // 5, M.shape[1]
    fx, fy = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
    lib().orc_fb_update_flow(M, fx, fy, w, h)
    return fx, fy


def fb_blur5(M, ksize, gaussian_sigma=None):
    M = _c(M)
    h, w = M.shape[0] // 5, M.shape[1]
    out = np.empty_like(M)
    if gaussian_sigma is None:
        lib().orc_fb_blur5(M, out, w, h, ksize // 2, None)
    else:
        k = np.ascontiguousarray(fb_gaussian_kernel(ksize, gaussian_sigma)[ksize // 2:])
        lib().orc_fb_blur5(M, out, w, h, ksize // 2, k.ctypes.data)
    return out


def fb_pyr_down(src):
    src = _c(src)
    h, w = src.shape
    dst = np.empty(((h + 1) // 2, (w + 1) // 2), np.float32)
    lib().orc_fb_pyr_down(src, w, h, dst, dst.shape[1], dst.shape[0])
    return dst


def fb_calc(I0, I1, params: FBParams | None = None, init_flow=None):
    p = params or fb_params()
    I0, I1 = np.ascontiguousarray(I0), np.ascontiguousarray(I1)
    if I0.ndim != 2 or I1.ndim != 2:
        raise ValueError("single-channel frames required")      # farneback.cpp:173
    if I0.shape != I1.shape or I0.dtype != I1.dtype:
        raise ValueError("frame size/type mismatch")             # :174
    if I0.dtype == np.uint8:
        typ = 0
    elif I0.dtype == np.float32:
        typ = 1
    else:
        raise ValueError("uint8 or float32 frames")
    h, w = I0.shape
    flow = np.zeros((h, w, 2), np.float32)
    if p.flags & OPTFLOW_USE_INITIAL_FLOW:
        if init_flow is None or init_flow.shape != (h, w, 2):
            raise ValueError("initial flow of the frame size required")   # :181-182
        flow[...] = init_flow
    rc = lib().orc_fb_calc(C.byref(p), I0.ctypes.data, I1.ctypes.data, typ, w, h, flow.reshape(-1))
    if rc:
        raise ValueError(f"orc_fb_calc failed: {rc}")
    return flow


# ---------------------------------------------------------------- SURF (cv::cuda::SURF_CUDA semantics)
def surf_params(**kw) -> SURFParams:
    """SURF_CUDA::create defaults (cuda.hpp:117-118), overridden by kw."""
    p = SURFParams()
    lib().orc_surf_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown SURF parameter {k}")
        setattr(p, k, v)
    return p


def surf_tables():
    ax, ay, aw, dw = (np.empty(113, np.float32) for _ in range(3)), None, None, np.empty(400, np.float32)
    ax, ay, aw = np.empty(113, np.float32), np.empty(113, np.float32), np.empty(113, np.float32)
    lib().orc_surf_tables(ax, ay, aw, dw)
    return ax, ay, aw, dw


def surf_integral(img):
    img = _u8(img)
    s = np.empty((img.shape[0] + 1, img.shape[1] + 1), np.uint32)
    lib().orc_surf_integral(img, img.shape[0], img.shape[1], s)
    return s


def surf_det_trace(sum_, octave, n_octave_layers=2):
    sum_ = np.ascontiguousarray(sum_, np.uint32)
    rows, cols = sum_.shape[0] - 1, sum_.shape[1] - 1
    lr = rows >> octave
    det = np.empty(((n_octave_layers + 2) * lr, cols), np.float32)
    tr = np.empty_like(det)
    lib().orc_surf_det_trace(sum_, rows, cols, octave, n_octave_layers, det, tr)
    return det, tr


def surf_detect_describe(img, params: SURFParams | None = None, mask=None, want_desc=True):
    """-> dict(x, y, laplacian, octave, size, angle, hessian, descriptors) in the deterministic scan order."""
    p = params or surf_params()
    img = _u8(img)
    rows, cols = img.shape
    maxf = min(int(np.float32(rows * cols) * np.float32(p.keypoints_ratio)), 65535)
    if maxf <= 0:
        raise ValueError("maxFeatures <= 0")
    kp = np.zeros((7, maxf), np.float32)
    dsz = 128 if p.extended else 64
    desc = np.zeros((maxf, dsz), np.float32) if want_desc else None
    m = _u8(mask) if mask is not None else None
    if m is not None and m.shape != img.shape:
        raise ValueError("mask size mismatch")    # surf.cuda.cpp:140
    n = lib().orc_surf_detect_describe(C.byref(p), img, m.ctypes.data if m is not None else None, rows, cols, kp.reshape(-1), maxf,
                                       desc.ctypes.data if want_desc else None, int(want_desc))
    if n < 0:
        raise ValueError(f"orc_surf_detect_describe failed: {n}")
    ki = kp.view(np.int32)
    return {"n": n, "x": kp[0, :n].copy(), "y": kp[1, :n].copy(), "laplacian": ki[2, :n].copy(), "octave": ki[3, :n].copy(),
            "size": kp[4, :n].copy(), "angle": kp[5, :n].copy(), "hessian": kp[6, :n].copy(),
            "descriptors": desc[:n].copy() if want_desc else None}


def pyr_down_u8(src):
    """cuda::pyrDown of a CV_8UC1 image (the pyramid of SparsePyrLKOpticalFlow)."""
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    dst = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyr_down_u8(src.ctypes.data_as(C.c_void_p), w, h, dst.ctypes.data_as(C.c_void_p), dst.shape[1], dst.shape[0])
    return dst


def pyrlk_sparse(prev, nxt, prev_pts, win_size=(21, 21), max_level=3, iters=30, next_pts=None):
    """cv::cuda::SparsePyrLKOpticalFlow on CV_8UC1 frames (oracle/pyrlk_ref.c).  prev_pts (N, 2) float32; next_pts given = useInitialFlow.
    -> (next_pts (N, 2), status (N,) uint8, err (N,) float32)."""
    prev, nxt = _u8(prev), _u8(nxt)
    if prev.shape != nxt.shape:
        raise ValueError("prevImg.size() == nextImg.size()")
    pp = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
    n = pp.shape[0]
    use_init = next_pts is not None
    out = np.ascontiguousarray(next_pts, np.float32).reshape(-1, 2).copy() if use_init else np.zeros_like(pp)
    if out.shape != pp.shape:
        raise ValueError("nextPts.size() == prevPts.size()")
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    L = lib()
    L.orc_pyrlk_sparse.restype = C.c_int
    L.orc_pyrlk_sparse.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rc = L.orc_pyrlk_sparse(prev.ctypes.data, nxt.ctypes.data, prev.shape[0], prev.shape[1], pp.ctypes.data, out.ctypes.data, n,
                            win_size[0], win_size[1], max_level, iters, int(use_init), st.ctypes.data, err.ctypes.data)
    if rc:
        raise ValueError(f"orc_pyrlk_sparse failed: {rc}")
    return out, st, err


# ------------------------------------------------------------------ the reference's CPU SURF class (oracle/surfcpu_ref.c)
def surfcpu_detect(img, hessian_threshold=100.0, n_octaves=4, n_octave_layers=3, mask=None, cap=65536):
    """xfeatures2d::SURF::detect without the orientation pass -> (n, 7) float32 rows {x, y, size, angle = -1, response, octave,
    class_id}, sorted like the reference (response descending)."""
    img = _u8(img)
    m = _u8(mask) if mask is not None else None
    kp = np.zeros((cap, 7), np.float32)
    L = lib()
    L.orc_surfcpu_detect.restype = C.c_int
    L.orc_surfcpu_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = L.orc_surfcpu_detect(img.ctypes.data, m.ctypes.data if m is not None else None, img.shape[0], img.shape[1],
                             hessian_threshold, n_octaves, n_octave_layers, kp.ctypes.data, cap)
    if n < 0:
        raise ValueError(f"orc_surfcpu_detect failed: {n}")
    return kp[:min(n, cap)].copy()


def surfcpu_compute(img, keypoints, extended=False, upright=False, want_desc=True):
    """SURFInvoker over the given keypoints: -> (keypoints with their angle, descriptors) with the keypoints the reference erases
    (no orientation sample inside the image) removed."""
    img = _u8(img)
    kp = np.ascontiguousarray(keypoints, np.float32).copy()
    n = kp.shape[0]
    desc = np.zeros((n, 128 if extended else 64), np.float32) if want_desc else None
    L = lib()
    L.orc_surfcpu_compute.restype = C.c_int
    L.orc_surfcpu_compute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rc = L.orc_surfcpu_compute(img.ctypes.data, img.shape[0], img.shape[1], kp.ctypes.data, n, int(extended), int(upright),
                               desc.ctypes.data if want_desc else None)
    if rc:
        raise ValueError(f"orc_surfcpu_compute failed: {rc}")
    keep = kp[:, 2] > 0
    return kp[keep], (desc[keep] if want_desc else None)


def surfcpu_detect_and_compute(img, hessian_threshold=100.0, n_octaves=4, n_octave_layers=3, extended=False, upright=False, mask=None,
                               want_desc=True):
    """SURF_Impl::detectAndCompute (surf.cpp:881-1015)."""
    kp = surfcpu_detect(img, hessian_threshold, n_octaves, n_octave_layers, mask)
    return surfcpu_compute(img, kp, extended, upright, want_desc)


# ------------------------------------------------------------------ superres adapter data formats
def superres_to_gray8(frame):
    """cv::superres::convertToType(frame, CV_8UC1) for GpuMat frames, restated in numpy:
    convertToCn = cuda::cvtColor BGR2GRAY / BGRA2GRAY (integer types CV_DESCALE(b*1868 + g*9617 + r*4899, 14); float types
    0.114 b + 0.587 g + 0.299 r in binary32, left to right), then convertToDepth = convertTo(CV_8U, 255 / maxVal(depth)) with
    saturate_cast<uchar> (round half to even, clamp) -- superres/src/input_array_utility.cpp:165-234,291-314."""
    f = np.asarray(frame)
    if f.ndim == 2:
        g = f
    else:
        if f.shape[2] not in (3, 4):
            raise ValueError("scn == 1 || scn == 3 || scn == 4")
        b, gch, r = f[..., 0], f[..., 1], f[..., 2]
        if f.dtype == np.float32:
            g = (b * np.float32(0.114) + gch * np.float32(0.587)).astype(np.float32) + r * np.float32(0.299)
            g = g.astype(np.float32)
        else:
            acc = b.astype(np.uint64) * 1868 + gch.astype(np.uint64) * 9617 + r.astype(np.uint64) * 4899 + (1 << 13)
            g = (acc >> 14).astype(f.dtype)
    if g.dtype == np.uint8:
        return np.ascontiguousarray(g)
    scale = np.float32(255.0 / 65535.0) if g.dtype == np.uint16 else np.float32(255.0)
    v = scale * g.astype(np.float32)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ DisparityBilateralFilter (part)
class DBFParams(C.Structure):
    _fields_ = [("ndisp", C.c_int), ("radius", C.c_int), ("iters", C.c_int), ("edge_threshold", C.c_float),
                ("max_disc_threshold", C.c_float), ("sigma_range", C.c_float)]


def dbf_params(ndisp=64, radius=3, iters=1, edge_threshold=0.1, max_disc_threshold=0.2, sigma_range=10.0):
    """createDisparityBilateralFilter defaults: cudastereo.hpp + disparity_bilateral_filter.cpp:125-136."""
    return DBFParams(ndisp, radius, iters, edge_threshold, max_disc_threshold, sigma_range)


def dbf_apply(disp, img, params: DBFParams | None = None):
    """cv::cuda::DisparityBilateralFilter::apply restated on the CPU.  disp: uint8 or int16 (H, W); img: uint8 (H, W) or (H, W, 3)."""
    p = params or dbf_params()
    d = np.ascontiguousarray(disp).copy()
    im = np.ascontiguousarray(img, dtype=np.uint8)
    if d.dtype not in (np.uint8, np.int16):
        raise ValueError("disp.type() == CV_8U || disp.type() == CV_16S")
    cn = 1 if im.ndim == 2 else im.shape[2]
    if im.shape[:2] != d.shape:
        raise ValueError("disp.size() == img.size()")
    L = lib()
    L.orc_dbf_apply.restype = C.c_int
    L.orc_dbf_apply.argtypes = [C.POINTER(DBFParams), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    rc = L.orc_dbf_apply(C.byref(p), d.ctypes.data, d.itemsize, im.ctypes.data, cn, d.shape[0], d.shape[1])
    if rc:
        raise ValueError(f"orc_dbf_apply failed: {rc}")
    return d


# ------------------------------------------------------------------ brute-force L2 matcher (part)
def bf_knn_match2(query, train, mask=None):
    """-> (train_idx (nq, 2) int32, distance (nq, 2) float32); column 0 is what match() returns."""
    q = np.ascontiguousarray(query, dtype=np.float32)
    t = np.ascontiguousarray(train, dtype=np.float32)
    if q.ndim != 2 or t.ndim != 2 or q.shape[1] != t.shape[1]:
        raise ValueError("query.cols == train.cols")
    m = None
    if mask is not None:
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        if m.shape != (q.shape[0], t.shape[0]):
            raise ValueError("mask must be query.rows x train.rows")
    idx = np.empty((q.shape[0], 2), np.int32)
    dist = np.empty((q.shape[0], 2), np.float32)
    L = lib()
    L.orc_bf_knn2.restype = C.c_int
    L.orc_bf_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.orc_bf_knn2(q.ctypes.data, q.shape[0], t.ctypes.data, t.shape[0], q.shape[1], m.ctypes.data if m is not None else None,
                       idx.ctypes.data, dist.ctypes.data)
    if rc:
        raise ValueError(f"orc_bf_knn2 failed: {rc}")
    return idx, dist


NORM_L1, NORM_L2 = 2, 4   # cv::NormTypes


NORM_HAMMING = 6
_BF_INT_DEPTHS = {"uint8": (NORM_L1, NORM_HAMMING), "uint16": (NORM_L1, NORM_HAMMING), "int16": (NORM_L1,), "int32": (NORM_L1, NORM_HAMMING)}


def _bf_is_int(query, norm):
    """(depth, norm) table of the reference (brute_force_matcher.cpp:336-356): float32 -> L1 / L2; integer depths -> the int path."""
    dt = np.asarray(query).dtype
    if dt == np.float32 or dt == np.float64:
        if norm not in (NORM_L1, NORM_L2):
            raise ValueError("unsupported combination of query.depth() and norm")
        return False
    if dt.name not in _BF_INT_DEPTHS or norm not in _BF_INT_DEPTHS[dt.name]:
        raise ValueError("unsupported combination of query.depth() and norm")
    return True


def _bf_collection(query, trains, masks, dtype=np.float32):
    q = np.ascontiguousarray(query, dtype=dtype)
    single = isinstance(trains, np.ndarray)
    ts = [np.ascontiguousarray(t, dtype=dtype) for t in ([trains] if single else trains)]
    if q.ndim != 2 or not ts or any(t.ndim != 2 or t.shape[1] != q.shape[1] for t in ts):
        raise ValueError("query.cols == train.cols")
    if masks is None:
        ms = None
    else:
        ms = [None if m is None else np.ascontiguousarray(m, dtype=np.uint8) for m in ([masks] if single else masks)]
        if len(ms) != len(ts) or any(m is not None and m.shape != (q.shape[0], t.shape[0]) for m, t in zip(ms, ts)):
            raise ValueError("masks must be query.rows x train.rows, one per train image")
    n = len(ts)
    tp = (C.c_void_p * n)(*[t.ctypes.data for t in ts])
    nts = (C.c_int * n)(*[t.shape[0] for t in ts])
    mp = None if ms is None else (C.c_void_p * n)(*[None if m is None else m.ctypes.data for m in ms])
    return q, ts, ms, n, tp, nts, mp


def bf_knn_match(query, trains, k, norm=NORM_L2, masks=None):
    """trains: one (nt, d) array or a list of them (collection).  -> (train_idx, img_idx, distance), each (nq, k); missing
    entries are (-1, -1, FLT_MAX)."""
    is_int = _bf_is_int(query, norm)
    q, ts, ms, n, tp, nts, mp = _bf_collection(query, trains, masks, np.int32 if is_int else np.float32)
    idx = np.empty((q.shape[0], k), np.int32)
    img = np.empty((q.shape[0], k), np.int32)
    dist = np.empty((q.shape[0], k), np.float32)
    L = lib()
    fn = L.orc_bf_knn_int if is_int else L.orc_bf_knn
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p]
    rc = fn(q.ctypes.data, q.shape[0], tp, nts, mp, n, q.shape[1], norm, k, idx.ctypes.data, img.ctypes.data, dist.ctypes.data)
    if rc:
        raise ValueError(f"orc_bf_knn failed: {rc}")
    return idx, img, dist


def bf_radius_match(query, trains, max_distance, cols, norm=NORM_L2, masks=None):
    """-> (train_idx, img_idx, distance) each (nq, cols) (entries past min(n, cols) are -1 / -1 / 0) and n (nq,), the number of
    train descriptors closer than max_distance; stored in ascending (image, train) order."""
    is_int = _bf_is_int(query, norm)
    q, ts, ms, n, tp, nts, mp = _bf_collection(query, trains, masks, np.int32 if is_int else np.float32)
    idx = np.full((q.shape[0], cols), -1, np.int32)
    img = np.full((q.shape[0], cols), -1, np.int32)
    dist = np.zeros((q.shape[0], cols), np.float32)
    cnt = np.zeros(q.shape[0], np.int32)
    L = lib()
    fn = L.orc_bf_radius_int if is_int else L.orc_bf_radius
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = fn(q.ctypes.data, q.shape[0], tp, nts, mp, n, q.shape[1], norm, max_distance, cols, idx.ctypes.data,
            img.ctypes.data, dist.ctypes.data, cnt.ctypes.data)
    if rc:
        raise ValueError(f"orc_bf_radius failed: {rc}")
    return idx, img, dist, cnt


# ------------------------------------------------------------------ StereoSGM
class SGMParams(C.Structure):
    _fields_ = [("min_disparity", C.c_int), ("num_disparities", C.c_int), ("P1", C.c_int), ("P2", C.c_int),
                ("uniqueness_ratio", C.c_int), ("mode", C.c_int), ("emulate_quirks", C.c_int)]


def sgm_params(min_disparity=0, num_disparities=128, P1=10, P2=120, uniqueness_ratio=5, mode=3, emulate_quirks=1):
    """createStereoSGM defaults (cudastereo.hpp); mode 1 = MODE_HH (8 paths), 3 = MODE_HH4 (4 paths)."""
    return SGMParams(min_disparity, num_disparities, P1, P2, uniqueness_ratio, mode, emulate_quirks)


def _img816(a):
    a = np.ascontiguousarray(a)
    if a.dtype not in (np.uint8, np.uint16) or a.ndim != 2:
        raise ValueError("left.type() == CV_8UC1 || left.type() == CV_16UC1")
    return a


def sgm_census(img):
    a = _img816(img)
    out = np.empty(a.shape, np.int32)
    lib().orc_sgm_census(C.c_void_p(a.ctypes.data), C.c_int(a.itemsize), C.c_int(a.shape[0]), C.c_int(a.shape[1]), C.c_void_p(out.ctypes.data))
    return out


def sgm_path(left_census, right_census, num_disparities, min_disparity, p1, p2, dx, dy):
    l = np.ascontiguousarray(left_census, dtype=np.int32)
    r = np.ascontiguousarray(right_census, dtype=np.int32)
    h, w = l.shape
    out = np.zeros(h * w * num_disparities, np.uint8)
    lib().orc_sgm_path(C.c_void_p(l.ctypes.data), C.c_void_p(r.ctypes.data), C.c_void_p(out.ctypes.data), C.c_int(w), C.c_int(h),
                       C.c_int(num_disparities), C.c_int(min_disparity), C.c_int(p1), C.c_int(p2), C.c_int(dx), C.c_int(dy))
    return out


def sgm_wta(aggregated, width, height, num_disparities, num_paths, uniqueness, subpixel):
    a = np.ascontiguousarray(aggregated, dtype=np.uint8).reshape(-1)
    assert a.size == width * height * num_disparities * num_paths
    left = np.empty((height, width), np.int16)
    right = np.empty((height, width), np.int16)
    lib().orc_sgm_wta(C.c_void_p(a.ctypes.data), C.c_void_p(left.ctypes.data), C.c_void_p(right.ctypes.data), C.c_int(width), C.c_int(height),
                      C.c_int(num_disparities), C.c_int(num_paths), C.c_float(uniqueness), C.c_int(int(subpixel)))
    return left, right


def sgm_compute(left, right, params: SGMParams | None = None):
    p = params or sgm_params()
    l, r = _img816(left), _img816(right)
    if l.shape != r.shape or l.dtype != r.dtype:
        raise ValueError("size == right.size() && left.type() == right.type()")
    disp = np.empty(l.shape, np.int16)
    L = lib()
    L.orc_sgm_compute.restype = C.c_int
    rc = L.orc_sgm_compute(C.byref(p), C.c_void_p(l.ctypes.data), C.c_void_p(r.ctypes.data), C.c_int(l.itemsize), C.c_int(l.shape[0]),
                           C.c_int(l.shape[1]), C.c_void_p(disp.ctypes.data))
    if rc:
        raise ValueError(f"orc_sgm_compute failed: {rc} (unsupported mode / number of disparities)")
    return disp


# ------------------------------------------------------------------ dense PyrLK
def pyrlk_dense(prev, nxt, win_size=(13, 13), max_level=3, iters=30):
    """cv::cuda::DensePyrLKOpticalFlow::create(winSize = (13, 13), maxLevel = 3, iters = 30)->calc restated on the CPU."""
    a, b = _u8(prev), _u8(nxt)
    if a.shape != b.shape:
        raise ValueError("prevImg.size() == nextImg.size()")
    flow = np.empty(a.shape + (2,), np.float32)
    L = lib()
    L.orc_pyrlk_dense.restype = C.c_int
    rc = L.orc_pyrlk_dense(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_int(a.shape[0]), C.c_int(a.shape[1]),
                           C.c_int(win_size[0]), C.c_int(win_size[1]), C.c_int(max_level), C.c_int(iters), C.c_void_p(flow.ctypes.data))
    if rc:
        raise ValueError("maxLevel >= 0 && winSize > 2")
    return flow
