"""ctypes front-end of oracle/_ref/libref_cu.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference's own CUDA kernel sources -- cudastereo/src/cuda/stereobm.cu, cudaoptflow/src/cuda/farneback.cu,
cudaoptflow/src/cuda/tvl1flow.cu, cudastereo/src/cuda/disparity_bilateral_filter.cu -- compiled for the HOST: oracle/Makefile.ref
rewrites only their launch sites (oracle/refshim/cu2host.py) into oracle/_ref/ and compiles the result against
oracle/refshim/cudashim (thread blocks as cooperatively scheduled contexts, __shared__ as static storage, the few main-repo device
headers stubbed).  Every line of kernel arithmetic that runs is the reference's.  Built where /root/reference exists; the .so
travels with the tree.  tests/test_ref_pin_cuda.py holds the restated oracles to it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import refocl

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_ref", "libref_cu.so")
_lib = None
_f32 = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def available() -> bool:
    return os.path.exists(LIB)


def lib():
    global _lib
    if _lib is None:
        if not available():
            refocl.build()
        L = C.CDLL(LIB)
        i, f, vp = C.c_int, C.c_float, C.c_void_p
        L.ref_cu_sbm_block_match.argtypes = [_u8, _u8, i, i, i, i, i, _u8, vp]
        L.ref_cu_sbm_prefilter_xsobel.argtypes = [_u8, i, i, i, _u8]
        L.ref_cu_sbm_prefilter_norm.argtypes = [_u8, i, i, i, i, _u8]
        L.ref_cu_sbm_textureness.argtypes = [_u8, i, i, i, f, _u8]
        L.ref_cu_fb_poly_exp.argtypes = [_f32, i, i, i, _f32, _f32, _f32, f, f, f, f, _f32]
        L.ref_cu_fb_update_matrices.argtypes = [_f32, _f32, _f32, _f32, i, i, _f32]
        L.ref_cu_fb_update_flow.argtypes = [_f32, i, i, _f32, _f32]
        L.ref_cu_fb_box5.argtypes = [_f32, i, i, i, _f32]
        L.ref_cu_fb_gaussian_blur.argtypes = [_f32, i, i, _f32, i, i, _f32]
        L.ref_cu_fb_gaussian_blur5.argtypes = [_f32, i, i, _f32, i, i, _f32]
        L.ref_cu_tvl1_centered_gradient.argtypes = [_f32, i, i, _f32, _f32]
        L.ref_cu_tvl1_warp.argtypes = [_f32] * 6 + [i, i] + [_f32] * 5
        L.ref_cu_tvl1_estimate_u.argtypes = [_f32] * 4 + [vp] * 10 + [i, i, f, f, f, i]
        L.ref_cu_tvl1_estimate_dual.argtypes = [vp] * 9 + [i, i, f, f]
        L.ref_cu_dbf_apply.restype = i
        L.ref_cu_dbf_apply.argtypes = [vp, i, _u8, i, i, i, i, i, i, f, f, f]
        L.ref_cu_resize_linear_f32.argtypes = [_f32, i, i, _f32, i, i, f, f]
        L.ref_cu_resize_linear_u8.argtypes = [_u8, i, i, _u8, i, i, f, f]
        L.ref_cu_pyr_down_f32.argtypes = [_f32, i, i, _f32, i, i]
        L.ref_cu_pyr_down_u8.argtypes = [_u8, i, i, _u8, i, i]
        d = C.c_double
        L.ref_cuhost_tvl1_calc.restype = i
        L.ref_cuhost_tvl1_calc.argtypes = [d, d, d, i, i, d, i, d, d, i, vp, vp, i, i, i, _f32, C.POINTER(i)]
        _lib = L
    return _lib


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


# ---------------------------------------------------------------------------------------------------------- StereoBM
def sbm_block_match(left, right, ndisp=64, winsz=19, uniqueness_ratio=0):
    left, right = _c(left, np.uint8), _c(right, np.uint8)
    d = np.zeros_like(left)
    ssd = np.zeros(left.shape, np.uint32)
    rc = lib().ref_cu_sbm_block_match(left, right, left.shape[0], left.shape[1], ndisp, winsz, uniqueness_ratio, d, ssd.ctypes.data)
    if rc:
        raise ValueError("stereoBM_CUDA threw")
    return d, ssd


def sbm_prefilter_xsobel(img, cap=31):
    img = _c(img, np.uint8)
    out = np.zeros_like(img)
    lib().ref_cu_sbm_prefilter_xsobel(img, img.shape[0], img.shape[1], cap, out)
    return out


def sbm_prefilter_norm(img, cap=31, winsize=9):
    img = _c(img, np.uint8)
    out = np.zeros_like(img)
    lib().ref_cu_sbm_prefilter_norm(img, img.shape[0], img.shape[1], cap, winsize, out)
    return out


def sbm_textureness(img, disp, winsz=19, avg_threshold=3.0):
    img, d = _c(img, np.uint8), _c(disp, np.uint8).copy()
    lib().ref_cu_sbm_textureness(img, img.shape[0], img.shape[1], winsz, avg_threshold, d)
    return d


# --------------------------------------------------------------------------------------------------------- Farneback
def fb_poly_exp(src, poly_n, g, xg, xxg, ig):
    src = _c(src)
    h, w = src.shape
    pad = lambda a: np.ascontiguousarray(np.concatenate([a, np.zeros(8 - len(a), np.float32)]))
    dst = np.zeros((5 * h, w), np.float32)
    lib().ref_cu_fb_poly_exp(src, h, w, poly_n, pad(g), pad(xg), pad(xxg), ig[0], ig[1], ig[2], ig[3], dst)
    return dst


def fb_update_matrices(flowx, flowy, R0, R1):
    flowx, flowy, R0, R1 = map(_c, (flowx, flowy, R0, R1))
    h, w = flowx.shape
    M = np.zeros((5 * h, w), np.float32)
    lib().ref_cu_fb_update_matrices(flowx, flowy, R0, R1, h, w, M)
    return M


def fb_update_flow(M):
    M = _c(M)
    h, w = M.shape[0] // 5, M.shape[1]
    fx, fy = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    lib().ref_cu_fb_update_flow(M, h, w, fx, fy)
    return fx, fy


def fb_box5(M, ksize):
    M = _c(M)
    out = np.zeros_like(M)
    lib().ref_cu_fb_box5(M, M.shape[0] // 5, M.shape[1], ksize // 2, out)
    return out


def fb_gaussian_blur(src, half_kernel, border):
    src, k = _c(src), _c(half_kernel)
    out = np.zeros_like(src)
    lib().ref_cu_fb_gaussian_blur(src, src.shape[0], src.shape[1], k, len(k) - 1, border, out)
    return out


def fb_gaussian_blur5(M, half_kernel, border=1):
    M, k = _c(M), _c(half_kernel)
    out = np.zeros_like(M)
    lib().ref_cu_fb_gaussian_blur5(M, M.shape[0] // 5, M.shape[1], k, len(k) - 1, border, out)
    return out


# ------------------------------------------------------------------------------------------------ TV-L1 (cv::cuda kernels)
def tvl1_centered_gradient(src):
    src = _c(src)
    dx, dy = np.zeros_like(src), np.zeros_like(src)
    lib().ref_cu_tvl1_centered_gradient(src, src.shape[0], src.shape[1], dx, dy)
    return dx, dy


def tvl1_warp(I0, I1, I1x, I1y, u1, u2):
    a = [_c(x) for x in (I0, I1, I1x, I1y, u1, u2)]
    h, w = a[0].shape
    out = [np.zeros((h, w), np.float32) for _ in range(5)]
    lib().ref_cu_tvl1_warp(*a, h, w, *out)
    return tuple(out)


def tvl1_iteration(I1wx, I1wy, grad, rho_c, u1, u2, p11, p12, p21, p22, l_t, theta, taut, gamma=0.0, u3=None, p31=None, p32=None):
    """estimateU (error plane on) then estimateDualVariables, on copies.  -> (error plane, u1, u2, p11..p22[, u3, p31, p32])."""
    st = [_c(x) for x in (I1wx, I1wy, grad, rho_c)]
    h, w = st[0].shape
    u = [_c(x).copy() for x in (u1, u2)]
    p = [_c(x).copy() for x in (p11, p12, p21, p22)]
    g3 = [_c(x).copy() if x is not None else np.zeros((h, w), np.float32) for x in (u3, p31, p32)]
    err = np.zeros((h, w), np.float32)
    ptr = lambda a: a.ctypes.data
    L = lib()
    L.ref_cu_tvl1_estimate_u(*st, ptr(p[0]), ptr(p[1]), ptr(p[2]), ptr(p[3]), ptr(g3[1]), ptr(g3[2]), ptr(u[0]), ptr(u[1]), ptr(g3[0]), ptr(err),
                             h, w, l_t, theta, gamma, 1)
    L.ref_cu_tvl1_estimate_dual(ptr(u[0]), ptr(u[1]), ptr(g3[0]), ptr(p[0]), ptr(p[1]), ptr(p[2]), ptr(p[3]), ptr(g3[1]), ptr(g3[2]), h, w, taut, gamma)
    out = (err, u[0], u[1], *p)
    return out + tuple(g3) if u3 is not None else out


# ------------------------------------------------------------------------------------------ DisparityBilateralFilter
def dbf_apply(disp, img, ndisp=64, radius=3, iters=1, edge_threshold=0.1, max_disc_threshold=0.2, sigma_range=10.0):
    disp = np.ascontiguousarray(disp).copy()
    img = np.ascontiguousarray(img, np.uint8)
    assert disp.dtype in (np.uint8, np.int16)
    ch = 1 if img.ndim == 2 else img.shape[2]
    rc = lib().ref_cu_dbf_apply(disp.ctypes.data, 0 if disp.dtype == np.uint8 else 3, img.reshape(-1), ch, disp.shape[0], disp.shape[1], ndisp, radius,
                                iters, edge_threshold, max_disc_threshold, sigma_range)
    if rc:
        raise ValueError("unsupported disparity type")
    return disp


def resize_linear(src, dsize):
    """cuda::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_32FC1 / CV_8UC1: the host arithmetic of cudawarping/src/resize.cpp:83-107
    (fx = dsize.width / cols in double, the kernel gets float(1 / fx)) around the reference's resize_linear kernel."""
    src = np.ascontiguousarray(src)
    sh, sw = src.shape
    dw, dh = dsize
    if (dw, dh) == (sw, sh):
        return src.copy()
    fx, fy = dw / sw, dh / sh
    dst = np.empty((dh, dw), src.dtype)
    fn = lib().ref_cu_resize_linear_f32 if src.dtype == np.float32 else lib().ref_cu_resize_linear_u8
    fn(src, sh, sw, dst, dh, dw, np.float32(1.0 / fy), np.float32(1.0 / fx))
    return dst


def pyr_down(src):
    """cuda::pyrDown (dst size (cols + 1) / 2 x (rows + 1) / 2, cudawarping/src/pyramids.cpp:66-94) on the reference's pyrDown kernel."""
    src = np.ascontiguousarray(src)
    sh, sw = src.shape
    dst = np.empty(((sh + 1) // 2, (sw + 1) // 2), src.dtype)
    fn = lib().ref_cu_pyr_down_f32 if src.dtype == np.float32 else lib().ref_cu_pyr_down_u8
    fn(src, sh, sw, dst, dst.shape[0], dst.shape[1])
    return dst


def cuda_class_tvl1_calc(I0, I1, tau=0.25, lambda_=0.15, theta=0.3, nscales=5, warps=5, epsilon=0.01, iterations=300, scale_step=0.8,
                         gamma=0.0, init_flow=None):
    """cv::cuda::OpticalFlowDual_TVL1::create(...)->calc(I0, I1, flow): the reference's HOST class (modules/cudaoptflow/src/tvl1flow.cpp,
    compiled verbatim) over the reference's kernels (tvl1flow.cu, resize.cu).  Returns (flow, nscales after the call)."""
    I0, I1 = np.ascontiguousarray(I0), np.ascontiguousarray(I1)
    assert I0.dtype == I1.dtype and I0.dtype in (np.uint8, np.float32) and I0.shape == I1.shape
    h, w = I0.shape
    flow = np.zeros((h, w, 2), np.float32)
    if init_flow is not None:
        flow[...] = init_flow
    ns = C.c_int(0)
    rc = lib().ref_cuhost_tvl1_calc(tau, lambda_, theta, nscales, warps, epsilon, iterations, scale_step, gamma, int(init_flow is not None),
                                    I0.ctypes.data, I1.ctypes.data, 0 if I0.dtype == np.uint8 else 1, w, h, flow.reshape(-1), C.byref(ns))
    if rc:
        raise ValueError("the reference class threw")
    return flow, ns.value


def cuda_class_farneback_calc(I0, I1, num_levels=5, pyr_scale=0.5, fast_pyramids=False, win_size=13, num_iters=10, poly_n=5, poly_sigma=1.1,
                              flags=0, init_flow=None):
    """cv::cuda::FarnebackOpticalFlow::create(...)->calc(I0, I1, flow): the reference's HOST class (modules/cudaoptflow/src/farneback.cpp,
    compiled verbatim) over the reference's kernels (farneback.cu, resize.cu, pyr_down.cu)."""
    I0, I1 = np.ascontiguousarray(I0), np.ascontiguousarray(I1)
    assert I0.dtype == I1.dtype and I0.dtype in (np.uint8, np.float32) and I0.shape == I1.shape
    h, w = I0.shape
    flow = np.zeros((h, w, 2), np.float32)
    if init_flow is not None:
        flow[...] = init_flow
    L = lib()
    L.ref_cuhost_farneback_calc.restype = C.c_int
    L.ref_cuhost_farneback_calc.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int, C.c_int, _f32]
    rc = L.ref_cuhost_farneback_calc(num_levels, pyr_scale, int(fast_pyramids), win_size, num_iters, poly_n, poly_sigma, flags,
                                     I0.ctypes.data, I1.ctypes.data, 0 if I0.dtype == np.uint8 else 1, w, h, flow.reshape(-1))
    if rc:
        raise ValueError("the reference class threw")
    return flow


def cuda_class_stereobm_compute(left, right, ndisp=64, block=19, prefilter_type=-1, prefilter_size=-1, prefilter_cap=-1, texture_threshold=-1,
                                uniqueness_ratio=-1):
    """cv::cuda::createStereoBM(ndisp, block)->compute(left, right, disp): the reference's HOST class (modules/cudastereo/src/stereobm.cpp,
    compiled verbatim) over the reference's kernels (stereobm.cu).  -1 = leave the constructor's value (no prefilter, cap 31, size 9,
    texture threshold 3, uniqueness 0)."""
    left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
    assert left.dtype == np.uint8 and right.dtype == np.uint8 and left.shape == right.shape
    h, w = left.shape
    disp = np.zeros((h, w), np.uint8)
    L = lib()
    L.ref_cuhost_stereobm_compute.restype = C.c_int
    L.ref_cuhost_stereobm_compute.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rc = L.ref_cuhost_stereobm_compute(ndisp, block, prefilter_type, prefilter_size, prefilter_cap, texture_threshold, uniqueness_ratio,
                                       left.ctypes.data, right.ctypes.data, w, h, disp.ctypes.data)
    if rc:
        raise ValueError("the reference class threw")
    return disp


def cuda_class_surf(img, hessian_threshold=100.0, n_octaves=4, n_octave_layers=2, extended=False, keypoints_ratio=0.01, upright=False, mask=None,
                    want_desc=True, provided=None):
    """cv::cuda::SURF_CUDA::create(...) then operator(): the reference's HOST class (modules/xfeatures2d/src/surf.cuda.cpp, compiled
    verbatim) over the reference's kernels (xfeatures2d/src/cuda/surf.cu on the fiber shim).  -> the dict of oracle.surf_detect_describe,
    sorted by (octave, y, x, size) -- the class appends through atomicInc, its own order is arbitrary.  provided = dict of keypoint
    rows (x, y, octave, size, angle) for useProvidedKeypoints."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    rows, cols = img.shape
    cap = 65536
    kp = np.zeros((7, cap), np.uint32)
    dsz = 128 if extended else 64
    n_in = 0
    if provided is not None:
        n_in = len(provided["x"])
        kf = kp.view(np.float32)
        ki = kp.view(np.int32)
        kf[0, :n_in] = provided["x"]; kf[1, :n_in] = provided["y"]; ki[2, :n_in] = 1; ki[3, :n_in] = provided["octave"]
        kf[4, :n_in] = provided["size"]; kf[5, :n_in] = provided["angle"]; kf[6, :n_in] = provided.get("hessian", 0.0)
    desc = np.zeros((cap if provided is None else max(n_in, 1), dsz), np.float32) if want_desc else None
    m = np.ascontiguousarray(mask) if mask is not None else None
    L = lib()
    L.ref_cuhost_surf.restype = C.c_int
    L.ref_cuhost_surf.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_int, C.c_void_p]
    n = L.ref_cuhost_surf(hessian_threshold, n_octaves, n_octave_layers, int(extended), keypoints_ratio, int(upright), img.ctypes.data,
                          m.ctypes.data if m is not None else None, cols, rows, int(provided is not None), n_in, kp.ctypes.data, cap,
                          desc.ctypes.data if want_desc else None)
    if n < 0:
        raise ValueError("the reference class threw" if n == -1 else "keypoint capacity")
    kf, ki = kp.view(np.float32), kp.view(np.int32)
    out = {"n": n, "x": kf[0, :n].copy(), "y": kf[1, :n].copy(), "laplacian": ki[2, :n].copy(), "octave": ki[3, :n].copy(),
           "size": kf[4, :n].copy(), "angle": kf[5, :n].copy(), "hessian": kf[6, :n].copy(),
           "descriptors": desc[:n].copy() if want_desc else None}
    if provided is None:
        order = np.lexsort((out["size"], out["x"], out["y"], out["octave"]))
        for k in ("x", "y", "laplacian", "octave", "size", "angle", "hessian"):
            out[k] = out[k][order]
        if want_desc:
            out["descriptors"] = out["descriptors"][order]
    return out


def cuda_class_dbf_apply(disp, img, ndisp=64, radius=3, iters=1, edge_threshold=-1.0, max_disc_threshold=-1.0, sigma_range=-1.0):
    """cv::cuda::createDisparityBilateralFilter(ndisp, radius, iters)->apply(disp, img, dst): the reference's HOST class
    (modules/cudastereo/src/disparity_bilateral_filter.cpp, compiled verbatim: weight tables, edge_disc / max_disc, type dispatch) over the
    reference kernel.  Negative thresholds keep the constructor's defaults (0.1, 0.2, 10)."""
    disp = np.ascontiguousarray(disp).copy()
    img = np.ascontiguousarray(img, np.uint8)
    assert disp.dtype in (np.uint8, np.int16) and img.shape[:2] == disp.shape
    ch = 1 if img.ndim == 2 else img.shape[2]
    L = lib()
    L.ref_cuhost_dbf_apply.restype = C.c_int
    L.ref_cuhost_dbf_apply.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    rc = L.ref_cuhost_dbf_apply(ndisp, radius, iters, edge_threshold, max_disc_threshold, sigma_range, disp.ctypes.data,
                                0 if disp.dtype == np.uint8 else 3, img.ctypes.data, ch, disp.shape[1], disp.shape[0])
    if rc:
        raise ValueError("the reference class threw")
    return disp
