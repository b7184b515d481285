"""ctypes front-end of oracle/_ref/libref_ocl*.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

oracle/_ref holds the reference's OWN OpenCL kernel sources (modules/optflow/src/opencl/optical_flow_tvl1.cl,
modules/xfeatures2d/src/opencl/surf.cl), compiled verbatim for x86-64 by `make -f oracle/Makefile.ref` and run on
the CPU through the execution shim in oracle/refshim/.  It is what the restated oracles are pinned against
(tests/test_ref_pin.py).  /root/reference exists only in the build container: the .so files are built there and
travel with the repository snapshot; nothing here reads /root/reference at run time.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "_ref")
REF_ROOT = os.environ.get("MIFLOW_REFERENCE", "/root/reference")
_libs = {}

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def lib_path(fma: bool = False) -> str:
    return os.path.join(_DIR, "libref_ocl_fma.so" if fma else "libref_ocl.so")


def can_build() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "modules", "optflow", "src", "opencl"))


def build(force: bool = False) -> bool:
    """Builds oracle/_ref when the reference tree is present (build container); True if the libraries exist."""
    if can_build():
        cmd = ["make", "-s", "-C", _HERE, "-f", "Makefile.ref", "REF=" + REF_ROOT] + (["-B"] if force else [])
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    return available()


def available() -> bool:
    return all(os.path.exists(p) for p in (lib_path(False), lib_path(True), cpu_lib_path(), os.path.join(_DIR, "libref_cu.so"), surfcpu_lib_path()))


def surfcpu_lib_path() -> str:
    return os.path.join(_DIR, "libref_surfcpu.so")


def cpu_lib_path() -> str:
    return os.path.join(_DIR, "libref_cpu.so")


def lib(fma: bool = False):
    key = bool(fma)
    if key not in _libs:
        if not available():
            build()
        L = C.CDLL(lib_path(fma))
        i, f, d = C.c_int, C.c_float, C.c_double
        L.ref_ocl_tvl1_centered_gradient.argtypes = [_f32p, i, i, _f32p, _f32p]
        L.ref_ocl_tvl1_warp.argtypes = [_f32p] * 6 + [i, i] + [_f32p] * 5
        L.ref_ocl_tvl1_estimate_u.argtypes = [_f32p] * 11 + [i, i, f, f, i]
        L.ref_ocl_tvl1_estimate_dual.argtypes = [_f32p] * 6 + [i, i, f]
        L.ref_ocl_tvl1_proc_one_scale.argtypes = [_f32p] * 4 + [i, i, d, d, d, d, i, i, i, C.c_void_p]
        u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
        i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        L.ref_ocl_surf_det_trace.argtypes = [u32p, i, i, i, i, _f32p, _f32p]
        L.ref_ocl_surf_find_maxima.restype = i
        L.ref_ocl_surf_find_maxima.argtypes = [_f32p, _f32p, i, i, i, i, f, i, i32p]
        L.ref_ocl_surf_interpolate.argtypes = [_f32p, i, i, i, i32p, i, _f32p, i, i, C.POINTER(i)]
        L.ref_ocl_surf_orientation.argtypes = [u32p, i, i, _f32p, i, i]
        L.ref_ocl_surf_descriptors.argtypes = [u8p, i, i, _f32p, i, i, i, _f32p]
        L.ref_ocl_surf_detect.restype = i
        L.ref_ocl_surf_detect.argtypes = [u32p, i, i, i, i, f, f, _f32p, i]
        _libs[key] = L
    return _libs[key]


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def tvl1_centered_gradient(src, fma=False):
    src = _c(src)
    h, w = src.shape
    dx, dy = np.empty_like(src), np.empty_like(src)
    lib(fma).ref_ocl_tvl1_centered_gradient(src, w, h, dx, dy)
    return dx, dy


def tvl1_warp(I0, I1, I1x, I1y, u1, u2, fma=False):
    a = [_c(x) for x in (I0, I1, I1x, I1y, u1, u2)]
    h, w = a[0].shape
    out = [np.empty((h, w), np.float32) for _ in range(5)]
    lib(fma).ref_ocl_tvl1_warp(*a, w, h, *out)
    return tuple(out)   # I1w, I1wx, I1wy, grad, rho_c


def tvl1_iteration(I1wx, I1wy, grad, rho_c, u1, u2, p11, p12, p21, p22, l_t, theta, taut, fma=False):
    """estimateUKernel (with the error plane) then estimateDualVariablesKernel, on copies.
    Returns (error plane, u1, u2, p11, p12, p21, p22)."""
    st = [_c(x) for x in (I1wx, I1wy, grad, rho_c)]
    u = [_c(x).copy() for x in (u1, u2)]
    p = [_c(x).copy() for x in (p11, p12, p21, p22)]
    h, w = st[0].shape
    err = np.zeros((h, w), np.float32)
    L = lib(fma)
    L.ref_ocl_tvl1_estimate_u(*st, *p, u[0], u[1], err, w, h, l_t, theta, 1)
    L.ref_ocl_tvl1_estimate_dual(u[0], u[1], *p, w, h, taut)
    return (err, u[0], u[1], *p)


def tvl1_proc_one_scale(I0, I1, u1, u2, tau=0.25, lambda_=0.15, theta=0.3, epsilon=0.01, warps=5, inner_iterations=1,
                        outer_iterations=300, fma=False):
    """OpticalFlowDual_TVL1::procOneScale_ocl on the reference kernels; returns (u1, u2, iterations per warp)."""
    I0, I1 = _c(I0), _c(I1)
    u1, u2 = _c(u1).copy(), _c(u2).copy()
    h, w = I0.shape
    iters = np.zeros(max(warps, 1), np.int32)
    lib(fma).ref_ocl_tvl1_proc_one_scale(I0, I1, u1, u2, w, h, tau, lambda_, theta, epsilon, warps, inner_iterations,
                                         outer_iterations, iters.ctypes.data)
    return u1, u2, iters[:warps]


# ---------------------------------------------------------------------------------------------------- SURF (surf.cl)
def surf_det_trace(sum_, octave, n_octave_layers=2, fma=False):
    """SURF_calcLayerDetAndTrace; planes of ((layers + 2) * (rows >> octave)) x cols like oracle.surf_det_trace."""
    sum_ = np.ascontiguousarray(sum_, np.uint32)
    rows, cols = sum_.shape[0] - 1, sum_.shape[1] - 1
    lr = rows >> octave
    det = np.zeros(((n_octave_layers + 2) * lr, cols), np.float32)
    tr = np.zeros_like(det)
    lib(fma).ref_ocl_surf_det_trace(sum_, rows, cols, octave, n_octave_layers, det, tr)
    return det, tr


def surf_find_maxima(det, trace, rows, cols, octave, n_octave_layers, hessian_threshold, max_candidates=65535, fma=False):
    """SURF_findMaximaInLayer -> (n, candidates[n, 4] = x, y, layer, laplacian) in launch order."""
    cand = np.zeros((max_candidates + 1, 4), np.int32)
    n = lib(fma).ref_ocl_surf_find_maxima(_c(det), _c(trace), rows, cols, octave, n_octave_layers, hessian_threshold, max_candidates,
                                          cand.reshape(-1))
    return n, cand[:min(n, max_candidates)].copy()


def surf_detect(sum_, n_octaves=4, n_octave_layers=2, hessian_threshold=100.0, keypoints_ratio=0.01, fma=False):
    """SURF_OCL::detectKeypoints without the orientation step -> keypoint matrix (7, n), rows as cuda.hpp:89-99."""
    sum_ = np.ascontiguousarray(sum_, np.uint32)
    rows, cols = sum_.shape[0] - 1, sum_.shape[1] - 1
    pitch = max(1, min(int(np.float32(rows * cols) * np.float32(keypoints_ratio)), 65535))
    kp = np.zeros((7, pitch), np.float32)
    n = lib(fma).ref_ocl_surf_detect(sum_, rows, cols, n_octaves, n_octave_layers, hessian_threshold, keypoints_ratio, kp.reshape(-1), pitch)
    return kp[:, :n].copy()


def surf_orientation(sum_, kp, fma=False):
    """SURF_calcOrientation on a (7, n) keypoint matrix; returns the ANGLE row."""
    sum_ = np.ascontiguousarray(sum_, np.uint32)
    rows, cols = sum_.shape[0] - 1, sum_.shape[1] - 1
    k = np.ascontiguousarray(kp, np.float32).copy()
    lib(fma).ref_ocl_surf_orientation(sum_, rows, cols, k.reshape(-1), k.shape[1], k.shape[1])
    return k[5].copy()


def surf_descriptors(img, kp, extended=False, fma=False):
    """SURF_computeDescriptors64/128 + SURF_normalizeDescriptors64/128 for a (7, n) keypoint matrix."""
    img = np.ascontiguousarray(img, np.uint8)
    k = np.ascontiguousarray(kp, np.float32)
    dsz = 128 if extended else 64
    desc = np.zeros((k.shape[1], dsz), np.float32)
    lib(fma).ref_ocl_surf_descriptors(img, img.shape[0], img.shape[1], k.reshape(-1), k.shape[1], k.shape[1], dsz, desc.reshape(-1))
    return desc


# ------------------------------------------------------------------------- the reference's CPU class (tvl1flow.cpp, verbatim)
_cpu = None


def cpu_lib():
    """oracle/_ref/libref_cpu.so: modules/optflow/src/tvl1flow.cpp compiled verbatim against the stub core headers of
    oracle/refshim/cvstub (cv::resize / remap / medianBlur forwarded to oracle/imgproc_ref.c)."""
    global _cpu
    if _cpu is None:
        if not os.path.exists(cpu_lib_path()):
            build()
        L = C.CDLL(cpu_lib_path())
        i, d = C.c_int, C.c_double
        L.ref_cpu_tvl1_calc.restype = i
        L.ref_cpu_tvl1_calc.argtypes = [d, d, d, i, i, d, i, i, d, d, i, i, C.c_void_p, C.c_void_p, i, i, i, _f32p, C.POINTER(i)]
        _cpu = L
    return _cpu


def cpu_tvl1_calc(I0, I1, tau=0.25, lambda_=0.15, theta=0.3, nscales=5, warps=5, epsilon=0.01, inner_iterations=30,
                  outer_iterations=10, scale_step=0.8, gamma=0.0, median_filtering=5, init_flow=None):
    """cv::optflow::DualTVL1OpticalFlow::create(...)->calc(I0, I1, flow) -- the reference class itself.  Returns (flow, nscales)."""
    I0, I1 = np.ascontiguousarray(I0), np.ascontiguousarray(I1)
    assert I0.dtype == I1.dtype and I0.dtype in (np.uint8, np.float32) and I0.shape == I1.shape
    h, w = I0.shape
    flow = np.zeros((h, w, 2), np.float32)
    if init_flow is not None:
        flow[...] = init_flow
    ns = C.c_int(0)
    rc = cpu_lib().ref_cpu_tvl1_calc(tau, lambda_, theta, nscales, warps, epsilon, inner_iterations, outer_iterations, scale_step, gamma,
                                     median_filtering, int(init_flow is not None), I0.ctypes.data, I1.ctypes.data,
                                     0 if I0.dtype == np.uint8 else 1, w, h, flow.reshape(-1), C.byref(ns))
    if rc:
        raise ValueError(f"the reference class threw (rc {rc})")
    return flow, ns.value


# ------------------------------------------------------------------------- the reference's CPU SURF class (surf.cpp, verbatim)
_surfcpu = None


def surfcpu_lib():
    """oracle/_ref/libref_surfcpu.so: modules/xfeatures2d/src/surf.cpp compiled verbatim (with the reference's own surf.hpp / nonfree.hpp)
    against the stub headers of oracle/refshim/cvsurf; integral / resize(INTER_AREA) / getGaussianKernel / phase / cvRound are the
    restatements of oracle/surfcpu_ref.c (the main repo is not under /root/reference)."""
    global _surfcpu
    if _surfcpu is None:
        if not os.path.exists(surfcpu_lib_path()):
            build()
        L = C.CDLL(surfcpu_lib_path())
        L.ref_surfcpu_detect_and_compute.restype = C.c_int
        L.ref_surfcpu_detect_and_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                                     C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _surfcpu = L
    return _surfcpu


def surfcpu_detect_and_compute(img, hessian_threshold=100.0, n_octaves=4, n_octave_layers=3, extended=False, upright=False, mask=None,
                               want_desc=True, keypoints=None, cap=65536):
    """cv::xfeatures2d::SURF::create(...)->detectAndCompute(img, mask, keypoints, descriptors, useProvidedKeypoints) -- the reference
    class itself.  keypoints (n x 7: x, y, size, angle, response, octave, class_id) given = useProvidedKeypoints.  Returns (keypoints,
    descriptors or None)."""
    img = np.ascontiguousarray(img, np.uint8)
    assert img.ndim == 2
    kp = np.zeros((cap, 7), np.float32)
    n_in = 0
    if keypoints is not None:
        n_in = len(keypoints)
        kp[:n_in] = np.asarray(keypoints, np.float32)
    dc = 128 if extended else 64
    d = np.zeros((cap, dc), np.float32) if want_desc else None
    m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
    n = surfcpu_lib().ref_surfcpu_detect_and_compute(img.ctypes.data, m.ctypes.data if m is not None else None, img.shape[0], img.shape[1],
                                                     float(hessian_threshold), n_octaves, n_octave_layers, int(extended), int(upright),
                                                     int(keypoints is not None), n_in, kp.ctypes.data, cap, d.ctypes.data if d is not None else None)
    if n < 0:
        raise RuntimeError(f"ref_surfcpu_detect_and_compute failed: {n}")
    return kp[:n].copy(), (d[:n].copy() if d is not None else None)
