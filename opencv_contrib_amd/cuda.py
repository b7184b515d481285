"""Python mirror of the reference's `cv2.cuda` classes for the hot path, over the miflow C-ABI.

Names, argument meaning and error behaviour follow the reference headers
(modules/cudaoptflow/include/opencv2/cudaoptflow.hpp:305-386 for OpticalFlowDual_TVL1); device
images are torch CUDA tensors standing in for cv::cuda::GpuMat (pitched row-major; see
capi.mat_from_tensor).  torch is plumbing only: allocation, streams, distributed.
"""
from __future__ import annotations

import ctypes as C

from . import capi
from .capi import MiError



def _destroy(obj, fn_name):
    """Shared __del__ body of the handle wrappers: never raises (interpreter shutdown, failed __init__, library gone)."""
    try:
        h = getattr(obj, "_h", None)
        if h is not None and getattr(h, "value", h):
            getattr(capi.lib(), fn_name)(h)
        obj._h = None
    except Exception:
        pass

def release_cached_memory():
    """miflow extension: hand the scratch blocks that destroyed handles left in libmiflow's process-wide cache (up to
    MIFLOW_CACHE_GB, default 24) back to the driver, e.g. before torch's allocator needs the memory."""
    capi.check(capi.lib().mi_release_cached_memory())


class OpticalFlowDual_TVL1:
    """cv::cuda::OpticalFlowDual_TVL1 (cudaoptflow.hpp:305-386).

    Extra, non-reference knobs (C-ABI extensions): `semantics` (0 = arithmetic of the CPU class
    cv::optflow::DualTVL1OpticalFlow, the acceptance reference and the DEFAULT; 1 = arithmetic of cv::cuda's kernels),
    `exactMath` (default False: fast device math, fused iterations), `innerIterations`/`medianFiltering` (CPU-class
    parameters), `lanes` (internal streams a batch is split over).
    """

    def __init__(self, params: capi.TVL1Params):
        self._h = C.c_void_p()
        capi.check(capi.lib().mi_tvl1_create(C.byref(params), C.byref(self._h)))
        self._p = params

    # -- factory with the reference's signature and defaults (cudaoptflow.hpp:375-385)
    @staticmethod
    def create(tau=0.25, lambda_=0.15, theta=0.3, nscales=5, warps=5, epsilon=0.01, iterations=300,
               scaleStep=0.8, gamma=0.0, useInitialFlow=False, *, semantics=None, exactMath=None, innerIterations=1,
               medianFiltering=1, timeBlock=0, lanes=0, stopSlack=0, hostFeedback=0) -> "OpticalFlowDual_TVL1":
        """The keyword-only arguments are miflow extensions; None = the library default (mi_tvl1_default_params):
        semantics MI_SEM_CPU_REF (the arithmetic of the CPU class, the acceptance reference; MI_SEM_CUDA_COMPAT = cv::cuda's
        own kernels, ~0.1 px mean EPE away, mostly at borders), fast device math (exactMath=True: IEEE operations in the
        reference's order -- fused in blocks of up to 5 iterations when the work is fixed, bit-identical to one launch per iteration); stopSlack > 0 lets the convergence-checked loop run up to that many
        iterations past the reference's stopping point (mi_tvl1_params.stop_slack)."""
        p = capi.TVL1Params()
        capi.lib().mi_tvl1_default_params(C.byref(p))
        p.tau, p.lambda_, p.theta, p.nscales, p.warps = tau, lambda_, theta, nscales, warps
        p.epsilon, p.iterations, p.scale_step, p.gamma = epsilon, iterations, scaleStep, gamma
        p.use_initial_flow = int(bool(useInitialFlow))
        if semantics is not None:
            p.semantics = semantics
        if exactMath is not None:
            p.exact_math = int(bool(exactMath))
        p.inner_iterations, p.median_filtering, p.time_block, p.lanes = innerIterations, medianFiltering, timeBlock, lanes
        p.stop_slack = stopSlack
        p.host_feedback = hostFeedback   # 0 automatic (a call of <= 2 pairs reads the converged flags back between launches), -1 never, 1 every single-lane call
        return OpticalFlowDual_TVL1(p)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                capi.lib().mi_tvl1_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def getDefaultName(self) -> str:  # cudaoptflow/src/tvl1flow.cpp:122
        return "DenseOpticalFlow.OpticalFlowDual_TVL1"

    def _set(self, **kw):
        for k, v in kw.items():
            setattr(self._p, k, v)
        capi.check(capi.lib().mi_tvl1_set_params(self._h, C.byref(self._p)))

    # getters / setters, cudaoptflow/src/tvl1flow.cpp:90-118
    def getTau(self): return self._p.tau
    def setTau(self, v): self._set(tau=v)
    def getLambda(self): return self._p.lambda_
    def setLambda(self, v): self._set(lambda_=v)
    def getGamma(self): return self._p.gamma
    def setGamma(self, v): self._set(gamma=v)
    def getTheta(self): return self._p.theta
    def setTheta(self, v): self._set(theta=v)
    def getNumScales(self): return self._p.nscales
    def setNumScales(self, v): self._set(nscales=v)
    def getNumWarps(self): return self._p.warps
    def setNumWarps(self, v): self._set(warps=v)
    def getEpsilon(self): return self._p.epsilon
    def setEpsilon(self, v): self._set(epsilon=v)
    def getNumIterations(self): return self._p.iterations
    def setNumIterations(self, v): self._set(iterations=v)
    def getScaleStep(self): return self._p.scale_step
    def setScaleStep(self, v): self._set(scale_step=v)
    def getUseInitialFlow(self): return bool(self._p.use_initial_flow)
    def setUseInitialFlow(self, v): self._set(use_initial_flow=int(bool(v)))

    def calc(self, I0, I1, flow=None, stream=None):
        """DenseOpticalFlow::calc(I0, I1, flow, stream) (cudaoptflow.hpp:80).  Returns flow (H,W,2) f32."""
        import torch
        if flow is None:
            if self._p.use_initial_flow:
                raise MiError(-1, "useInitialFlow requires a flow argument")
            flow = torch.empty((I0.shape[0], I0.shape[1], 2), dtype=torch.float32, device=I0.device)
        m0, m1, mf = capi.mat_from_tensor(I0), capi.mat_from_tensor(I1), capi.mat_from_tensor(flow)
        sp = C.c_void_p(stream) if stream is not None else capi.current_stream_ptr()
        capi.check(capi.lib().mi_tvl1_calc(self._h, C.byref(m0), C.byref(m1), C.byref(mf), sp))
        capi.check(capi.lib().mi_tvl1_get_params(self._h, C.byref(self._p)))   # nscales shrinks like the reference's nscales_
        return flow

    def calc_batch(self, I0s, I1s, flows=None, stream=None):
        """n independent pairs in one pass (batched-frames mode).  I0s/I1s: sequences of tensors, or
        one (N,H,W) tensor each; flows: list or (N,H,W,2)."""
        import torch
        n = len(I0s)
        if flows is None:
            flows = torch.empty((n, I0s[0].shape[0], I0s[0].shape[1], 2), dtype=torch.float32, device=I0s[0].device)
        A0 = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in I0s])
        A1 = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in I1s])
        AF = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in flows])
        sp = C.c_void_p(stream) if stream is not None else capi.current_stream_ptr()
        capi.check(capi.lib().mi_tvl1_calc_batch(self._h, n, A0, A1, AF, sp))
        capi.check(capi.lib().mi_tvl1_get_params(self._h, C.byref(self._p)))
        return flows

    def setProfiling(self, on=True):
        capi.check(capi.lib().mi_tvl1_set_profiling(self._h, int(bool(on))))

    def getProfile(self, kind=0, level=-1):
        """(ms inside the iteration-launch regions [kind 0] or the warp launches [kind 1], launches, algorithmic bytes) of the
        last calc; level >= 0 restricts it to one pyramid level (0 = finest)."""
        ms, n, by = C.c_double(), C.c_longlong(), C.c_double()
        capi.check(capi.lib().mi_tvl1_get_profile_level(self._h, kind, level, C.byref(ms), C.byref(n), C.byref(by)))
        return ms.value, n.value, by.value

    def lastIterations(self, pair=0):
        """Executed inner iterations [scale][warp] of the last calc (epsilon > 0: device-decided)."""
        ns = C.c_int()
        cap = 32 * 64
        buf = (C.c_int * cap)()
        capi.check(capi.lib().mi_tvl1_last_iterations(self._h, pair, C.byref(ns), buf, cap, capi.current_stream_ptr()))
        nw = self._p.warps
        return [[buf[s * nw + w] for w in range(nw)] for s in range(ns.value)]


class TVL1MultiDevice:
    """Batched-frames mode over the GPUs of one node through the C-ABI (mi_tvl1_multi_*): contiguous shards of independent pairs,
    one host thread + handle + stream pair per device, staging from / to the ROOT device (devices[0]) over RCCL point-to-point
    messages (where librccl is present and the worker sits on another GPU) or peer-to-peer copies; no reduction.
    `alg` is an OpticalFlowDual_TVL1 whose parameters are used (create it with the reference's factory arguments).
    The same device id may appear several times (several workers on one GPU)."""

    def __init__(self, alg: "OpticalFlowDual_TVL1", devices=None, chunk=16):
        self._h = C.c_void_p()
        ids = None if devices is None else (C.c_int * len(devices))(*devices)
        capi.check(capi.lib().mi_tvl1_multi_create(C.byref(alg._p), 0 if devices is None else len(devices), ids, C.byref(self._h)))
        capi.check(capi.lib().mi_tvl1_multi_set_chunk(self._h, chunk))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                capi.lib().mi_tvl1_multi_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def deviceCount(self):
        return capi.lib().mi_tvl1_multi_device_count(self._h)

    def transport(self):
        """(workers linked to the root over RCCL, workers using peer copies)."""
        a, b = C.c_int(), C.c_int()
        capi.check(capi.lib().mi_tvl1_multi_transport(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    @staticmethod
    def transportWhy():
        """Why the most recent link of this process fell back to peer copies ("" if none did)."""
        return (capi.lib().mi_tvl1_multi_transport_why() or b"").decode()

    def calc_batch(self, I0s, I1s, flows=None):
        """All tensors on the root device.  Synchronous: earlier work on the inputs must be complete (synchronised here)."""
        import torch
        n = len(I0s)
        if flows is None:
            flows = torch.empty((n, I0s[0].shape[0], I0s[0].shape[1], 2), dtype=torch.float32, device=I0s[0].device)
        torch.cuda.synchronize()
        A0 = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in I0s])
        A1 = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in I1s])
        AF = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in flows])
        capi.check(capi.lib().mi_tvl1_multi_calc_batch(self._h, n, A0, A1, AF))
        return flows


# ---------------------------------------------------------------------------------------------
# stage-level functions (the reference's internal device-layer boundary), used by parity tests
def _m(t):
    return capi.mat_from_tensor(t)


def tvl1_centeredGradient(src):
    import torch
    dx, dy = torch.empty_like(src), torch.empty_like(src)
    capi.check(capi.lib().mi_tvl1_centered_gradient(C.byref(_m(src)), C.byref(_m(dx)), C.byref(_m(dy)), capi.current_stream_ptr()))
    return dx, dy


def tvl1_warpBackward(semantics, I0, I1, I1x, I1y, u1, u2):
    """I1x = I1y = None: the derivative planes are formed from I1 inside the kernel (the kernel calc() runs)."""
    import torch
    torch.cuda.synchronize()
    outs = [torch.empty_like(I0) for _ in range(5)]
    args = [C.byref(_m(t)) if t is not None else None for t in (I0, I1, I1x, I1y, u1, u2, *outs)]
    capi.check(capi.lib().mi_tvl1_warp_backward(semantics, *args))
    return tuple(outs)


def tvl1_iterate(I1wx, I1wy, grad, rho_c, u, p, l_t, theta, taut, niter=1, exact=True, time_block=0, want_err=True):
    """u: [u1,u2], p: [p11,p12,p21,p22] -> (u_out, p_out, err[niter])."""
    import torch
    u_out = [torch.empty_like(t) for t in u]
    p_out = [torch.empty_like(t) for t in p]
    U = (capi.Mat * 2)(*[_m(t) for t in u]); P = (capi.Mat * 4)(*[_m(t) for t in p])
    UO = (capi.Mat * 2)(*[_m(t) for t in u_out]); PO = (capi.Mat * 4)(*[_m(t) for t in p_out])
    err = (C.c_double * niter)()
    capi.check(capi.lib().mi_tvl1_iterate(int(exact), time_block, niter, C.byref(_m(I1wx)), C.byref(_m(I1wy)),
                                          C.byref(_m(grad)), C.byref(_m(rho_c)), U, P, UO, PO, l_t, theta, taut,
                                          err if want_err else None, capi.current_stream_ptr()))
    torch.cuda.synchronize()
    return u_out, p_out, list(err)


def resize_linear(src, dsize=None, fx=0.0, fy=0.0, semantics=capi.MI_SEM_CPU_REF, post_scale=1.0):
    """cv::resize / cv::cuda::resize (INTER_LINEAR, CV_32FC1).  dsize = (width, height)."""
    import torch
    h, w = src.shape
    if dsize is None:
        dw, dh = int(round_half_even(w * fx)), int(round_half_even(h * fy))
        explicit = 0
    else:
        dw, dh = dsize
        explicit = 1
    dst = torch.empty((dh, dw), dtype=torch.float32, device=src.device)
    capi.check(capi.lib().mi_resize_linear(semantics, C.byref(_m(src)), C.byref(_m(dst)), fx, fy, explicit,
                                           post_scale, capi.current_stream_ptr()))
    return dst


def round_half_even(v: float) -> int:
    import numpy as np
    return int(np.rint(v))


# =============================================================================================
class StereoBM:
    """cv::cuda::StereoBM (cudastereo.hpp:72-90; StereoBMImpl cudastereo/src/stereobm.cpp:67-132).

    Setters that are no-ops in the reference (minDisparity, speckle*, disp12MaxDiff, smallerBlockSize,
    ROI1/2: stereobm.cpp:78-115) are no-ops here and their getters return the same constants."""

    PREFILTER_NORMALIZED_RESPONSE, PREFILTER_XSOBEL = 0, 1  # cv::StereoBM (main repo calib3d.hpp)

    def __init__(self, numDisparities=64, blockSize=19, *, emulateCudaEdge=True):
        p = capi.StereoBMParams()
        capi.lib().mi_stereobm_default_params(C.byref(p))
        p.num_disparities, p.block_size, p.emulate_cuda_edge = numDisparities, blockSize, int(bool(emulateCudaEdge))
        self._h = C.c_void_p()
        capi.check(capi.lib().mi_stereobm_create(C.byref(p), C.byref(self._h)))
        self._p = p

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                capi.lib().mi_stereobm_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _set(self, **kw):
        for k, v in kw.items():
            setattr(self._p, k, v)
        capi.check(capi.lib().mi_stereobm_set_params(self._h, C.byref(self._p)))

    def getMinDisparity(self): return 0
    def setMinDisparity(self, v): pass
    def getNumDisparities(self): return self._p.num_disparities
    def setNumDisparities(self, v): self._set(num_disparities=v)
    def getBlockSize(self): return self._p.block_size
    def setBlockSize(self, v): self._set(block_size=v)
    def getSpeckleWindowSize(self): return 0
    def setSpeckleWindowSize(self, v): pass
    def getSpeckleRange(self): return 0
    def setSpeckleRange(self, v): pass
    def getDisp12MaxDiff(self): return 0
    def setDisp12MaxDiff(self, v): pass
    def getPreFilterType(self): return self._p.prefilter_type
    def setPreFilterType(self, v): self._set(prefilter_type=v)
    def getPreFilterSize(self): return self._p.prefilter_size
    def setPreFilterSize(self, v): self._set(prefilter_size=v)
    def getPreFilterCap(self): return self._p.prefilter_cap
    def setPreFilterCap(self, v): self._set(prefilter_cap=v)
    def getTextureThreshold(self): return int(self._p.texture_threshold)
    def setTextureThreshold(self, v): self._set(texture_threshold=float(v))
    def getUniquenessRatio(self): return self._p.uniqueness_ratio
    def setUniquenessRatio(self, v): self._set(uniqueness_ratio=v)
    def getSmallerBlockSize(self): return 0
    def setSmallerBlockSize(self, v): pass

    def compute(self, left, right, disparity=None, stream=None):
        """compute(left, right, disparity[, stream]) (cudastereo.hpp:80-82).  Returns CV_8UC1 disparity."""
        import torch
        if disparity is None:  # _disparity.create(left.size(), CV_8UC1)  stereobm.cpp:154
            disparity = torch.empty((left.shape[0], left.shape[1]), dtype=torch.uint8, device=left.device)
        ml, mr, md = capi.mat_from_tensor(left), capi.mat_from_tensor(right), capi.mat_from_tensor(disparity)
        sp = C.c_void_p(stream) if stream is not None else capi.current_stream_ptr()
        capi.check(capi.lib().mi_stereobm_compute(self._h, C.byref(ml), C.byref(mr), C.byref(md), sp))
        return disparity

    def compute_batch(self, lefts, rights, disparities=None, stream=None):
        """n stereo pairs through one handle (miflow extension): sequences of tensors or (N,H,W) tensors."""
        import torch
        n = len(lefts)
        if disparities is None:
            disparities = torch.empty((n, lefts[0].shape[0], lefts[0].shape[1]), dtype=torch.uint8, device=lefts[0].device)
        AL = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in lefts])
        AR = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in rights])
        AD = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in disparities])
        sp = C.c_void_p(stream) if stream is not None else capi.current_stream_ptr()
        capi.check(capi.lib().mi_stereobm_compute_batch(self._h, n, AL, AR, AD, sp))
        return disparities


def createStereoBM(numDisparities=64, blockSize=19, **kw) -> StereoBM:
    """cv::cuda::createStereoBM (cudastereo.hpp:90)."""
    return StereoBM(numDisparities, blockSize, **kw)


class DensePyrLKOpticalFlow:
    """cv::cuda::DensePyrLKOpticalFlow (cudaoptflow.hpp; cudaoptflow/src/pyrlk.cpp:238-299,354-406)."""

    def __init__(self, winSize=(13, 13), maxLevel=3, iters=30, useInitialFlow=False):
        self._p = capi.DensePyrLKParams()
        capi.lib().mi_densepyrlk_default_params(C.byref(self._p))
        self._p.win_width, self._p.win_height, self._p.max_level, self._p.iters = winSize[0], winSize[1], maxLevel, iters
        self._p.use_initial_flow = int(bool(useInitialFlow))
        self._h = C.c_void_p()
        capi.check(capi.lib().mi_densepyrlk_create(C.byref(self._p), C.byref(self._h)))

    @classmethod
    def create(cls, winSize=(13, 13), maxLevel=3, iters=30, useInitialFlow=False):
        return cls(winSize, maxLevel, iters, useInitialFlow)

    def __del__(self):
        _destroy(self, "mi_densepyrlk_destroy")

    def _set(self, **kw):
        for k, v in kw.items():
            setattr(self._p, k, v)
        capi.check(capi.lib().mi_densepyrlk_set_params(self._h, C.byref(self._p)))

    def getDefaultName(self): return "DenseOpticalFlow.DensePyrLKOpticalFlow"   # pyrlk.cpp:394
    def getWinSize(self): return (self._p.win_width, self._p.win_height)
    def setWinSize(self, v): self._set(win_width=v[0], win_height=v[1])
    def getMaxLevel(self): return self._p.max_level
    def setMaxLevel(self, v): self._set(max_level=v)
    def getNumIters(self): return self._p.iters
    def setNumIters(self, v): self._set(iters=v)
    def getUseInitialFlow(self): return bool(self._p.use_initial_flow)
    def setUseInitialFlow(self, v): self._set(use_initial_flow=int(bool(v)))

    def calc(self, prevImg, nextImg, flow=None):
        import torch
        if flow is None:
            flow = torch.empty((prevImg.shape[0], prevImg.shape[1], 2), dtype=torch.float32, device=prevImg.device)
        capi.check(capi.lib().mi_densepyrlk_calc(self._h, C.byref(_m(prevImg)), C.byref(_m(nextImg)), C.byref(_m(flow)),
                                                 capi.current_stream_ptr()))
        return flow


class SparsePyrLKOpticalFlow:
    """cv::cuda::SparsePyrLKOpticalFlow for CV_8UC1 frames (cudaoptflow.hpp:85-104,203-223; cudaoptflow/src/pyrlk.cpp:149-231,308-352).
    calc(prevImg, nextImg, prevPts (1, N, 2) or (N, 2) float32[, nextPts]) -> (nextPts (1, N, 2), status (1, N) uint8, err (1, N))."""

    def __init__(self, winSize=(21, 21), maxLevel=3, iters=30, useInitialFlow=False):
        self._p = capi.SparsePyrLKParams()
        capi.lib().mi_sparsepyrlk_default_params(C.byref(self._p))
        self._p.win_width, self._p.win_height, self._p.max_level, self._p.iters = winSize[0], winSize[1], maxLevel, iters
        self._p.use_initial_flow = int(bool(useInitialFlow))
        self._h = C.c_void_p()
        capi.check(capi.lib().mi_sparsepyrlk_create(C.byref(self._p), C.byref(self._h)))

    @classmethod
    def create(cls, winSize=(21, 21), maxLevel=3, iters=30, useInitialFlow=False):
        return cls(winSize, maxLevel, iters, useInitialFlow)

    def __del__(self):
        _destroy(self, "mi_sparsepyrlk_destroy")

    def _set(self, **kw):
        for k, v in kw.items():
            setattr(self._p, k, v)
        capi.check(capi.lib().mi_sparsepyrlk_set_params(self._h, C.byref(self._p)))

    def getDefaultName(self): return "SparseOpticalFlow.SparsePyrLKOpticalFlow"   # pyrlk.cpp:350
    def getWinSize(self): return (self._p.win_width, self._p.win_height)
    def setWinSize(self, v): self._set(win_width=v[0], win_height=v[1])
    def getMaxLevel(self): return self._p.max_level
    def setMaxLevel(self, v): self._set(max_level=v)
    def getNumIters(self): return self._p.iters
    def setNumIters(self, v): self._set(iters=v)
    def getUseInitialFlow(self): return bool(self._p.use_initial_flow)
    def setUseInitialFlow(self, v): self._set(use_initial_flow=int(bool(v)))

    def calc(self, prevImg, nextImg, prevPts, nextPts=None, wantErr=True):
        import torch
        pp = prevPts.reshape(1, -1, 2).contiguous()
        n = pp.shape[1]
        if n == 0:      # pyrlk.cpp:209-215: empty input releases the outputs
            e = torch.empty((1, 0), device=prevImg.device)
            return torch.empty((1, 0, 2), device=prevImg.device), e.to(torch.uint8), (e if wantErr else None)
        if self._p.use_initial_flow:
            if nextPts is None or nextPts.numel() != pp.numel():
                raise capi.MiError(-1, "nextPts.size() == prevPts.size() (useInitialFlow)")   # :160
            npts = nextPts.reshape(1, -1, 2).contiguous().clone()
        else:
            npts = torch.empty_like(pp)
        status = torch.empty((1, n), dtype=torch.uint8, device=prevImg.device)
        err = torch.empty((1, n), dtype=torch.float32, device=prevImg.device) if wantErr else None
        capi.check(capi.lib().mi_sparsepyrlk_calc(self._h, C.byref(_m(prevImg)), C.byref(_m(nextImg)), C.byref(_m(pp)), C.byref(_m(npts)),
                                                  C.byref(_m(status)), C.byref(_m(err)) if wantErr else None, capi.current_stream_ptr()))
        return npts, status, err


class StereoSGM:
    """cv::cuda::StereoSGM (cudastereo.hpp; cudastereo/src/stereosgm.cpp:20-153): semi-global matching on census costs,
    4 (MODE_HH4) or 8 (MODE_HH) paths, CV_16SC1 output with 4 fractional bits."""

    MODE_HH, MODE_HH4 = 1, 3

    def __init__(self, minDisparity=0, numDisparities=128, P1=10, P2=120, uniquenessRatio=5, mode=3, emulateCudaQuirks=True):
        self._p = capi.StereoSGMParams()
        capi.lib().mi_stereosgm_default_params(C.byref(self._p))
        self._p.min_disparity, self._p.num_disparities, self._p.P1, self._p.P2 = minDisparity, numDisparities, P1, P2
        self._p.uniqueness_ratio, self._p.mode, self._p.emulate_cuda_quirks = uniquenessRatio, mode, int(bool(emulateCudaQuirks))
        self._h = C.c_void_p()
        capi.check(capi.lib().mi_stereosgm_create(C.byref(self._p), C.byref(self._h)))

    def __del__(self):
        _destroy(self, "mi_stereosgm_destroy")

    def _set(self, **kw):
        for k, v in kw.items():
            setattr(self._p, k, v)
        capi.check(capi.lib().mi_stereosgm_set_params(self._h, C.byref(self._p)))

    # stereosgm.cpp:38-72 (the fixed getters of the reference included)
    def getBlockSize(self): return -1
    def setBlockSize(self, v): pass
    def getDisp12MaxDiff(self): return 1
    def setDisp12MaxDiff(self, v): pass
    def getMinDisparity(self): return self._p.min_disparity
    def setMinDisparity(self, v): self._set(min_disparity=v)
    def getNumDisparities(self): return self._p.num_disparities
    def setNumDisparities(self, v): self._set(num_disparities=v)
    def getSpeckleWindowSize(self): return 0
    def setSpeckleWindowSize(self, v): pass
    def getSpeckleRange(self): return 0
    def setSpeckleRange(self, v): pass
    def getP1(self): return self._p.P1
    def setP1(self, v): self._set(P1=v)
    def getP2(self): return self._p.P2
    def setP2(self, v): self._set(P2=v)
    def getUniquenessRatio(self): return self._p.uniqueness_ratio
    def setUniquenessRatio(self, v): self._set(uniqueness_ratio=v)
    def getMode(self): return self._p.mode
    def setMode(self, v): self._set(mode=v)
    def getPreFilterCap(self): return -1
    def setPreFilterCap(self, v): pass

    def compute(self, left, right, disparity=None):
        import torch
        if disparity is None:
            disparity = torch.empty(left.shape[:2], dtype=torch.int16, device=left.device)
        capi.check(capi.lib().mi_stereosgm_compute(self._h, C.byref(_m(left)), C.byref(_m(right)), C.byref(_m(disparity)),
                                                   capi.current_stream_ptr()))
        return disparity


def createStereoSGM(minDisparity=0, numDisparities=128, P1=10, P2=120, uniquenessRatio=5, mode=3, **kw) -> StereoSGM:
    """cv::cuda::createStereoSGM (cudastereo.hpp)."""
    return StereoSGM(minDisparity, numDisparities, P1, P2, uniquenessRatio, mode, **kw)


def sgm_census(img):
    import torch
    out = torch.empty(img.shape, dtype=torch.int32, device=img.device)
    capi.check(capi.lib().mi_sgm_census(C.byref(_m(img)), C.byref(_m(out)), capi.current_stream_ptr()))
    return out


def sgm_aggregate_path(left_census, right_census, num_disparities, min_disparity, p1, p2, dx, dy):
    import torch
    h, w = left_census.shape
    out = torch.empty((1, h * w * num_disparities), dtype=torch.uint8, device=left_census.device)
    capi.check(capi.lib().mi_sgm_aggregate_path(C.byref(_m(left_census)), C.byref(_m(right_census)), C.byref(_m(out)), num_disparities,
                                                min_disparity, p1, p2, dx, dy, capi.current_stream_ptr()))
    return out


def sgm_winner_takes_all(aggregated, width, height, num_disparities, num_paths, uniqueness, subpixel):
    import torch
    left = torch.empty((height, width), dtype=torch.int16, device=aggregated.device)
    right = torch.empty_like(left)
    capi.check(capi.lib().mi_sgm_winner_takes_all(C.byref(_m(aggregated)), C.byref(_m(left)), C.byref(_m(right)), num_disparities, num_paths,
                                                  C.c_float(uniqueness), int(subpixel), capi.current_stream_ptr()))
    return left, right


class DMatch:
    """cv::DMatch (queryIdx, trainIdx, imgIdx, distance); ordered by distance like the reference's operator<."""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx=-1, trainIdx=-1, imgIdx=-1, distance=float("inf")):
        self.queryIdx, self.trainIdx, self.imgIdx, self.distance = int(queryIdx), int(trainIdx), int(imgIdx), float(distance)

    def __lt__(self, other):
        return self.distance < other.distance

    def __repr__(self):
        return f"DMatch(queryIdx={self.queryIdx}, trainIdx={self.trainIdx}, imgIdx={self.imgIdx}, distance={self.distance:.6g})"


class BFMatcher:
    """cv::cuda::DescriptorMatcher::createBFMatcher(NORM_L1 | NORM_L2 | NORM_HAMMING): float descriptors (L1 / L2) and integer ones
    (L1 / Hamming: the reference's (depth, norm) table) (cudafeatures2d.hpp;
    cudafeatures2d/src/brute_force_matcher.cpp:143-1070; kernels cuda/bf_match.cu, bf_knnmatch.cu, bf_radius_match.cu).

    Three forms of every query, like the reference: `match / knnMatch / radiusMatch` return DMatch lists (host),
    `*Async` return the reference's packed device matrix and `*Convert` unpack it; `*Device` return the separate device tensors
    the C-ABI fills (no host transfer).  With `trainDescriptors=None` the collection added with add() is searched."""

    NORM_L1 = 2
    NORM_L2 = 4
    NORM_HAMMING = 6      # integer descriptors (uint8 / uint16 / int32); NORM_L1 also takes uint8 / uint16 / int16 / int32

    def __init__(self, normType=4):
        self._h = C.c_void_p()
        capi.check(capi.lib().mi_bf_create(int(normType), C.byref(self._h)))
        self._train = []

    def __del__(self):
        _destroy(self, "mi_bf_destroy")

    # ---- train collection (brute_force_matcher.cpp:187-211)
    def isMaskSupported(self): return True
    def add(self, descriptors): self._train.extend(descriptors)
    def getTrainDescriptors(self): return self._train
    def clear(self): self._train = []
    def empty(self): return not self._train
    def train(self): pass

    def _collection(self, train, mask, masks):
        """-> (trains list, masks list or None, is_collection)"""
        if train is not None:
            return [train], ([mask] if mask is not None else None), False
        if masks is not None and len(masks) and len(masks) != len(self._train):
            raise capi.MiError(-1, "masks.size() == trainDescCollection.size()")          # makeGpuCollection, :156
        return list(self._train), (list(masks) if masks else None), True

    @staticmethod
    def _mats(tensors):
        arr = (capi.Mat * len(tensors))()
        for j, t in enumerate(tensors):
            arr[j] = _m(t) if t is not None and t.numel() else capi.Mat(None, 0, 0, 0, 0)
        return arr

    # ---- device forms
    def knnMatchDevice(self, queryDescriptors, trainDescriptors=None, k=2, mask=None, masks=None):
        """-> (trainIdx (nq, k) int32, imgIdx (nq, k) int32, distance (nq, k) float32); entries past the candidates -1 / -1 / FLT_MAX."""
        import torch
        trains, ms, _ = self._collection(trainDescriptors, mask, masks)
        q = queryDescriptors
        if q.numel() == 0 or not trains:
            e = torch.empty((0, k), dtype=torch.int32, device=q.device)
            return e, e.clone(), torch.empty((0, k), dtype=torch.float32, device=q.device)
        nq = q.shape[0]
        idx = torch.empty((nq, k), dtype=torch.int32, device=q.device)
        img = torch.empty((nq, k), dtype=torch.int32, device=q.device)
        dist = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        capi.check(capi.lib().mi_bf_knn_match(self._h, C.byref(_m(q)), self._mats(trains), self._mats(ms) if ms else None, len(trains), int(k),
                                              C.byref(_m(idx)), C.byref(_m(img)), C.byref(_m(dist)), capi.current_stream_ptr()))
        return idx, img, dist

    def matchDevice(self, queryDescriptors, trainDescriptors=None, mask=None, masks=None):
        """-> (trainIdx (nq,), imgIdx (nq,), distance (nq,))"""
        idx, img, dist = self.knnMatchDevice(queryDescriptors, trainDescriptors, 1, mask, masks)
        return idx[:, 0], img[:, 0], dist[:, 0]

    def radiusMatchDevice(self, queryDescriptors, trainDescriptors=None, maxDistance=0.0, mask=None, masks=None):
        """-> (trainIdx, imgIdx, distance) (nq, cols) and nMatches (nq,); cols as the reference sizes it (max(nTrain / 100, nQuery),
        nQuery for a collection: brute_force_matcher.cpp:897,969).  Row q holds its first min(nMatches[q], cols) hits in ascending
        (image, train) order."""
        import torch
        trains, ms, coll = self._collection(trainDescriptors, mask, masks)
        q = queryDescriptors
        if q.numel() == 0 or not trains:
            e = torch.empty((0, 0), dtype=torch.int32, device=q.device)
            return e, e.clone(), torch.empty((0, 0), dtype=torch.float32, device=q.device), torch.empty((0,), dtype=torch.int32, device=q.device)
        nq = q.shape[0]
        cols = nq if coll else max(trains[0].shape[0] // 100, nq)
        idx = torch.full((nq, cols), -1, dtype=torch.int32, device=q.device)
        img = torch.full((nq, cols), -1, dtype=torch.int32, device=q.device)
        dist = torch.zeros((nq, cols), dtype=torch.float32, device=q.device)
        n = torch.empty((1, nq), dtype=torch.int32, device=q.device)
        capi.check(capi.lib().mi_bf_radius_match(self._h, C.byref(_m(q)), self._mats(trains), self._mats(ms) if ms else None, len(trains),
                                                 float(maxDistance), C.byref(_m(idx)), C.byref(_m(img)), C.byref(_m(dist)), C.byref(_m(n)),
                                                 capi.current_stream_ptr()))
        return idx, img, dist, n[0]

    # ---- the reference's packed device matrices (matchAsync / knnMatchAsync / radiusMatchAsync)
    def matchAsync(self, queryDescriptors, trainDescriptors=None, mask=None, masks=None):
        """CV_32SC1 2 x nq {trainIdx; distance bits} (:367-374), 3 x nq {trainIdx; imgIdx; distance bits} for the collection (:429-437)."""
        import torch
        idx, img, dist = self.matchDevice(queryDescriptors, trainDescriptors, mask, masks)
        rows = [idx, dist.view(torch.int32)] if trainDescriptors is not None else [idx, img, dist.view(torch.int32)]
        return torch.stack(rows, 0)

    @staticmethod
    def matchConvert(gpu_matches):
        """brute_force_matcher.cpp:440-495: unmatched queries (trainIdx -1) are dropped."""
        import numpy as np
        g = gpu_matches.cpu().numpy()
        if g.size == 0:
            return []
        assert g.dtype == np.int32 and g.shape[0] in (2, 3)
        idx, dist = g[0], g[-1].view(np.float32)
        img = g[1] if g.shape[0] == 3 else np.zeros_like(idx)
        return [DMatch(q, idx[q], img[q], dist[q]) for q in range(g.shape[1]) if idx[q] != -1]

    def match(self, queryDescriptors, trainDescriptors=None, mask=None, masks=None):
        return self.matchConvert(self.matchAsync(queryDescriptors, trainDescriptors, mask, masks))

    def knnMatchAsync(self, queryDescriptors, trainDescriptors=None, k=2, mask=None, masks=None):
        """k = 2: CV_32SC2 2 x nq (3 x nq with imgIdx for the collection); other k (single train set only, like the reference:
        :662-665): CV_32SC1 2 nq x k, trainIdx rows then distance rows (:634-642)."""
        import torch
        if trainDescriptors is None and k != 2:
            raise capi.MiError(-1, "only k=2 mode is supported for now (knnMatchAsync over the collection); use knnMatch")
        idx, img, dist = self.knnMatchDevice(queryDescriptors, trainDescriptors, k, mask, masks)
        bits = dist.view(torch.int32)
        if k == 2:
            return torch.stack([idx, bits] if trainDescriptors is not None else [idx, img, bits], 0)
        return torch.cat([idx, bits], 0)

    @staticmethod
    def knnMatchConvert(gpu_matches, compactResult=False):
        """brute_force_matcher.cpp:727-812"""
        import numpy as np
        g = gpu_matches.cpu().numpy()
        if g.size == 0:
            return []
        if g.ndim == 3:
            idx, dist = g[0], g[-1].view(np.float32)
            img = g[1] if g.shape[0] == 3 else np.zeros_like(idx)
        else:
            nq = g.shape[0] // 2
            idx, dist = g[:nq], g[nq:].view(np.float32)
            img = np.zeros_like(idx)
        out = []
        for q in range(idx.shape[0]):
            cur = [DMatch(q, idx[q, j], img[q, j], dist[q, j]) for j in range(idx.shape[1]) if idx[q, j] != -1]
            if cur or not compactResult:
                out.append(cur)
        return out

    def knnMatch(self, queryDescriptors, trainDescriptors=None, k=2, mask=None, masks=None, compactResult=False):
        """The reference merges per-image k-lists on the host for k != 2 over a collection (:512-571); here one device pass."""
        import numpy as np
        if trainDescriptors is not None or k == 2:
            return self.knnMatchConvert(self.knnMatchAsync(queryDescriptors, trainDescriptors, k, mask, masks), compactResult)
        idx, img, dist = (t.cpu().numpy() for t in self.knnMatchDevice(queryDescriptors, None, k, None, masks))
        out = []
        for q in range(idx.shape[0]):
            cur = [DMatch(q, idx[q, j], img[q, j], dist[q, j]) for j in range(k) if idx[q, j] != -1]
            if cur or not compactResult:
                out.append(cur)
        return out

    def radiusMatchAsync(self, queryDescriptors, trainDescriptors=None, maxDistance=0.0, mask=None, masks=None):
        """CV_32SC1 (2 nq + 1) x cols {trainIdx rows; distance rows; nMatches} (:897-905); collection: (3 nq + 1) x nq with imgIdx rows
        (the reference types that one CV_32FC1, :969-975; the bits are the same)."""
        import torch
        idx, img, dist, n = self.radiusMatchDevice(queryDescriptors, trainDescriptors, maxDistance, mask, masks)
        if idx.numel() == 0:
            return idx
        last = torch.zeros((1, idx.shape[1]), dtype=torch.int32, device=idx.device)
        last[0, :n.shape[0]] = n
        parts = [idx, dist.view(torch.int32), last] if trainDescriptors is not None else [idx, img, dist.view(torch.int32), last]
        return torch.cat(parts, 0)

    @staticmethod
    def radiusMatchConvert(gpu_matches, compactResult=False, collection=False):
        """brute_force_matcher.cpp:982-1063: min(nMatches, cols) entries per query, sorted by distance (stable here: ties keep the
        ascending (image, train) order)."""
        import numpy as np
        g = gpu_matches.cpu().numpy()
        if g.size == 0:
            return []
        per = 3 if collection else 2
        nq = (g.shape[0] - 1) // per
        idx, dist, n = g[:nq], g[(per - 1) * nq:per * nq].view(np.float32), g[per * nq]
        img = g[nq:2 * nq] if collection else np.zeros_like(idx)
        out = []
        for q in range(nq):
            m = min(int(n[q]), g.shape[1])
            if m == 0:
                if not compactResult:
                    out.append([])
                continue
            cur = [DMatch(q, idx[q, j], img[q, j], dist[q, j]) for j in range(m)]
            cur.sort(key=lambda d: d.distance)
            out.append(cur)
        return out

    def radiusMatch(self, queryDescriptors, trainDescriptors=None, maxDistance=0.0, mask=None, masks=None, compactResult=False):
        g = self.radiusMatchAsync(queryDescriptors, trainDescriptors, maxDistance, mask, masks)
        return self.radiusMatchConvert(g, compactResult, collection=trainDescriptors is None)


def createBFMatcher(normType=4) -> BFMatcher:
    """cv::cuda::DescriptorMatcher::createBFMatcher (cudafeatures2d.hpp)."""
    return BFMatcher(normType)


class DescriptorMatcher:
    """Name parity with cv2.cuda_DescriptorMatcher: `cuda.DescriptorMatcher.createBFMatcher(cuda.NORM_L2)`."""
    createBFMatcher = staticmethod(createBFMatcher)


NORM_L1, NORM_L2, NORM_HAMMING = 2, 4, 6


class DisparityBilateralFilter:
    """cv::cuda::DisparityBilateralFilter (cudastereo.hpp:298-330; cudastereo/src/disparity_bilateral_filter.cpp:58-190):
    joint bilateral refinement of a disparity map at its discontinuities, guided by the image."""

    def __init__(self, ndisp=64, radius=3, iters=1):
        self._p = capi.DispBilateralParams()
        capi.lib().mi_disp_bilateral_default_params(C.byref(self._p))
        self._p.ndisp, self._p.radius, self._p.iters = ndisp, radius, iters
        self._h = C.c_void_p()
        capi.check(capi.lib().mi_disp_bilateral_create(C.byref(self._p), C.byref(self._h)))

    def __del__(self):
        _destroy(self, "mi_disp_bilateral_destroy")

    def _set(self, **kw):
        for k, v in kw.items():
            setattr(self._p, k, v)
        capi.check(capi.lib().mi_disp_bilateral_set_params(self._h, C.byref(self._p)))

    def getNumDisparities(self): return self._p.ndisp
    def setNumDisparities(self, v): self._set(ndisp=v)
    def getRadius(self): return self._p.radius
    def setRadius(self, v): self._set(radius=v)
    def getNumIters(self): return self._p.iters
    def setNumIters(self, v): self._set(iters=v)
    def getEdgeThreshold(self): return float(self._p.edge_threshold)
    def setEdgeThreshold(self, v): self._set(edge_threshold=float(v))
    def getMaxDiscThreshold(self): return float(self._p.max_disc_threshold)
    def setMaxDiscThreshold(self, v): self._set(max_disc_threshold=float(v))
    def getSigmaRange(self): return float(self._p.sigma_range)
    def setSigmaRange(self, v): self._set(sigma_range=float(v))

    def apply(self, disparity, image, dst=None):
        """disparity: uint8 or int16 (H, W); image: uint8 (H, W) or (H, W, 3); returns dst (same type as disparity)."""
        import torch
        if dst is None:
            dst = torch.empty_like(disparity)
        capi.check(capi.lib().mi_disp_bilateral_apply(self._h, C.byref(_m(disparity)), C.byref(_m(image)), C.byref(_m(dst)),
                                                      capi.current_stream_ptr()))
        return dst


def createDisparityBilateralFilter(ndisp=64, radius=3, iters=1) -> DisparityBilateralFilter:
    """cv::cuda::createDisparityBilateralFilter (cudastereo.hpp:338-339)."""
    return DisparityBilateralFilter(ndisp, radius, iters)


def stereobm_prefilter_xsobel(img, cap=31):
    import torch
    out = torch.empty_like(img)
    capi.check(capi.lib().mi_stereobm_prefilter_xsobel(C.byref(_m(img)), C.byref(_m(out)), cap, capi.current_stream_ptr()))
    return out


def stereobm_prefilter_norm(img, cap=31, winsize=9):
    import torch
    out = torch.empty_like(img)
    capi.check(capi.lib().mi_stereobm_prefilter_norm(C.byref(_m(img)), C.byref(_m(out)), cap, winsize, capi.current_stream_ptr()))
    return out


def stereobm_block_match(left, right, ndisp=64, winsz=19, uniqueness_ratio=0, emulate_cuda_edge=True):
    """-> (disp CV_8UC1, minSSD CV_32SC1 holding uint32 bit patterns)."""
    import torch
    disp = torch.empty_like(left)
    ssd = torch.empty(left.shape, dtype=torch.int32, device=left.device)
    capi.check(capi.lib().mi_stereobm_block_match(C.byref(_m(left)), C.byref(_m(right)), C.byref(_m(disp)), C.byref(_m(ssd)),
                                                  ndisp, winsz, uniqueness_ratio, int(bool(emulate_cuda_edge)),
                                                  capi.current_stream_ptr()))
    return disp, ssd


def stereobm_textureness(img, disp, winsz=19, avg_threshold=3.0):
    out = disp.clone()
    capi.check(capi.lib().mi_stereobm_textureness(C.byref(_m(img)), C.byref(_m(out)), winsz, avg_threshold,
                                                  capi.current_stream_ptr()))
    return out


# =============================================================================================
OPTFLOW_USE_INITIAL_FLOW, OPTFLOW_FARNEBACK_GAUSSIAN = 4, 256   # cv::OPTFLOW_* (main repo video/tracking.hpp)


class FarnebackOpticalFlow:
    """cv::cuda::FarnebackOpticalFlow (cudaoptflow.hpp:258-294; impl cudaoptflow/src/farneback.cpp:96-165)."""

    def __init__(self, params: capi.FarnebackParams):
        self._h = C.c_void_p()
        capi.check(capi.lib().mi_farneback_create(C.byref(params), C.byref(self._h)))
        self._p = params

    @staticmethod
    def create(numLevels=5, pyrScale=0.5, fastPyramids=False, winSize=13, numIters=10, polyN=5, polySigma=1.1,
               flags=0) -> "FarnebackOpticalFlow":
        p = capi.FarnebackParams()
        capi.lib().mi_farneback_default_params(C.byref(p))
        p.num_levels, p.pyr_scale, p.fast_pyramids, p.win_size = numLevels, pyrScale, int(bool(fastPyramids)), winSize
        p.num_iters, p.poly_n, p.poly_sigma, p.flags = numIters, polyN, polySigma, flags
        return FarnebackOpticalFlow(p)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                capi.lib().mi_farneback_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def getDefaultName(self) -> str:  # farneback.cpp:132
        return "DenseOpticalFlow.FarnebackOpticalFlow"

    def _set(self, **kw):
        for k, v in kw.items():
            setattr(self._p, k, v)
        capi.check(capi.lib().mi_farneback_set_params(self._h, C.byref(self._p)))

    # getters / setters, farneback.cpp:106-128
    def getNumLevels(self): return self._p.num_levels
    def setNumLevels(self, v): self._set(num_levels=v)
    def getPyrScale(self): return self._p.pyr_scale
    def setPyrScale(self, v): self._set(pyr_scale=v)
    def getFastPyramids(self): return bool(self._p.fast_pyramids)
    def setFastPyramids(self, v): self._set(fast_pyramids=int(bool(v)))
    def getWinSize(self): return self._p.win_size
    def setWinSize(self, v): self._set(win_size=v)
    def getNumIters(self): return self._p.num_iters
    def setNumIters(self, v): self._set(num_iters=v)
    def getPolyN(self): return self._p.poly_n
    def setPolyN(self, v): self._set(poly_n=v)
    def getPolySigma(self): return self._p.poly_sigma
    def setPolySigma(self, v): self._set(poly_sigma=v)
    def getFlags(self): return self._p.flags
    def setFlags(self, v): self._set(flags=v)

    def calc(self, I0, I1, flow=None, stream=None):
        """DenseOpticalFlow::calc(I0, I1, flow, stream) (cudaoptflow.hpp:80).  Returns flow (H,W,2) f32."""
        import torch
        if flow is None:
            if self._p.flags & OPTFLOW_USE_INITIAL_FLOW:
                raise MiError(-1, "OPTFLOW_USE_INITIAL_FLOW requires a flow argument")
            flow = torch.empty((I0.shape[0], I0.shape[1], 2), dtype=torch.float32, device=I0.device)
        m0, m1, mf = capi.mat_from_tensor(I0), capi.mat_from_tensor(I1), capi.mat_from_tensor(flow)
        sp = C.c_void_p(stream) if stream is not None else capi.current_stream_ptr()
        capi.check(capi.lib().mi_farneback_calc(self._h, C.byref(m0), C.byref(m1), C.byref(mf), sp))
        return flow

    def calc_batch(self, I0s, I1s, flows=None, stream=None):
        """n independent pairs of one size and type in one pass (miflow extension; blockIdx.z = pair in every kernel)."""
        import torch
        n = len(I0s)
        if flows is None:
            if self._p.flags & OPTFLOW_USE_INITIAL_FLOW:
                raise MiError(-1, "OPTFLOW_USE_INITIAL_FLOW requires the flows argument")
            flows = torch.empty((n, I0s[0].shape[0], I0s[0].shape[1], 2), dtype=torch.float32, device=I0s[0].device)
        A0 = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in I0s])
        A1 = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in I1s])
        AF = (capi.Mat * n)(*[capi.mat_from_tensor(t) for t in flows])
        sp = C.c_void_p(stream) if stream is not None else capi.current_stream_ptr()
        capi.check(capi.lib().mi_farneback_calc_batch(self._h, n, A0, A1, AF, sp))
        return flows


def farneback_polyExp(src, polyN=5, polySigma=1.1):
    import torch
    dst = torch.empty((5 * src.shape[0], src.shape[1]), dtype=torch.float32, device=src.device)
    capi.check(capi.lib().mi_farneback_poly_exp(C.byref(_m(src)), C.byref(_m(dst)), polyN, polySigma, capi.current_stream_ptr()))
    return dst


def farneback_updateMatrices(flowx, flowy, R0, R1):
    import torch
    M = torch.empty_like(R0)
    capi.check(capi.lib().mi_farneback_update_matrices(C.byref(_m(flowx)), C.byref(_m(flowy)), C.byref(_m(R0)), C.byref(_m(R1)),
                                                       C.byref(_m(M)), capi.current_stream_ptr()))
    return M


def farneback_blur5(M, ksize, gaussian=False):
    import torch
    out = torch.empty_like(M)
    capi.check(capi.lib().mi_farneback_blur5(C.byref(_m(M)), C.byref(_m(out)), ksize, int(bool(gaussian)), capi.current_stream_ptr()))
    return out


def farneback_updateFlow(M):
    import torch
    h = M.shape[0] // 5
    fx = torch.empty((h, M.shape[1]), dtype=torch.float32, device=M.device)
    fy = torch.empty_like(fx)
    capi.check(capi.lib().mi_farneback_update_flow(C.byref(_m(M)), C.byref(_m(fx)), C.byref(_m(fy)), capi.current_stream_ptr()))
    return fx, fy


def farneback_iterate(M, R0, R1, ksize, gaussian=False, update=True):
    """One fused inner iteration -> (flowx, flowy, M')."""
    import torch
    h = M.shape[0] // 5
    fx = torch.empty((h, M.shape[1]), dtype=torch.float32, device=M.device)
    fy = torch.empty_like(fx)
    Mo = torch.empty_like(M)
    capi.check(capi.lib().mi_farneback_iterate(C.byref(_m(M)), C.byref(_m(R0)), C.byref(_m(R1)), C.byref(_m(fx)), C.byref(_m(fy)),
                                               C.byref(_m(Mo)), ksize, int(bool(gaussian)), int(bool(update)),
                                               capi.current_stream_ptr()))
    return fx, fy, Mo


def farneback_gaussianBlur(src, ksize, sigma, border=4):
    import torch
    dst = torch.empty_like(src)
    capi.check(capi.lib().mi_farneback_gaussian_blur(C.byref(_m(src)), C.byref(_m(dst)), ksize, float(sigma), border,
                                                     capi.current_stream_ptr()))
    return dst


def pyrDown(src):
    """cv::cuda::pyrDown on CV_32FC1 (cudawarping/src/pyramids.cpp:66-94)."""
    import torch
    dst = torch.empty(((src.shape[0] + 1) // 2, (src.shape[1] + 1) // 2), dtype=torch.float32, device=src.device)
    capi.check(capi.lib().mi_pyr_down(C.byref(_m(src)), C.byref(_m(dst)), capi.current_stream_ptr()))
    return dst


# =============================================================================================
class SURF_CUDA:
    """cv::cuda::SURF_CUDA (xfeatures2d/cuda.hpp:86-196).  Keypoints live in a 7 x N CV_32FC1 device matrix
    (rows X, Y, LAPLACIAN, OCTAVE, SIZE, ANGLE, HESSIAN; LAPLACIAN/OCTAVE hold int bit patterns)."""

    X_ROW, Y_ROW, LAPLACIAN_ROW, OCTAVE_ROW, SIZE_ROW, ANGLE_ROW, HESSIAN_ROW, ROWS_COUNT = range(8)

    def __init__(self, _hessianThreshold=100.0, _nOctaves=4, _nOctaveLayers=2, _extended=True, _keypointsRatio=0.01,
                 _upright=False):
        # defaults of the default constructor (surf.cuda.cpp:257-265): extended = true
        p = capi.SURFParams()
        p.hessian_threshold, p.n_octaves, p.n_octave_layers = _hessianThreshold, _nOctaves, _nOctaveLayers
        p.extended, p.keypoints_ratio, p.upright = int(bool(_extended)), _keypointsRatio, int(bool(_upright))
        self._h = C.c_void_p()
        capi.check(capi.lib().mi_surf_create(C.byref(p), C.byref(self._h)))
        self._p = p

    @staticmethod
    def create(_hessianThreshold, _nOctaves=4, _nOctaveLayers=2, _extended=False, _keypointsRatio=0.01, _upright=False):
        """cuda.hpp:117-118 (extended defaults to false here)."""
        return SURF_CUDA(_hessianThreshold, _nOctaves, _nOctaveLayers, _extended, _keypointsRatio, _upright)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                capi.lib().mi_surf_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _sync(self):
        capi.check(capi.lib().mi_surf_set_params(self._h, C.byref(self._p)))

    # public fields of the reference class
    hessianThreshold = property(lambda s: s._p.hessian_threshold, lambda s, v: (setattr(s._p, "hessian_threshold", v), s._sync()))
    nOctaves = property(lambda s: s._p.n_octaves, lambda s, v: (setattr(s._p, "n_octaves", v), s._sync()))
    nOctaveLayers = property(lambda s: s._p.n_octave_layers, lambda s, v: (setattr(s._p, "n_octave_layers", v), s._sync()))
    extended = property(lambda s: bool(s._p.extended), lambda s, v: (setattr(s._p, "extended", int(bool(v))), s._sync()))
    upright = property(lambda s: bool(s._p.upright), lambda s, v: (setattr(s._p, "upright", int(bool(v))), s._sync()))
    keypointsRatio = property(lambda s: s._p.keypoints_ratio, lambda s, v: (setattr(s._p, "keypoints_ratio", v), s._sync()))

    def descriptorSize(self): return 128 if self._p.extended else 64   # surf.cuda.cpp:277-280
    def defaultNorm(self): return 4                                    # NORM_L2, surf.cuda.cpp:282-285
    def releaseMemory(self): capi.lib().mi_surf_release_memory(self._h)

    def detect(self, img, mask=None, stream=None):
        """operator()(img, mask, keypoints) -> keypoints (7, nFeatures) float32 device tensor."""
        import torch
        mf = C.c_int()
        capi.check(capi.lib().mi_surf_max_features(self._h, img.shape[0], img.shape[1], C.byref(mf)))
        kp = torch.empty((7, mf.value), dtype=torch.float32, device=img.device)   # ensureSizeIsEnough(ROWS_COUNT, maxFeatures)
        n = C.c_int()
        sp = C.c_void_p(stream) if stream is not None else capi.current_stream_ptr()
        mm = C.byref(capi.mat_from_tensor(mask)) if mask is not None else None
        mk = capi.mat_from_tensor(kp)
        capi.check(capi.lib().mi_surf_detect(self._h, C.byref(capi.mat_from_tensor(img)), mm, C.byref(mk), C.byref(n), sp))
        return kp[:, : n.value]   # keypoints.cols = featureCounter (surf.cuda.cpp:209)

    def detectWithDescriptors(self, img, mask=None, keypoints=None, useProvidedKeypoints=False, stream=None):
        """operator()(img, mask, keypoints, descriptors, useProvidedKeypoints) (surf.cuda.cpp:380-397)."""
        import torch
        sp = C.c_void_p(stream) if stream is not None else capi.current_stream_ptr()
        if not useProvidedKeypoints:
            # one enqueue for the frame: the descriptor kernels read the feature count on the device (mi_surf_detect_and_compute)
            mf = C.c_int()
            capi.check(capi.lib().mi_surf_max_features(self._h, img.shape[0], img.shape[1], C.byref(mf)))
            kp = torch.empty((7, mf.value), dtype=torch.float32, device=img.device)
            desc = torch.empty((mf.value, self.descriptorSize()), dtype=torch.float32, device=img.device)
            n = C.c_int()
            mm = C.byref(capi.mat_from_tensor(mask)) if mask is not None else None
            mk, md = capi.mat_from_tensor(kp), capi.mat_from_tensor(desc)
            capi.check(capi.lib().mi_surf_detect_and_compute(self._h, C.byref(capi.mat_from_tensor(img)), mm, C.byref(mk), C.byref(md), C.byref(n), sp))
            return kp[:, : n.value], desc[: n.value]
        elif not self._p.upright:
            capi.check(capi.lib().mi_surf_compute_orientation(self._h, C.byref(capi.mat_from_tensor(img)),
                                                              C.byref(capi.mat_from_tensor(keypoints)), keypoints.shape[1], sp))
        n = keypoints.shape[1]
        desc = torch.empty((n, self.descriptorSize()), dtype=torch.float32, device=img.device)
        if n:
            capi.check(capi.lib().mi_surf_compute_descriptors(self._h, C.byref(capi.mat_from_tensor(img)),
                                                              C.byref(capi.mat_from_tensor(keypoints)), n,
                                                              C.byref(capi.mat_from_tensor(desc)), sp))
        return keypoints, desc

    # ---- the CPU class's orientation / descriptor arithmetic (xfeatures2d::SURF_Impl, surf.cpp:568-866) on the same keypoint matrix
    def cpuClassOrientation(self, img, keypoints, stream=None):
        """Writes the ANGLE row of `keypoints` (7, n) in place as SURFInvoker does (270 when upright) and sets SIZE to -1 for the
        keypoints the CPU class erases."""
        sp = C.c_void_p(stream) if stream is not None else capi.current_stream_ptr()
        if keypoints.shape[1] == 0:
            return keypoints
        s = surf_integral(img)
        capi.check(capi.lib().mi_surfcpu_orientation(C.byref(capi.mat_from_tensor(s)), C.byref(capi.mat_from_tensor(keypoints)),
                                                     keypoints.shape[1], int(self._p.upright), sp))
        return keypoints

    def cpuClassDescriptors(self, img, keypoints, stream=None):
        import torch
        sp = C.c_void_p(stream) if stream is not None else capi.current_stream_ptr()
        n = keypoints.shape[1]
        desc = torch.empty((n, self.descriptorSize()), dtype=torch.float32, device=img.device)
        if n:
            capi.check(capi.lib().mi_surfcpu_descriptors(C.byref(capi.mat_from_tensor(img)), C.byref(capi.mat_from_tensor(keypoints)), n,
                                                         int(self._p.extended), int(self._p.upright), C.byref(capi.mat_from_tensor(desc)), sp))
        return desc

    def detectAndComputeCpuClass(self, img, mask=None, keypoints=None, useProvidedKeypoints=False):
        """cv::xfeatures2d::SURF::detectAndCompute (surf.cpp:881-1015) on the GPU: this class's detector (the same fast-Hessian
        responses: tests/test_zz_surf_cpu_class.py), then the CPU class's orientation and descriptor; keypoints it erases are
        removed.  -> (keypoints (7, n), descriptors (n, 64 | 128))."""
        if not useProvidedKeypoints:
            keypoints = self.detect(img, mask)
        keypoints = keypoints.contiguous().clone()
        self.cpuClassOrientation(img, keypoints)
        desc = self.cpuClassDescriptors(img, keypoints)
        keep = keypoints[4] > 0
        return keypoints[:, keep], desc[keep]

    @staticmethod
    def downloadKeypoints(keypointsGPU):
        """-> dict of host arrays (the fields of cv::KeyPoint the reference fills, surf.cuda.cpp:319-356)."""
        import numpy as np
        k = keypointsGPU.detach().cpu().numpy()
        ki = k.view(np.int32)
        return {"x": k[0].copy(), "y": k[1].copy(), "laplacian": ki[2].copy(), "octave": ki[3].copy(), "size": k[4].copy(),
                "angle": k[5].copy(), "hessian": k[6].copy()}


def surf_integral(img, clamp_to_one=False):
    import torch
    alg = SURF_CUDA(100)
    s = torch.empty((img.shape[0] + 1, img.shape[1] + 1), dtype=torch.int32, device=img.device)
    capi.check(capi.lib().mi_surf_integral(alg._h, C.byref(_m(img)), int(bool(clamp_to_one)), C.byref(_m(s)), capi.current_stream_ptr()))
    return s


def surf_detTrace(sum_, octave, nOctaveLayers=2):
    import torch
    alg = SURF_CUDA(100)
    rows, cols = sum_.shape[0] - 1, sum_.shape[1] - 1
    det = torch.empty(((nOctaveLayers + 2) * (rows >> octave), cols), dtype=torch.float32, device=sum_.device)
    tr = torch.empty_like(det)
    capi.check(capi.lib().mi_surf_det_trace(alg._h, C.byref(_m(sum_)), octave, nOctaveLayers, C.byref(_m(det)), C.byref(_m(tr)),
                                            capi.current_stream_ptr()))
    torch.cuda.synchronize()
    return det, tr
