"""Mirror of the reference's superres GPU optical-flow adapters: the one in-tree CALLER of the hot path.

`cv::superres::createOptFlow_DualTVL1_CUDA()` / `createOptFlow_Farneback_CUDA()` (superres/src/optical_flow.cpp:665-845) wrap
the `cv::cuda` flow classes behind `DenseOpticalFlowExt::calc(frame0, frame1, flow1, flow2)`: frames of any supported
depth/channel count are converted to CV_8UC1 (`convertToType`, input_array_utility.cpp:291-314), the class runs, and the
CV_32FC2 result is split into two CV_32FC1 planes (or returned merged when flow2 is not requested, optical_flow.cpp:475-493).
Frames are torch CUDA tensors here where the reference takes GpuMat; everything runs through the C-ABI (no CPU fallback).
"""
from __future__ import annotations

import ctypes as C

from . import capi, cuda


def _m(t):
    return capi.mat_from_tensor(t)


def convertToGray8(frame):
    """convertToType(frame, CV_8UC1): BGR/BGRA -> gray, then depth -> 8U with scale 255 / maxVal(depth)."""
    import torch
    if frame.dtype == torch.uint8 and frame.dim() == 2:
        return frame   # input_array_utility.cpp:293-294: same type, no copy
    dst = torch.empty(frame.shape[:2], dtype=torch.uint8, device=frame.device)
    capi.check(capi.lib().mi_superres_to_gray8(C.byref(_m(frame)), C.byref(_m(dst)), capi.current_stream_ptr()))
    return dst


def splitFlow(flow):
    """cuda::split(flow, flows): CV_32FC2 -> two CV_32FC1 planes."""
    import torch
    u = torch.empty(flow.shape[:2], dtype=torch.float32, device=flow.device)
    v = torch.empty_like(u)
    capi.check(capi.lib().mi_split_flow(C.byref(_m(flow)), C.byref(_m(u)), C.byref(_m(v)), capi.current_stream_ptr()))
    return u, v


class _GpuOpticalFlow:
    """cv::superres GpuOpticalFlow (optical_flow.cpp:436-505), work type CV_8UC1."""

    def calc(self, frame0, frame1, want_flow2: bool = True):
        """Returns (flow1, flow2) = (u, v) planes, or the merged CV_32FC2 flow when want_flow2 is False
        (`_flow2.needed()` in the reference)."""
        if frame0.dtype != frame1.dtype or frame0.shape != frame1.shape:
            raise capi.MiError(-3, "frame1.type() == frame0.type() && frame1.size() == frame0.size()")   # optical_flow.cpp:466-467
        in0, in1 = convertToGray8(frame0), convertToGray8(frame1)
        flow = self._impl(in0, in1)
        return splitFlow(flow) if want_flow2 else flow

    def collectGarbage(self):
        self._alg = self._create()


class DualTVL1_CUDA(_GpuOpticalFlow):
    """cv::superres::createOptFlow_DualTVL1_CUDA() (optical_flow.cpp:757-845): getters/setters of
    cv::superres::DualTVL1OpticalFlow, parameters pushed into the cuda class at every calc (:817-826)."""

    def __init__(self):
        self._alg = self._create()
        a = self._alg
        self._p = dict(Tau=a.getTau(), Lambda=a.getLambda(), Theta=a.getTheta(), ScalesNumber=a.getNumScales(),
                       WarpingsNumber=a.getNumWarps(), Epsilon=a.getEpsilon(), Iterations=a.getNumIterations(),
                       UseInitialFlow=a.getUseInitialFlow())

    @staticmethod
    def _create():
        return cuda.OpticalFlowDual_TVL1.create()

    def _impl(self, in0, in1):
        a, p = self._alg, self._p
        a.setTau(p["Tau"]); a.setLambda(p["Lambda"]); a.setTheta(p["Theta"]); a.setNumScales(p["ScalesNumber"])
        a.setNumWarps(p["WarpingsNumber"]); a.setEpsilon(p["Epsilon"]); a.setNumIterations(p["Iterations"])
        a.setUseInitialFlow(p["UseInitialFlow"])
        return a.calc(in0, in1)


class Farneback_CUDA(_GpuOpticalFlow):
    """cv::superres::createOptFlow_Farneback_CUDA() (optical_flow.cpp:665-750)."""

    def __init__(self):
        self._alg = self._create()
        a = self._alg
        self._p = dict(PyrScale=a.getPyrScale(), LevelsNumber=a.getNumLevels(), WindowSize=a.getWinSize(),
                       Iterations=a.getNumIters(), PolyN=a.getPolyN(), PolySigma=a.getPolySigma(), Flags=a.getFlags())

    @staticmethod
    def _create():
        return cuda.FarnebackOpticalFlow.create()

    def _impl(self, in0, in1):
        a, p = self._alg, self._p
        a.setPyrScale(p["PyrScale"]); a.setNumLevels(p["LevelsNumber"]); a.setWinSize(p["WindowSize"])
        a.setNumIters(p["Iterations"]); a.setPolyN(p["PolyN"]); a.setPolySigma(p["PolySigma"]); a.setFlags(p["Flags"])
        return a.calc(in0, in1)


def _add_accessors(cls, names):
    for n in names:
        setattr(cls, "get" + n, (lambda n: lambda self: self._p[n])(n))
        setattr(cls, "set" + n, (lambda n: lambda self, val: self._p.__setitem__(n, val))(n))


_add_accessors(DualTVL1_CUDA, ["Tau", "Lambda", "Theta", "ScalesNumber", "WarpingsNumber", "Epsilon", "Iterations", "UseInitialFlow"])
_add_accessors(Farneback_CUDA, ["PyrScale", "LevelsNumber", "WindowSize", "Iterations", "PolyN", "PolySigma", "Flags"])


def createOptFlow_DualTVL1_CUDA():
    return DualTVL1_CUDA()


def createOptFlow_Farneback_CUDA():
    return Farneback_CUDA()
