"""Middlebury `.flo` optical-flow files and the reference's flow acceptance checks.

The on-disk format of the reference's TV-L1 golden (`optflow/tvl1_flow.flo`, optflow/test/test_tvl1optflow.cpp:49-107):
4 bytes "PIEH" (= float 202021.25 little endian), int32 width, int32 height, then height*width interleaved (u, v) float32,
row-major.  Host-side only (numpy): a data format either side of the hot path, not a kernel.
"""
from __future__ import annotations

import numpy as np

FLO_TAG_FLOAT = np.float32(202021.25)   # test_tvl1optflow.cpp:50
FLO_TAG_STRING = b"PIEH"                # test_tvl1optflow.cpp:57


def writeOpticalFlow(path: str, flow) -> None:
    """writeOpticalFlowToFile (test_tvl1optflow.cpp:55-75).  flow: (H, W, 2) float32."""
    f = np.ascontiguousarray(flow, dtype="<f4")
    if f.ndim != 3 or f.shape[2] != 2:
        raise ValueError("flow must be (H, W, 2)")
    with open(path, "wb") as fh:
        fh.write(FLO_TAG_STRING)
        fh.write(np.array([f.shape[1], f.shape[0]], dtype="<i4").tobytes())
        fh.write(f.tobytes())


def readOpticalFlow(path: str) -> np.ndarray:
    """readOpticalFlowFromFile (test_tvl1optflow.cpp:80-107): asserts the tag, returns (H, W, 2) float32."""
    with open(path, "rb") as fh:
        raw = fh.read()
    if len(raw) < 12 or np.frombuffer(raw[:4], dtype="<f4")[0] != FLO_TAG_FLOAT:
        raise ValueError("tag == FLO_TAG_FLOAT")   # CV_Assert in the reference
    w, h = (int(v) for v in np.frombuffer(raw[4:12], dtype="<i4"))
    if w <= 0 or h <= 0 or len(raw) < 12 + 8 * w * h:
        raise ValueError("truncated .flo file")
    return np.frombuffer(raw[12:12 + 8 * w * h], dtype="<f4").reshape(h, w, 2).astype(np.float32)


def isFlowCorrect(flow) -> np.ndarray:
    """Per-pixel validity mask (test_tvl1optflow.cpp:109-112): both components finite and |.| < 1e9."""
    f = np.asarray(flow)
    return ~np.isnan(f[..., 0]) & ~np.isnan(f[..., 1]) & (np.abs(f[..., 0]) < 1e9) & (np.abs(f[..., 1]) < 1e9)


def accuracy(gold, flow, threshold: float = 0.1) -> float:
    """Fraction of valid gold pixels whose flow is valid and within `threshold` px (test_tvl1optflow.cpp:114-141);
    the reference accepts >= 0.95 at threshold 0.1."""
    gold, flow = np.asarray(gold, dtype=np.float64), np.asarray(flow, dtype=np.float64)
    if gold.shape != flow.shape:
        raise ValueError("gold.size() == flow.size()")
    g_ok = isFlowCorrect(gold)
    both = g_ok & isFlowCorrect(flow)
    d = np.where(both[..., None], gold - flow, 0.0)
    err = (d * d).sum(-1)
    return float((both & (err <= threshold * threshold)).sum()) / max(int(g_ok.sum()), 1)


def calcRMSE(flow1, flow2) -> float:
    """optflow/test/test_OF_accuracy.cpp:58-85: RMSE of the endpoint error over pixels valid in both flows."""
    a, b = np.asarray(flow1, dtype=np.float64), np.asarray(flow2, dtype=np.float64)
    ok = isFlowCorrect(a) & isFlowCorrect(b)
    d = (a - b)[ok]
    return float(np.sqrt((d * d).sum(-1).mean())) if ok.any() else 0.0
