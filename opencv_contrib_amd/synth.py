"""Synthetic inputs for the hot path (no datasets are reachable: opencv_extra is absent).

Generators follow SURVEY.md section 8d:
  * flow pairs: band-limited random texture (white noise -> Gaussian sigma -> stretch to
    0..255) and the same texture warped by a known smooth field, so the true flow is
    analytic;
  * stereo pairs: random texture + integer disparity field (bit-exact truth);
  * SURF frames: sum of Gaussian blobs + noise; and the reference's own synthetic "cross"
    known-answer image (xfeatures2d/test/test_rotation_and_scale_invariance.cpp:259-285).
numpy/scipy only -- usable on the CPU-only container and on the GPU box.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage


def texture(h: int, w: int, seed: int, sigma: float = 2.0) -> np.ndarray:
    """Band-limited random texture, float64 in [0, 255]."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(h, w)).astype(np.float64)
    img = ndimage.gaussian_filter(img, sigma, mode="reflect")
    lo, hi = img.min(), img.max()
    return (img - lo) * (255.0 / (hi - lo))


def flow_field(h: int, w: int, scale: float = 1.0, x=None, y=None):
    """u = 2 + 1.5 sin(2 pi y/(h/2)), v = -1 + cos(2 pi x/(w/2)), times `scale` (px),
    evaluated on the pixel grid or at the given coordinates."""
    if x is None:
        y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    u = scale * (2.0 + 1.5 * np.sin(2 * np.pi * y / (h / 2.0)))
    v = scale * (-1.0 + 1.0 * np.cos(2 * np.pi * x / (w / 2.0)))
    return u, v


def flow_pair(h: int, w: int, seed: int = 1234, dtype: str = "f32", flow_scale: float | None = None,
              sigma: float | None = None):
    """Returns (I0, I1, flow_gt[h,w,2]).

    I1(q) = I0(q - F(q)) with F the analytic field; the flow TV-L1 estimates, w with
    I1(p + w(p)) = I0(p), is the fixed point w = F(p + w), solved here to ~1e-9 px.
    dtype 'u8' -> CV_8UC1; 'f32' -> CV_32FC1 in [0, 1] (the API multiplies floats by 255,
    cudaoptflow/src/tvl1flow.cpp:200-201).
    """
    if flow_scale is None:
        flow_scale = max(1.0, w / 640.0)
    if sigma is None:
        sigma = 2.0 * max(1.0, w / 640.0)
    I0 = texture(h, w, seed, sigma)
    u, v = flow_field(h, w, flow_scale)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    # I1(p) = I0(p - flow(p)) : then I1(p + flow) ~= I0(p) for smooth small flow gradients
    I1 = ndimage.map_coordinates(I0, [y - v, x - u], order=3, mode="reflect")
    I1 = np.clip(I1, 0, 255)
    gu, gv = u.copy(), v.copy()
    for _ in range(12):
        gu, gv = flow_field(h, w, flow_scale, x + gu, y + gv)
    gt = np.stack([gu, gv], axis=-1).astype(np.float32)
    if dtype == "u8":
        return np.rint(I0).astype(np.uint8), np.rint(I1).astype(np.uint8), gt
    return (I0 / 255.0).astype(np.float32), (I1 / 255.0).astype(np.float32), gt


def stereo_pair(h: int, w: int, seed: int = 42, max_disp: int = 70):
    """Left/right CV_8UC1 with an integer disparity field d in [10, max_disp]:
    right(x, y) = left(x + d, y)  <=>  left(x,y) matches right(x - d, y).  Columns with no
    source are filled with independent noise (seed+1). Returns (left, right, disp_gt)."""
    rng = np.random.default_rng(seed + 1)
    left = np.rint(texture(h, w, seed, 1.5)).astype(np.uint8)
    y, x = np.mgrid[0:h, 0:w]
    amp = (max_disp - 10) / 2.0
    d = np.rint(10 + amp + amp * np.sin(2 * np.pi * x / (w / 2.0)) * np.cos(2 * np.pi * y / (h / 2.0))).astype(np.int64)
    src = x + d
    right = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    ok = src < w
    right[ok] = left[y[ok], src[ok]]
    return left, right, d.astype(np.int32)


def blob_image(h: int, w: int, seed: int = 7, n_blobs: int | None = None) -> np.ndarray:
    """SURF test frame: Gaussian blobs of log-uniform scale on mid-grey + N(0,2) noise, u8."""
    rng = np.random.default_rng(seed)
    if n_blobs is None:
        n_blobs = max(50, int(20000 * (h * w) / (3840 * 2160)))
    img = np.full((h, w), 128.0)
    cx = rng.uniform(0, w, n_blobs)
    cy = rng.uniform(0, h, n_blobs)
    sg = np.exp(rng.uniform(np.log(1.5), np.log(24.0), n_blobs))
    amp = rng.uniform(30, 110, n_blobs) * rng.choice([-1.0, 1.0], n_blobs)
    for i in range(n_blobs):
        r = int(np.ceil(4 * sg[i]))
        x0, x1 = max(0, int(cx[i]) - r), min(w, int(cx[i]) + r + 1)
        y0, y1 = max(0, int(cy[i]) - r), min(h, int(cy[i]) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        img[y0:y1, x0:x1] += amp[i] * np.exp(-((xx - cx[i]) ** 2 + (yy - cy[i]) ** 2) / (2 * sg[i] ** 2))
    img += rng.normal(0, 2.0, size=(h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def cross_image() -> np.ndarray:
    """100x100 white image with two 3-px-thick lines of value 100 forming a cross
    (cv::line((30,50),(69,50),100,3) and ((50,30),(50,69),100,3)); restated rasterisation:
    thickness-3 axis-aligned lines cover +-1 px about the axis and, with OpenCV's round
    end caps, one extra pixel on the axis at each end."""
    img = np.full((100, 100), 255, np.uint8)
    img[49:52, 30:70] = 100
    img[30:70, 49:52] = 100
    img[50, 29] = img[50, 70] = 100
    img[29, 50] = img[70, 50] = 100
    return img


def epe(flow_a: np.ndarray, flow_b: np.ndarray) -> float:
    """Mean end-point error (optflow/samples/optical_flow_evaluation.cpp:32-49,117-122)."""
    d = flow_a.astype(np.float64) - flow_b.astype(np.float64)
    return float(np.sqrt((d ** 2).sum(-1)).mean())


def ccorr_dissimilarity(a: np.ndarray, b: np.ndarray) -> float:
    """|1 - TM_CCORR_NORMED(a,b)| -- the reference's EXPECT_MAT_SIMILAR metric
    (cudaoptflow/test/test_optflow.cpp:465)."""
    a = a.astype(np.float64).ravel()
    b = b.astype(np.float64).ravel()
    return float(abs(1.0 - (a @ b) / np.sqrt((a @ a) * (b @ b))))
