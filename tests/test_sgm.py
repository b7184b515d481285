"""cv::cuda::StereoSGM.  The stage tests restate the reference's own unit tests (cudastereo/test/test_sgm_funcs.cpp):
random census / cost volumes through each device stage against the CPU twin the reference ships for it, bit-exact
(EXPECT_MAT_NEAR(gold, dst, 0)); the oracle's census, path aggregation and left winner-takes-all ARE those twins
(oracle/sgm_ref.c), so these stages are pinned on the reference's own tests."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opencv_contrib_amd import synth  # noqa: E402

DIRS = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, 1), (-1, -1), (1, -1)]   # test_sgm_funcs.cpp:301-339


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_census_bit_order_and_border(oracle):
    """A single bright pixel: bit b of the census of a pixel is set iff the b-th compared neighbour (row-major over the upper
    half window, test_sgm_funcs.cpp:139-147) is brighter than its point reflection; the 4 / 3 pixel border stays 0."""
    img = np.zeros((15, 21), np.uint8)
    img[7, 10] = 255
    c = oracle.sgm_census(img)
    assert (c[:3] == 0).all() and (c[-3:] == 0).all() and (c[:, :4] == 0).all() and (c[:, -4:] == 0).all()
    # centre pixel (7, 10): no compared pair contains it (pairs are point reflections about the centre) -> 0
    assert c[7, 10] == 0
    # pixel (8, 10): its neighbour (dy, dx) = (-1, 0) is the bright pixel = comparison index 9 + 9 + 4 = 22 of 31 -> bit 30 - 22
    assert c[8, 10] == 1 << (30 - 22)
    # pixel (6, 10): the bright pixel is its (+1, 0) neighbour = the reflected operand b of that same comparison -> a > b false
    assert c[6, 10] == 0


def test_oracle_path_first_pixel_and_saturation(oracle):
    rng = np.random.default_rng(0)
    l = rng.integers(0, 2 ** 31 - 1, (5, 9), dtype=np.int64).astype(np.int32)
    r = rng.integers(0, 2 ** 31 - 1, (5, 9), dtype=np.int64).astype(np.int32)
    out = oracle.sgm_path(l, r, 64, 0, 10, 120, 1, 0).reshape(5, 9, 64)
    # first pixel of a left-to-right path: no predecessor, cost = hamming(l, 0) for every disparity > 0 (k > j -> r = 0)
    assert out[2, 0, 5] == bin(int(l[2, 0]) & 0xffffffff).count("1")
    assert out[2, 0, 0] == bin((int(l[2, 0]) ^ int(r[2, 0])) & 0xffffffff).count("1")
    big = oracle.sgm_path(l, r, 64, 0, 10, 250, 1, 0)      # P2 + popcount can exceed 255: static_cast<uint8_t> wraps
    assert big.dtype == np.uint8


def test_oracle_compute_rejects_unsupported(oracle):
    img = np.zeros((32, 64), np.uint8)
    with pytest.raises(ValueError):
        oracle.sgm_compute(img, img, oracle.sgm_params(num_disparities=96))      # stereosgm.cpp:138 "Unsupported num of disparities"
    with pytest.raises(ValueError):
        oracle.sgm_compute(img, img, oracle.sgm_params(mode=0))                  # stereosgm.cpp:102-105 "Unsupported mode"


def test_oracle_recovers_synthetic_disparity(oracle):
    left, right, gt = synth.stereo_pair(96, 224, seed=42, max_disp=30)
    d = oracle.sgm_compute(left, right, oracle.sgm_params(num_disparities=64)).astype(np.float64) / 16
    ys, xs = np.mgrid[0:96, 0:224]
    valid = d > 0
    xr = np.clip(np.round(xs - d).astype(int), 0, 223)
    assert valid.mean() > 0.7 and np.median(np.abs(d - gt[ys, xr])[valid]) < 1.0


# ------------------------------------------------------------------ stages, HIP vs the reference's CPU twins (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("shape", [(128, 128), (113, 131), (8, 12), (6, 30)])
def test_census_random(gpu, oracle, dtype, shape):
    """StereoSGM_CensusTransformRandom, test_sgm_funcs.cpp:190-206."""
    import torch
    from opencv_contrib_amd import cuda
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256 if dtype == np.uint8 else 65536, shape).astype(dtype)
    out = cuda.sgm_census(torch.from_numpy(img).to(gpu)).cpu().numpy()
    np.testing.assert_array_equal(out, oracle.sgm_census(img))


@pytest.mark.gpu
@pytest.mark.parametrize("dxdy", DIRS)
@pytest.mark.parametrize("min_disp", [0, 1, 10])
def test_path_aggregation_random(gpu, oracle, dxdy, min_disp):
    """StereoSGM_PathAggregation.Random*, test_sgm_funcs.cpp:262-345: DISPARITY 128, P1 10, P2 120, random 31-bit census values."""
    import torch
    from opencv_contrib_amd import cuda
    rng = np.random.default_rng(7 + min_disp)
    h, w = 67, 113
    l = rng.integers(0, 2 ** 31 - 1, (h, w), dtype=np.int64).astype(np.int32)
    r = rng.integers(0, 2 ** 31 - 1, (h, w), dtype=np.int64).astype(np.int32)
    out = cuda.sgm_aggregate_path(torch.from_numpy(l).to(gpu), torch.from_numpy(r).to(gpu), 128, min_disp, 10, 120, *dxdy).cpu().numpy()
    np.testing.assert_array_equal(out.reshape(-1), oracle.sgm_path(l, r, 128, min_disp, 10, 120, *dxdy))


@pytest.mark.gpu
@pytest.mark.parametrize("D", [64, 256])
def test_path_aggregation_other_disparity_counts(gpu, oracle, D):
    import torch
    from opencv_contrib_amd import cuda
    rng = np.random.default_rng(D)
    l = rng.integers(0, 2 ** 31 - 1, (21, 300), dtype=np.int64).astype(np.int32)
    r = rng.integers(0, 2 ** 31 - 1, (21, 300), dtype=np.int64).astype(np.int32)
    for dxdy in [(1, 0), (0, -1), (-1, 1)]:
        out = cuda.sgm_aggregate_path(torch.from_numpy(l).to(gpu), torch.from_numpy(r).to(gpu), D, 2, 7, 200, *dxdy).cpu().numpy()
        np.testing.assert_array_equal(out.reshape(-1), oracle.sgm_path(l, r, D, 2, 7, 200, *dxdy))


@pytest.mark.gpu
@pytest.mark.parametrize("subpixel", [False, True])
@pytest.mark.parametrize("npaths", [4, 8])
@pytest.mark.parametrize("D", [64, 128, 256])
def test_winner_takes_all_random(gpu, oracle, subpixel, npaths, D):
    """StereoSGM_WinnerTakesAll.RandomLeft, test_sgm_funcs.cpp:405-441 (costs 0..32, uniqueness 0.95) -- and the right map."""
    import torch
    from opencv_contrib_amd import cuda
    rng = np.random.default_rng(D + npaths)
    h, w = 19, 141
    agg = rng.integers(0, 32, (1, w * h * D * npaths)).astype(np.uint8)
    left, right = cuda.sgm_winner_takes_all(torch.from_numpy(agg).to(gpu), w, h, D, npaths, 0.95, subpixel)
    rl, rr = oracle.sgm_wta(agg, w, h, D, npaths, 0.95, subpixel)
    np.testing.assert_array_equal(left.cpu().numpy(), rl)
    np.testing.assert_array_equal(right.cpu().numpy(), rr)


@pytest.mark.gpu
@pytest.mark.parametrize("D,w", [(64, 600), (128, 777), (256, 1100)])
def test_winner_takes_all_wide_rows_are_segmented(gpu, oracle, D, w):
    """Rows wider than one segment (256 / 512 px): segments walk D - 1 pixels past their end for the right minima."""
    import torch
    from opencv_contrib_amd import cuda
    rng = np.random.default_rng(D + w)
    h = 5
    agg = rng.integers(0, 32, (1, w * h * D * 4)).astype(np.uint8)
    left, right = cuda.sgm_winner_takes_all(torch.from_numpy(agg).to(gpu), w, h, D, 4, 0.95, True)
    rl, rr = oracle.sgm_wta(agg, w, h, D, 4, 0.95, True)
    np.testing.assert_array_equal(left.cpu().numpy(), rl)
    np.testing.assert_array_equal(right.cpu().numpy(), rr)


# ------------------------------------------------------------------ full pipeline
@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 3])
@pytest.mark.parametrize("D,min_disp", [(64, 0), (128, 3)])
@pytest.mark.parametrize("quirks", [True, False])
def test_compute_bit_exact(gpu, oracle, mode, D, min_disp, quirks):
    import torch
    from opencv_contrib_amd import cuda
    left, right, _ = synth.stereo_pair(101, 237, seed=11, max_disp=40)      # not multiples of 16: the consistency-check quirk matters
    sgm = cuda.createStereoSGM(min_disp, D, 10, 120, 5, mode, emulateCudaQuirks=quirks)
    assert (sgm.getMinDisparity(), sgm.getNumDisparities(), sgm.getP1(), sgm.getP2(), sgm.getUniquenessRatio(), sgm.getMode()) == \
        (min_disp, D, 10, 120, 5, mode)
    assert (sgm.getBlockSize(), sgm.getDisp12MaxDiff(), sgm.getPreFilterCap()) == (-1, 1, -1)       # stereosgm.cpp:38-72
    out = sgm.compute(torch.from_numpy(left).to(gpu), torch.from_numpy(right).to(gpu)).cpu().numpy()
    ref = oracle.sgm_compute(left, right, oracle.sgm_params(min_disp, D, 10, 120, 5, mode, int(quirks)))
    np.testing.assert_array_equal(out, ref)
    assert (out >= (min_disp - 1) * 16).all() and (out > 0).mean() > 0.5


@pytest.mark.gpu
def test_compute_16bit_images_pitched_and_errors(gpu, oracle):
    import torch
    from opencv_contrib_amd import capi, cuda
    left, right, _ = synth.stereo_pair(64, 160, seed=12, max_disp=30)
    l16, r16 = (left.astype(np.uint16) * 257), (right.astype(np.uint16) * 257)
    sgm = cuda.createStereoSGM(0, 64)
    bigl = torch.zeros((70, 200), dtype=torch.uint16, device=gpu); bigr = torch.zeros_like(bigl)
    bigl[3:67, 20:180] = torch.from_numpy(l16).to(gpu); bigr[3:67, 20:180] = torch.from_numpy(r16).to(gpu)
    out = sgm.compute(bigl[3:67, 20:180], bigr[3:67, 20:180]).cpu().numpy()        # pitched ROI views
    np.testing.assert_array_equal(out, oracle.sgm_compute(l16, r16, oracle.sgm_params(num_disparities=64)))
    tl = torch.from_numpy(left).to(gpu)
    sgm.setNumDisparities(96)
    with pytest.raises(capi.MiError):
        sgm.compute(tl, tl)                                    # "Unsupported num of disparities"
    sgm.setNumDisparities(64); sgm.setMode(0)
    with pytest.raises(capi.MiError):
        sgm.compute(tl, tl)                                    # "Unsupported mode"
    sgm.setMode(3)
    with pytest.raises(capi.MiError):
        sgm.compute(tl, torch.from_numpy(right[:, :100].copy()).to(gpu))           # size mismatch
