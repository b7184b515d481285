"""cv::cuda::DensePyrLKOpticalFlow (second part): HIP vs the CPU restatement, bit-exact (the texture reads are defined
identically on both sides, oracle/pyrlk_ref.c)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opencv_contrib_amd import synth  # noqa: E402


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_recovers_translation_and_analytic_flow(oracle):
    rng = np.random.default_rng(0)
    base = synth.blob_image(140, 180, seed=3)
    I0, I1 = base[10:130, 12:172].copy(), base[10:130, 9:169].copy()      # I1(x) = I0(x - 3): pure translation u = +3
    f = oracle.pyrlk_dense(I0, I1)
    inner = f[24:-24, 24:-24]
    assert abs(np.median(inner[..., 0]) - 3.0) < 0.1 and abs(np.median(inner[..., 1])) < 0.1
    A0, A1, gt = synth.flow_pair(120, 160, seed=5, dtype="u8")
    assert synth.epe(oracle.pyrlk_dense(A0, A1)[20:-20, 20:-20], gt[20:-20, 20:-20]) < 0.3


def test_oracle_textureless_pixels_stay_zero_and_bad_args(oracle):
    """Singular structure tensor (D < FLT_EPSILON): the kernel returns without writing (pyrlk.cu:777-782) -> the zero the
    buffers were initialised with."""
    flat = np.full((48, 64), 90, np.uint8)
    assert (oracle.pyrlk_dense(flat, flat) == 0).all()
    with pytest.raises(ValueError):
        oracle.pyrlk_dense(flat, flat, win_size=(2, 13))       # winSize > 2, pyrlk.cpp:243
    with pytest.raises(ValueError):
        oracle.pyrlk_dense(flat, flat, max_level=-1)


# ------------------------------------------------------------------ HIP vs oracle (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(120, 160), (67, 101)])
@pytest.mark.parametrize("win,levels,iters", [((13, 13), 3, 30), ((21, 21), 2, 10), ((7, 9), 1, 5), ((13, 5), 0, 30)])
def test_calc_bit_exact(gpu, oracle, shape, win, levels, iters):
    import torch
    from opencv_contrib_amd import cuda
    I0, I1, _ = synth.flow_pair(shape[0], shape[1], seed=21, dtype="u8")
    alg = cuda.DensePyrLKOpticalFlow.create(win, levels, iters)
    assert alg.getWinSize() == win and alg.getMaxLevel() == levels and alg.getNumIters() == iters and not alg.getUseInitialFlow()
    flow = alg.calc(torch.from_numpy(I0).to(gpu), torch.from_numpy(I1).to(gpu)).cpu().numpy()
    np.testing.assert_array_equal(flow, oracle.pyrlk_dense(I0, I1, win, levels, iters))


@pytest.mark.gpu
def test_calc_pitched_inputs_reuse_and_errors(gpu, oracle):
    import torch
    from opencv_contrib_amd import capi, cuda
    I0, I1, _ = synth.flow_pair(96, 128, seed=22, dtype="u8")
    alg = cuda.DensePyrLKOpticalFlow.create()
    big0 = torch.zeros((110, 160), dtype=torch.uint8, device=gpu); big1 = torch.zeros_like(big0)
    big0[7:103, 16:144] = torch.from_numpy(I0).to(gpu); big1[7:103, 16:144] = torch.from_numpy(I1).to(gpu)
    ref = oracle.pyrlk_dense(I0, I1)
    for _ in range(2):                                        # second call: scratch reuse must re-zero the (u, v) buffers
        flow = alg.calc(big0[7:103, 16:144], big1[7:103, 16:144]).cpu().numpy()
        np.testing.assert_array_equal(flow, ref)
    with pytest.raises(capi.MiError):
        alg.calc(torch.from_numpy(I0).to(gpu), torch.from_numpy(I1[:, :100].copy()).to(gpu))
    with pytest.raises(capi.MiError):
        alg.calc(torch.from_numpy(I0.astype(np.float32)).to(gpu), torch.from_numpy(I1.astype(np.float32)).to(gpu))   # CV_8UC1 only
    alg.setWinSize((2, 13))
    with pytest.raises(capi.MiError):
        alg.calc(torch.from_numpy(I0).to(gpu), torch.from_numpy(I1).to(gpu))
