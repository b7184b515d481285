"""HIP product vs the REFERENCE'S OWN CUDA classes, directly (VERDICT r02 weak #1c: every `-m gpu` test compared the HIP path with the
restatement only).

oracle/_ref/libref_cu.so holds the reference's four host classes -- cudaoptflow/src/tvl1flow.cpp, cudaoptflow/src/farneback.cpp,
cudastereo/src/stereobm.cpp, xfeatures2d/src/surf.cuda.cpp, compiled verbatim -- over the reference's own kernels (tvl1flow.cu,
farneback.cu, stereobm.cu, surf.cu, resize.cu, pyr_down.cu) executed on the host (oracle/Makefile.ref).  The library is built where
/root/reference exists and travels to the GPU box with the snapshot; these tests skip where it is absent.  Tolerances are those of the
HIP-vs-oracle tests of each class (integer paths bit-exact) -- the oracle equals these classes bit for bit
(tests/test_ref_pin_cuda.py), so this is the same statement made without the restatement in between.
"""
import numpy as np
import pytest

from opencv_contrib_amd import synth
from oracle import refcu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not refcu.available(), reason="oracle/_ref/libref_cu.so not built (needs /root/reference)")]


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("shape,seed,dtype,kw", [
    ((120, 160), 3, "f32", dict(iterations=10, epsilon=0.0)),          # the headline setting under cv::cuda's semantics
    ((96, 128), 5, "u8", dict(iterations=10)),                           # the reference test's literal setting: epsilon stays 0.01
    ((64, 88), 7, "f32", dict()),                                        # class defaults: 300 iterations, epsilon 0.01, sparse check schedule
    # Gamma(1.0), the other half of the reference's test matrix (cudaoptflow/test/test_optflow.cpp:451,530-532): the blocked kernel with the
    # illumination channel (round 6) against the reference class's own estimateU / estimateDualVariables (tvl1flow.cu:209-348)
    ((240, 320), 9, "f32", dict(iterations=10, epsilon=0.0, gamma=1.0)),
    ((96, 128), 11, "u8", dict(iterations=10, gamma=1.0)),               # ... convergence-checked: speculative blocks with the channel
    ((64, 88), 13, "f32", dict(gamma=0.5)),
])
def test_tvl1_hip_vs_the_reference_cuda_class(gpu, shape, seed, dtype, kw):
    from opencv_contrib_amd import capi, cuda
    I0, I1, _ = synth.flow_pair(shape[0], shape[1], seed=seed, dtype=dtype)
    if kw.get("gamma"):   # a brightness change between the frames, so that u3 is exercised
        I1 = np.clip(I1.astype(np.float32) * 1.05 + (3 if dtype == "u8" else 0.012), 0, 255 if dtype == "u8" else 1).astype(I1.dtype)
    ref, _ = refcu.cuda_class_tvl1_calc(I0, I1, **kw)
    alg = cuda.OpticalFlowDual_TVL1.create(semantics=capi.MI_SEM_CUDA_COMPAT, **kw)
    flow = N(alg.calc(T(I0, gpu), T(I1, gpu)))
    d = np.sqrt(((flow - ref) ** 2).sum(-1))
    # bounds of tests/test_tvl1_gpu.py::test_calc_cuda_compat_check_schedule (fast-math HIP path against cv::cuda's arithmetic)
    assert np.isfinite(flow).all() and d.mean() <= 5e-3 and (d <= 0.02).mean() >= 0.99, (float(d.mean()), float((d <= 0.02).mean()))
    # convergence-checked with the illumination channel: an iteration count that differs by one at a warp (fast-math error sums decide a
    # borderline check differently) moves more than it does without the channel -- 1e-3, a quarter of what the reference accepts (4e-3)
    assert synth.ccorr_dissimilarity(flow, ref) <= (1e-3 if kw.get("gamma") and kw.get("epsilon", 0.01) > 0 else 1e-4)


@pytest.mark.parametrize("kw_ref,kw_hip", [
    (dict(), dict()),
    (dict(fast_pyramids=1), dict(fastPyramids=True)),
    (dict(flags=256, poly_n=7, poly_sigma=1.5), dict(flags=256, polyN=7, polySigma=1.5)),
])
def test_farneback_hip_vs_the_reference_cuda_class(gpu, kw_ref, kw_hip):
    from opencv_contrib_amd import cuda
    I0, I1, _ = synth.flow_pair(240, 320, seed=31, dtype="u8")
    ref = refcu.cuda_class_farneback_calc(I0, I1, **kw_ref)
    flow = N(cuda.FarnebackOpticalFlow.create(**kw_hip).calc(T(I0, gpu), T(I1, gpu)))
    d = np.sqrt(((flow - ref) ** 2).sum(-1))
    assert np.isfinite(flow).all() and d.mean() <= 2e-3, float(d.mean())       # tests/test_farneback.py::_assert_flow_close
    assert synth.ccorr_dissimilarity(flow, ref) <= 1e-5


@pytest.mark.parametrize("kw", [dict(), dict(prefilter_type=1), dict(ndisp=128, block=15, uniqueness_ratio=10, texture_threshold=0)])
def test_stereobm_hip_vs_the_reference_cuda_class(gpu, kw):
    from opencv_contrib_amd import cuda
    left, right, _ = synth.stereo_pair(120, 300, seed=11, max_disp=40)
    ref = refcu.cuda_class_stereobm_compute(left, right, **kw)
    bm = cuda.createStereoBM(kw.get("ndisp", 64), kw.get("block", 19))
    if "prefilter_type" in kw:
        bm.setPreFilterType(kw["prefilter_type"])
    if "uniqueness_ratio" in kw:
        bm.setUniquenessRatio(kw["uniqueness_ratio"])
    if "texture_threshold" in kw:
        bm.setTextureThreshold(kw["texture_threshold"])
    np.testing.assert_array_equal(N(bm.compute(T(left, gpu), T(right, gpu))), ref)      # integer path: bit-exact


@pytest.mark.parametrize("extended,upright", [(False, False), (True, True)])
def test_surf_hip_vs_the_reference_cuda_class(gpu, extended, upright):
    from opencv_contrib_amd import cuda
    img = np.rint(synth.texture(200, 260, 11, 2.0)).astype(np.uint8)
    ref = refcu.cuda_class_surf(img, hessian_threshold=50.0, extended=extended, upright=upright)
    surf = cuda.SURF_CUDA.create(50.0, 4, 2, extended, 0.01, upright)
    kp, desc = surf.detectWithDescriptors(T(img, gpu))
    kp, desc = N(kp), N(desc)
    assert kp.shape[1] == ref["n"] > 100
    ki = kp.view(np.int32)
    order = np.lexsort((kp[4], kp[0], kp[1], ki[3]))       # (octave, y, x, size): the order refcu sorts the reference's features by
    kp, ki, desc = kp[:, order], ki[:, order], desc[order]
    np.testing.assert_array_equal(ki[2], ref["laplacian"])
    np.testing.assert_array_equal(ki[3], ref["octave"])
    np.testing.assert_array_equal(kp[4], ref["size"])
    for row, name in ((0, "x"), (1, "y"), (6, "hessian")):
        np.testing.assert_allclose(kp[row], ref[name], rtol=1e-6, atol=1e-4, err_msg=name)
    # the bounds of tests/test_surf.py::_compare (device atan2f / sincosf differ from glibc's by an ulp, which can move a sample across
    # a 5-degree window edge or a texel boundary)
    da = np.abs(kp[5] - ref["angle"]); da = np.minimum(da, 360 - da)
    assert (da <= 1e-2).mean() >= 0.99 or (da > 1e-2).sum() <= 2
    dd = np.abs(desc - ref["descriptors"]).max(1)
    ok = (dd <= 1e-4) | (da > 1e-2)
    assert ok.mean() >= 0.99 or (~ok).sum() <= 2, (float(dd.max()), int((~ok).sum()))
