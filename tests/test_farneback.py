"""Farneback: oracle self-tests (CPU) and HIP-vs-oracle parity (GPU) for cv::cuda::FarnebackOpticalFlow.

Tolerances (float path, stated):
  * stage level: same binary32 operations in the same order on both sides (-ffp-contract=off); the only
    freedom is libm exp() in the host-side kernel tables (identical code path: both computed on the host
    with the same formula) -> atol 1e-5 relative to plane magnitude (rtol 2e-6);
  * full calc vs oracle: mean EPE <= 2e-3 px and |1-CCORR| <= 1e-5 -- far inside the reference's own
    CUDA-vs-CPU acceptance |1-CCORR| <= 1e-4 (box) / 2e-2 (Gaussian), cudaoptflow/test/test_optflow.cpp:349.
"""
import os

import numpy as np
import pytest

from opencv_contrib_amd import synth


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_gaussian_kernel_tables_and_normalisation(oracle):
    np.testing.assert_array_equal(oracle.fb_gaussian_kernel(3, 0), np.array([0.25, 0.5, 0.25], np.float32))
    k = oracle.fb_gaussian_kernel(13, 13 // 2 * 0.3)
    assert abs(k.sum() - 1) < 1e-6 and np.allclose(k, k[::-1]) and k.argmax() == 6
    k = oracle.fb_gaussian_kernel(9, 0)      # sigma <= 0, n > 7: sigma = ((n-1)*0.5 - 1)*0.3 + 0.8
    s = ((9 - 1) * 0.5 - 1) * 0.3 + 0.8
    x = np.arange(9) - 4
    ref = np.exp(-x * x / (2 * s * s)); ref /= ref.sum()
    np.testing.assert_allclose(k, ref, rtol=1e-6)


def test_oracle_prepare_gaussian_inverts_moment_matrix(oracle):
    g, xg, xxg, ig = oracle.fb_prepare_gaussian(5, 1.1)
    full = np.concatenate([g[:0:-1], g])
    assert abs(full.sum() - 1) < 1e-6
    # ig11 = 1 / sum(g_y g_x x^2) (the x-moment block of G is diagonal)
    x = np.arange(-5, 6)
    m2 = (full[:, None] * full[None, :] * (x[None, :] ** 2)).sum()
    assert abs(ig[0] - 1 / m2) < 1e-5


def test_oracle_polyexp_recovers_quadratic(oracle):
    """f(x,y) = c + ax + by + (1/2)(r4 x^2... ): the expansion of an exact quadratic returns its coefficients
    in the layout [b_y?]: plane0 = d/dy, plane1 = d/dx, plane2 = yy/..., here checked via a pure linear ramp."""
    h, w = 40, 50
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = (3.0 + 0.5 * x - 0.25 * y).astype(np.float32)
    R = oracle.fb_poly_exp(img, 5, 1.1).reshape(5, h, w)
    c = (slice(8, -8), slice(8, -8))
    np.testing.assert_allclose(R[0][c], -0.25, atol=1e-4)   # b3*ig11: linear term in y
    np.testing.assert_allclose(R[1][c], 0.5, atol=1e-4)     # b2*ig11: linear term in x
    for k in (2, 3, 4):
        np.testing.assert_allclose(R[k][c], 0.0, atol=1e-4)


def test_oracle_pyr_down_constant_and_size(oracle):
    img = np.full((31, 45), 7.0, np.float32)
    d = oracle.fb_pyr_down(img)
    assert d.shape == (16, 23)
    np.testing.assert_allclose(d, 7.0, rtol=1e-6)


@pytest.mark.parametrize("flags", [0, 256])
def test_oracle_recovers_analytic_flow_config1(oracle, flags):
    """BASELINE configs[0]: 640x480 synthetic pair, class defaults."""
    I0, I1, gt = synth.flow_pair(480, 640, seed=1234, dtype="u8")
    f = oracle.fb_calc(I0, I1, oracle.fb_params(flags=flags))
    assert synth.epe(f[40:-40, 40:-40], gt[40:-40, 40:-40]) < 0.08


def test_oracle_initial_flow_and_fast_pyramids(oracle):
    I0, I1, gt = synth.flow_pair(240, 320, seed=5, dtype="u8")
    f = oracle.fb_calc(I0, I1, oracle.fb_params(fast_pyramids=1))
    assert synth.epe(f[30:-30, 30:-30], gt[30:-30, 30:-30]) < 0.12
    f2 = oracle.fb_calc(I0, I1, oracle.fb_params(flags=4, num_iters=3), init_flow=gt)
    assert synth.epe(f2[30:-30, 30:-30], gt[30:-30, 30:-30]) < 0.12


def test_oracle_argument_errors(oracle):
    I = np.zeros((64, 64), np.uint8)
    with pytest.raises(ValueError):
        oracle.fb_calc(I, I, oracle.fb_params(poly_n=6))              # CV_Assert(polyN == 5 || 7)  farneback.cpp:316
    with pytest.raises(ValueError):
        oracle.fb_calc(I, I, oracle.fb_params(fast_pyramids=1, pyr_scale=0.8))   # :317
    with pytest.raises(ValueError):
        oracle.fb_calc(I, I[:, :32])
    with pytest.raises(ValueError):
        oracle.fb_calc(I, I, oracle.fb_params(flags=4))               # initial flow missing


# ------------------------------------------------------------------ HIP vs oracle (GPU)
gpu_mark = pytest.mark.gpu


def _close(a, b, name=""):
    scale = max(float(np.abs(b).max()), 1e-6)
    np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-6 * scale, err_msg=name)


@gpu_mark
@pytest.mark.parametrize("shape", [(48, 70), (97, 531)])
@pytest.mark.parametrize("polyN,sigma", [(5, 1.1), (7, 1.5)])
def test_poly_exp_matches_oracle(gpu, oracle, shape, polyN, sigma):
    from opencv_contrib_amd import cuda
    img = np.random.default_rng(1).random(shape, dtype=np.float32) * 255
    ref = oracle.fb_poly_exp(img, polyN, sigma)
    _close(N(cuda.farneback_polyExp(T(img, gpu), polyN, sigma)), ref)


def _fb_state(h, w, seed, amp=2.0):
    rng = np.random.default_rng(seed)
    I0 = synth.texture(h, w, seed, 2.0).astype(np.float32)
    I1 = synth.texture(h, w, seed + 1, 2.0).astype(np.float32)
    fx = ((rng.random((h, w), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
    fy = ((rng.random((h, w), dtype=np.float32) - 0.5) * 2 * amp).astype(np.float32)
    return I0, I1, fx, fy


@gpu_mark
@pytest.mark.parametrize("shape,amp", [((40, 64), 1.5), ((67, 300), 30.0)])
def test_update_matrices_matches_oracle(gpu, oracle, shape, amp):
    from opencv_contrib_amd import cuda
    I0, I1, fx, fy = _fb_state(*shape, seed=3, amp=amp)
    R0, R1 = oracle.fb_poly_exp(I0), oracle.fb_poly_exp(I1)
    ref = oracle.fb_update_matrices(fx, fy, R0, R1)
    out = cuda.farneback_updateMatrices(T(fx, gpu), T(fy, gpu), T(R0, gpu), T(R1, gpu))
    _close(N(out), ref)


@gpu_mark
@pytest.mark.parametrize("ksize,gauss", [(13, False), (13, True), (5, False), (31, True)])
def test_blur5_update_flow_and_fused_iteration_match_oracle(gpu, oracle, ksize, gauss):
    from opencv_contrib_amd import cuda
    h, w = 61, 277
    I0, I1, fx, fy = _fb_state(h, w, seed=7)
    R0, R1 = oracle.fb_poly_exp(I0), oracle.fb_poly_exp(I1)
    M = oracle.fb_update_matrices(fx, fy, R0, R1)
    sig = (ksize // 2 * np.float32(0.3)) if gauss else None
    Mb = oracle.fb_blur5(M, ksize, float(sig) if gauss else None)
    _close(N(cuda.farneback_blur5(T(M, gpu), ksize, gauss)), Mb, "blur5")
    rfx, rfy = oracle.fb_update_flow(Mb)
    gfx, gfy = cuda.farneback_updateFlow(T(Mb, gpu))
    _close(N(gfx), rfx, "flowx"); _close(N(gfy), rfy, "flowy")
    M2 = oracle.fb_update_matrices(rfx, rfy, R0, R1)
    ifx, ify, iM = cuda.farneback_iterate(T(M, gpu), T(R0, gpu), T(R1, gpu), ksize, gauss, True)
    # the fused kernel divides once per pixel like updateFlow; flows agree to float rounding, M' follows
    np.testing.assert_allclose(N(ifx), rfx, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(N(ify), rfy, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(N(iM), M2, rtol=1e-3, atol=1e-3 * np.abs(M2).max())


@gpu_mark
@pytest.mark.parametrize("border", [1, 4])
@pytest.mark.parametrize("ksize,sigma", [(3, 0.0), (9, 1.5), (39, 7.5)])
def test_gaussian_blur_and_pyr_down_match_oracle(gpu, oracle, border, ksize, sigma):
    from opencv_contrib_amd import cuda
    img = np.random.default_rng(2).random((53, 300), dtype=np.float32) * 255
    _close(N(cuda.farneback_gaussianBlur(T(img, gpu), ksize, sigma, border)), oracle.fb_gaussian_blur(img, ksize, sigma, border))
    if border == 4 and ksize == 3:
        for shp in ((53, 300), (32, 33), (7, 9)):
            im = np.random.default_rng(3).random(shp, dtype=np.float32)
            np.testing.assert_array_equal(N(cuda.pyrDown(T(im, gpu))), oracle.fb_pyr_down(im))


def _assert_flow_close(flow, ref, mean_epe=2e-3, ccorr=1e-5):
    assert np.isfinite(flow).all()
    d = np.sqrt(((flow - ref) ** 2).sum(-1))
    assert d.mean() <= mean_epe, f"mean EPE {d.mean()}"
    assert synth.ccorr_dissimilarity(flow, ref) <= ccorr


@gpu_mark
@pytest.mark.parametrize("pyrScale", [0.3, 0.5, 0.8])
@pytest.mark.parametrize("polyN,sigma", [(5, 1.1), (7, 1.5)])
@pytest.mark.parametrize("flags", [0, 256])
def test_calc_matches_oracle_reference_test_parameters(gpu, oracle, pyrScale, polyN, sigma, flags):
    """Parameter grid of cudaoptflow/test/test_optflow.cpp:307-356 (pyrScale x polyN x flags), RubberWhale-size
    synthetic pair (584x388)."""
    from opencv_contrib_amd import cuda
    I0, I1, _ = synth.flow_pair(388, 584, seed=31, dtype="u8")
    p = oracle.fb_params(pyr_scale=pyrScale, poly_n=polyN, poly_sigma=sigma, flags=flags)
    ref = oracle.fb_calc(I0, I1, p)
    alg = cuda.FarnebackOpticalFlow.create(pyrScale=pyrScale, polyN=polyN, polySigma=sigma, flags=flags)
    flow = N(alg.calc(T(I0, gpu), T(I1, gpu)))
    _assert_flow_close(flow, ref)


@gpu_mark
def test_calc_config1_defaults_640x480(gpu, oracle):
    """BASELINE configs[0] workload on the GPU path vs the oracle, plus accuracy against the analytic flow."""
    from opencv_contrib_amd import cuda
    I0, I1, gt = synth.flow_pair(480, 640, seed=1234, dtype="u8")
    ref = oracle.fb_calc(I0, I1)
    alg = cuda.FarnebackOpticalFlow.create()
    assert alg.getDefaultName() == "DenseOpticalFlow.FarnebackOpticalFlow" and alg.getWinSize() == 13
    flow = N(alg.calc(T(I0, gpu), T(I1, gpu)))
    _assert_flow_close(flow, ref)
    assert synth.epe(flow[40:-40, 40:-40], gt[40:-40, 40:-40]) < 0.08


@gpu_mark
def test_calc_initial_flow_fast_pyramids_f32_and_pitched(gpu, oracle):
    import torch
    from opencv_contrib_amd import cuda
    I0, I1, gt = synth.flow_pair(240, 320, seed=5, dtype="u8")
    # fastPyramids
    ref = oracle.fb_calc(I0, I1, oracle.fb_params(fast_pyramids=1))
    alg = cuda.FarnebackOpticalFlow.create(fastPyramids=True)
    _assert_flow_close(N(alg.calc(T(I0, gpu), T(I1, gpu))), ref)
    # initial flow (flags = USE_INITIAL_FLOW), pitched float frames and pitched flow
    F0, F1 = I0.astype(np.float32), I1.astype(np.float32)
    ref = oracle.fb_calc(F0, F1, oracle.fb_params(flags=4, num_iters=3, num_levels=3), init_flow=gt)
    buf0 = torch.zeros((240, 384), dtype=torch.float32, device=gpu); buf1 = torch.zeros((240, 352), dtype=torch.float32, device=gpu)
    fbuf = torch.zeros((240, 330, 2), dtype=torch.float32, device=gpu)
    buf0[:, 5:325] = T(F0, gpu); buf1[:, 1:321] = T(F1, gpu); fbuf[:, 3:323] = T(gt, gpu)
    alg = cuda.FarnebackOpticalFlow.create(numLevels=3, numIters=3, flags=cuda.OPTFLOW_USE_INITIAL_FLOW)
    out = alg.calc(buf0[:, 5:325], buf1[:, 1:321], fbuf[:, 3:323])
    _assert_flow_close(N(out), ref)


@gpu_mark
def test_calc_argument_errors_and_determinism(gpu):
    import torch
    from opencv_contrib_amd import cuda, capi
    a = torch.zeros((64, 80), dtype=torch.uint8, device=gpu)
    with pytest.raises(capi.MiError):
        cuda.FarnebackOpticalFlow.create(polyN=6).calc(a, a)
    with pytest.raises(capi.MiError):
        cuda.FarnebackOpticalFlow.create(fastPyramids=True, pyrScale=0.8).calc(a, a)
    with pytest.raises(capi.MiError):
        cuda.FarnebackOpticalFlow.create().calc(a, a[:, :40])
    I0, I1, _ = synth.flow_pair(200, 300, seed=9, dtype="u8")
    alg = cuda.FarnebackOpticalFlow.create()
    f1 = N(alg.calc(T(I0, gpu), T(I1, gpu)))
    f2 = N(cuda.FarnebackOpticalFlow.create().calc(T(I0, gpu), T(I1, gpu)))
    np.testing.assert_array_equal(f1, f2)


@gpu_mark
@pytest.mark.parametrize("kw", [dict(), dict(fastPyramids=True), dict(flags=256, winSize=9), dict(flags=4)],
                         ids=["defaults", "fast_pyramids", "gaussian", "initial_flow"])
def test_calc_batch_equals_single_calcs(gpu, oracle, kw):
    """mi_farneback_calc_batch (blockIdx.z = pair in every kernel of the level loop): every pair of a batch of distinct
    pairs equals, bit for bit, the single calc() of that pair -- and with it the oracle -- for the default path, fast pyramids,
    the Gaussian window and a caller-supplied initial flow."""
    import torch
    from opencv_contrib_amd import cuda
    pairs = [synth.flow_pair(240, 320, seed=60 + k, dtype="u8")[:2] for k in range(5)]
    I0s, I1s = [T(p[0], gpu) for p in pairs], [T(p[1], gpu) for p in pairs]
    alg, one = cuda.FarnebackOpticalFlow.create(**kw), cuda.FarnebackOpticalFlow.create(**kw)
    flows = None
    if kw.get("flags", 0) & 4:   # OPTFLOW_USE_INITIAL_FLOW: flows carry the initial guess in, the result out
        rng = np.random.default_rng(3)
        init = [torch.from_numpy((rng.standard_normal((240, 320, 2)) * 0.5).astype(np.float32)).to(gpu) for _ in range(5)]
        flows = torch.stack(init).clone()
        singles = [one.calc(I0s[k], I1s[k], init[k].clone()) for k in range(5)]
    else:
        singles = [one.calc(I0s[k], I1s[k]).clone() for k in range(5)]
    out = alg.calc_batch(I0s, I1s, flows)
    torch.cuda.synchronize()
    for k in range(5):
        assert torch.equal(out[k], singles[k]), f"pair {k}"
    assert not torch.equal(out[0], out[1])
    if not kw.get("flags", 0) & 4:
        p = oracle.fb_params(fast_pyramids=int(kw.get("fastPyramids", False)), flags=kw.get("flags", 0), win_size=kw.get("winSize", 13))
        _assert_flow_close(N(out[3]), oracle.fb_calc(pairs[3][0], pairs[3][1], p))
    # a smaller batch through the same handle afterwards (capacity is kept), and a larger one (arena regrown)
    again = alg.calc_batch(I0s[:2], I1s[:2], None if flows is None else torch.stack(init[:2]).clone())
    assert torch.equal(again[1], singles[1])


@gpu_mark
@pytest.mark.parametrize("kw", [dict(), dict(flags=4, fastPyramids=True)], ids=["defaults", "initial_flow_fast_pyramids"])
def test_large_batch_runs_in_cache_sized_groups_and_equals_single_calcs(gpu, kw):
    """Round 5: a large batch walks the level loop in groups of pairs whose planes fit the last-level cache (MIFLOW_FB_GROUP_MB,
    farneback_api.cpp calc) and converts / merges 64 pairs per launch.  70 pairs of 240 x 320 are three groups with a ragged last one
    and two convert launches with a ragged second one: every pair still equals its single calc() bit for bit."""
    import torch
    from opencv_contrib_amd import cuda
    pairs = [synth.flow_pair(240, 320, seed=80 + k, dtype="u8")[:2] for k in range(5)]
    I0s, I1s = [T(p[0], gpu) for p in pairs], [T(p[1], gpu) for p in pairs]
    alg, one = cuda.FarnebackOpticalFlow.create(**kw), cuda.FarnebackOpticalFlow.create(**kw)
    B, flows = 70, None
    if kw.get("flags", 0) & 4:
        rng = np.random.default_rng(5)
        init = [torch.from_numpy((rng.standard_normal((240, 320, 2)) * 0.5).astype(np.float32)).to(gpu) for _ in range(5)]
        flows = torch.stack([init[k % 5] for k in range(B)]).clone()
        singles = [one.calc(I0s[k], I1s[k], init[k].clone()).clone() for k in range(5)]
    else:
        singles = [one.calc(I0s[k], I1s[k]).clone() for k in range(5)]
    out = alg.calc_batch([I0s[k % 5] for k in range(B)], [I1s[k % 5] for k in range(B)], flows)
    torch.cuda.synchronize()
    for k in range(B):
        assert torch.equal(out[k], singles[k % 5]), f"pair {k}"


def test_call_plan_host_arithmetic(tmp_path):
    """tests/cpp/fb_plan_test.cpp: the plan of a mi_farneback_calc_batch call (csrc/fb_plan.h, round 6 -- the level crop and geometry of
    cudaoptflow/src/farneback.cpp:330-395, how each level's flow starts, fused / plain zoom, pair groups, two iterations per launch) for
    the class defaults, the bench's batch, the initial-flow and zero-iteration cases, fast pyramids, forced switches, and its invariants
    over a sweep of shapes.  The level loop (enqueue_level) only executes a plan.  Plain C++, no device."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fb_plan_test")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(root, "opencv_contrib_amd", "csrc"),
                        os.path.join(root, "tests", "cpp", "fb_plan_test.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "fb_plan_test: ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(winSize=9), dict(winSize=15), dict(winSize=21), dict(winSize=11), dict(winSize=5, numIters=1),
                                dict(flags=256), dict(polyN=7, polySigma=1.5), dict(fastPyramids=True), dict(numLevels=1), dict(pyrScale=0.8, numLevels=8)],
                         ids=["defaults_win13", "win9", "win15", "win21", "win11_generic", "win5_one_iteration", "gaussian", "poly7", "fast_pyramids",
                              "one_level", "scale0.8"])
@pytest.mark.parametrize("shape,B", [((120, 160), 1), ((97, 203), 3), ((480, 640), 1), ((240, 320), 40)])
def test_no_plane_is_read_before_it_is_written(gpu, kw, shape, B):
    """the coarsest level's flow planes are never cleared -- the first matrix update takes the flow as zero and every iterate
    variant (tiled, narrow 64 x 4 tiles, two iterations per launch, the generic window sizes, Gaussian windows, pair groups on two
    streams) must write all w x h pixels of both planes before anything reads them.  A calc whose scratch arena was filled with NaNs
    (miflow_selftest_farneback_poison) must give the bytes of the calc before it: a stale read would turn up as NaNs or as a changed flow."""
    import torch
    from opencv_contrib_amd import capi, cuda
    pairs = [synth.flow_pair(*shape, seed=600 + k, dtype="u8")[:2] for k in range(min(B, 4))]
    I0s = [torch.from_numpy(pairs[k % len(pairs)][0]).to(gpu) for k in range(B)]
    I1s = [torch.from_numpy(pairs[k % len(pairs)][1]).to(gpu) for k in range(B)]
    alg = cuda.FarnebackOpticalFlow.create(**kw)
    run = (lambda: alg.calc(I0s[0], I1s[0]).clone()) if B == 1 else (lambda: alg.calc_batch(I0s, I1s).clone())
    a = run()
    torch.cuda.synchronize()
    capi.check(capi.lib().miflow_selftest_farneback_poison(alg._h, capi.current_stream_ptr()))
    b = run()
    torch.cuda.synchronize()
    assert torch.isfinite(b).all()
    assert torch.equal(a, b)


def test_pair_group_plan_host_arithmetic(tmp_path):
    """tests/cpp/fb_groups_test.cpp: the plan of the batched level loop (csrc/fb_groups.h -- pairs per launch group, one or two chains)
    for the bench's shape, the sizes where no groups are formed, and its invariants over a sweep (groups cover the batch, the groups in
    flight stay inside the cache budget).  Plain C++, no device."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fb_groups_test")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(root, "opencv_contrib_amd", "csrc"),
                        os.path.join(root, "tests", "cpp", "fb_groups_test.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "fb_groups_test: ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_pair_group_budget_never_changes_a_flow(gpu):
    """MIFLOW_FB_GROUP_MB (read once per process, hence the subprocesses) only decides how many pairs a launch of a large level covers
    and on which of the handle's two streams a group's chain runs: whole-batch launches (0), the default and a budget that leaves one
    pair per group give the same bytes for a 12-pair 640 x 480 batch."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for mb in ("0", None, "30"):
        env = dict(os.environ)
        env.pop("MIFLOW_FB_GROUP_MB", None)
        if mb is not None:
            env["MIFLOW_FB_GROUP_MB"] = mb
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "fb_batch.py"), "12", "1", "640", "480"], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        digests[mb] = re.search(r"digest ([0-9a-f]+)", r.stdout).group(1)
    assert len(set(digests.values())) == 1, digests


@pytest.mark.gpu
@pytest.mark.parametrize("mb", ["6", "20"])
def test_random_batches_in_many_small_pair_groups_equal_single_calcs(gpu, mb):
    """tools/fb_stress.py under a pair-group budget of a few MB (every level of every call is cut into many groups on two streams): random
    sizes, 2..40 pairs, 8-bit / float frames, pitched ROI inputs (the pre-blur reads them in place), window sizes, pyramid parameters, the fast
    pyramid -- each batch equals the single calc()s of its pairs bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fb_stress.py"), "8", mb], capture_output=True, text=True,
                       env=dict(os.environ, MIFLOW_FB_GROUP_MB=mb), timeout=900)
    assert r.returncode == 0 and "fb_stress: ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@gpu_mark
@pytest.mark.parametrize("batch", [1, 12], ids=["one_pair", "batch_in_groups"])
def test_calc_captured_into_a_graph_by_the_caller_replays_the_same_flow(gpu, batch):
    """A caller may capture calc() / calc_batch() into a HIP graph (torch.cuda.CUDAGraph here) and replay it: no call of the level loop
    synchronises or allocates once the handle has its arena, and a batch whose levels run in pair groups stays ONE chain while its
    stream is being captured (the second chain lives on a stream of the handle's own, which a capture must not touch).  The replay
    writes the bytes of the plain call."""
    import torch
    from opencv_contrib_amd import cuda
    pairs = [synth.flow_pair(480, 640, seed=300 + k, dtype="u8")[:2] for k in range(min(batch, 3))]
    I0 = [T(pairs[i % len(pairs)][0], gpu) for i in range(batch)]
    I1 = [T(pairs[i % len(pairs)][1], gpu) for i in range(batch)]
    alg = cuda.FarnebackOpticalFlow.create()
    run = (lambda out: alg.calc(I0[0], I1[0], out)) if batch == 1 else (lambda out: alg.calc_batch(I0, I1, out))
    out = run(None)
    torch.cuda.synchronize()
    ref = out.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run(out)                      # the handle's arena and streams exist before the capture
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        run(out)
    for _ in range(2):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


def _random_fb_configs():
    rng = np.random.default_rng(int(os.environ.get("MIFLOW_SWEEP_SEED", "7702")))
    out = []
    for k in range(int(os.environ.get("MIFLOW_SWEEP_N", "24"))):
        out.append(dict(shape=(int(rng.integers(40, 300)), int(rng.integers(40, 420))), seed=int(rng.integers(1, 10 ** 6)),
                        winSize=int((9, 13, 15, 21, 11, 5, 27)[int(rng.integers(7))]),   # 9 / 13 / 15 / 21: tiled kernel; others: one-row kernel
                        flags=int((0, 256)[int(rng.integers(2))]), numLevels=int(rng.integers(1, 6)), numIters=int(rng.integers(1, 7)),
                        pyrScale=float((0.5, 0.7, 0.35)[int(rng.integers(3))]), fastPyramids=bool(rng.integers(2)),
                        batch=int((1, 1, 3)[int(rng.integers(3))])))
    return out


@gpu_mark
@pytest.mark.parametrize("cfg", _random_fb_configs(), ids=lambda c: f"{c['shape'][0]}x{c['shape'][1]}-w{c['winSize']}-f{c['flags']}-l{c['numLevels']}"
                                                                   f"-i{c['numIters']}-b{c['batch']}")
def test_random_configuration_matches_oracle(gpu, oracle, cfg):
    """Seeded sweep over frame sizes that are multiples of nothing (row / column tile edges of the tiled iteration kernel, its
    XCD-contiguous tile order, the mirror-step border indices of the pyramid blur), window sizes of the tiled and of the one-row
    kernel, box and Gaussian windows, pyramid depth and scale, fast pyramids, and batches."""
    from opencv_contrib_amd import cuda
    if cfg["fastPyramids"]:
        cfg = dict(cfg, pyrScale=0.5)   # CV_Assert(!fastPyramids || std::abs(pyrScale - 0.5) < 1e-6), farneback.cpp:317
    kw = dict(numLevels=cfg["numLevels"], pyrScale=cfg["pyrScale"], fastPyramids=cfg["fastPyramids"], winSize=cfg["winSize"],
              numIters=cfg["numIters"], flags=cfg["flags"])
    okw = dict(num_levels=cfg["numLevels"], pyr_scale=cfg["pyrScale"], fast_pyramids=int(cfg["fastPyramids"]), win_size=cfg["winSize"],
               num_iters=cfg["numIters"], flags=cfg["flags"])
    pairs = [synth.flow_pair(*cfg["shape"], seed=cfg["seed"] + b, dtype="u8") for b in range(cfg["batch"])]
    alg = cuda.FarnebackOpticalFlow.create(**kw)
    if cfg["batch"] == 1:
        flows = [alg.calc(T(pairs[0][0], gpu), T(pairs[0][1], gpu))]
    else:
        flows = list(alg.calc_batch([T(p[0], gpu) for p in pairs], [T(p[1], gpu) for p in pairs]))
    for b, p in enumerate(pairs):
        _assert_flow_close(N(flows[b]), oracle.fb_calc(p[0], p[1], oracle.fb_params(**okw)))


@pytest.mark.gpu
def test_few_launch_forms_of_a_single_pair_are_bit_identical(gpu):
    """Round 4: a single pair is a chain of launch latencies, so small calls run forms with fewer / shorter
    launches -- 64 x 4 tiles (MIFLOW_FB_NARROW), the resize sampled inside poly_exp and the first matrix update + the merge written by
    the last iteration (MIFLOW_FB_FUSE), two iterations per launch on the coarse levels (MIFLOW_FB_PAIR).  Every one of them performs the same operations in the same order: the flow
    of a whole calc must not change by a bit.  The switches are read once per process, hence the subprocesses."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for name, env in (("plain", {"MIFLOW_FB_NARROW": "0", "MIFLOW_FB_FUSE": "0", "MIFLOW_FB_PAIR": "0"}), ("default", {}),
                      ("all_on", {"MIFLOW_FB_NARROW": "1", "MIFLOW_FB_FUSE": "1", "MIFLOW_FB_PAIR": "1"})):
        for (w, h) in ((640, 480), (333, 217)):
            r = subprocess.run([sys.executable, os.path.join(root, "tools", "fb_single.py"), str(w), str(h), "3"], capture_output=True, text=True,
                               env=dict(os.environ, **env), timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            digests[(name, w)] = [l.split("digest")[1].strip() for l in r.stdout.splitlines() if "digest" in l][0]
    for w in (640, 333):
        assert digests[("plain", w)] == digests[("default", w)] == digests[("all_on", w)], digests
