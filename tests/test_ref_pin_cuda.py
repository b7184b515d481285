"""PINNING, part 2: the restated oracles against the reference's own CUDA kernel SOURCES, executed here on the CPU.

oracle/_ref/libref_cu.so = cudastereo/src/cuda/stereobm.cu, cudaoptflow/src/cuda/farneback.cu, cudaoptflow/src/cuda/tvl1flow.cu and
cudastereo/src/cuda/disparity_bilateral_filter.cu compiled for the host (oracle/Makefile.ref): cu2host.py rewrites only the
`k<<<...>>>(...)` launch sites and the `extern __shared__` declarations; oracle/refshim/cudashim supplies the execution model
(thread blocks as cooperatively scheduled contexts, __syncthreads = yield, __shared__ = static storage) and stand-ins for the
few main-repo device headers (PtrStepSz, border index maps, numeric_limits, the texture fetch rules of the CUDA programming guide).
Every line of kernel arithmetic that runs is the reference's.

Asserted: oracle/stereobm_ref.c, oracle/farneback_ref.c, the CUDA_COMPAT half of oracle/tvl1_ref.c and oracle/dbf_ref.c equal those
kernels BIT FOR BIT on the same seeded inputs (the one exception, a data race inside the reference's bilateral filter, is
isolated below).
"""
import numpy as np
import pytest

from opencv_contrib_amd import synth
from oracle import refcu, refocl

pytestmark = pytest.mark.skipif(not (refcu.available() or refocl.can_build()), reason="oracle/_ref not built and /root/reference absent")


# ------------------------------------------------------------------------------------------------------------ StereoBM
@pytest.mark.parametrize("nd,ws,uq", [(64, 15, 0), (32, 11, 0), (64, 9, 10), (128, 19, 0), (16, 5, 25), (64, 21, 3)])
@pytest.mark.parametrize("kind", ["ramp", "noise"])
def test_stereobm_block_match_equals_reference_kernel(oracle, nd, ws, uq, kind):
    """stereoKernel<RADIUS> (SSD over the window, winner-take-all in batches of 8 disparities with its tie rules, the
    uniqueness-ratio state machine, the 128-column block edge) on textured pairs and on pure noise (ties everywhere)."""
    if kind == "ramp":
        left, right, _ = synth.stereo_pair(96, 300, seed=42 + nd, max_disp=min(30, nd - 2))
    else:
        rng = np.random.default_rng(nd * 31 + ws)
        left = rng.integers(0, 4, size=(60, 300)).astype(np.uint8)      # 2-bit noise: many equal SSDs
        right = rng.integers(0, 4, size=(60, 300)).astype(np.uint8)
    ref, ssd = refcu.sbm_block_match(left, right, nd, ws, uq)
    got = oracle.sbm_block_match(left, right, ndisp=nd, winsz=ws, uniqueness_ratio=uq, emulate_edge=True)
    np.testing.assert_array_equal(got, ref)
    assert (ref > 0).any()


@pytest.mark.parametrize("cap", [31, 15, 63])
def test_stereobm_prefilters_equal_reference_kernels(oracle, cap):
    img = np.rint(synth.texture(90, 130, 7, 1.5)).astype(np.uint8)
    np.testing.assert_array_equal(oracle.sbm_prefilter_xsobel(img, cap), refcu.sbm_prefilter_xsobel(img, cap))
    for win in (9, 5, 15):
        np.testing.assert_array_equal(oracle.sbm_prefilter_norm(img, cap, win), refcu.sbm_prefilter_norm(img, cap, win))


@pytest.mark.parametrize("winsz,thr", [(15, 3.0), (9, 10.0), (19, 1.0)])
def test_stereobm_textureness_equals_reference_kernel(oracle, winsz, thr):
    """postfilter_textureness on an image with flat regions (they must be zeroed) and textured ones.  The kernel reads the image
    through a linearly filtered texture at integer coordinates, i.e. the mean of a 2 x 2 texel block (CUDA programming guide,
    texture fetching; 1.8 fixed-point weights are exact at 0.5): the oracle's exact-integer quarter-weight definition."""
    img = np.rint(synth.texture(100, 260, 5, 1.5)).astype(np.uint8)
    img[20:60, 40:130] = 117                     # flat patch
    img[70:95, 150:250] = (img[70:95, 150:250] // 64) * 64   # weak texture
    disp = np.random.default_rng(2).integers(1, 64, size=img.shape).astype(np.uint8)
    ref = refcu.sbm_textureness(img, disp, winsz, thr)
    got = oracle.sbm_textureness(img, disp, winsz, thr)
    np.testing.assert_array_equal(got, ref)
    z = (ref == 0) & (disp != 0)
    assert 0.02 < z.mean() < 0.9, z.mean()


# -------------------------------------------------------------- StereoBM: the reference's HOST class over its kernels, end to end
@pytest.mark.parametrize("kw", [
    dict(),                                                                        # createStereoBM(64, 19): no prefilter, texture threshold 3
    dict(prefilter_type=1),                                                        # PREFILTER_XSOBEL, cap 31
    dict(prefilter_type=0, prefilter_size=7, prefilter_cap=20),                    # PREFILTER_NORMALIZED_RESPONSE
    dict(ndisp=128, block=11, texture_threshold=0, uniqueness_ratio=10),           # no textureness pass; the uniqueness rule
    dict(ndisp=32, block=7, texture_threshold=10),
])
def test_stereobm_oracle_equals_the_reference_cuda_host_class(oracle, kw):
    """VERDICT r02 "next" #6: `StereoBMImpl::compute` (modules/cudastereo/src/stereobm.cpp:139-191) compiled VERBATIM against the
    reference's own public header (cudastereo.hpp) and the stub core of oracle/refshim/cudahost, driving the reference's own kernels
    (stereobm.cu): the prefilter choice, the buffer reuse (ensureSizeIsEnough; the entry computes twice), the block matching and the
    textureness pass are reference code end to end.  oracle.sbm_compute must give the same disparity map bit for bit; the class's own
    CV_Asserts reject what the oracle rejects."""
    left, right, _ = synth.stereo_pair(96, 260, seed=11, max_disp=30)
    ref = refcu.cuda_class_stereobm_compute(left, right, **kw)
    names = {"ndisp": "num_disparities", "block": "block_size"}
    got = oracle.sbm_compute(left, right, oracle.sbm_params(**{names.get(k, k): v for k, v in kw.items()}))
    np.testing.assert_array_equal(got, ref)
    assert int((ref > 0).sum()) > ref.size // 4
    if not kw:
        for bad in (dict(ndisp=60), dict(block=10), dict(ndisp=0), dict(ndisp=264)):      # stereobm.cpp:143-146
            with pytest.raises(ValueError):
                refcu.cuda_class_stereobm_compute(left, right, **bad)
            with pytest.raises(ValueError):
                oracle.sbm_compute(left, right, oracle.sbm_params(**{names.get(k, k): v for k, v in bad.items()}))


# ----------------------------------------------------------------------------------------------------------- Farneback
@pytest.fixture(scope="module")
def fb_inputs(oracle):
    I0, I1, _ = synth.flow_pair(120, 300, seed=5, dtype="u8")
    a, b = I0.astype(np.float32), I1.astype(np.float32)
    rng = np.random.default_rng(0)
    fx = (rng.standard_normal(a.shape) * 2).astype(np.float32)
    fy = (rng.standard_normal(a.shape) * 2).astype(np.float32)
    fx[:, :3] = -9.0; fy[-2:, :] = 7.5          # flows leaving the image: the border-scale table of updateMatrices
    return a, b, fx, fy


@pytest.mark.parametrize("n,sigma", [(5, 1.1), (7, 1.5), (5, 0.8)])
def test_farneback_poly_exp_equals_reference_kernel(oracle, fb_inputs, n, sigma):
    a = fb_inputs[0]
    g, xg, xxg, ig = oracle.fb_prepare_gaussian(n, sigma)
    np.testing.assert_array_equal(oracle.fb_poly_exp(a, n, sigma), refcu.fb_poly_exp(a, n, g, xg, xxg, ig))


def test_farneback_update_matrices_and_flow_equal_reference_kernels(oracle, fb_inputs):
    a, b, fx, fy = fb_inputs
    R0, R1 = oracle.fb_poly_exp(a), oracle.fb_poly_exp(b)
    M = oracle.fb_update_matrices(fx, fy, R0, R1)
    np.testing.assert_array_equal(M, refcu.fb_update_matrices(fx, fy, R0, R1))
    B = oracle.fb_blur5(M, 13)
    ox, oy = oracle.fb_update_flow(B)
    rx, ry = refcu.fb_update_flow(B)
    np.testing.assert_array_equal(ox, rx)
    np.testing.assert_array_equal(oy, ry)


@pytest.mark.parametrize("ksize", [5, 13, 21, 33])
def test_farneback_window_filters_equal_reference_kernels(oracle, fb_inputs, ksize):
    """boxFilter5 and gaussianBlur5<BrdReplicate> on the 5-plane matrix stack (the two window types of updateFlow)."""
    a, b, fx, fy = fb_inputs
    M = oracle.fb_update_matrices(fx, fy, oracle.fb_poly_exp(a), oracle.fb_poly_exp(b))
    np.testing.assert_array_equal(oracle.fb_blur5(M, ksize), refcu.fb_box5(M, ksize))
    sigma = ksize // 2 * 0.3
    half = oracle.fb_gaussian_kernel(ksize, sigma)[ksize // 2:]
    np.testing.assert_array_equal(oracle.fb_blur5(M, ksize, sigma), refcu.fb_gaussian_blur5(M, half, 1))


@pytest.mark.parametrize("ksize,sigma,border", [(5, 1.0, 4), (9, 2.0, 4), (7, 1.5, 1), (3, 0.5, 4), (41, 7.5, 4)])
def test_farneback_pyramid_blur_equals_reference_kernel(oracle, fb_inputs, ksize, sigma, border):
    a = fb_inputs[0]
    half = oracle.fb_gaussian_kernel(ksize, sigma)[ksize // 2:]
    np.testing.assert_array_equal(oracle.fb_gaussian_blur(a, ksize, sigma, border), refcu.fb_gaussian_blur(a, half, border))


# ------------------------------------------------------------------------------------------- TV-L1: the cv::cuda kernels
@pytest.mark.parametrize("h,w,seed", [(77, 101, 3), (64, 64, 11), (33, 250, 7)])
@pytest.mark.parametrize("gamma", [0.0, 1.0])
def test_tvl1_cuda_kernels_equal_oracle(oracle, h, w, seed, gamma):
    """centeredGradientKernel, warpBackwardKernel (point-sampled clamp-addressed textures), estimateUKernel (with the error
    plane) and estimateDualVariablesKernel of cudaoptflow/src/cuda/tvl1flow.cu -- the kernels of the class being replaced --
    against the CUDA_COMPAT oracle, including the illumination channel (gamma != 0: u3, p31, p32)."""
    I0, I1, _ = synth.flow_pair(h, w, seed=seed)
    I0, I1 = (I0 * np.float32(255)).astype(np.float32), (I1 * np.float32(255)).astype(np.float32)
    rx, ry = refcu.tvl1_centered_gradient(I1)
    ox, oy = oracle.tvl1_centered_gradient(I1)
    np.testing.assert_array_equal(ox, rx); np.testing.assert_array_equal(oy, ry)
    rng = np.random.default_rng(seed)
    for amp in (0.0, 2.5, 300.0):
        u1 = (rng.standard_normal((h, w)) * amp).astype(np.float32); u2 = (rng.standard_normal((h, w)) * amp).astype(np.float32)
        if amp == 0.0:
            u1[::3, ::5] = 1.0; u2[1::4, ::2] = -2.0
        r = refcu.tvl1_warp(I0, I1, rx, ry, u1, u2)
        o = oracle.tvl1_warp(1, I0, I1, ox, oy, u1, u2)
        for name, a, b in zip(("I1w", "I1wx", "I1wy", "grad", "rho_c"), o, r):
            np.testing.assert_array_equal(a, b, err_msg=f"{name} amp {amp}")
    p = [(rng.standard_normal((h, w)) * 0.4).astype(np.float32) for _ in range(4)]
    g3 = [(rng.standard_normal((h, w)) * 0.1).astype(np.float32) for _ in range(3)] if gamma else [None] * 3
    l_t, theta, taut = np.float32(0.045), np.float32(0.3), np.float32(0.25 / 0.3)
    u1 = (rng.standard_normal((h, w)) * 2).astype(np.float32); u2 = (rng.standard_normal((h, w)) * 2).astype(np.float32)
    _, wx, wy, gr, rc = refcu.tvl1_warp(I0, I1, rx, ry, u1, u2)
    gr[::7, ::3] = 0
    r = refcu.tvl1_iteration(wx, wy, gr, rc, u1, u2, *p, l_t, theta, taut, gamma, *g3)
    o = oracle.tvl1_iteration(1, wx, wy, gr, rc, u1, u2, *p, l_t, theta, taut, gamma, *g3)
    for k in range(1, len(o)):
        np.testing.assert_array_equal(o[k], r[k], err_msg=f"plane {k}")
    assert np.float32(r[0].astype(np.float64).sum()) == np.float32(o[0])   # cuda::calcSum: float terms, double accumulator


# -------------------------------------------------------------- TV-L1: the reference's HOST class over its kernels, end to end
@pytest.mark.parametrize("h,w,seed,dtype,kw", [
    (120, 160, 3, "f32", dict(iterations=10, epsilon=0.0)),                 # the setting of the headline metric
    (64, 88, 5, "u8", dict()),                                             # class defaults: 300 iterations, epsilon 0.01, cv::cuda's sparse check schedule
    (60, 47, 9, "f32", dict(iterations=40, epsilon=0.02)),
    (30, 40, 2, "f32", dict(iterations=7, epsilon=0.0, gamma=0.5)),         # illumination channel; the 16-px rule shrinks nscales to 3
    (50, 70, 4, "u8", dict(iterations=12, epsilon=0.0, tau=0.2, lambda_=0.1, theta=0.25, nscales=4, warps=3, scale_step=0.7)),
    (41, 33, 8, "f32", dict(iterations=25, epsilon=0.05, nscales=2, warps=2)),
])
def test_tvl1_oracle_equals_the_reference_cuda_host_class(oracle, h, w, seed, dtype, kw):
    """VERDICT r02 missing #4: `OpticalFlowDual_TVL1_Impl::calc / calcImpl / procOneScale` (modules/cudaoptflow/src/tvl1flow.cpp:170-382)
    were restated only.  oracle/_ref/libref_cu.so now holds that file compiled VERBATIM against the reference's own public header
    and a stub core (oracle/refshim/cudahost), driving the reference's own kernels (tvl1flow.cu, resize.cu): the pyramid and its
    16-px rule, the per-warp loop with the sparse convergence schedule (error summed at odd iterations while prevError < threshold),
    the flow upsampling and its 1 / scaleStep multiplies are reference code end to end.  oracle.tvl1_calc(semantics = CUDA_COMPAT)
    must give the same flow bit for bit, and stop the pyramid at the same scale."""
    I0, I1, _ = synth.flow_pair(h, w, seed=seed, dtype=dtype)
    ref, ns = refcu.cuda_class_tvl1_calc(I0, I1, **kw)
    okw = dict(kw)
    okw.setdefault("iterations", 300)
    got, st = oracle.tvl1_calc(I0, I1, oracle.tvl1_params(semantics=1, **okw), return_stats=True)
    assert st["nscales"] == ns
    np.testing.assert_array_equal(got, ref)
    assert np.isfinite(ref).all() and float(np.abs(ref).max()) > 0.1


# ----------------------------------------------------------- Farneback: the reference's HOST class over its kernels, end to end
@pytest.mark.parametrize("h,w,seed,dtype,kw", [
    (120, 160, 3, "u8", dict()),                                                       # class defaults: 5 levels, 0.5, box window 13, 10 iterations
    (120, 160, 3, "f32", dict(fast_pyramids=1)),                                       # the cuda::pyrDown pyramid
    (96, 130, 5, "u8", dict(flags=256)),                                               # OPTFLOW_FARNEBACK_GAUSSIAN
    (90, 75, 7, "f32", dict(poly_n=7, poly_sigma=1.5, win_size=9, num_iters=3, num_levels=3, pyr_scale=0.7)),
    (70, 100, 9, "u8", dict(num_levels=6, pyr_scale=0.6, win_size=21, num_iters=2)),   # MIN_SIZE 32 crops the pyramid
    (64, 64, 2, "f32", dict(num_levels=1, poly_sigma=0.0)),                            # poly_sigma < eps -> n * 0.3
    (80, 112, 4, "u8", dict(flags=4, num_levels=3)),                                   # OPTFLOW_USE_INITIAL_FLOW
    (80, 112, 4, "f32", dict(flags=4 | 256, fast_pyramids=1, num_levels=2, win_size=7)),
])
def test_farneback_oracle_equals_the_reference_cuda_host_class(oracle, h, w, seed, dtype, kw):
    """VERDICT r02 "next" #6: `FarnebackOpticalFlowImpl::calc / calcImpl / prepareGaussian / updateFlow_*`
    (modules/cudaoptflow/src/farneback.cpp:170-480) compiled VERBATIM against the reference's own public header and the stub core of
    oracle/refshim/cudahost, driving the reference's own kernels (farneback.cu, resize.cu, pyr_down.cu): the level cropping (MIN_SIZE
    32), the per-level Gaussian blur + cuda::resize or the pyrDown pyramid, the flow upsampling (resize, then convertTo by 1 / pyrScale),
    the normal-matrix inverse of prepareGaussian and the iteration loop are reference code end to end.  oracle.fb_calc must give the same
    flow bit for bit."""
    I0, I1, gt = synth.flow_pair(h, w, seed=seed, dtype="u8")
    if dtype == "f32":
        I0, I1 = I0.astype(np.float32), I1.astype(np.float32)     # 0..255: Farneback's determinant regulariser assumes 8-bit range
    init = (0.5 * gt).astype(np.float32) if kw.get("flags", 0) & 4 else None
    ref = refcu.cuda_class_farneback_calc(I0, I1, init_flow=init, **kw)
    got = oracle.fb_calc(I0, I1, oracle.fb_params(**kw), init_flow=init)
    np.testing.assert_array_equal(got, ref)
    assert np.isfinite(ref).all() and float(np.abs(ref).max()) > 0.1


# ------------------------------------------------------------------- SURF: the reference's HOST class over surf.cu, end to end
def _surf_tables_of_the_reference():
    """c_aptX / c_aptY / c_aptW / c_DW parsed out of modules/xfeatures2d/src/cuda/surf.cu (None when /root/reference is absent)."""
    import os
    import re
    path = "/root/reference/modules/xfeatures2d/src/cuda/surf.cu"
    if not os.path.exists(path):
        return None
    src = open(path).read()
    out = []
    for name in ("c_aptX", "c_aptY", "c_aptW", "c_DW"):
        body = re.search(name + r"\s*\[[^\]]*\]\s*=\s*\{(.*?)\};", src, re.S).group(1)
        out.append(np.array([np.float32(float(v.rstrip("f"))) for v in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?f?", body)], np.float32))
    return out


def test_surf_weight_tables_equal_the_literals_of_surf_cu(oracle):
    """surf.cu:520-522,685-707 hold the orientation sample offsets / weights and the 20 x 20 descriptor weights as LITERALS.  The
    oracle (and csrc/surf_api.cpp, same code) GENERATES them -- outer products in float of the Gaussian kernel as OpenCV 2.4 rounded it
    (exp -> float, sum in double, float * (1 / sum) -> float) for sigma = 2.5f / 3.3f -- and must reproduce every literal bit for bit.
    (Today's single-rounding getGaussianKernel, which the CPU class calls at run time, differs in 60 / 113 and 188 / 400 entries.)"""
    ref = _surf_tables_of_the_reference()
    if ref is None:
        pytest.skip("/root/reference absent")
    for r, mine, n in zip(ref, oracle.surf_tables(), (113, 113, 113, 400)):
        assert r.shape == (n,)
        np.testing.assert_array_equal(mine, r)


def _sorted_oracle(o):
    order = np.lexsort((o["size"], o["x"], o["y"], o["octave"]))
    return {k: (v[order] if isinstance(v, np.ndarray) else v) for k, v in o.items()}


@pytest.mark.parametrize("shape,seed,kw", [
    ((240, 320), 7, dict(hessian_threshold=100.0)),                                              # SURF_CUDA::create defaults
    ((240, 320), 7, dict(hessian_threshold=400.0, extended=True)),                               # 128-float descriptors
    ((200, 260), 11, dict(hessian_threshold=50.0, mask=True)),                                   # Mask<true>: maskSum through cuda::min / integral
    ((200, 260), 11, dict(hessian_threshold=50.0, upright=True)),
    ((200, 260), 11, dict(hessian_threshold=20.0, n_octaves=3, n_octave_layers=4, keypoints_ratio=0.05)),
    ((150, 170), 3, dict(hessian_threshold=300.0, n_octaves=2, n_octave_layers=1)),
])
def test_surf_oracle_equals_the_reference_cuda_host_class(oracle, shape, seed, kw):
    """VERDICT r02 "next" #6: `SURF_CUDA_Invoker` and the SURF_CUDA operators (modules/xfeatures2d/src/surf.cuda.cpp:134-452) compiled
    VERBATIM against the reference's own header (xfeatures2d/cuda.hpp) and the stub core, driving xfeatures2d/src/cuda/surf.cu ITSELF on
    the fiber shim (compiled as for sm_30+: double Haar sums, shfl_down reductions restated in refshim/cudashim/.../reduce.hpp).
    Everything must agree BIT FOR BIT as a set (the class appends through atomicInc): x, y, laplacian, octave, size, hessian, the
    orientation and all 64 / 128 descriptor floats.  This pin is what moved the oracle (and the HIP kernels) from the OpenCL twin's
    patch sampling to the CUDA class's: floor-addressed texture reads, patch samples rounded to 8 bits by saturate_cast<uchar>,
    AreaFilter's edge normalisation, the warp-then-partials order of normalize_descriptors, the literal weight tables."""
    kw = dict(kw)
    img = np.rint(synth.texture(shape[0], shape[1], seed, 1.5 if seed == 7 else 2.0)).astype(np.uint8)
    mask = None
    if kw.pop("mask", False):
        mask = np.zeros_like(img)
        mask[30:170, 40:220] = 255
        mask[80:100, 100:140] = 0
    ref = refcu.cuda_class_surf(img, mask=mask, **kw)
    o = _sorted_oracle(oracle.surf_detect_describe(img, oracle.surf_params(**{k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}), mask=mask))
    assert ref["n"] == o["n"] and ref["n"] > 100
    for k in ("x", "y", "laplacian", "octave", "size", "angle", "hessian", "descriptors"):
        np.testing.assert_array_equal(o[k], ref[k], err_msg=k)


@pytest.mark.parametrize("extended,upright", [(False, False), (True, False), (False, True)])
def test_surf_oracle_equals_the_reference_cuda_host_class_on_provided_keypoints(oracle, extended, upright):
    """operator()(img, mask, keypoints, descriptors, useProvidedKeypoints = true) (surf.cuda.cpp:380-397, uploadKeypoints /
    downloadKeypoints): orientation (unless upright) and descriptors of the caller's keypoints -- sizes down to 4 px put s <= 1, the
    LinearFilter branch of calc_dx_dy (surf.cu:775-779) no detected keypoint reaches; positions next to the border exercise the
    texture clamp."""
    img = np.rint(synth.texture(200, 260, 11, 2.0)).astype(np.uint8)
    rng = np.random.default_rng(5)
    n = 48
    prov = {"x": rng.uniform(1, 259, n).astype(np.float32), "y": rng.uniform(1, 199, n).astype(np.float32), "octave": np.zeros(n, np.int32),
            "size": rng.choice([4.0, 6.0, 7.5, 9.0, 15.0, 30.0, 60.0], n).astype(np.float32), "angle": rng.uniform(0, 360, n).astype(np.float32)}
    ref = refcu.cuda_class_surf(img, extended=extended, upright=upright, provided=prov)
    assert ref["n"] == n
    np.testing.assert_array_equal(ref["x"], prov["x"])
    sum_ = oracle.surf_integral(img)
    ax, ay, aw, dw = oracle.surf_tables()
    L = oracle.lib()
    dsz = 128 if extended else 64
    for i in range(n):
        ang = prov["angle"][i] if upright else L.orc_surf_orientation(sum_, 200, 260, prov["x"][i], prov["y"][i], prov["size"][i], ax, ay, aw)
        assert np.float32(ang) == ref["angle"][i], i
        d = np.empty(dsz, np.float32)
        L.orc_surf_descriptor(img, 200, 260, prov["x"][i], prov["y"][i], prov["size"][i], np.float32(ang), int(extended), dw, d)
        np.testing.assert_array_equal(d, ref["descriptors"][i], err_msg=f"keypoint {i} size {prov['size'][i]}")


# ------------------------------------------------------------------------------------------ cuda::resize / cuda::pyrDown
@pytest.mark.parametrize("shape,dsize", [((1080, 1920), (1536, 864)), ((864, 1536), (1229, 691)), ((97, 131), (105, 78)),
                                          ((60, 80), (160, 120)), ((33, 47), (47, 33)), ((240, 320), (160, 120))])
def test_cuda_resize_linear_equals_reference_kernel(oracle, shape, dsize):
    """resize_linear<float> (cudawarping/src/cuda/resize.cu:234-269): the pyramid of cv::cuda::OpticalFlowDual_TVL1 (scale 0.8 per
    level, the first two 1080p levels included), its flow upsampling, Farneback's level resize.  No half-pixel centres, the source
    coordinate is dst * (1 / scale) in float, the right / bottom neighbour is clamped."""
    rng = np.random.default_rng(shape[0] + dsize[0])
    src = (rng.random(shape) * 255).astype(np.float32)
    np.testing.assert_array_equal(oracle.resize_linear_cuda(src, dsize=dsize), refcu.resize_linear(src, dsize))


@pytest.mark.parametrize("shape", [(480, 640), (97, 131), (66, 258), (5, 7), (301, 517)])
def test_cuda_pyr_down_equals_reference_kernel(oracle, shape):
    """pyrDown<T, BrdReflect101> (cudawarping/src/cuda/pyr_down.cu:54-175), float (Farneback's fastPyramids) and 8-bit with
    round-half-even saturation (the pyramid of SparsePyrLKOpticalFlow); widths that are not multiples of the 256-column block and
    odd sizes exercise the border branch and the dst_cols guard."""
    rng = np.random.default_rng(shape[1])
    f = (rng.random(shape) * 255).astype(np.float32)
    np.testing.assert_array_equal(oracle.fb_pyr_down(f), refcu.pyr_down(f))
    u = rng.integers(0, 256, size=shape).astype(np.uint8)
    np.testing.assert_array_equal(oracle.pyr_down_u8(u), refcu.pyr_down(u))


# --------------------------------------------------------------------------------------------- DisparityBilateralFilter
@pytest.mark.parametrize("radius,iters", [(3, 1), (3, 2), (5, 1)])
@pytest.mark.parametrize("dtype,bgr", [(np.uint8, False), (np.int16, True)])
def test_disparity_bilateral_filter_equals_reference_kernel_where_it_is_deterministic(oracle, radius, iters, dtype, bgr):
    """The reference kernel refines a pixel IN PLACE while neighbouring threads of the same red/black pass read the window it
    lies in (disparity_bilateral_filter.cu:118-131): a data race whenever two pixels of one pass within a window both change.
    The oracle (and the HIP kernel) read the pass's input snapshot.  On inputs where refined pixels are isolated -- spikes on
    plateaus -- the two definitions coincide and the whole arithmetic (weights, truncated costs, the ordered minimum search)
    is compared bit for bit."""
    rng = np.random.default_rng(radius * 10 + iters)
    h, w = 90, 150
    disp = np.full((h, w), 20, dtype)
    ys, xs = np.meshgrid(np.arange(radius + 3, h - radius - 3, 2 * radius + 3), np.arange(radius + 3, w - radius - 3, 2 * radius + 3), indexing="ij")
    disp[ys, xs] = rng.integers(30, 60, size=ys.shape).astype(dtype)
    scale = 16 if dtype == np.int16 else 1
    disp = (disp * scale).astype(dtype)
    g = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    img = np.stack([g, 255 - g, g // 2], -1) if bgr else g
    ref = refcu.dbf_apply(disp, img, 64 * scale, radius, iters)
    got = oracle.dbf_apply(disp, img, oracle.dbf_params(ndisp=64 * scale, radius=radius, iters=iters))
    np.testing.assert_array_equal(got, ref)
    assert (ref != disp).sum() >= ys.size // 2


@pytest.mark.parametrize("radius,iters,dtype,bgr,thr", [(3, 1, np.uint8, False, None), (2, 3, np.int16, True, None), (5, 2, np.uint8, True, (0.25, 0.4, 4.0)),
                                                        (1, 1, np.int16, False, (0.05, 0.1, 25.0))])
def test_disparity_bilateral_filter_oracle_equals_the_reference_cuda_host_class(oracle, radius, iters, dtype, bgr, thr):
    """`DispBilateralFilterImpl` (modules/cudastereo/src/disparity_bilateral_filter.cpp:58-191, compiled verbatim against the reference's
    own cudastereo.hpp and the stub core): the colour / space weight tables (exp in double -> float, exp(-sqrt(float) / dist) in float),
    edge_disc = max(1, short(ndisp * edge_threshold + 0.5)), max_disc, the in-place copy and the type dispatch are reference code; the
    kernel is the reference's (inputs with isolated refined pixels, where its in-place race does not show).  oracle.dbf_apply must
    give the same map bit for bit, also with non-default thresholds and sigma."""
    rng = np.random.default_rng(radius * 10 + iters)
    h, w = 90, 150
    disp = np.full((h, w), 20, dtype)
    ys, xs = np.meshgrid(np.arange(radius + 3, h - radius - 3, 2 * radius + 3), np.arange(radius + 3, w - radius - 3, 2 * radius + 3), indexing="ij")
    disp[ys, xs] = rng.integers(30, 60, size=ys.shape).astype(dtype)
    scale = 16 if dtype == np.int16 else 1
    disp = (disp * scale).astype(dtype)
    g = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    img = np.stack([g, 255 - g, g // 2], -1) if bgr else g
    kw = {} if thr is None else dict(edge_threshold=thr[0], max_disc_threshold=thr[1], sigma_range=thr[2])
    ref = refcu.cuda_class_dbf_apply(disp, img, 64 * scale, radius, iters, **kw)
    got = oracle.dbf_apply(disp, img, oracle.dbf_params(ndisp=64 * scale, radius=radius, iters=iters, **kw))
    np.testing.assert_array_equal(got, ref)
    assert (ref != disp).sum() >= ys.size // 4
    for bad in (dict(ndisp=0), dict(radius=0), dict(iters=0)):     # disparity_bilateral_filter.cpp:179
        with pytest.raises(ValueError):
            refcu.cuda_class_dbf_apply(disp, img, **{**dict(ndisp=64, radius=3, iters=1), **bad})


def test_disparity_bilateral_filter_on_a_real_map_differs_only_by_the_reference_race(oracle):
    """On a block-matching disparity map (edges: neighbouring refined pixels) the sequential execution of the reference kernel
    realises ONE outcome of its race; the snapshot definition agrees with it on all but a fraction of a percent of the pixels."""
    left, right, _ = synth.stereo_pair(96, 224, seed=42, max_disp=30)
    d = oracle.sbm_compute(left, right, oracle.sbm_params(num_disparities=64, block_size=15))
    ref = refcu.dbf_apply(d, left, 64, 3, 1)
    got = oracle.dbf_apply(d, left, oracle.dbf_params(ndisp=64, radius=3, iters=1))
    assert (ref != d).mean() > 0.02
    assert (ref != got).mean() < 0.02
