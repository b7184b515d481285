"""PINNING: the restated oracles against REFERENCE CODE EXECUTED HERE.

oracle/_ref/libref_ocl.so is the reference's own OpenCL kernel source -- modules/optflow/src/opencl/optical_flow_tvl1.cl:46-378
and modules/xfeatures2d/src/opencl/surf.cl -- compiled verbatim for x86-64 by clang's OpenCL C front end (oracle/Makefile.ref)
and run on the CPU through the NDRange shim of oracle/refshim/.  The TV-L1 kernels are the OpenCL twins of
cudaoptflow/src/cuda/tvl1flow.cu:59-348, i.e. the arithmetic the `MI_SEM_CUDA_COMPAT` oracle restates; the SURF kernels are the
twins of xfeatures2d/src/cuda/surf.cu:122-942, which oracle/surf_ref.c restates (second half of this file).

What is asserted, on the same seeded inputs:
  * every stage of oracle/tvl1_ref.c (semantics CUDA_COMPAT) equals the reference kernel BIT FOR BIT when both are
    compiled with separately rounded operations (-ffp-contract=off);
  * the whole per-scale loop (gradient, warps x (warp + estimateU / estimateDualVariables with the sparse convergence
    check of procOneScale_ocl, optflow/src/tvl1flow.cpp:1224-1310 == cudaoptflow/src/tvl1flow.cpp:304-382)) yields the
    same flow bit for bit and the same iteration counts;
  * the freedom a device compiler has (OpenCL C's default FP_CONTRACT ON, nvcc -fmad=true: a*b+c may be fused) is
    bounded: a second build of the same reference source with contraction allowed stays within a stated distance.
The `-m gpu` leg holds the HIP kernels to the reference kernels directly.
"""
import numpy as np
import pytest

from opencv_contrib_amd import synth
from oracle import refocl

pytestmark = pytest.mark.skipif(not (refocl.available() or refocl.can_build()),
                                reason="oracle/_ref not built and /root/reference absent")

SIZES = [(77, 101, 3), (96, 128, 5), (64, 64, 11), (33, 250, 7)]


def _pair(h, w, seed):
    I0, I1, _ = synth.flow_pair(h, w, seed=seed)
    return (I0 * np.float32(255)).astype(np.float32), (I1 * np.float32(255)).astype(np.float32)


def _flow(h, w, seed, amp):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((h, w)) * amp).astype(np.float32), (rng.standard_normal((h, w)) * amp).astype(np.float32)


@pytest.mark.parametrize("h,w,seed", SIZES)
def test_centered_gradient_equals_reference_kernel(oracle, h, w, seed):
    _, I1 = _pair(h, w, seed)
    rx, ry = refocl.tvl1_centered_gradient(I1)
    ox, oy = oracle.tvl1_centered_gradient(I1)
    np.testing.assert_array_equal(ox, rx)
    np.testing.assert_array_equal(oy, ry)


@pytest.mark.parametrize("amp", [0.0, 1.5, 8.0, 400.0])   # 400 px: every tap clamped far outside the image
@pytest.mark.parametrize("h,w,seed", SIZES)
def test_warp_equals_reference_kernel(oracle, h, w, seed, amp):
    I0, I1 = _pair(h, w, seed)
    I1x, I1y = refocl.tvl1_centered_gradient(I1)
    u1, u2 = _flow(h, w, seed + 100, amp)
    if amp == 0.0:
        u1[::3, ::5] = 0.5   # exact half-pixel phases: the 5-tap rows of the [ceil(w-2), floor(w+2)] window
        u2[1::4, ::2] = -1.0
    ref = refocl.tvl1_warp(I0, I1, I1x, I1y, u1, u2)
    got = oracle.tvl1_warp(1, I0, I1, I1x, I1y, u1, u2)
    for name, r, g in zip(("I1w", "I1wx", "I1wy", "grad", "rho_c"), ref, got):
        np.testing.assert_array_equal(g, r, err_msg=name)


@pytest.mark.parametrize("h,w,seed", SIZES)
def test_iteration_equals_reference_kernels(oracle, h, w, seed):
    I0, I1 = _pair(h, w, seed)
    I1x, I1y = refocl.tvl1_centered_gradient(I1)
    u1, u2 = _flow(h, w, seed + 1, 2.0)
    _, I1wx, I1wy, grad, rho_c = refocl.tvl1_warp(I0, I1, I1x, I1y, u1, u2)
    grad[::7, ::3] = 0   # the gradVal <= FLT_EPSILON branch of the thresholding step
    p = [x for pair in (_flow(h, w, seed + 2, 0.4), _flow(h, w, seed + 3, 0.4)) for x in pair]
    l_t, theta, taut = np.float32(0.15 * 0.3), np.float32(0.3), np.float32(0.25 / 0.3)
    u1r, u2r, pr = u1, u2, p
    u1o, u2o, po = u1, u2, p
    for it in range(3):
        r = refocl.tvl1_iteration(I1wx, I1wy, grad, rho_c, u1r, u2r, *pr, l_t, theta, taut)
        o = oracle.tvl1_iteration(1, I1wx, I1wy, grad, rho_c, u1o, u2o, *po, l_t, theta, taut)
        for k, name in enumerate(("u1", "u2", "p11", "p12", "p21", "p22")):
            np.testing.assert_array_equal(o[1 + k], r[1 + k], err_msg=f"{name} after iteration {it + 1}")
        # cv::sum(diff)[0] of the reference's error plane (double accumulator) == the oracle's error
        assert np.float32(r[0].astype(np.float64).sum()) == np.float32(o[0])
        u1r, u2r, pr = r[1], r[2], list(r[3:])
        u1o, u2o, po = o[1], o[2], list(o[3:])


@pytest.mark.parametrize("eps,iters", [(0.0, 10), (0.01, 300), (0.05, 40)])
@pytest.mark.parametrize("h,w,seed", SIZES[:2])
def test_proc_one_scale_equals_reference_loop(oracle, h, w, seed, eps, iters):
    """The per-scale loop on the reference kernels vs oracle.tvl1 (CUDA_COMPAT): flow bit for bit, iteration counts equal."""
    I0, I1 = _pair(h, w, seed)
    u1, u2 = _flow(h, w, seed + 9, 0.5)
    P = oracle.tvl1_params(semantics=1, epsilon=eps, outer_iterations=iters, inner_iterations=1, median_filtering=1)
    o1, o2, oit = oracle.tvl1_proc_one_scale(I0, I1, u1, u2, P)
    r1, r2, rit = refocl.tvl1_proc_one_scale(I0, I1, u1, u2, epsilon=eps, outer_iterations=iters)
    np.testing.assert_array_equal(oit, rit)
    if eps > 0:
        assert 2 <= rit.min() < iters, "the convergence rule must be what stops some of these warps"
    np.testing.assert_array_equal(o1, r1)
    np.testing.assert_array_equal(o2, r2)


def test_contraction_freedom_is_bounded(oracle):
    """Same reference source, built with a*b+c fusion allowed (what a device compiler may do): stage outputs move by a few
    ulp of the plane's scale, and a 10-iteration scale by < 1e-4 px -- two orders below the parity bounds used on the GPU."""
    h, w, seed = 96, 128, 5
    I0, I1 = _pair(h, w, seed)
    I1x, I1y = refocl.tvl1_centered_gradient(I1)
    u1, u2 = _flow(h, w, seed + 100, 1.5)
    a = refocl.tvl1_warp(I0, I1, I1x, I1y, u1, u2)
    b = refocl.tvl1_warp(I0, I1, I1x, I1y, u1, u2, fma=True)
    assert any(not np.array_equal(x, y) for x, y in zip(a, b)), "the contracted build must really differ"
    for name, x, y in zip(("I1w", "I1wx", "I1wy"), a, b):
        assert np.abs(x - y).max() <= 4e-6 * max(1.0, np.abs(x).max()), name
    z = np.zeros((h, w), np.float32)
    r1, r2, _ = refocl.tvl1_proc_one_scale(I0, I1, z, z, epsilon=0.0, outer_iterations=10)
    f1, f2, _ = refocl.tvl1_proc_one_scale(I0, I1, z, z, epsilon=0.0, outer_iterations=10, fma=True)
    assert synth.epe(np.stack([r1, r2], -1), np.stack([f1, f2], -1)) < 1e-4


# ------------------------------------------------------------------------------------------- SURF: oracle vs surf.cl
SURF_CASES = [(300, 400, 11, 100.0, 4, 2), (200, 264, 5, 300.0, 3, 3), (481, 640, 7, 400.0, 4, 2)]


def _kp_matrix(ro):
    K = np.zeros((7, ro["n"]), np.float32)
    K[0], K[1], K[4], K[5], K[6] = ro["x"], ro["y"], ro["size"], ro["angle"], ro["hessian"]
    K[2:3].view(np.int32)[0] = ro["laplacian"]
    K[3:4].view(np.int32)[0] = ro["octave"]
    return K


@pytest.mark.parametrize("h,w,seed,thr,octaves,layers", SURF_CASES)
def test_surf_det_trace_and_maxima_equal_reference_kernels(oracle, h, w, seed, thr, octaves, layers):
    """SURF_calcLayerDetAndTrace (built with DOUBLE_SUPPORT, as on a device with doubles) and SURF_findMaximaInLayer against
    oracle.surf_det_trace / orc_surf_find_maxima: determinant and trace planes bit for bit wherever the kernel writes, the same
    set of (x, y, layer, laplacian) candidates in every octave."""
    img = synth.blob_image(h, w, seed=seed)
    S = oracle.surf_integral(img)
    L = oracle.lib()
    for octave in range(octaves):
        rd, rt = refocl.surf_det_trace(S, octave, layers)
        od, ot = oracle.surf_det_trace(S, octave, layers)
        m = (rd != 0) | (rt != 0)
        assert m.mean() > 0.02
        np.testing.assert_array_equal(od[m], rd[m])
        np.testing.assert_array_equal(ot[m], rt[m])
        n, cand = refocl.surf_find_maxima(od, ot, h, w, octave, layers, thr)
        oc = np.zeros((65536, 4), np.int32)
        no = L.orc_surf_find_maxima(od, ot, None, h, w, octave, layers, thr, 65535, oc.reshape(-1))
        assert n == no
        assert set(map(tuple, cand.tolist())) == set(map(tuple, oc[:no].tolist()))


@pytest.mark.parametrize("h,w,seed,thr,octaves,layers", SURF_CASES)
def test_surf_keypoints_equal_reference_detector(oracle, h, w, seed, thr, octaves, layers):
    """SURF_OCL::detectKeypoints on the reference kernels (det/trace -> maxima -> SURF_interpolateKeypoint, the octave loop of
    surf.ocl.cpp:152-200) against oracle.surf_detect_describe: the same set of keypoints, size / response / laplacian / octave bit
    for bit (the reference appends with atomic_inc, so order is not compared); x / y to one unit in the last place -- the oracle
    follows the CUDA class bit for bit (tests/test_ref_pin_cuda.py: surf.cu itself), whose solve3x3 (core/cuda/utility.hpp) takes
    the reciprocal of the determinant in double where the OpenCL twin (surf.cl:413-441) stays in float."""
    img = synth.blob_image(h, w, seed=seed)
    S = oracle.surf_integral(img)
    ro = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=thr, n_octaves=octaves, n_octave_layers=layers,
                                                             keypoints_ratio=0.05), want_desc=False)
    kp = refocl.surf_detect(S, octaves, layers, thr, 0.05)
    assert kp.shape[1] == ro["n"] > 10
    ko = _kp_matrix(ro)
    order = lambda K: np.lexsort((K[6], K[4], K[1], K[0]))
    a, b = kp[:, order(kp)], ko[:, order(ko)]
    for row in (4, 6):
        np.testing.assert_array_equal(a[row], b[row])
    for row in (0, 1):
        assert (np.abs(a[row] - b[row]) <= np.spacing(np.abs(b[row]))).all()
        assert (a[row] != b[row]).mean() < 0.02
    np.testing.assert_array_equal(a[2:4].view(np.int32), b[2:4].view(np.int32))


@pytest.mark.parametrize("extended", [False, True])
@pytest.mark.parametrize("h,w,seed,thr,octaves,layers", SURF_CASES[:2])
def test_surf_orientation_and_descriptors_against_reference_kernels(oracle, h, w, seed, thr, octaves, layers, extended):
    """SURF_calcOrientation and SURF_computeDescriptors64/128 + normalize of the OpenCL class on the oracle's keypoints -- a CROSS-CHECK
    of a sibling implementation, not the pin: the oracle is pinned bit for bit on the CUDA class itself (tests/test_ref_pin_cuda.py runs
    surf.cu and surf.cuda.cpp).  Orientation: the twin reads the integral image as float and reduces in another order: within 1e-3
    degrees.  Descriptors: the twin samples the rotated window differently from the CUDA class -- coordinates rounded (surf.cl:62-68)
    instead of floor-addressed texture reads, patch samples kept in float where core/cuda/filters.hpp rounds them to 8 bits
    (WinReader::elem_type = uchar), 1 / s^2 everywhere where AreaFilter normalises the last patch row / column by the window's
    remainder -- so the two unit vectors agree to a few hundredths per element: L2 distance < 0.35 for every keypoint, < 0.1 on
    average, cosine > 0.93 (64 floats; the 128-float form splits the sums by the SIGN of the other derivative, which the 8-bit patch
    often makes exactly 0: < 0.7, < 0.25, > 0.75).  (Until round 3 the oracle followed the twin here and this test held 1e-6; the CUDA source pin showed the
    CUDA class differs.)"""
    img = synth.blob_image(h, w, seed=seed)
    S = oracle.surf_integral(img)
    ro = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=thr, n_octaves=octaves, n_octave_layers=layers,
                                                             extended=int(extended), keypoints_ratio=0.05))
    K = _kp_matrix(ro)
    ang = refocl.surf_orientation(S, K)
    d = np.abs(ang - ro["angle"]); d = np.minimum(d, 360 - d)
    assert d.max() <= 1e-3, d.max()
    desc = refocl.surf_descriptors(img, K, extended)
    assert desc.shape == ro["descriptors"].shape
    dist = np.linalg.norm(desc - ro["descriptors"], axis=1)
    assert dist.max() < (0.7 if extended else 0.35) and dist.mean() < (0.25 if extended else 0.1), (dist.max(), dist.mean())
    assert (desc * ro["descriptors"]).sum(1).min() > (0.75 if extended else 0.93)
    np.testing.assert_allclose(np.linalg.norm(desc, axis=1), 1.0, atol=1e-5)


# ------------------------------------------------------------------------------- the CPU class: oracle vs tvl1flow.cpp itself
CPU_CASES = [dict(),                                                                         # class defaults: median 5, inner 30, outer 10
             dict(inner_iterations=1, outer_iterations=10, median_filtering=1, epsilon=0.0),  # the cv::cuda-equivalent test setting
             dict(inner_iterations=1, outer_iterations=300, median_filtering=1),              # convergence-checked, no median filter
             dict(inner_iterations=3, outer_iterations=4, median_filtering=3, epsilon=0.0),
             dict(inner_iterations=1, outer_iterations=10, median_filtering=1, epsilon=0.0, gamma=1.0),
             dict(nscales=3, warps=2, scale_step=0.5, tau=0.2, lambda_=0.3, theta=0.25, inner_iterations=1, outer_iterations=8,
                  median_filtering=1, epsilon=0.0)]


@pytest.mark.parametrize("kw", CPU_CASES, ids=["defaults", "test_setting", "eps_check", "median3", "gamma", "other_params"])
@pytest.mark.parametrize("h,w,seed,dtype", [(96, 128, 3, "f32"), (120, 160, 5, "u8"), (50, 70, 9, "f32")])
def test_cpu_class_oracle_equals_reference_class(oracle, h, w, seed, dtype, kw):
    """oracle.tvl1_calc (semantics CPU_REF) against cv::optflow::DualTVL1OpticalFlow ITSELF: modules/optflow/src/tvl1flow.cpp
    compiled verbatim (oracle/_ref/libref_cpu.so) against the stub core of oracle/refshim/cvstub.  Every line of arithmetic in
    that file -- gradients, divergence, thresholding, primal / dual updates with the f64 hypot, the serial float error sum and
    the loops around them, level sizes, flow rescaling -- is the reference's; cv::resize / cv::remap / cv::medianBlur are
    main-repo functions (absent from /root/reference) and both sides use the restatement of oracle/imgproc_ref.c for them.
    Flows equal bit for bit."""
    I0, I1, _ = synth.flow_pair(h, w, seed=seed, dtype=dtype)
    ref, ns = refocl.cpu_tvl1_calc(I0, I1, **kw)
    okw = dict(kw)
    got, st = oracle.tvl1_calc(I0, I1, oracle.tvl1_params(**okw), return_stats=True)
    np.testing.assert_array_equal(got, ref)
    assert st["nscales"] == ns


def test_cpu_class_oracle_equals_reference_class_initial_flow_and_shrinking_scales(oracle):
    I0, I1, gt = synth.flow_pair(40, 56, seed=13)
    init = (gt * 0.7).astype(np.float32)
    kw = dict(inner_iterations=1, outer_iterations=6, median_filtering=1, epsilon=0.0, nscales=8)   # levels fall below 16 px: nscales shrinks
    ref, ns = refocl.cpu_tvl1_calc(I0, I1, init_flow=init, **kw)
    got, st = oracle.tvl1_calc(I0, I1, oracle.tvl1_params(use_initial_flow=1, **kw), init_flow=init, return_stats=True)
    np.testing.assert_array_equal(got, ref)
    assert ns == st["nscales"] < 8


# ------------------------------------------------------------------------------------------- HIP vs the reference kernels
@pytest.mark.gpu
@pytest.mark.parametrize("h,w,seed", SIZES[:2])
def test_hip_stages_against_reference_kernels(gpu, h, w, seed):
    import torch
    from opencv_contrib_amd import capi, cuda
    I0, I1 = _pair(h, w, seed)
    rx, ry = refocl.tvl1_centered_gradient(I1)
    dx, dy = cuda.tvl1_centeredGradient(torch.from_numpy(I1).to(gpu))
    np.testing.assert_array_equal(dx.cpu().numpy(), rx)
    np.testing.assert_array_equal(dy.cpu().numpy(), ry)
    u1, u2 = _flow(h, w, seed + 100, 1.5)
    ref = refocl.tvl1_warp(I0, I1, rx, ry, u1, u2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    got = cuda.tvl1_warpBackward(capi.MI_SEM_CUDA_COMPAT, t(I0), t(I1), t(rx), t(ry), t(u1), t(u2))
    for name, r, g in zip(("I1w", "I1wx", "I1wy", "grad", "rho_c"), ref, got):
        g = g.cpu().numpy()
        scale = max(1.0, float(np.abs(r).max()))
        assert np.abs(g - r).max() <= 2e-5 * scale, name
    # one exact-math iteration on the reference's warp planes
    p = [x for pair in (_flow(h, w, seed + 2, 0.4), _flow(h, w, seed + 3, 0.4)) for x in pair]
    l_t, theta, taut = np.float32(0.15 * 0.3), np.float32(0.3), np.float32(0.25 / 0.3)
    r = refocl.tvl1_iteration(ref[1], ref[2], ref[3], ref[4], u1, u2, *p, l_t, theta, taut)
    uo, po, _ = cuda.tvl1_iterate(t(ref[1]), t(ref[2]), t(ref[3]), t(ref[4]), [t(u1), t(u2)], [t(x) for x in p],
                                  float(l_t), float(theta), float(taut), niter=1, exact=True)
    for k, (name, g) in enumerate(zip(("u1", "u2", "p11", "p12", "p21", "p22"), uo + po)):
        assert np.abs(g.cpu().numpy() - r[1 + k]).max() <= 2e-6 * max(1.0, float(np.abs(r[1 + k]).max())), name


# ------------------------------------------------------------------ the reference's CPU SURF class, verbatim (round 4)
def _java_cross():
    img = np.full((100, 100), 255, np.uint8)   # getTestImg() of the Java tests, see tests/test_zz_surf_cpu_class.py::java_cross
    img[49:52, 20:80] = 100
    img[20:80, 49:52] = 100
    return img


def test_surf_cpu_class_verbatim_reproduces_the_java_golden_vectors():
    """oracle/_ref/libref_surfcpu.so = modules/xfeatures2d/src/surf.cpp compiled VERBATIM against oracle/refshim/cvsurf (VERDICT r03,
    missing item 4): the class itself -- not a restatement -- gives the known answers of SURFFeatureDetectorTest.java:52-57,100-126 and
    SURFDescriptorExtractorTest.java:38-69 within those tests' EPS."""
    from oracle import refocl
    from tests.test_zz_surf_cpu_class import JAVA_TRUTH, JAVA_DESCRIPTOR, EPS
    kp, _ = refocl.surfcpu_detect_and_compute(_java_cross(), 8000, 3, 4, extended=True, upright=False, want_desc=False)
    kp = kp[np.argsort(kp[:, 3])]
    assert kp.shape == (4, 7)
    np.testing.assert_allclose(kp[:, :5], JAVA_TRUTH[:, :5], rtol=0, atol=EPS)
    np.testing.assert_array_equal(kp[:, 5:], JAVA_TRUTH[:, 5:])
    mask = np.full((100, 100), 255, np.uint8)
    mask[:, 50:] = 0
    km, _ = refocl.surfcpu_detect_and_compute(_java_cross(), 8000, 3, 4, extended=True, upright=False, mask=mask, want_desc=False)
    km = km[np.argsort(km[:, 3])]
    np.testing.assert_allclose(km[:, :5], JAVA_TRUTH[1:3, :5], rtol=0, atol=EPS)
    one = np.array([[55.775577545166016, 44.224422454833984, 16, 9.754629, 8617.863, 1, -1]], np.float32)
    k2, d = refocl.surfcpu_detect_and_compute(_java_cross(), 100, 2, 4, extended=True, upright=False, keypoints=one)
    assert d.shape == (1, 128) and np.abs(d[0] - JAVA_DESCRIPTOR).max() < 1e-6 and abs(k2[0, 3] - 350.24573) < EPS


@pytest.mark.parametrize("shape,seed", [((240, 320), 7), ((300, 400), 11), ((480, 640), 3)])
@pytest.mark.parametrize("extended,upright", [(False, False), (True, False), (False, True)])
def test_surf_cpu_class_restatement_equals_the_verbatim_class(oracle, shape, seed, extended, upright):
    """oracle/surfcpu_ref.c (what the GPU-side acceptance tests use as the CPU class) against the class itself on blob images, detector and
    descriptors, also through a mask and for provided keypoints: BIT-identical keypoints (position, size, angle, response, octave,
    Laplacian sign) and descriptors.  (This pin found the restatement's one deviation: the descriptor weights' sigma is the FLOAT
    constant 3.3f promoted to double, surf.cpp:121,560 -- with 3.3 a fifth of the descriptor entries were 1 ulp off.)"""
    from oracle import refocl
    from opencv_contrib_amd import synth
    img = synth.blob_image(shape[0], shape[1], seed=seed)
    a_kp, a_d = refocl.surfcpu_detect_and_compute(img, 400, 4, 2, extended=extended, upright=upright)
    b_kp, b_d = oracle.surfcpu_detect_and_compute(img, 400, 4, 2, extended=extended, upright=upright)
    assert len(a_kp) > 30
    np.testing.assert_array_equal(a_kp, b_kp)
    np.testing.assert_array_equal(a_d, b_d)
    mask = np.zeros(shape, np.uint8)
    mask[shape[0] // 5:, : 3 * shape[1] // 4] = 255
    m_kp, _ = refocl.surfcpu_detect_and_compute(img, 400, 4, 2, extended=extended, upright=upright, mask=mask, want_desc=False)
    n_kp, _ = oracle.surfcpu_detect_and_compute(img, 400, 4, 2, extended=extended, upright=upright, mask=mask, want_desc=False)
    assert 0 < len(m_kp) < len(a_kp)
    np.testing.assert_array_equal(m_kp, n_kp)
    p_kp, p_d = refocl.surfcpu_detect_and_compute(img, 400, 4, 2, extended=extended, upright=upright, keypoints=a_kp[::3])
    q_kp, q_d = oracle.surfcpu_compute(img, a_kp[::3], extended=extended, upright=upright)
    keep = q_kp[:, 2] > 0
    np.testing.assert_array_equal(p_kp, q_kp[keep])
    np.testing.assert_array_equal(p_d, q_d[keep])
