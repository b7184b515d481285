"""SURF against the reference's CPU class (xfeatures2d::SURF_Impl) -- the comparison the reference's own CUDA test makes.
(File name: collected last, after the HIP-vs-oracle parity tests it builds on.)

* oracle/surfcpu_ref.c restates the CPU class and is PINNED on the reference's known-answer vectors
  (xfeatures2d/misc/java/test/SURFFeatureDetectorTest.java:52-57,100-126 and SURFDescriptorExtractorTest.java:38-69, tolerance = those
  tests' EPS 1e-3; the restatement reproduces them to the last printed digit / 3e-8).
* The CUDA-class oracle (oracle/surf_ref.c, what the HIP kernels are held to) and the HIP kernels themselves are then accepted the way
  xfeatures2d/test/test_surf.cuda.cpp:81-170 accepts cv::cuda::SURF_CUDA: matched-keypoint ratio > 0.95 against the CPU class,
  descriptor nearest-neighbour match ratio > 0.6 for descriptors computed at the CPU class's keypoints; parameter grid :176-187.
  keyPointsEquals of the reference's test support (opencv/opencv modules/ts/src/cuda_test.cpp, un-vendored) is restated: point distance
  < 1 px, size difference < 1, angle difference < 2 degrees, response difference < 0.1, same octave and class_id (Laplacian sign).
"""
import ctypes as C

import numpy as np
import pytest

from opencv_contrib_amd import synth

JAVA_TRUTH = np.array([[55.775578, 55.775578, 16, 80.245735, 8617.8633, 0, -1],
                       [44.224422, 55.775578, 16, 170.24574, 8617.8633, 0, -1],
                       [44.224422, 44.224422, 16, 260.24573, 8617.8633, 0, -1],
                       [55.775578, 44.224422, 16, 350.24573, 8617.8633, 0, -1]], np.float32)
JAVA_DESCRIPTOR = np.array([
    0, 0, 0, 0, 0, 0, 0, 0, 0.058821894, 0.058821894, -0.045962855, 0.046261817, 0.0085156476,
    0.0085754395, -0.0064509804, 0.0064509804, 0.00044069235, 0.00044069235, 0, 0, 0.00025723741,
    0.00025723741, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.00025723741, 0.00025723741, -0.00044069235,
    0.00044069235, 0, 0, 0.36278215, 0.36278215, -0.24688604, 0.26173124, 0.052068226, 0.052662034,
    -0.032815345, 0.032815345, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -0.0064523756,
    0.0064523756, 0.0082002236, 0.0088908644, -0.059001274, 0.059001274, 0.045789491, 0.04648013,
    0.11961588, 0.22789426, -0.01322381, 0.18291828, -0.14042182, 0.23973691, 0.073782086, 0.23769434,
    -0.027880307, 0.027880307, 0.049587864, 0.049587864, -0.33991757, 0.33991757, 0.21437603, 0.21437603,
    -0.0020763327, 0.0020763327, 0.006245892, 0.006245892, -0.04067041, 0.04067041, 0.019361559,
    0.019361559, 0, 0, -0.0035977389, 0.0035977389, 0, 0, -0.00099993451, 0.00099993451, 0.040670406,
    0.040670406, -0.019361559, 0.019361559, 0.006245892, 0.006245892, -0.0020763327, 0.0020763327,
    -0.00034532088, 0.00034532088, 0, 0, 0, 0, 0.00034532088, 0.00034532088, -0.00099993451,
    0.00099993451, 0, 0, 0, 0, 0.0035977389, 0.0035977389], np.float32)
EPS = 1e-3      # OpenCVTestCase.EPS of the Java tests


def java_cross():
    """getTestImg() of the Java tests: 100 x 100, value 255, Imgproc.line((20, 50) - (79, 50)) and ((50, 20) - (50, 79)) with value
    100 and thickness 2.  cv::line is main-repository code (un-vendored); an axis-aligned thickness-2 line covers the axis row and one
    row on either side -- the only rasterisation under which any keypoint survives the 8000 threshold, and under it the CPU-class
    restatement reproduces every digit of the golden vectors below."""
    img = np.full((100, 100), 255, np.uint8)
    img[49:52, 20:80] = 100
    img[20:80, 49:52] = 100
    return img


# ------------------------------------------------------------------ the CPU class, pinned (CPU)
def test_cpu_class_reproduces_the_reference_golden_keypoints(oracle):
    """SURFFeatureDetectorTest.testDetectMatListOfKeyPoint: hessianThreshold 8000, nOctaves 3, nOctaveLayers 4, upright false."""
    kp, _ = oracle.surfcpu_detect_and_compute(java_cross(), 8000, 3, 4, extended=True, upright=False, want_desc=False)
    kp = kp[np.argsort(kp[:, 3])]                 # the test orders by angle
    assert kp.shape == (4, 7)
    np.testing.assert_allclose(kp[:, :5], JAVA_TRUTH[:, :5], rtol=0, atol=EPS)
    np.testing.assert_array_equal(kp[:, 5:], JAVA_TRUTH[:, 5:])      # octave, class_id
    # testDetectMatListOfKeyPointMat: the right half masked out leaves truth[1], truth[2]
    mask = np.full((100, 100), 255, np.uint8)
    mask[:, 50:] = 0
    km, _ = oracle.surfcpu_detect_and_compute(java_cross(), 8000, 3, 4, extended=True, upright=False, mask=mask, want_desc=False)
    km = km[np.argsort(km[:, 3])]
    np.testing.assert_allclose(km[:, :5], JAVA_TRUTH[1:3, :5], rtol=0, atol=EPS)


def test_cpu_class_reproduces_the_reference_golden_descriptor(oracle):
    """SURFDescriptorExtractorTest.testComputeMatListOfKeyPointMat: SURF(100, 2, 4, extended, not upright).compute on one keypoint."""
    kp = np.array([[55.775577545166016, 44.224422454833984, 16, 9.754629, 8617.863, 1, -1]], np.float32)
    k2, d = oracle.surfcpu_compute(java_cross(), kp, extended=True, upright=False)
    assert d.shape == (1, 128)
    np.testing.assert_allclose(d[0], JAVA_DESCRIPTOR, rtol=0, atol=EPS)
    assert np.abs(d[0] - JAVA_DESCRIPTOR).max() < 1e-6            # in fact to the last digit
    assert abs(k2[0, 3] - 350.24573) < EPS                         # compute() re-estimates the orientation


def test_cuda_class_oracle_on_the_golden_cross(oracle):
    """The CUDA-class restatement finds the same four keypoints (position, size, response, octave within the Java tests' EPS).
    Their orientation has two exactly tied maxima (the image is symmetric under transposition): the CPU class's sequential strict-<
    search and the CUDA class's per-thread / cross-thread reduction (surf.cu:595-640) keep different twins for two of them."""
    r = oracle.surf_detect_describe(java_cross(), oracle.surf_params(hessian_threshold=8000, n_octaves=3, n_octave_layers=4, extended=1,
                                                                     keypoints_ratio=0.05), want_desc=False)
    assert r["n"] == 4
    for i in range(4):
        j = int(np.argmin(np.hypot(JAVA_TRUTH[:, 0] - r["x"][i], JAVA_TRUTH[:, 1] - r["y"][i])))
        t = JAVA_TRUTH[j]
        assert abs(r["x"][i] - t[0]) < EPS and abs(r["y"][i] - t[1]) < EPS and r["size"][i] == t[2] and r["octave"][i] == 0
        assert abs(r["hessian"][i] - t[4]) < 5e-3 and r["laplacian"][i] == -1
        # the tied twin is the truth's mirror image about the diagonal through the keypoint
        diag = 45.0 if (t[0] > 50) == (t[1] > 50) else 135.0
        twin = (2 * diag - t[3]) % 360
        da = min(abs(r["angle"][i] - t[3]), 360 - abs(r["angle"][i] - t[3]))
        db = min(abs(r["angle"][i] - twin), 360 - abs(r["angle"][i] - twin))
        assert min(da, db) < 2e-3, (r["angle"][i], t[3], twin)


# ------------------------------------------------------------------ the reference's acceptance of the CUDA class (test_surf.cuda.cpp)
def keypoints_equal(g, a):
    """keyPointsEquals: g, a = rows {x, y, size, angle, response, octave, class_id}"""
    da = np.abs(g[:, 3] - a[:, 3])
    return ((np.hypot(g[:, 0] - a[:, 0], g[:, 1] - a[:, 1]) < 1.0) & (np.abs(g[:, 2] - a[:, 2]) < 1.0) & (np.minimum(da, 360 - da) < 2.0) &
            (np.abs(g[:, 4] - a[:, 4]) < 0.1) & (g[:, 5] == a[:, 5]) & (g[:, 6] == a[:, 6]))


def matched_points_ratio(gold, actual):
    """getMatchedPointsCount(gold, actual) / gold.size(): both lists sorted by (x, y), compared pairwise."""
    n = min(len(gold), len(actual))
    g = gold[np.lexsort((gold[:, 1], gold[:, 0]))][:n]
    a = actual[np.lexsort((actual[:, 1], actual[:, 0]))][:n]
    return float(keypoints_equal(g, a).sum()) / len(gold)


def cuda_rows(r):
    return np.stack([r["x"], r["y"], r["size"], r["angle"], r["hessian"], r["octave"].astype(np.float32),
                     r["laplacian"].astype(np.float32)], 1)


GRID = [(thr, o, l, e, u) for thr in (100.0, 500.0, 1000.0) for o in (3, 4) for l in (2, 3) for e in (False, True) for u in (False, True)]


def _detector_case(cuda_detect, oracle, thr, octaves, layers, upright, mask=None):
    img = synth.blob_image(360, 480, seed=21)
    gold, _ = oracle.surfcpu_detect_and_compute(img, thr, octaves, layers, upright=upright, mask=mask, want_desc=False)
    act = cuda_detect(img, thr, octaves, layers, upright, mask)
    assert len(gold) > 50
    assert abs(len(gold)) - len(act) <= 1                        # lengthDiff, test_surf.cuda.cpp:100-101
    assert matched_points_ratio(gold, act) > 0.95                 # :102-105


def _descriptor_case(cuda_describe, oracle, thr, octaves, layers, extended, upright):
    img = synth.blob_image(360, 480, seed=22)
    kp, gold = oracle.surfcpu_detect_and_compute(img, thr, octaves, layers, extended=extended, upright=upright)
    act = cuda_describe(img, kp, thr, octaves, layers, extended, upright)
    assert act.shape == gold.shape and len(kp) > 50
    idx, _, _ = oracle.bf_knn_match(gold, act, 1)                  # cv::BFMatcher(NORM_L2).match(descriptors_gold, descriptors)
    matched = keypoints_equal(kp, kp[idx[:, 0]])
    assert float(matched.sum()) / len(kp) > 0.6                   # :166-169


def _oracle_detect(oracle):
    def f(img, thr, octaves, layers, upright, mask):
        return cuda_rows(oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=thr, n_octaves=octaves, n_octave_layers=layers,
                                                                              upright=int(upright), keypoints_ratio=0.05), mask=mask, want_desc=False))
    return f


def _oracle_describe(oracle):
    L = oracle.lib()
    L.orc_surf_orientation.restype = C.c_float
    ax, ay, aw, dw = oracle.surf_tables()

    def f(img, kp, thr, octaves, layers, extended, upright):
        s = oracle.surf_integral(img)
        out = np.zeros((len(kp), 128 if extended else 64), np.float32)
        for i, k in enumerate(kp):      # operator()(img, mask, keypoints, descriptors, useProvidedKeypoints = true): surf.cuda.cpp:380-397
            a = 270.0 if upright else L.orc_surf_orientation(s, img.shape[0], img.shape[1], float(k[0]), float(k[1]), float(k[2]), ax, ay, aw)
            L.orc_surf_descriptor(img, img.shape[0], img.shape[1], float(k[0]), float(k[1]), float(k[2]), float(a), int(extended), dw, out[i])
        return out
    return f


@pytest.mark.parametrize("thr,octaves,layers,upright", sorted({(t, o, l, u) for t, o, l, _, u in GRID}))
def test_cuda_class_oracle_detector_accepted_against_cpu_class(oracle, thr, octaves, layers, upright):
    _detector_case(_oracle_detect(oracle), oracle, thr, octaves, layers, upright)


def test_cuda_class_oracle_masked_detector_accepted_against_cpu_class(oracle):
    mask = np.ones((360, 480), np.uint8)
    mask[:180, :240] = 0                                          # Detector_Masked, test_surf.cuda.cpp:112-113
    _detector_case(_oracle_detect(oracle), oracle, 100.0, 4, 2, False, mask)


@pytest.mark.parametrize("thr,octaves,layers,extended,upright", [g for g in GRID if g[0] != 500.0 and g[1] == 4])
def test_cuda_class_oracle_descriptors_accepted_against_cpu_class(oracle, thr, octaves, layers, extended, upright):
    _descriptor_case(_oracle_describe(oracle), oracle, thr, octaves, layers, extended, upright)


def _binding_test_case(cuda_detect, cuda_describe, oracle):
    """xfeatures2d/misc/python/test/test_cuda_xfeatures2d.py:26-49 (its image, aloe.png, is in opencv_extra: a synthetic one here):
    SURF_CUDA and SURF find the same NUMBER of keypoints, and descriptors computed for provided keypoints have the CPU class's shape."""
    img = synth.blob_image(420, 560, seed=31)
    gold, dgold = oracle.surfcpu_detect_and_compute(img, 100.0, 3, 2, extended=False, upright=False)
    act = cuda_detect(img, 100.0, 3, 2, False, None)
    assert len(act) == len(gold) > 100
    assert cuda_describe(img, gold, 100.0, 3, 2, False, False).shape == dgold.shape


def test_cuda_class_oracle_passes_the_reference_python_binding_test(oracle):
    _binding_test_case(_oracle_detect(oracle), _oracle_describe(oracle), oracle)


# ------------------------------------------------------------------ the same acceptance on the HIP kernels (GPU)
def _hip_detect(gpu):
    import torch
    from opencv_contrib_amd import cuda

    def f(img, thr, octaves, layers, upright, mask):
        alg = cuda.SURF_CUDA.create(thr, octaves, layers, False, 0.05, upright)
        t = torch.from_numpy(img).to(gpu)
        kp = cuda.SURF_CUDA.downloadKeypoints(alg.detect(t, None if mask is None else torch.from_numpy(mask).to(gpu)))
        return cuda_rows(kp)
    return f


def _hip_describe(gpu):
    import torch
    from opencv_contrib_amd import cuda

    def f(img, kp, thr, octaves, layers, extended, upright):
        alg = cuda.SURF_CUDA.create(thr, octaves, layers, extended, 0.05, upright)
        k = np.zeros((7, len(kp)), np.float32)                    # SURF_CUDA keypoint matrix rows, cuda.hpp:89-99
        k[0], k[1], k[4], k[5], k[6] = kp[:, 0], kp[:, 1], kp[:, 2], kp[:, 3], kp[:, 4]     # uploadKeypoints, surf.cuda.cpp:287-317
        k.view(np.int32)[2] = kp[:, 6].astype(np.int32)
        k.view(np.int32)[3] = kp[:, 5].astype(np.int32)
        _, desc = alg.detectWithDescriptors(torch.from_numpy(img).to(gpu), None, torch.from_numpy(k).to(gpu), True)
        return desc.cpu().numpy()
    return f


@pytest.mark.gpu
@pytest.mark.parametrize("thr,octaves,layers,upright", [(100.0, 4, 2, False), (500.0, 3, 3, False), (1000.0, 4, 3, True)])
def test_hip_detector_accepted_against_cpu_class(gpu, oracle, thr, octaves, layers, upright):
    _detector_case(_hip_detect(gpu), oracle, thr, octaves, layers, upright)


@pytest.mark.gpu
def test_hip_masked_detector_accepted_against_cpu_class(gpu, oracle):
    mask = np.ones((360, 480), np.uint8)
    mask[:180, :240] = 0
    _detector_case(_hip_detect(gpu), oracle, 100.0, 4, 2, False, mask)


@pytest.mark.gpu
@pytest.mark.parametrize("thr,octaves,layers,extended,upright", [(100.0, 4, 2, False, False), (1000.0, 4, 3, True, False),
                                                                 (100.0, 4, 3, True, True)])
def test_hip_descriptors_accepted_against_cpu_class(gpu, oracle, thr, octaves, layers, extended, upright):
    _descriptor_case(_hip_describe(gpu), oracle, thr, octaves, layers, extended, upright)


@pytest.mark.gpu
def test_hip_passes_the_reference_python_binding_test(gpu, oracle):
    _binding_test_case(_hip_detect(gpu), _hip_describe(gpu), oracle)
