"""Committed golden fixtures (tests/golden/*.npz, produced by tools/make_golden.py FROM THE ORACLE -- the reference's own
goldens live in opencv_extra, absent here): the oracle must keep reproducing them (CPU) and the HIP path must match
them (GPU) with the same tolerances as the direct HIP-vs-oracle tests."""
import glob
import json
import os

import numpy as np
import pytest

from opencv_contrib_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _files(prefix):
    return sorted(glob.glob(os.path.join(GOLD, prefix + "_*.npz")))


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------ oracle vs golden (CPU)
@pytest.mark.parametrize("path", _files("sbm"))
def test_stereobm_oracle_matches_golden(oracle, path):
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    np.testing.assert_array_equal(oracle.sbm_compute(z["left"], z["right"], oracle.sbm_params(**kw)), z["disp"])


@pytest.mark.parametrize("path", _files("fb"))
def test_farneback_oracle_matches_golden(oracle, path):
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    np.testing.assert_allclose(oracle.fb_calc(z["I0"], z["I1"], oracle.fb_params(**kw)), z["flow"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("path", _files("surf"))
def test_surf_oracle_matches_golden(oracle, path):
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    r = oracle.surf_detect_describe(z["img"], oracle.surf_params(**kw))
    assert r["n"] == len(z["x"])
    for k in ("laplacian", "octave", "size"):
        np.testing.assert_array_equal(r[k], z[k])
    for k in ("x", "y", "hessian", "angle"):
        np.testing.assert_allclose(r[k], z[k], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(r["descriptors"], z["descriptors"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("path", _files("sgm"))
def test_stereosgm_oracle_matches_golden(oracle, path):
    z = np.load(path)
    np.testing.assert_array_equal(oracle.sgm_compute(z["left"], z["right"], oracle.sgm_params(**json.loads(str(z["params"])))), z["disp"])


@pytest.mark.parametrize("path", _files("dbf"))
def test_disp_bilateral_oracle_matches_golden(oracle, path):
    z = np.load(path)
    np.testing.assert_array_equal(oracle.dbf_apply(z["disp"], z["img"], oracle.dbf_params(**json.loads(str(z["params"])))), z["out"])


@pytest.mark.parametrize("path", _files("bf"))
def test_bfmatch_oracle_matches_golden(oracle, path):
    z = np.load(path)
    idx, dist = oracle.bf_knn_match2(z["query"], z["train"])
    np.testing.assert_array_equal(idx, z["idx"]); np.testing.assert_array_equal(dist, z["dist"])


# ------------------------------------------------------------------ HIP vs golden (GPU)
@pytest.mark.parametrize("path", _files("surfcpu"))
def test_cpu_class_surf_oracle_matches_golden(oracle, path):
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    kp = oracle.surfcpu_detect(z["img"], kw["hessian_threshold"], kw["n_octaves"], kw["n_octave_layers"])
    np.testing.assert_array_equal(kp, z["detected"])
    k2, d2 = oracle.surfcpu_compute(z["img"], kp, kw["extended"], kw["upright"])
    np.testing.assert_array_equal(k2, z["keypoints"]); np.testing.assert_array_equal(d2, z["descriptors"])


@pytest.mark.parametrize("path", _files("sparselk"))
def test_sparse_pyrlk_oracle_matches_golden(oracle, path):
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    nxt, st, err = oracle.pyrlk_sparse(z["I0"], z["I1"], z["prev_pts"], tuple(kw["win_size"]), kw["max_level"], kw["iters"])
    np.testing.assert_array_equal(nxt, z["next_pts"]); np.testing.assert_array_equal(st, z["status"]); np.testing.assert_array_equal(err, z["err"])


@pytest.mark.parametrize("path", _files("bfint"))
def test_integer_matcher_oracle_matches_golden(oracle, path):
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    idx, img, dist = oracle.bf_knn_match(z["query"], [z["train0"], z["train1"]], kw["k"], kw["norm"])
    np.testing.assert_array_equal(idx, z["idx"]); np.testing.assert_array_equal(img, z["img"]); np.testing.assert_array_equal(dist, z["dist"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", _files("sgm"))
def test_stereosgm_hip_matches_golden(gpu, path):
    from opencv_contrib_amd import cuda
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    sgm = cuda.createStereoSGM(kw.get("min_disparity", 0), kw["num_disparities"], 10, 120, 5, kw["mode"],
                               emulateCudaQuirks=bool(kw.get("emulate_quirks", 1)))
    np.testing.assert_array_equal(sgm.compute(T(z["left"], gpu), T(z["right"], gpu)).cpu().numpy(), z["disp"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", _files("dbf"))
def test_disp_bilateral_hip_matches_golden(gpu, path):
    from opencv_contrib_amd import cuda
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    f = cuda.createDisparityBilateralFilter(kw["ndisp"], kw["radius"], kw["iters"])
    np.testing.assert_array_equal(f.apply(T(z["disp"], gpu), T(z["img"], gpu)).cpu().numpy(), z["out"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", _files("bf"))
def test_bfmatch_hip_matches_golden(gpu, path):
    from opencv_contrib_amd import cuda
    z = np.load(path)
    idx, _, dist = cuda.createBFMatcher().knnMatchDevice(T(z["query"], gpu), T(z["train"], gpu), k=2)
    np.testing.assert_array_equal(idx.cpu().numpy(), z["idx"]); np.testing.assert_array_equal(dist.cpu().numpy(), z["dist"])



@pytest.mark.gpu
@pytest.mark.parametrize("path", _files("sbm"))
def test_stereobm_hip_matches_golden(gpu, path):
    from opencv_contrib_amd import cuda
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    bm = cuda.createStereoBM(kw["num_disparities"], kw["block_size"], emulateCudaEdge=bool(kw.get("emulate_edge", 1)))
    if "prefilter_type" in kw:
        bm.setPreFilterType(kw["prefilter_type"])
    if "uniqueness_ratio" in kw:
        bm.setUniquenessRatio(kw["uniqueness_ratio"])
    if "texture_threshold" in kw:
        bm.setTextureThreshold(int(kw["texture_threshold"]))
    np.testing.assert_array_equal(bm.compute(T(z["left"], gpu), T(z["right"], gpu)).cpu().numpy(), z["disp"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", _files("fb"))
def test_farneback_hip_matches_golden(gpu, path):
    from opencv_contrib_amd import cuda
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    alg = cuda.FarnebackOpticalFlow.create(numLevels=kw.get("num_levels", 5), fastPyramids=bool(kw.get("fast_pyramids", 0)),
                                           polyN=kw.get("poly_n", 5), polySigma=kw.get("poly_sigma", 1.1), flags=kw.get("flags", 0))
    flow = alg.calc(T(z["I0"], gpu), T(z["I1"], gpu)).cpu().numpy()
    d = np.sqrt(((flow - z["flow"]) ** 2).sum(-1))
    assert d.mean() <= 2e-3 and synth.ccorr_dissimilarity(flow, z["flow"]) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("path", _files("surf"))
def test_surf_hip_matches_golden(gpu, path):
    from opencv_contrib_amd import cuda
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    alg = cuda.SURF_CUDA.create(kw["hessian_threshold"], kw.get("n_octaves", 4), 2, bool(kw.get("extended", 0)),
                                kw.get("keypoints_ratio", 0.01), bool(kw.get("upright", 0)))
    kpg, desc = alg.detectWithDescriptors(T(z["img"], gpu))
    kp = cuda.SURF_CUDA.downloadKeypoints(kpg)
    assert kp["x"].shape[0] == len(z["x"])
    for k in ("laplacian", "octave", "size"):
        np.testing.assert_array_equal(kp[k], z[k])
    for k in ("x", "y", "hessian"):
        np.testing.assert_allclose(kp[k], z[k], rtol=1e-6, atol=1e-4)
    da = np.abs(kp["angle"] - z["angle"]); da = np.minimum(da, 360 - da)
    assert (da <= 1e-2).mean() >= 0.95
    dd = np.abs(desc.cpu().numpy() - z["descriptors"]).max(1)
    assert ((dd <= 1e-4) | (da > 1e-2)).mean() >= 0.95
