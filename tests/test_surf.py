"""SURF: oracle self-tests (CPU) and HIP-vs-oracle parity (GPU) for cv::cuda::SURF_CUDA.

Stated tolerances:
  * integral image, candidate set, laplacian/octave/size rows: exact;
  * det/trace and x/y/hessian rows: same double/float operations in the same order on both sides -> exact or 1 ulp
    (assert_allclose rtol 1e-6);
  * angle: atan2f of the device vs glibc may differ by 1 ulp, which can move a sample across a 5-degree window edge
    -> >= 99 % of the features within 1e-2 degrees;
  * descriptors: >= 99 % of the features with max |diff| <= 1e-4 (sincosf ulp differences can flip a
    nearest-texel read for a few samples).
The reference's own CUDA-vs-CPU acceptance is far looser (matched-keypoint ratio > 0.95, descriptor match ratio > 0.6,
xfeatures2d/test/test_surf.cuda.cpp:102-107,166-173).
"""
import os

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
def test_oracle_tables_match_reference_literals(oracle):
    """Generated tables vs literal entries of the reference's constant tables (xfeatures2d/src/cuda/surf.cu:520-522,
    685-707): c_aptX[0..8], c_aptY[0..8], c_aptW[0], c_aptW[56] (centre), c_DW[0], c_DW[21], c_DW[210]."""
    ax, ay, aw, dw = oracle.surf_tables()
    np.testing.assert_array_equal(ax[:9], [-6, -5, -5, -5, -5, -5, -5, -5, -4])
    np.testing.assert_array_equal(ay[:9], [0, -3, -2, -1, 0, 1, 2, 3, -4])
    assert ax[-1] == 6 and ay[-1] == 0 and len(ax) == 113
    # bit for bit (round 3: the generator reproduces the literals' own rounding; all 113 + 400 entries are compared against the parsed
    # reference file in tests/test_ref_pin_cuda.py::test_surf_weight_tables_equal_the_literals_of_surf_cu)
    np.testing.assert_array_equal([aw[0], aw[56]], np.array([0.001455130288377404, 0.02592208795249462], np.float32))
    np.testing.assert_array_equal([dw[0], dw[21], dw[210]], np.array([3.695352233989979e-06, 1.929736572492402e-05, 0.01435048412531614], np.float32))


def test_oracle_integral(oracle):
    img = np.random.default_rng(0).integers(0, 256, size=(37, 53)).astype(np.uint8)
    s = oracle.surf_integral(img)
    ref = np.zeros((38, 54), np.uint64)
    ref[1:, 1:] = img.astype(np.uint64).cumsum(0).cumsum(1)
    np.testing.assert_array_equal(s, ref.astype(np.uint32))


def test_oracle_cross_fixture(oracle):
    """A self-contained SURF fixture of the reference (xfeatures2d/test/test_rotation_and_scale_invariance.cpp:259-285; the one
    with known answers, the Java tests' cross, is in tests/test_zz_surf_cpu_class.py:
    100x100 white image, two 3-px dark bars, SURF(8000, 3, 4, extended, upright=false)).  The CPU class reports 5
    keypoints there; under the CUDA class's strict 26-neighbour maximum the centre is a plateau (det = 15376 on a 3x3
    patch of layer 1) and only the 4 symmetric keypoints survive -- with equal responses, as that test requires."""
    img = synth.cross_image()
    r = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=8000, n_octaves=3, n_octave_layers=4, extended=1,
                                                            keypoints_ratio=0.05))
    assert r["n"] == 4
    assert np.ptp(r["hessian"]) <= 1e-6 * r["hessian"][0]
    c = np.stack([r["x"], r["y"]], 1)
    np.testing.assert_allclose(np.abs(c - 50.0), np.abs(c[0] - 50.0)[None].repeat(4, 0), rtol=1e-6)   # 4-fold symmetric about (50,50)
    assert set(map(tuple, np.sign(c - 50.0).astype(int))) == {(-1, -1), (1, -1), (-1, 1), (1, 1)}


def test_oracle_blobs_scale_and_descriptor_norm(oracle):
    img = synth.blob_image(240, 320, seed=7)
    r = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=400))
    assert r["n"] > 50 and len(np.unique(r["octave"])) >= 2
    np.testing.assert_allclose(np.linalg.norm(r["descriptors"], axis=1), 1.0, atol=1e-5)
    assert ((r["angle"] >= 0) & (r["angle"] < 360)).all()
    r2 = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=400, upright=1, extended=1))
    assert (r2["angle"] == 270).all() and r2["descriptors"].shape[1] == 128 and r2["n"] == r["n"]


def test_oracle_mask_and_errors(oracle):
    img = synth.blob_image(200, 260, seed=3)
    full = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=300), want_desc=False)
    mask = np.zeros_like(img); mask[:, :130] = 255
    half = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=300), mask=mask, want_desc=False)
    assert 0 < half["n"] < full["n"] and half["x"].max() < 140
    with pytest.raises(ValueError):
        oracle.surf_detect_describe(img[:60, :60], oracle.surf_params())          # too small for 4 octaves (surf.cuda.cpp:144-151)
    with pytest.raises(ValueError):
        oracle.surf_detect_describe(img, oracle.surf_params(n_octaves=0))


# ------------------------------------------------------------------ HIP vs oracle (GPU)
gpu_mark = pytest.mark.gpu


@gpu_mark
def test_wave_scan_semantics(gpu):
    import ctypes as C
    from opencv_contrib_amd import capi
    v = np.random.default_rng(1).integers(0, 1000, size=64).astype(np.uint32)
    inp = (C.c_uint * 64)(*[int(x) for x in v]); out = (C.c_uint * 64)()
    capi.check(capi.lib().miflow_selftest_wave_scan(inp, out))
    np.testing.assert_array_equal(np.array(out[:], np.uint64), np.cumsum(v.astype(np.uint64)))


@gpu_mark
@pytest.mark.parametrize("shape", [(37, 53), (100, 64), (481, 1283)])
def test_integral_bit_exact(gpu, oracle, shape):
    from opencv_contrib_amd import cuda
    img = np.random.default_rng(2).integers(0, 256, size=shape).astype(np.uint8)
    np.testing.assert_array_equal(N(cuda.surf_integral(T(img, gpu))).view(np.uint32), oracle.surf_integral(img))
    m = (img > 100).astype(np.uint8) * 255
    np.testing.assert_array_equal(N(cuda.surf_integral(T(m, gpu), True)).view(np.uint32), oracle.surf_integral(np.minimum(m, 1)))


@gpu_mark
@pytest.mark.parametrize("octave", [0, 1, 2])
def test_det_trace_matches_oracle(gpu, oracle, octave):
    from opencv_contrib_amd import cuda
    img = synth.blob_image(200, 264, seed=5)
    S = oracle.surf_integral(img)
    rd, rt = oracle.surf_det_trace(S, octave, 2)
    gd, gt = cuda.surf_detTrace(T(S.view(np.int32), gpu), octave, 2)
    lc = 264 >> octave
    np.testing.assert_allclose(N(gd)[:, :lc], rd[:, :lc], rtol=1e-6, atol=1e-3)
    np.testing.assert_allclose(N(gt)[:, :lc], rt[:, :lc], rtol=1e-6, atol=1e-3)


def _compare(kp, desc, ref, check_desc=True):
    assert kp["x"].shape[0] == ref["n"], (kp["x"].shape[0], ref["n"])
    for k in ("laplacian", "octave", "size"):
        np.testing.assert_array_equal(kp[k], ref[k], err_msg=k)
    for k in ("x", "y", "hessian"):
        np.testing.assert_allclose(kp[k], ref[k], rtol=1e-6, atol=1e-4, err_msg=k)
    d = np.abs(kp["angle"] - ref["angle"]); d = np.minimum(d, 360 - d)
    # >= 99 % of the features, or all but two on small sets (a fast-math angle / a sign decision of the extended descriptor on a value
    # within rounding of zero flips a bin)
    assert (d <= 1e-2).mean() >= 0.99 or (d > 1e-2).sum() <= 2, (d > 1e-2).sum()
    if check_desc:
        dd = np.abs(desc - ref["descriptors"]).max(1)
        ok = (dd <= 1e-4) | (d > 1e-2)     # a feature whose angle differs is allowed a different descriptor
        assert ok.mean() >= 0.99 or (~ok).sum() <= 2, (float(dd.max()), int((~ok).sum()))


@gpu_mark
@pytest.mark.parametrize("thr,octaves,layers,extended,upright", [(100, 4, 2, False, False), (500, 3, 3, True, False),
                                                                 (1000, 4, 2, False, True), (100, 3, 2, True, True)])
def test_detect_and_describe_match_oracle(gpu, oracle, thr, octaves, layers, extended, upright):
    """Parameter grid after xfeatures2d/test/test_surf.cuda.cpp:176-187, keypointsRatio 0.05 as there (:92)."""
    from opencv_contrib_amd import cuda
    img = synth.blob_image(300, 400, seed=11)
    ref = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=thr, n_octaves=octaves, n_octave_layers=layers,
                                                              extended=int(extended), upright=int(upright), keypoints_ratio=0.05))
    assert ref["n"] > 20
    alg = cuda.SURF_CUDA.create(thr, octaves, layers, extended, 0.05, upright)
    kpg, desc = alg.detectWithDescriptors(T(img, gpu))
    assert desc.shape == (ref["n"], alg.descriptorSize()) and alg.defaultNorm() == 4
    _compare(cuda.SURF_CUDA.downloadKeypoints(kpg), N(desc), ref)


@gpu_mark
@pytest.mark.parametrize("octaves,layers", [(3, 5), (4, 4), (2, 1)])
def test_detect_per_octave_and_all_octave_launch_plans_agree_with_the_oracle(gpu, oracle, octaves, layers):
    """Round 3: one launch per detector stage covers all octaves where the kernel arguments hold them (<= 6 octaves, nOctaveLayers + 2
    <= 6 layers: every default); otherwise the octaves run one after the other through one set of planes (the round-2 plan).
    (3, 5) takes the per-octave plan, (4, 4) and (2, 1) the all-octave plan at its layer limit / smallest size; one handle re-used
    across a change of the plan must re-size its scratch."""
    from opencv_contrib_amd import cuda
    img = synth.blob_image(300, 400, seed=21)
    for (o, l) in ((octaves, layers), (2, 2)):
        ref = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=100.0, n_octaves=o, n_octave_layers=l, keypoints_ratio=0.05))
        alg = cuda.SURF_CUDA.create(100.0, o, l, False, 0.05, False)
        kpg, desc = alg.detectWithDescriptors(T(img, gpu))
        assert ref["n"] > 20
        _compare(cuda.SURF_CUDA.downloadKeypoints(kpg), N(desc), ref)


@gpu_mark
@pytest.mark.parametrize("shape,thr,extended", [((300, 400), 100.0, False), ((1080, 1920), 400.0, False), ((720, 1283), 50.0, True)])
def test_detect_and_compute_with_the_count_on_the_device_is_bit_identical(gpu, shape, thr, extended):
    """Round 5: `mi_surf_detect_and_compute` leaves the feature count on the device between detectKeypoints and
    computeDescriptors (the reference reads keypoints.cols back in between, surf.cuda.cpp:205-209) -- the descriptor kernels read it
    themselves and a fixed grid walks the features.  Keypoints and descriptors must equal, bit for bit, detect() followed by
    compute_descriptors with the host-known count (one workgroup per feature); overflow (tiny keypointsRatio) and a mask included."""
    from opencv_contrib_amd import cuda
    img = synth.blob_image(*shape, seed=31)
    t = T(img, gpu)
    for ratio, mask in ((0.05 if shape[0] < 1000 else 0.01, None), (0.0005, None), (0.05 if shape[0] < 1000 else 0.01, "m")):
        m = None
        if mask:
            mm = np.zeros_like(img); mm[shape[0] // 8: shape[0] // 2, shape[1] // 6: shape[1] - 40] = 1
            m = T(mm, gpu)
        alg = cuda.SURF_CUDA.create(thr, 4, 2, extended, ratio, False)
        kp1, d1 = alg.detectWithDescriptors(t, m)
        kp0 = alg.detect(t, m)
        assert kp0.shape == kp1.shape and kp0.shape[1] > 5
        np.testing.assert_array_equal(N(kp0), N(kp1))
        _, d0 = alg.detectWithDescriptors(t, None, kp0.clone(), True)   # provided keypoints: host-known count, one workgroup per feature
        assert d0.shape == d1.shape == (kp0.shape[1], alg.descriptorSize())
        np.testing.assert_array_equal(N(d0), N(d1))


@gpu_mark
def test_maxima_flagged_inside_the_det_kernel_are_bit_identical(gpu):
    """Round 5: octave 0's det / trace values stay in the LDS tile of the kernel that computes them, the 26
    comparisons run there and only the flag words (+ the sign of the trace at each maximum) leave it; the sub-pixel refinement
    evaluates its 27 values from the integral image (MIFLOW_SURF_NMS0=1: an opt-in -- measured slower than the plane form, see
    surf_api.cpp).  Keypoints (order included) and descriptors must not change by a bit against the default plane form -- 4K at the BASELINE setting, small and odd sizes, one
    layer, a mask, the overflow of the candidate list.  The switch is read once per process, hence the subprocesses."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for tag, env in (("fused", {"MIFLOW_SURF_NMS0": "1"}), ("planes", {}), ("strided", {"MIFLOW_SURF_POLY": "0"})):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "surf_digest.py")], capture_output=True, text=True,
                           env=dict(os.environ, **env), timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = re.findall(r"n=(\d+) digest ([0-9a-f]{16})", r.stdout)
        assert len(out[tag]) == 5 and all(int(n) > 5 for n, _ in out[tag]), r.stdout
    # ... and the polyphase planes of the integral image (octaves >= 1 read consecutive words, the default since round 5) against the
    # strided gathers (MIFLOW_SURF_POLY=0): the same integers, the same det / trace values
    assert out["fused"] == out["planes"] == out["strided"], out


@gpu_mark
def test_detect_mask_overflow_and_provided_keypoints(gpu, oracle):
    from opencv_contrib_amd import cuda
    img = synth.blob_image(260, 330, seed=13)
    mask = np.zeros_like(img); mask[40:200, 30:220] = 7
    p = oracle.surf_params(hessian_threshold=200, keypoints_ratio=0.05)
    ref = oracle.surf_detect_describe(img, p, mask=mask)
    alg = cuda.SURF_CUDA.create(200, _keypointsRatio=0.05)
    kpg = alg.detect(T(img, gpu), T(mask, gpu))
    _compare(cuda.SURF_CUDA.downloadKeypoints(kpg), None, ref, check_desc=False)
    # overflow: tiny keypointsRatio -> deterministic prefix of the scan-ordered feature list
    full = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=50, keypoints_ratio=0.05), want_desc=False)
    few = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=50, keypoints_ratio=0.0004), want_desc=False)
    assert few["n"] == int(np.float32(260 * 330) * np.float32(0.0004)) < full["n"]
    kpf = cuda.SURF_CUDA.downloadKeypoints(cuda.SURF_CUDA.create(50, _keypointsRatio=0.0004).detect(T(img, gpu)))
    np.testing.assert_allclose(kpf["x"], few["x"], rtol=1e-6, atol=1e-4)
    # useProvidedKeypoints: orientation + descriptors recomputed for the given keypoints
    ref2 = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=200, keypoints_ratio=0.05))
    kpg2 = alg.detect(T(img, gpu))
    kpg2c = kpg2.clone(); kpg2c[5] = 0
    kpo, desc = alg.detectWithDescriptors(T(img, gpu), None, kpg2c, True)
    _compare(cuda.SURF_CUDA.downloadKeypoints(kpo), N(desc), ref2)


@gpu_mark
def test_cross_fixture_and_errors(gpu, oracle):
    import torch
    from opencv_contrib_amd import cuda, capi
    img = synth.cross_image()
    alg = cuda.SURF_CUDA.create(8000, 3, 4, True, 0.05, False)
    kp = cuda.SURF_CUDA.downloadKeypoints(alg.detect(T(img, gpu)))
    assert kp["x"].shape[0] == 4 and np.ptp(kp["hessian"]) <= 1e-6 * kp["hessian"][0]
    with pytest.raises(capi.MiError):
        cuda.SURF_CUDA.create(100).detect(T(img[:60, :60], gpu))                  # too small for 4 octaves
    with pytest.raises(capi.MiError):
        cuda.SURF_CUDA.create(100).detect(T(img.astype(np.float32), gpu))         # CV_8UC1 only (surf.cuda.cpp:139)
    with pytest.raises(capi.MiError):
        cuda.SURF_CUDA.create(100).detect(T(img, gpu), T(img[:50], gpu))          # mask size (:140)
    # determinism: two runs, identical bytes
    b = synth.blob_image(300, 400, seed=4)
    a1, alg2 = cuda.SURF_CUDA.create(100), cuda.SURF_CUDA.create(100)
    k1, d1 = a1.detectWithDescriptors(T(b, gpu)); k2, d2 = alg2.detectWithDescriptors(T(b, gpu))
    # compare bit patterns: the LAPLACIAN row stores int -1 = 0xFFFFFFFF, a NaN when read as float
    assert torch.equal(k1.view(torch.int32), k2.view(torch.int32)) and torch.equal(d1, d2)


def _random_surf_configs():
    rng = np.random.default_rng(int(os.environ.get("MIFLOW_SWEEP_SEED", "90210")))
    out = []
    for k in range(int(os.environ.get("MIFLOW_SWEEP_N", "12"))):
        out.append(dict(shape=(int(rng.integers(120, 520)), int(rng.integers(160, 700))), seed=int(rng.integers(1, 10 ** 6)),
                        thr=float((50.0, 100.0, 400.0, 1500.0)[int(rng.integers(4))]), octaves=int(rng.integers(2, 5)), layers=int(rng.integers(1, 4)),
                        extended=bool(rng.integers(2)), upright=bool(rng.integers(2))))
    return out


@gpu_mark
@pytest.mark.parametrize("cfg", _random_surf_configs(), ids=lambda c: f"{c['shape'][0]}x{c['shape'][1]}-t{int(c['thr'])}-o{c['octaves']}l{c['layers']}"
                                                                     f"-e{int(c['extended'])}u{int(c['upright'])}")
def test_random_configuration_matches_oracle(gpu, oracle, cfg):
    """Seeded sweep over image sizes (band / chunk edges of the integral, NMS rows, octave sub-sampling remainders), thresholds,
    octave and layer counts, 64 / 128-element descriptors, oriented and upright: same keypoint set, >= 99 % of the orientations and
    descriptors within the stated tolerances."""
    from opencv_contrib_amd import cuda
    from opencv_contrib_amd import capi
    img = synth.blob_image(*cfg["shape"], seed=cfg["seed"])
    alg = cuda.SURF_CUDA.create(cfg["thr"], cfg["octaves"], cfg["layers"], cfg["extended"], 0.05, cfg["upright"])
    try:
        ref = oracle.surf_detect_describe(img, oracle.surf_params(hessian_threshold=cfg["thr"], n_octaves=cfg["octaves"], n_octave_layers=cfg["layers"],
                                                                  extended=int(cfg["extended"]), upright=int(cfg["upright"]), keypoints_ratio=0.05))
    except ValueError:
        # the image is too small for the last octave's filters: CV_Assert(layer_rows - 2 * min_margin > 0), surf.cuda.cpp:154-156
        with pytest.raises(capi.MiError):
            alg.detect(T(img, gpu))
        return
    if ref["n"] == 0:
        kpg = alg.detect(T(img, gpu))
        assert cuda.SURF_CUDA.downloadKeypoints(kpg)["x"].shape[0] == 0
        return
    kpg, desc = alg.detectWithDescriptors(T(img, gpu))
    _compare(cuda.SURF_CUDA.downloadKeypoints(kpg), N(desc), ref)


def test_box_division_by_reciprocal_is_exact(tmp_path):
    """k_det_trace (round 3) divides a box sum by its area as q = a y, q += fma(-q, b, a) y with y = RN(1 / b) instead of a / b.  For
    the integers that occur -- |a| = box sum x weight < 2^35, b = box area < 2^24 -- this must BE the correctly rounded quotient,
    or det / trace would stop being bit-identical to the reference's double-precision division (surf.cu:148): every area of every
    filter size of 5 octaves x 6 layers against 20 000 dividends each (incl. near multiples), and 2e7 random (a, b)."""
    import subprocess
    src = tmp_path / "divtest.c"
    src.write_text(r'''
#include <math.h>
#include <stdio.h>
#include <stdint.h>
static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static long check(double a, double b)
{
    const double y = 1.0 / b, q = a * y;
    return fma(fma(-q, b, a), y, q) != a / b;
}
int main(void)
{
    long bad = 0, n = 0;
    for (int octave = 0; octave < 5; ++octave)
        for (int layer = 0; layer < 6; ++layer) {
            const int size = (9 + 6 * layer) << octave;
            const float ratio = (float)size / 9;
            int e[10];
            for (int k = 0; k < 10; ++k) e[k] = (int)rintf(ratio * (float)k);
            const int areas[6] = {(e[3] - e[0]) * (e[7] - e[2]), (e[6] - e[3]) * (e[7] - e[2]), (e[9] - e[6]) * (e[7] - e[2]),
                                  (e[4] - e[1]) * (e[4] - e[1]), (e[8] - e[5]) * (e[4] - e[1]), (e[8] - e[5]) * (e[8] - e[5])};
            for (int ai = 0; ai < 6; ++ai)
                for (int t = 0; t < 20000; ++t) {
                    int64_t a = (int64_t)(rnd() % (1ull << 35)) - (1ll << 34);
                    if (t < 200) a = (int64_t)areas[ai] * (t - 100) * 977 + (t % 3) - 1;
                    bad += check((double)a, (double)areas[ai]); ++n;
                }
        }
    for (long t = 0; t < 20000000; ++t) {
        const int64_t a = (int64_t)(rnd() % (1ull << 35)) - (1ll << 34);
        bad += check((double)a, (double)(1 + rnd() % (1u << 24))); ++n;
    }
    printf("%ld %ld\n", n, bad);
    return bad != 0;
}
''')
    exe = str(tmp_path / "divtest")
    r = subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", str(src), "-o", exe, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    n, bad = (int(x) for x in r.stdout.split())
    assert r.returncode == 0 and n > 2.3e7 and bad == 0, r.stdout


@gpu_mark
def test_staged_descriptor_kernel_is_bit_identical_to_the_global_memory_kernel(gpu):
    """Round 3: features with a cell side s >= 5 build their 21 x 21 patch in `k_descriptors_staged` -- the texels of a strip of patch rows
    staged through LDS by lanes arranged as 8 x 8 blocks of the rotated window lattice, then accumulated per sample in the reference's
    order -- instead of every thread walking its own cell through global memory.  Same texel per (dy, dx), same order of the float
    adds: the descriptors must be identical to the last bit.  tools/surf_stage_check.py prints digests of the 64- and 128-float
    descriptors of the 4K blob frame and of 300 provided keypoints of sizes 4 .. 700 px (cells from below the threshold to larger than a
    tile holds, which stay on the global path); MIFLOW_SURF_STAGE_S is read once per process, hence two subprocesses."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for stage in ("0", "5"):
        env = dict(os.environ, MIFLOW_SURF_STAGE_S=stage)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "surf_stage_check.py")], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith(("4K blob", "provided sizes"))])
    assert len(outs[0]) == 4 and all(l.endswith("True") for l in outs[0][2:])
    assert outs[0] == outs[1]
