"""StereoBM: oracle self-tests (CPU) and bit-exact HIP-vs-oracle parity (GPU) for cv::cuda::StereoBM.

Integer path: every comparison is exact (np.array_equal).  The oracle restates
modules/cudastereo/src/cuda/stereobm.cu sequentially (oracle/stereobm_ref.c)."""
import os

import numpy as np
import pytest

from opencv_contrib_amd import synth


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def _planes_pair(h, w, seed, disps=(12, 25, 40), noise=0):
    """left/right with piecewise-constant integer disparity bands (block matching recovers them exactly
    away from the band borders)."""
    left = np.rint(synth.texture(h, w, seed, 1.5)).astype(np.uint8)
    rng = np.random.default_rng(seed + 1)
    right = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    gt = np.zeros((h, w), np.int32)
    nb = len(disps)
    for k, d in enumerate(disps):
        y0, y1 = k * h // nb, (k + 1) * h // nb
        right[y0:y1, : w - d] = left[y0:y1, d:]
        gt[y0:y1] = d
    if noise:
        right = np.clip(right.astype(int) + rng.integers(-noise, noise + 1, size=right.shape), 0, 255).astype(np.uint8)
    return left, right, gt


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_recovers_piecewise_constant_disparity(oracle):
    left, right, gt = _planes_pair(120, 240, seed=5)
    R = 7
    d = oracle.sbm_block_match(left, right, ndisp=64, winsz=15)
    assert d.dtype == np.uint8 and d.shape == left.shape
    # outside the valid region the output stays 0 (stereobm.cu:506 memset)
    assert not d[:R].any() and not d[-R:].any() and not d[:, : 64 + R].any() and not d[:, -R:].any()
    for k, dd in enumerate((12, 25, 40)):
        y0, y1 = k * 40 + R + 1, (k + 1) * 40 - R - 1
        blk = d[y0:y1, 64 + R : 240 - R]
        assert (blk == dd).mean() > 0.999, (dd, (blk == dd).mean())


def test_oracle_tie_break_last_in_batch_first_batch(oracle):
    """Constant images: every SSD is 0 -> the reference keeps the LAST index of the FIRST batch of 8: 7
    (stereobm.cu:120-125 last-wins inside a batch, :349 strict < across batches)."""
    img = np.full((40, 120), 77, np.uint8)
    d = oracle.sbm_block_match(img, img, ndisp=32, winsz=9)
    assert (d[4:-4, 32 + 4 : -4] == 7).all()


def test_oracle_edge_emulation_only_touches_the_right_strip(oracle):
    left, right, _ = _planes_pair(60, 260, seed=9, noise=3)
    R = 5
    a = oracle.sbm_block_match(left, right, ndisp=32, winsz=11, emulate_edge=True)
    b = oracle.sbm_block_match(left, right, ndisp=32, winsz=11, emulate_edge=False)
    diff = np.argwhere(a != b)
    assert len(diff) > 0
    assert diff[:, 1].min() >= 260 - 2 * R and diff[:, 1].max() < 260 - R


def test_oracle_uniqueness_rejects_ambiguous_matches(oracle):
    # periodic texture (period 8 px): several disparities match equally well -> rejected
    row = np.tile(np.array([128, 198, 228, 198, 128, 57, 28, 57], np.uint8), 25)
    img = np.tile(row, (50, 1))
    right = np.roll(img, -4, axis=1)   # SSD == 0 at d = 4, 12, 20, 28
    plain = oracle.sbm_block_match(img, right, ndisp=32, winsz=9)
    uniq = oracle.sbm_block_match(img, right, ndisp=32, winsz=9, uniqueness_ratio=15)
    valid = (slice(4, -4), slice(32 + 4, -4 - 8))
    assert (plain[valid] == 4).all()
    assert not uniq[valid].any()


def test_oracle_prefilters(oracle):
    img = np.random.default_rng(3).integers(0, 256, size=(33, 47)).astype(np.uint8)
    xs = oracle.sbm_prefilter_xsobel(img, 31)
    assert xs.min() >= 0 and xs.max() <= 62
    flat = np.full((20, 30), 100, np.uint8)
    assert (oracle.sbm_prefilter_xsobel(flat, 31) == 31).all()
    nr = oracle.sbm_prefilter_norm(img, 31, 9)
    assert nr.min() >= 0 and nr.max() <= 62


def test_oracle_textureness_zeroes_flat_regions(oracle):
    img = np.full((60, 160), 90, np.uint8)
    img[:, 80:] = np.random.default_rng(1).integers(0, 256, size=(60, 80))
    disp = np.full((60, 160), 9, np.uint8)
    out = oracle.sbm_textureness(img, disp, winsz=11, avg_threshold=3.0)
    assert not out[10:-10, 5:60].any()          # flat part -> zeroed
    assert (out[10:-10, 100:150] == 9).all()    # textured part kept


def test_oracle_argument_errors(oracle):
    img = np.zeros((40, 100), np.uint8)
    with pytest.raises(ValueError):
        oracle.sbm_block_match(img, img, ndisp=12, winsz=9)      # ndisp % 8
    with pytest.raises(ValueError):
        oracle.sbm_block_match(img, img, ndisp=16, winsz=8)      # even window
    with pytest.raises(ValueError):
        oracle.sbm_block_match(img, img, ndisp=96, winsz=9)      # no valid column
    with pytest.raises(ValueError):
        oracle.sbm_block_match(img, img[:, :50], ndisp=16, winsz=9)


# ------------------------------------------------------------------ HIP vs oracle (GPU, bit-exact)
gpu_mark = pytest.mark.gpu


@gpu_mark
def test_wave_min_and_tie_break_semantics(gpu):
    """DPP row_shr/row_bcast reduction + ballot tie-break used by the block matcher."""
    import ctypes as C
    from opencv_contrib_amd import capi
    rng = np.random.default_rng(0)
    for trial in range(6):
        v = rng.integers(5, 1000, size=64).astype(np.uint32)
        if trial == 1:
            v[:] = 42                      # all equal -> lane 7 (last of the first batch)
        if trial == 2:
            v[[13, 14, 40, 47]] = 1        # first batch holding the min is batch 1 -> last index 14
        if trial == 3:
            v[63] = 0
        inp = (C.c_uint * 64)(*[int(x) for x in v])
        out = (C.c_uint * 65)()
        capi.check(capi.lib().miflow_selftest_wave_min(inp, out))
        m = int(v.min())
        assert all(out[i] == m for i in range(64)), (trial, list(out[:64]))
        idx = np.flatnonzero(v == m)
        b = idx[0] // 8
        want = max(i for i in idx if i // 8 == b)
        assert out[64] == want, (trial, out[64], want)


@gpu_mark
def test_transposed_max16_semantics(gpu):
    """16 values per lane -> lane j holds max over the 64 lanes of value (j & 15) (DPP quad_perm / row_shl+shr with
    bank masks / cross-row permutes used by the packed winner-take-all)."""
    import ctypes as C
    from opencv_contrib_amd import capi
    rng = np.random.default_rng(5)
    v = rng.integers(0, 2 ** 32, size=(16, 64), dtype=np.uint64).astype(np.uint32)
    inp = (C.c_uint * 1024)(*[int(x) for x in v.reshape(-1)])
    out = (C.c_uint * 64)()
    capi.check(capi.lib().miflow_selftest_tmax16(inp, out))
    want = v.max(axis=1)
    assert [int(out[l]) for l in range(64)] == [int(want[l & 15]) for l in range(64)]


@gpu_mark
@pytest.mark.parametrize("shape,ndisp,winsz", [((60, 200), 32, 9), ((97, 331), 64, 15), ((70, 300), 128, 19),
                                               ((64, 420), 48, 7), ((80, 400), 256, 5), ((120, 260), 64, 51), ((90, 300), 64, 25), ((90, 300), 64, 27), ((75, 290), 128, 11),
                                               ((50, 180), 8, 3)])
def test_block_match_bit_exact(gpu, oracle, shape, ndisp, winsz):
    h, w = shape
    left, right, _ = _planes_pair(h, w, seed=h + w, disps=(3, ndisp // 3, ndisp - 2), noise=2)
    ref, rssd = oracle.sbm_block_match(left, right, ndisp=ndisp, winsz=winsz, return_ssd=True)
    from opencv_contrib_amd import cuda
    d, ssd = cuda.stereobm_block_match(T(left, gpu), T(right, gpu), ndisp=ndisp, winsz=winsz)
    np.testing.assert_array_equal(N(d), ref)
    np.testing.assert_array_equal(N(ssd).view(np.uint32), rssd)


@gpu_mark
def test_block_match_ties_on_noise_free_constant_images(gpu, oracle):
    from opencv_contrib_amd import cuda
    img = np.full((48, 300), 77, np.uint8)
    ref = oracle.sbm_block_match(img, img, ndisp=128, winsz=9)
    d, _ = cuda.stereobm_block_match(T(img, gpu), T(img, gpu), ndisp=128, winsz=9)
    np.testing.assert_array_equal(N(d), ref)
    assert (ref[4:-4, 128 + 4 : -4] == 7).all()
    # pure noise: many near-ties
    rng = np.random.default_rng(4)
    a = rng.integers(0, 4, size=(60, 280)).astype(np.uint8)
    b = rng.integers(0, 4, size=(60, 280)).astype(np.uint8)
    ref = oracle.sbm_block_match(a, b, ndisp=64, winsz=5)
    d, _ = cuda.stereobm_block_match(T(a, gpu), T(b, gpu), ndisp=64, winsz=5)
    np.testing.assert_array_equal(N(d), ref)


@gpu_mark
@pytest.mark.parametrize("emulate", [True, False])
def test_block_match_edge_modes(gpu, oracle, emulate):
    from opencv_contrib_amd import cuda
    left, right, _ = _planes_pair(70, 64 + 2 * 7 + 128 * 2 + 37, seed=11, noise=3)   # several 128-wide reference blocks
    ref = oracle.sbm_block_match(left, right, ndisp=64, winsz=15, emulate_edge=emulate)
    d, _ = cuda.stereobm_block_match(T(left, gpu), T(right, gpu), ndisp=64, winsz=15, emulate_cuda_edge=emulate)
    np.testing.assert_array_equal(N(d), ref)


@gpu_mark
@pytest.mark.parametrize("ratio", [5, 15, 40])
@pytest.mark.parametrize("ndisp", [32, 128])
def test_block_match_uniqueness_bit_exact(gpu, oracle, ratio, ndisp):
    from opencv_contrib_amd import cuda
    left, right, _ = _planes_pair(72, 320, seed=21 + ratio, disps=(5, ndisp // 2, ndisp - 3), noise=6)
    ref = oracle.sbm_block_match(left, right, ndisp=ndisp, winsz=11, uniqueness_ratio=ratio)
    base = oracle.sbm_block_match(left, right, ndisp=ndisp, winsz=11)
    assert (ref != base).any(), "test input does not exercise the uniqueness rejection"
    d, _ = cuda.stereobm_block_match(T(left, gpu), T(right, gpu), ndisp=ndisp, winsz=11, uniqueness_ratio=ratio)
    np.testing.assert_array_equal(N(d), ref)


@gpu_mark
def test_prefilters_and_textureness_bit_exact(gpu, oracle):
    from opencv_contrib_amd import cuda
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, size=(131, 203)).astype(np.uint8)
    np.testing.assert_array_equal(N(cuda.stereobm_prefilter_xsobel(T(img, gpu), 31)), oracle.sbm_prefilter_xsobel(img, 31))
    for ws in (5, 9, 21):
        np.testing.assert_array_equal(N(cuda.stereobm_prefilter_norm(T(img, gpu), 31, ws)), oracle.sbm_prefilter_norm(img, 31, ws))
    img2 = np.rint(synth.texture(131, 203, 3, 4.0)).astype(np.uint8)
    img2[:, :70] = 120
    disp = rng.integers(1, 60, size=img2.shape).astype(np.uint8)
    for ws, thr in ((9, 3.0), (19, 3.0), (15, 10.0), (51, 3.0), (5, 6.0), (33, 4.0)):   # 51: the largest window (round 4: the one-launch form's LDS tile)
        ref = oracle.sbm_textureness(img2, disp, ws, thr)
        assert (ref != disp).any() and (ref == disp).any()
        np.testing.assert_array_equal(N(cuda.stereobm_textureness(T(img2, gpu), T(disp, gpu), ws, thr)), ref)


@gpu_mark
@pytest.mark.parametrize("prefilter", [-1, 0, 1])
def test_compute_matches_oracle_full_pipeline(gpu, oracle, prefilter):
    """createStereoBM(128, 19) as in cudastereo/test/test_stereo.cpp:64-119 (golden-PNG tests there;
    synthetic pair + oracle here), default textureness filter on; pitched device images."""
    import torch
    from opencv_contrib_amd import cuda
    left, right, _ = synth.stereo_pair(180, 420, seed=42, max_disp=40)
    p = oracle.sbm_params(num_disparities=128, block_size=19, prefilter_type=prefilter)
    ref = oracle.sbm_compute(left, right, p)
    bm = cuda.createStereoBM(128, 19)
    if prefilter >= 0:
        bm.setPreFilterType(prefilter)
    assert bm.getPreFilterType() == prefilter and bm.getNumDisparities() == 128 and bm.getBlockSize() == 19
    buf = torch.zeros((180, 512), dtype=torch.uint8, device=gpu)
    buf2 = torch.zeros((180, 448), dtype=torch.uint8, device=gpu)
    L, Rr = buf[:, 7:427], buf2[:, 3:423]
    L.copy_(T(left, gpu)); Rr.copy_(T(right, gpu))
    out = bm.compute(L, Rr)
    np.testing.assert_array_equal(N(out), ref)
    assert (ref > 0).mean() > 0.2


@gpu_mark
def test_compute_uniqueness_and_setters(gpu, oracle):
    from opencv_contrib_amd import cuda
    left, right, _ = synth.stereo_pair(150, 400, seed=7, max_disp=50)
    bm = cuda.createStereoBM(64, 15)
    bm.setUniquenessRatio(15)
    bm.setTextureThreshold(0)
    assert bm.getUniquenessRatio() == 15 and bm.getTextureThreshold() == 0 and bm.getMinDisparity() == 0
    ref = oracle.sbm_compute(left, right, oracle.sbm_params(num_disparities=64, block_size=15, uniqueness_ratio=15,
                                                            texture_threshold=0.0))
    np.testing.assert_array_equal(N(bm.compute(T(left, gpu), T(right, gpu))), ref)


@gpu_mark
def test_compute_argument_errors(gpu):
    import torch
    from opencv_contrib_amd import cuda, capi
    a = torch.zeros((60, 200), dtype=torch.uint8, device=gpu)
    with pytest.raises(capi.MiError):
        cuda.createStereoBM(12, 9).compute(a, a)          # ndisp % 8 (stereobm.cpp:145)
    with pytest.raises(capi.MiError):
        cuda.createStereoBM(16, 8).compute(a, a)          # even window (:146)
    with pytest.raises(capi.MiError):
        cuda.createStereoBM(16, 9).compute(a, a[:, :100])  # size mismatch (:152)
    with pytest.raises(capi.MiError):
        cuda.createStereoBM(16, 9).compute(a.float(), a.float())  # type (:151)
    with pytest.raises(capi.MiError):
        cuda.createStereoBM(256, 9).compute(a, a)         # no valid column


@gpu_mark
def test_1080p_config3_properties(gpu):
    """BASELINE configs[2] size (1920x1080, ndisp 128, block 15): the oracle takes ~10 s here, so check
    size-independent properties: determinism, zero frame outside the valid region, recovery of a known
    constant shift, and agreement of the two d-set waves with a 64-disparity run on d < 64."""
    import torch
    from opencv_contrib_amd import cuda
    left = np.rint(synth.texture(1080, 1920, 42, 1.5)).astype(np.uint8)
    right = np.roll(left, -37, axis=1)
    bm = cuda.createStereoBM(128, 15)
    bm.setTextureThreshold(0)
    d1 = N(bm.compute(T(left, gpu), T(right, gpu)))
    d2 = N(bm.compute(T(left, gpu), T(right, gpu)))
    np.testing.assert_array_equal(d1, d2)
    R = 7
    assert not d1[:R].any() and not d1[-R:].any() and not d1[:, : 128 + R].any() and not d1[:, -R:].any()
    assert (d1[R:-R, 128 + R : -R - 37] == 37).all()


@gpu_mark
def test_compute_batch_equals_single_computes(gpu):
    """mi_stereobm_compute_batch: n pairs through one handle, each equal to its own compute()."""
    import torch
    from opencv_contrib_amd import cuda
    pairs = [synth.stereo_pair(120, 260, seed=70 + k, max_disp=30)[:2] for k in range(4)]
    L, R = [T(p[0], gpu) for p in pairs], [T(p[1], gpu) for p in pairs]
    bm = cuda.createStereoBM(64, 11)
    out = bm.compute_batch(L, R)
    one = cuda.createStereoBM(64, 11)
    for k in range(4):
        assert torch.equal(out[k], one.compute(L[k], R[k]))
    assert not torch.equal(out[0], out[1])


@gpu_mark
@pytest.mark.parametrize("kw", [dict(), dict(uniqueness_ratio=10), dict(prefilter_type=1, texture_threshold=0), dict(prefilter_type=0, uniqueness_ratio=5)],
                         ids=["defaults", "uniqueness", "xsobel", "norm_uniqueness"])
@pytest.mark.parametrize("n", [2, 5])
def test_compute_batch_one_launch_block_matching(gpu, kw, n):
    """The batched block matching (blockIdx.z = pair, taller row bands) with every optional stage: uniqueness second pass reading the
    per-pair winner-SSD planes, both prefilters writing per-pair buffers, the textureness post-filter; pitched outputs; a handle
    reused for a smaller batch afterwards."""
    import torch
    from opencv_contrib_amd import cuda
    pairs = [synth.stereo_pair(150, 420, seed=90 + k, max_disp=40)[:2] for k in range(n)]
    L, R = [T(p[0], gpu) for p in pairs], [T(p[1], gpu) for p in pairs]
    bm = cuda.createStereoBM(64, 15)
    one = cuda.createStereoBM(64, 15)
    for alg in (bm, one):
        alg._set(**kw)
    out = bm.compute_batch(L, R)
    for k in range(n):
        assert torch.equal(out[k], one.compute(L[k], R[k])), k
    out2 = bm.compute_batch(L[:2], R[:2])
    assert torch.equal(out2[0], out[0]) and torch.equal(out2[1], out[1])


def _random_sbm_configs():
    rng = np.random.default_rng(int(os.environ.get("MIFLOW_SWEEP_SEED", "4242")))
    out = []
    for k in range(int(os.environ.get("MIFLOW_SWEEP_N", "24"))):
        nd = int((16, 32, 64, 72, 128, 200, 256)[int(rng.integers(7))])
        bs = int((5, 9, 11, 15, 19, 21, 31, 51)[int(rng.integers(8))])
        h = int(rng.integers(bs + 8, 260))
        w = int(rng.integers(nd + bs + 8, nd + bs + 400))
        out.append(dict(shape=(h, w), seed=int(rng.integers(1, 10 ** 6)), nd=nd, bs=bs, uniq=int((0, 0, 5, 15)[int(rng.integers(4))]),
                        pre=int((-1, -1, 0, 1)[int(rng.integers(4))]), tex=float((0.0, 3.0, 10.0)[int(rng.integers(3))]),
                        batch=int((1, 2, 5)[int(rng.integers(3))]), noise=bool(rng.integers(4) == 0)))
    return out


@gpu_mark
@pytest.mark.parametrize("cfg", _random_sbm_configs(), ids=lambda c: f"{c['shape'][0]}x{c['shape'][1]}-d{c['nd']}-b{c['bs']}-u{c['uniq']}-p{c['pre']}"
                                                                    f"-t{c['tex']}-n{c['batch']}")
def test_random_configuration_bit_exact(gpu, oracle, cfg):
    """Seeded sweep: image sizes down to the smallest the parameters allow (one tile, one band), every disparity-range / block
    size family of the kernel (packed and generic winner search, one to four disparity sets), uniqueness, both prefilters,
    textureness on / off, 2-bit-noise images (ties everywhere), single pairs and batches: bit-exact against the oracle."""
    import torch
    from opencv_contrib_amd import cuda
    pairs = []
    for b in range(cfg["batch"]):
        if cfg["noise"]:
            rng = np.random.default_rng(cfg["seed"] + b)
            pairs.append((rng.integers(0, 4, size=cfg["shape"]).astype(np.uint8), rng.integers(0, 4, size=cfg["shape"]).astype(np.uint8)))
        else:
            pairs.append(synth.stereo_pair(*cfg["shape"], seed=cfg["seed"] + b, max_disp=min(cfg["nd"] - 2, 60))[:2])
    p = oracle.sbm_params(num_disparities=cfg["nd"], block_size=cfg["bs"], uniqueness_ratio=cfg["uniq"], prefilter_type=cfg["pre"],
                          texture_threshold=cfg["tex"])
    bm = cuda.createStereoBM(cfg["nd"], cfg["bs"])
    bm._set(uniqueness_ratio=cfg["uniq"], prefilter_type=cfg["pre"], texture_threshold=cfg["tex"])
    L, R = [T(q[0], gpu) for q in pairs], [T(q[1], gpu) for q in pairs]
    out = [bm.compute(L[0], R[0])] if cfg["batch"] == 1 else list(bm.compute_batch(L, R))
    for b, q in enumerate(pairs):
        np.testing.assert_array_equal(out[b].cpu().numpy(), oracle.sbm_compute(q[0], q[1], p), err_msg=f"pair {b}")
