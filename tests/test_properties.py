"""Property tests (hypothesis) of the oracles and the host-side logic: invariants that must hold for ANY input, on small random cases
with many exact ties, empty / ragged shapes and masks.  CPU only."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from opencv_contrib_amd import flowio, parallel

FLT_MAX = np.finfo(np.float32).max
FAST = settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


def _ints(rng, shape, hi):
    return rng.integers(0, hi, shape).astype(np.float32)       # integer-valued floats: distances tie exactly


@FAST
@given(seed=st.integers(0, 2 ** 31 - 1), nq=st.integers(1, 9), nts=st.lists(st.integers(1, 12), min_size=1, max_size=4),
       d=st.integers(1, 9), k=st.integers(1, 15), norm=st.sampled_from([2, 4]), masked=st.booleans())
def test_matcher_oracle_invariants(oracle, seed, nq, nts, d, k, norm, masked):
    rng = np.random.default_rng(seed)
    q = _ints(rng, (nq, d), 3)
    trains = [_ints(rng, (n, d), 3) for n in nts]
    masks = [(rng.random((nq, n)) < 0.6).astype(np.uint8) for n in nts] if masked else None
    idx, img, dist = oracle.bf_knn_match(q, trains, k, norm, masks)
    total = sum(nts)
    full = oracle.bf_knn_match(q, trains, total, norm, masks)      # every candidate, in the matcher's order
    for i in range(nq):
        n_valid = int((idx[i] >= 0).sum())
        allowed = total if masks is None else int(sum(m[i].sum() for m in masks))
        assert n_valid == min(k, allowed)                                            # short lists end in (-1, -1, FLT_MAX)
        assert (idx[i, n_valid:] == -1).all() and (img[i, n_valid:] == -1).all() and (dist[i, n_valid:] == FLT_MAX).all()
        key = [(float(dist[i, j]), int(img[i, j]), int(idx[i, j])) for j in range(n_valid)]
        assert key == sorted(key)                                                    # ordered by (distance, image, index)
        assert len(set(key)) == len(key)
        np.testing.assert_array_equal(idx[i, :n_valid], full[0][i, :n_valid])         # k-list = prefix of the complete list
        np.testing.assert_array_equal(dist[i, :n_valid], full[2][i, :n_valid])
        if masks is not None:
            for j in range(n_valid):
                assert masks[img[i, j]][i, idx[i, j]] != 0                            # masked pairs never appear
        # a collection is the concatenation of its images
        if masks is None:
            cat = oracle.bf_knn_match(q[i:i + 1], np.concatenate(trains), min(k, total), norm)
            offs = np.cumsum([0] + nts)
            flat = [int(offs[img[i, j]] + idx[i, j]) for j in range(min(n_valid, cat[0].shape[1]))]
            assert flat == cat[0][0, :len(flat)].tolist()


@FAST
@given(seed=st.integers(0, 2 ** 31 - 1), nq=st.integers(1, 6), nt=st.integers(1, 30), d=st.integers(1, 6), cols=st.integers(1, 12),
       norm=st.sampled_from([2, 4]), radius=st.floats(0.0, 6.0))
def test_radius_oracle_is_the_thresholded_scan(oracle, seed, nq, nt, d, cols, norm, radius):
    rng = np.random.default_rng(seed)
    q, t = _ints(rng, (nq, d), 3), _ints(rng, (nt, d), 3)
    full = oracle.bf_knn_match(q, t, nt, norm)
    dall = np.empty((nq, nt), np.float32)
    np.put_along_axis(dall, full[0], full[2], 1)
    idx, img, dist, n = oracle.bf_radius_match(q, t, radius, cols, norm)
    for i in range(nq):
        hits = np.nonzero(dall[i] < np.float32(radius))[0]
        assert n[i] == len(hits)                                                     # counts every hit, even past `cols`
        m = min(len(hits), cols)
        np.testing.assert_array_equal(idx[i, :m], hits[:m])                          # ascending train order, first `cols` kept
        np.testing.assert_array_equal(dist[i, :m], dall[i, hits[:m]])
        assert (idx[i, m:] == -1).all()                                              # the rest untouched


@FAST
@given(h=st.integers(1, 9), w=st.integers(1, 9), seed=st.integers(0, 10 ** 6))
def test_flo_files_round_trip(tmp_path_factory, h, w, seed):
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((h, w, 2)).astype(np.float32) * 50
    if h * w > 2:
        f[0, 0, 0] = np.nan
        f[-1, -1, 1] = 1e10                                    # the reference's "unknown flow" markers
    p = str(tmp_path_factory.mktemp("flo") / "x.flo")
    flowio.writeOpticalFlow(p, f)
    g = flowio.readOpticalFlow(p)
    assert g.shape == f.shape and g.dtype == np.float32
    np.testing.assert_array_equal(np.isnan(g), np.isnan(f))
    np.testing.assert_array_equal(g[~np.isnan(g)], f[~np.isnan(f)])
    ok = flowio.isFlowCorrect(g)
    assert ok.sum() == int((~np.isnan(f).any(-1) & (np.abs(f) < 1e9).all(-1)).sum())
    assert flowio.accuracy(f, g) == 1.0                         # a flow is as accurate as itself wherever it is valid


@FAST
@given(n=st.integers(0, 600), world=st.integers(1, 9))
def test_shards_partition_the_batch_in_order(n, world):
    parts = [parallel.shard_range(n, world, r) for r in range(world)]
    assert [i for p in parts for i in p] == list(range(n))
    sizes = [len(p) for p in parts]
    assert max(sizes) <= -(-n // world)                                               # block partition: ceil(n / world) per rank at most
    assert all(a >= b for a, b in zip(sizes, sizes[1:]))                              # only trailing ranks run short


@FAST
@given(seed=st.integers(0, 10 ** 6), h=st.integers(1, 6), w=st.integers(1, 7), cn=st.sampled_from([1, 3, 4]),
       dtype=st.sampled_from(["uint8", "uint16", "float32"]))
def test_superres_gray8_oracle_properties(oracle, seed, h, w, cn, dtype):
    rng = np.random.default_rng(seed)
    shape = (h, w) if cn == 1 else (h, w, cn)
    if dtype == "float32":
        x = rng.random(shape).astype(np.float32)
    else:
        x = rng.integers(0, np.iinfo(dtype).max + 1, shape).astype(dtype)
    g = oracle.superres_to_gray8(x)
    assert g.shape == (h, w) and g.dtype == np.uint8
    if cn > 1:                                                   # a grey image in colour form converts like the grey image
        grey = x[..., 0]
        rep = np.stack([grey] * cn, -1) if dtype != "float32" else None
        if rep is not None:
            np.testing.assert_array_equal(oracle.superres_to_gray8(rep), oracle.superres_to_gray8(grey))
    if dtype == "uint8" and cn == 1:
        np.testing.assert_array_equal(g, x)


@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_identical_frames_give_exactly_zero_flow(oracle, dtype):
    """Zero motion is a fixed point of the iterations: rho vanishes identically for TV-L1 (both semantics), the residual for PyrLK.
    Farneback is NOT exactly zero there: its matrix update drops the second frame's polynomial at the last row / column
    (farneback.cu:176: `x1 < width - 1 && y1 < height - 1`), an asymmetry of the reference that leaks a sub-pixel flow inwards from the border."""
    from opencv_contrib_amd import synth
    I0, _, _ = synth.flow_pair(72, 96, seed=3, dtype=dtype)
    for sem in (0, 1):
        f = oracle.tvl1_calc(I0, I0, oracle.tvl1_params(iterations=8, epsilon=0.0, semantics=sem))
        assert np.abs(f).max() == 0.0
    if dtype == "u8":
        assert np.abs(oracle.pyrlk_dense(I0, I0)).max() == 0.0
        fb = np.abs(oracle.fb_calc(I0, I0)).max(-1)
        assert fb[12:-12, 12:-12].max() < 1e-3 and 0 < fb.max() < 0.5


def test_xcd_contiguous_tile_order_is_a_permutation():
    """The workgroup remap of k_iterate_tile / k_warp6 / k_block_match / k_iterate_tbr (round 4: `lid` from the launch-order index
    `orig`; XCD = orig mod 8 takes a contiguous run of tiles): for every grid size it must be a bijection of [0, nwg) -- every tile
    computed exactly once -- and the tiles of one XCD must be consecutive."""
    import numpy as np
    for nwg in list(range(1, 300)) + [1020, 1023, 1024, 1025, 4400, 35937]:
        orig = np.arange(nwg, dtype=np.int64)
        xcd, qq, rr = orig & 7, nwg >> 3, nwg & 7
        lid = np.where(xcd < rr, xcd * (qq + 1), rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3)
        assert np.array_equal(np.sort(lid), orig), nwg
        for k in range(min(8, nwg)):
            mine = np.sort(lid[xcd == k])
            assert len(mine) == 0 or np.array_equal(mine, np.arange(mine[0], mine[0] + len(mine))), (nwg, k)
