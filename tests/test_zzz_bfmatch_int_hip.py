"""HIP integer-descriptor matcher against the oracle: exact integer distances, identical lists.  The kernel logic is checked bit for
bit on the CPU (tests/test_bfmatch_int.py, the same source compiled for the host).  (Collected last; written after the round's GPU
budget was spent: first execution is the driver's round-end run.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NORM_L1, NORM_HAMMING = 2, 6
CASES = [("uint8", NORM_HAMMING, 256), ("uint8", NORM_L1, 256), ("uint16", NORM_HAMMING, 65536), ("int16", NORM_L1, 3000),
         ("int32", NORM_HAMMING, 2 ** 20), ("int32", NORM_L1, 2 ** 20)]


def _torch_dtype_ok(dt):
    import torch
    return hasattr(torch, dt)


@pytest.mark.parametrize("dt,norm,hi", CASES)
def test_hip_integer_matcher_equals_the_oracle(gpu, oracle, dt, norm, hi):
    import torch
    from opencv_contrib_amd import cuda
    if not _torch_dtype_ok(dt):
        pytest.skip(f"torch has no {dt}")
    rng = np.random.default_rng(9)
    lo = -hi if dt in ("int16", "int32") else 0
    nq, nts, d = 150, (333, 5, 64), 32
    q = rng.integers(lo, hi, (nq, d)).astype(dt)
    trains = [rng.integers(lo, hi, (n, d)).astype(dt) for n in nts]
    trains[0][3] = trains[0][1]
    masks = [(rng.random((nq, n)) < 0.7).astype(np.uint8) for n in nts]
    masks[1] = None
    T = lambda a: None if a is None else torch.from_numpy(a).to(gpu)
    m = cuda.createBFMatcher(norm)
    m.add([T(t) for t in trains])
    for ms in (None, masks):
        tm = None if ms is None else [T(x) for x in ms]
        for k in (1, 2, 7, 16):
            r = oracle.bf_knn_match(q, trains, k, norm, ms)
            idx, img, dist = m.knnMatchDevice(T(q), None, k=k, masks=tm)
            np.testing.assert_array_equal(idx.cpu().numpy(), r[0])
            np.testing.assert_array_equal(img.cpu().numpy(), r[1])
            np.testing.assert_array_equal(dist.cpu().numpy(), r[2])
        full = oracle.bf_knn_match(q, trains, 16, norm, ms)
        radius = float(np.percentile(full[2][:, -1], 60)) + 1
        rr = oracle.bf_radius_match(q, trains, radius, nq, norm, ms)
        i2, m2, d2, n2 = m.radiusMatchDevice(T(q), None, radius, masks=tm)
        np.testing.assert_array_equal(n2.cpu().numpy(), rr[3])
        np.testing.assert_array_equal(i2.cpu().numpy(), rr[0])
        np.testing.assert_array_equal(m2.cpu().numpy(), rr[1])
        np.testing.assert_array_equal(d2.cpu().numpy(), rr[2])
    # single train set through the host-list forms
    ms1 = m.match(T(q), T(trains[0]))
    r1 = oracle.bf_knn_match(q, trains[0], 1, norm)
    assert [x.trainIdx for x in ms1] == r1[0][:, 0].tolist()


def test_hip_reference_binding_test_and_type_table(gpu):
    """cudafeatures2d/misc/python/test/test_cudafeatures2d.py:46-54 on random 32-byte descriptors + the (depth, norm) table's errors."""
    import torch
    from opencv_contrib_amd import capi, cuda
    rng = np.random.default_rng(4)
    d1 = rng.integers(0, 256, (500, 32)).astype(np.uint8)
    d2 = d1.copy()
    d2[::3] ^= rng.integers(0, 4, d2[::3].shape).astype(np.uint8)
    t1, t2 = torch.from_numpy(d1).to(gpu), torch.from_numpy(d2).to(gpu)
    bf = cuda.DescriptorMatcher.createBFMatcher(cuda.NORM_HAMMING)
    assert len(bf.match(t1, t2)) == 500
    assert len(bf.knnMatch(t1, t2, 2)) == 500
    assert sum(len(r) for r in bf.radiusMatch(t1, t2, 0.1)) >= 300
    with pytest.raises(capi.MiError):
        cuda.createBFMatcher(cuda.NORM_L2).matchDevice(t1, t2)                      # L2 on CV_8U: unsupported combination
    with pytest.raises(capi.MiError):
        cuda.createBFMatcher(cuda.NORM_HAMMING).matchDevice(t1.to(torch.int16), t2.to(torch.int16))   # Hamming on CV_16S
    with pytest.raises(capi.MiError):
        bf.knnMatchDevice(t1, t2, k=17)                                             # integer path: k <= 16
    with pytest.raises(capi.MiError):
        bf.matchDevice(torch.zeros((4, 200), dtype=torch.uint8, device=gpu), torch.zeros((4, 200), dtype=torch.uint8, device=gpu))   # > 128 elements
