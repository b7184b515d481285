"""Integer descriptors in the brute-force matcher (NORM_L1 on 8U / 16U / 16S / 32S, NORM_HAMMING on 8U / 16U / 32S -- the reference's
(depth, norm) table, brute_force_matcher.cpp:336-356): oracle against numpy, the kernel's host build against the oracle bit for bit
(CPU), the reference's Python binding test restated.  HIP against the oracle: tests/test_zzz_bfmatch_int_hip.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NORM_L1, NORM_L2, NORM_HAMMING = 2, 4, 6
DEPTH = {"uint8": 0, "uint16": 2, "int16": 3, "int32": 4}
CASES = [("uint8", NORM_HAMMING, 256), ("uint8", NORM_L1, 256), ("uint16", NORM_HAMMING, 65536), ("uint16", NORM_L1, 65536),
         ("int16", NORM_L1, 3000), ("int32", NORM_HAMMING, 2 ** 20), ("int32", NORM_L1, 2 ** 20)]


def _data(rng, dt, hi, nq, nts, d):
    lo = -hi if dt in ("int16", "int32") else 0
    q = rng.integers(lo, hi, (nq, d)).astype(dt)
    trains = [rng.integers(lo, hi, (n, d)).astype(dt) for n in nts]
    if nts[0] > 3:
        trains[0][3] = trains[0][1]                      # an exact tie
    return q, trains


@pytest.mark.parametrize("dt,norm,hi", CASES)
def test_oracle_integer_distances_against_numpy(oracle, dt, norm, hi):
    rng = np.random.default_rng(1)
    q, (t,) = _data(rng, dt, hi, 9, (40,), 17)
    if norm == NORM_HAMMING:
        bits = 8 * q.dtype.itemsize
        x = (q[:, None].astype(np.int64) ^ t[None].astype(np.int64)) & ((1 << bits) - 1 if dt != "int32" else 0xFFFFFFFF)
        ref = np.array([[sum(bin(int(v)).count("1") for v in row) for row in qq] for qq in x])
    else:
        ref = np.abs(q[:, None].astype(np.int64) - t[None].astype(np.int64)).sum(-1)
    order = np.argsort(ref, axis=1, kind="stable")
    idx, img, dist = oracle.bf_knn_match(q, t, 4, norm)
    np.testing.assert_array_equal(idx, order[:, :4])
    np.testing.assert_array_equal(dist, np.take_along_axis(ref, order[:, :4], 1).astype(np.float32))
    assert (img == 0).all()
    for bad in ([NORM_L2] + ([NORM_HAMMING] if dt == "int16" else [])):
        with pytest.raises(ValueError):
            oracle.bf_knn_match(q, t, 2, bad)            # unsupported combination of query.depth() and norm
    with pytest.raises(ValueError):
        oracle.bf_knn_match(q.astype(np.float32), t.astype(np.float32), 2, NORM_HAMMING)


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emul") / "libbfint_emul.so")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "opencv_contrib_amd", "csrc"),
           os.path.join(ROOT, "tests", "cpp", "bfint_emul.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    L = C.CDLL(out)
    L.emul_bfint_knn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
    L.emul_bfint_radius.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


@pytest.mark.parametrize("dt,norm,hi", CASES)
@pytest.mark.parametrize("d,nq,nts,k", [(32, 130, (300, 7, 65), 5), (61, 64, (90,), 1), (128, 3, (40, 40), 16)])
def test_emulated_kernels_equal_the_oracle(emul, oracle, dt, norm, hi, d, nq, nts, k):
    rng = np.random.default_rng(d + nq)
    q, trains = _data(rng, dt, hi, nq, nts, d)
    masks = [(rng.random((nq, n)) < 0.7).astype(np.uint8) for n in nts]
    if len(nts) > 1:
        masks[0] = None
    n = len(nts)
    tp = (C.c_void_p * n)(*[t.ctypes.data for t in trains])
    ntc = (C.c_int * n)(*nts)
    mp = (C.c_void_p * n)(*[None if m is None else m.ctypes.data for m in masks])
    r = oracle.bf_knn_match(q, trains, k, norm, masks)
    idx = np.full((nq, k), -7, np.int32); img = np.full((nq, k), -7, np.int32); dist = np.zeros((nq, k), np.float32)
    assert emul.emul_bfint_knn(q.ctypes.data, nq, tp, ntc, mp, n, d, DEPTH[dt], norm, k, idx.ctypes.data, img.ctypes.data, dist.ctypes.data) == 0
    np.testing.assert_array_equal(idx, r[0]); np.testing.assert_array_equal(img, r[1]); np.testing.assert_array_equal(dist, r[2])
    cols = 9
    radius = float(np.percentile(r[2][:, -1][r[0][:, -1] >= 0], 70)) + 1            # some rows overflow `cols`, some stay short
    rr = oracle.bf_radius_match(q, trains, radius, cols, norm, masks)
    i2 = np.full((nq, cols), -1, np.int32); m2 = np.full((nq, cols), -1, np.int32); d2 = np.zeros((nq, cols), np.float32)
    n2 = np.zeros(nq, np.int32)
    assert emul.emul_bfint_radius(q.ctypes.data, nq, tp, ntc, mp, n, d, DEPTH[dt], norm, radius, cols, i2.ctypes.data, m2.ctypes.data,
                                  d2.ctypes.data, n2.ctypes.data) == 0
    np.testing.assert_array_equal(n2, rr[3]); np.testing.assert_array_equal(i2, rr[0])
    np.testing.assert_array_equal(m2, rr[1]); np.testing.assert_array_equal(d2, rr[2])


def test_python_mirror_runs_the_reference_binding_test_on_binary_descriptors(oracle):
    """cudafeatures2d/misc/python/test/test_cudafeatures2d.py:46-54: createBFMatcher(NORM_HAMMING); match, knnMatch(.., 2), radiusMatch(.., 0.1)
    on ORB descriptors (8-bit, 32 bytes; random ones here) must run and return matches -- through the Python mirror with the oracle
    standing in for the two C entry points."""
    import torch
    from test_bfmatch import _oracle_backed_mirror
    rng = np.random.default_rng(4)
    d1 = rng.integers(0, 256, (120, 32)).astype(np.uint8)
    d2 = d1.copy()
    d2[::3] ^= rng.integers(0, 4, d2[::3].shape).astype(np.uint8)                  # a third of the descriptors slightly perturbed
    bf = _oracle_backed_mirror(oracle, NORM_HAMMING)
    t1, t2 = torch.from_numpy(d1), torch.from_numpy(d2)
    matches = bf.match(t1, t2)
    assert len(matches) == 120 and all(m.trainIdx == m.queryIdx for m in matches)
    knn = bf.knnMatch(t1, t2, 2)
    assert len(knn) == 120 and all(len(row) == 2 and row[0].distance <= row[1].distance for row in knn)
    rad = bf.radiusMatch(t1, t2, 0.1)                                              # only exact duplicates are closer than 0.1
    assert len(rad) == 120 and sum(len(r) for r in rad) >= 80 and all(m.distance == 0 for r in rad for m in r)
