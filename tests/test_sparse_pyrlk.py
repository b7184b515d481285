"""cv::cuda::SparsePyrLKOpticalFlow (CV_8UC1): oracle self-tests, the kernel's host build against the oracle bit for bit (CPU), and the HIP
kernel against the oracle (GPU; collected in tests/test_zzz_sparse_pyrlk_hip.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from opencv_contrib_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _points(rng, n, w, h, margin=-12.0):
    return np.stack([rng.uniform(margin, w - margin, n), rng.uniform(margin, h - margin, n)], 1).astype(np.float32)


def test_oracle_tracks_an_analytic_flow(oracle):
    I0, I1, gt = synth.flow_pair(240, 320, seed=5, dtype="u8")
    pts = _points(np.random.default_rng(0), 300, 320, 240, margin=25.0)
    nxt, st, err = oracle.pyrlk_sparse(I0, I1, pts)
    assert st.all()
    g = gt[pts[:, 1].astype(int), pts[:, 0].astype(int)]
    epe = np.hypot(*(nxt - pts - g).T)
    assert np.median(epe) < 0.15 and np.percentile(epe, 95) < 0.5
    assert (err >= 0).all() and np.median(err) < 10                       # mean |J - I| over the window, 8-bit units
    # identical frames: every point stays where it is and the residual is zero
    n2, s2, e2 = oracle.pyrlk_sparse(I0, I0, pts)
    np.testing.assert_allclose(n2, pts, atol=1e-4)
    assert s2.all() and np.abs(e2).max() < 1e-4


def test_oracle_status_and_initial_flow(oracle):
    I0, I1, _ = synth.flow_pair(120, 160, seed=7, dtype="u8")
    pts = np.array([[-3.0, 10.0], [10.0, -0.5], [160.0, 50.0], [50.0, 120.0], [80.0, 60.0]], np.float32)
    nxt, st, _ = oracle.pyrlk_sparse(I0, I1, pts)
    assert st.tolist() == [0, 0, 0, 0, 1]                                  # prevPt outside [0, cols) x [0, rows) at level 0 (pyrlk.cu:165-171)
    flat = np.full((120, 160), 77, np.uint8)
    _, st2, _ = oracle.pyrlk_sparse(flat, flat, pts[4:])
    assert st2.tolist() == [0]                                             # singular structure tensor (:229-235)
    # useInitialFlow: a good guess and no guess converge to the same place
    good = (nxt[4:] + 0.3).astype(np.float32)
    n3, s3, _ = oracle.pyrlk_sparse(I0, I1, pts[4:], next_pts=good)
    assert s3.all() and np.abs(n3 - nxt[4:]).max() < 0.05
    with pytest.raises(ValueError):
        oracle.pyrlk_sparse(I0, I1[:100], pts)


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emul") / "libsparselk_emul.so")
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror",
           "-I" + os.path.join(ROOT, "opencv_contrib_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "sparselk_emul.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    L = C.CDLL(out)
    L.emul_sparse_lk.restype = C.c_int
    L.emul_sparse_lk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p, C.c_void_p]
    return L


@pytest.mark.parametrize("win,max_level,iters,use_init", [((21, 21), 3, 30, False), ((13, 9), 2, 10, False), ((31, 31), 4, 30, True),
                                                          ((3, 3), 0, 5, False), ((32, 32), 1, 8, False), ((5, 27), 3, 3, True)])
def test_emulated_kernel_equals_the_oracle(emul, oracle, win, max_level, iters, use_init):
    """The HIP kernel's source compiled for the host (phases as loops): next points, status and error bit for bit, including points
    outside the image, on the border, and tracks that leave the image."""
    I0, I1, _ = synth.flow_pair(203, 317, seed=11, dtype="u8")            # odd sizes: uneven pyramid levels
    pts = _points(np.random.default_rng(3), 500, 317, 203)
    init = (pts + np.float32(1.5)).astype(np.float32) if use_init else None
    rn, rs, re_ = oracle.pyrlk_sparse(I0, I1, pts, win, max_level, iters, init)
    nxt = init.copy() if use_init else np.zeros_like(pts)
    st = np.zeros(500, np.uint8)
    err = np.zeros(500, np.float32)
    rc = emul.emul_sparse_lk(I0.ctypes.data, I1.ctypes.data, 203, 317, pts.ctypes.data, nxt.ctypes.data, 500, win[0], win[1], max_level, iters,
                             int(use_init), st.ctypes.data, err.ctypes.data)
    assert rc == 0 and 0 < rs.mean() < 1
    np.testing.assert_array_equal(st, rs)
    np.testing.assert_array_equal(nxt, rn)
    np.testing.assert_array_equal(err, re_)
