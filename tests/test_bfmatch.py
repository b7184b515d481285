"""Brute-force L2 matcher for SURF descriptors (first part): HIP vs the CPU restatement, bit-exact distances
(same fma chain order) and identical indices."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _desc(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)     # unit norm like SURF descriptors


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_against_numpy_and_tie_rule(oracle):
    rng = np.random.default_rng(0)
    q, t = _desc(rng, 37, 64), _desc(rng, 91, 64)
    idx, dist = oracle.bf_knn_match2(q, t)
    d = np.sqrt(((q[:, None, :].astype(np.float64) - t[None].astype(np.float64)) ** 2).sum(-1))
    order = np.argsort(d, axis=1, kind="stable")
    np.testing.assert_array_equal(idx, order[:, :2])
    np.testing.assert_allclose(dist, np.take_along_axis(d, order[:, :2], 1), rtol=1e-6)
    # exact ties: the lowest train index wins, the duplicate becomes the second best
    t2 = np.concatenate([t[:5], t[:5], t[5:]])
    idx2, dist2 = oracle.bf_knn_match2(t[:5], t2)
    np.testing.assert_array_equal(idx2, np.stack([np.arange(5), np.arange(5) + 5], 1))
    assert (dist2 == 0).all()


def test_oracle_mask_and_no_candidate(oracle):
    rng = np.random.default_rng(1)
    q, t = _desc(rng, 4, 64), _desc(rng, 6, 64)
    mask = np.ones((4, 6), np.uint8); mask[0] = 0; mask[1, 1:] = 0
    idx, dist = oracle.bf_knn_match2(q, t, mask)
    assert idx[0].tolist() == [-1, -1] and (dist[0] == np.finfo(np.float32).max).all()      # bf_match.cu:150-151 initial values
    assert idx[1].tolist() == [0, -1]


# ------------------------------------------------------------------ HIP vs oracle (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("nq,nt,d", [(1, 1, 64), (70, 33, 64), (300, 1000, 64), (257, 519, 128), (64, 200, 17), (129, 40, 100)])
def test_match_and_knn_bit_exact(gpu, oracle, nq, nt, d):
    import torch
    from opencv_contrib_amd import cuda
    rng = np.random.default_rng(nq * 7 + nt)
    q, t = _desc(rng, nq, d), _desc(rng, nt, d)
    if nt > 4:
        t[3] = t[1]                                  # a duplicate train descriptor: exercises the tie rule
    m = cuda.createBFMatcher(cuda.BFMatcher.NORM_L2)
    tq, tt = torch.from_numpy(q).to(gpu), torch.from_numpy(t).to(gpu)
    ridx, rdist = oracle.bf_knn_match2(q, t)
    i1, d1 = m.match(tq, tt)
    np.testing.assert_array_equal(i1.cpu().numpy(), ridx[:, 0])
    np.testing.assert_array_equal(d1.cpu().numpy(), rdist[:, 0])
    i2, d2 = m.knnMatch(tq, tt, k=2)
    np.testing.assert_array_equal(i2.cpu().numpy(), ridx)
    np.testing.assert_array_equal(d2.cpu().numpy(), rdist)


@pytest.mark.gpu
def test_mask_pitched_inputs_and_errors(gpu, oracle):
    import torch
    from opencv_contrib_amd import capi, cuda
    rng = np.random.default_rng(5)
    q, t = _desc(rng, 90, 64), _desc(rng, 150, 64)
    mask = (rng.random((90, 150)) < 0.3).astype(np.uint8); mask[7] = 0
    m = cuda.createBFMatcher()
    big = torch.zeros((100, 80), device=gpu); big[5:95, 8:72] = torch.from_numpy(q).to(gpu)
    i2, d2 = m.knnMatch(big[5:95, 8:72], torch.from_numpy(t).to(gpu), k=2, mask=torch.from_numpy(mask).to(gpu))   # pitched query rows
    ridx, rdist = oracle.bf_knn_match2(q, t, mask)
    np.testing.assert_array_equal(i2.cpu().numpy(), ridx)
    np.testing.assert_array_equal(d2.cpu().numpy(), rdist)
    assert i2[7].tolist() == [-1, -1]
    with pytest.raises(capi.MiError):
        m.match(torch.from_numpy(q).to(gpu), torch.from_numpy(_desc(rng, 10, 32)).to(gpu))     # query.cols != train.cols
    with pytest.raises(capi.MiError):
        m.match(torch.zeros((4, 200), device=gpu), torch.zeros((4, 200), device=gpu))           # longer than 128
    with pytest.raises(capi.MiError):
        cuda.createBFMatcher(6)                                                                 # NORM_HAMMING: not built


@pytest.mark.gpu
def test_surf_descriptors_match_themselves_and_a_shifted_view(gpu):
    """The step after detect/describe (the reference's own SURF test matches descriptors, test_surf.cuda.cpp:166-168):
    descriptors of an image against those of the same image are an identity match at distance 0."""
    import torch
    from opencv_contrib_amd import cuda, synth
    img = synth.blob_image(240, 320, seed=7)
    surf = cuda.SURF_CUDA.create(300, 3, 2, False, 0.05)
    _, desc = surf.detectWithDescriptors(torch.from_numpy(img).to(gpu))
    assert desc.shape[0] > 20
    idx, dist = cuda.createBFMatcher().match(desc, desc)
    assert torch.equal(idx.cpu(), torch.arange(desc.shape[0], dtype=torch.int32)) and float(dist.max()) == 0.0
