"""HIP kernels of the CPU-class SURF stages (csrc/surfcpu_kernels.hip) against the pinned oracle and the reference's golden vectors.
The kernel logic itself is checked bit for bit on the CPU (tests/test_surfcpu_emulation.py, the same source compiled for the host);
what can differ on the device is sinf / cosf of the rotated window (1 ulp), so the tolerances are: orientation exact for >= 99 % of
the keypoints, descriptors within 1e-4 for >= 99 %, the golden vectors within the Java tests' EPS 1e-3.
(File name: collected last.  Written after the round's GPU budget was spent: its first execution is the driver's round-end run.)"""
import numpy as np
import pytest

from opencv_contrib_amd import synth

pytestmark = pytest.mark.gpu


def _kp_matrix(kp):
    """oracle rows {x, y, size, angle, response, octave, class_id} -> SURF_CUDA 7 x n matrix (cuda.hpp:89-99)"""
    k = np.zeros((7, len(kp)), np.float32)
    k[0], k[1], k[4], k[5], k[6] = kp[:, 0], kp[:, 1], kp[:, 2], kp[:, 3], kp[:, 4]
    k.view(np.int32)[2] = kp[:, 6].astype(np.int32)
    k.view(np.int32)[3] = kp[:, 5].astype(np.int32)
    return k


@pytest.mark.parametrize("extended,upright", [(False, False), (True, False), (True, True)])
def test_hip_cpu_class_stages_match_the_oracle(gpu, oracle, extended, upright):
    import torch
    from opencv_contrib_amd import cuda
    img = synth.blob_image(360, 480, seed=21)
    kp = oracle.surfcpu_detect(img, 200.0, 4, 3)
    k_ref, d_ref = oracle.surfcpu_compute(img, kp, extended, upright)
    alg = cuda.SURF_CUDA.create(200.0, 4, 3, extended, 0.05, upright)
    kg, dg = alg.detectAndComputeCpuClass(torch.from_numpy(img).to(gpu), None, torch.from_numpy(_kp_matrix(kp)).to(gpu), True)
    kg, dg = kg.cpu().numpy(), dg.cpu().numpy()
    assert kg.shape[1] == len(k_ref) and dg.shape == d_ref.shape
    np.testing.assert_array_equal(kg[0], k_ref[:, 0]); np.testing.assert_array_equal(kg[4], k_ref[:, 2])
    assert (kg[5] == k_ref[:, 3]).mean() >= 0.99 and np.abs(kg[5] - k_ref[:, 3]).max() < 360
    ok = np.abs(dg - d_ref).max(1) <= 1e-4
    assert ok.mean() >= 0.99, (float(np.abs(dg - d_ref).max()), int((~ok).sum()))


def test_hip_reproduces_the_reference_golden_vectors(gpu, oracle):
    """The Java tests' known answers (see tests/test_zz_surf_cpu_class.py) through the product: this class's detector, then the CPU
    class's orientation and descriptor."""
    import torch
    from opencv_contrib_amd import cuda
    from test_zz_surf_cpu_class import EPS, JAVA_DESCRIPTOR, JAVA_TRUTH, java_cross
    img = torch.from_numpy(java_cross()).to(gpu)
    alg = cuda.SURF_CUDA.create(8000, 3, 4, True, 0.05, False)
    kg, _ = alg.detectAndComputeCpuClass(img)
    k = cuda.SURF_CUDA.downloadKeypoints(kg)
    order = np.argsort(k["angle"])
    got = np.stack([k["x"], k["y"], k["size"], k["angle"], k["hessian"]], 1)[order]
    assert got.shape == (4, 5)
    np.testing.assert_allclose(got[:, :4], JAVA_TRUTH[:, :4], rtol=0, atol=EPS)
    np.testing.assert_allclose(got[:, 4], JAVA_TRUTH[:, 4], rtol=0, atol=5e-3)     # response: 8617.86 in binary32 has 1e-3 resolution
    assert (k["octave"] == 0).all() and (k["laplacian"] == -1).all()
    # SURFDescriptorExtractorTest: SURF(100, 2, 4, extended, not upright).compute on the given keypoint
    kp = np.array([[55.775577545166016, 44.224422454833984, 16, 9.754629, 8617.863, 1, -1]], np.float32)
    k1 = np.zeros((7, 1), np.float32)
    k1[0], k1[1], k1[4], k1[5], k1[6] = kp[:, 0], kp[:, 1], kp[:, 2], kp[:, 3], kp[:, 4]
    alg2 = cuda.SURF_CUDA.create(100, 2, 4, True, 0.05, False)
    k2, d2 = alg2.detectAndComputeCpuClass(img, None, torch.from_numpy(k1).to(gpu), True)
    assert abs(float(k2[5, 0]) - 350.24573) < EPS
    np.testing.assert_allclose(d2.cpu().numpy()[0], JAVA_DESCRIPTOR, rtol=0, atol=EPS)
