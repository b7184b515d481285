"""The C++ drop-in headers (include/opencv2/*.hpp over the C-ABI): caller code written against the
reference's class names compiles unchanged; without a GPU it fails loudly; on a GPU it produces the same
bytes as the Python mirror (both are thin bindings of the same C-ABI)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "opencv_contrib_amd")


def _build(tmp_path):
    exe = str(tmp_path / "shim_smoke")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "shim_smoke.cpp"), "-o", exe, "-L" + LIBDIR, "-lmiflow",
           "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_shim_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


def test_public_signatures_match_the_reference_headers():
    """tests/cpp/api_conformance.cpp: member-function pointer types, factory defaults and public fields of every class of the
    drop-in headers, transcribed from the reference's headers (cited there), checked at compile time."""
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "api_conformance.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_gpumat_layout_matches_reference_struct(tmp_path):
    """{int flags; int rows, cols; size_t step; uchar* data; int* refcount; uchar* datastart;
    const uchar* dataend; Allocator* allocator;} -- offsets on LP64."""
    src = tmp_path / "layout.cpp"
    src.write_text('#include <cstdio>\n#include <cstddef>\n#include "opencv2/core/cuda.hpp"\n'
                   'int main(){ using G = cv::cuda::GpuMat; printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", offsetof(G, flags),'
                   'offsetof(G, rows), offsetof(G, cols), offsetof(G, step), offsetof(G, data), offsetof(G, refcount),'
                   'offsetof(G, datastart), offsetof(G, dataend), offsetof(G, allocator), sizeof(G)); }\n')
    exe = str(tmp_path / "layout")
    r = subprocess.run(["g++", "-std=c++17", "-Wno-invalid-offsetof", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe,
                        "-L" + LIBDIR, "-lmiflow", "-Wl,-rpath," + LIBDIR], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True).stdout.split()
    assert [int(x) for x in out] == [0, 4, 8, 16, 24, 32, 40, 48, 56, 64]


@pytest.mark.gpu
def test_shim_matches_python_mirror_on_gpu(gpu, tmp_path):
    import torch
    from opencv_contrib_amd import cuda, synth
    exe = _build(tmp_path)
    left, right, _ = synth.stereo_pair(96, 160, seed=3, max_disp=24)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("ii", 96, 160))
        f.write(left.tobytes())
        f.write(right.tobytes())
    fsr = tmp_path / "sr.bin"
    r = subprocess.run([exe, str(fin), str(fout), str(fsr)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    raw = open(fout, "rb").read()
    flow = np.frombuffer(raw[: 96 * 160 * 8], np.float32).reshape(96, 160, 2)
    disp = np.frombuffer(raw[96 * 160 * 8: 96 * 160 * 9], np.uint8).reshape(96, 160)
    fbflow = np.frombuffer(raw[96 * 160 * 9: 96 * 160 * 17], np.float32).reshape(96, 160, 2)
    rest = raw[96 * 160 * 17:]
    nk = struct.unpack("i", rest[:4])[0]
    kps = np.frombuffer(rest[4: 4 + nk * 20], np.float32).reshape(nk, 5)
    sdesc = np.frombuffer(rest[4 + nk * 20:], np.float32).reshape(nk, 64)
    tl, tr = torch.from_numpy(left).to(gpu), torch.from_numpy(right).to(gpu)
    pf = cuda.OpticalFlowDual_TVL1.create(iterations=10, epsilon=0.0).calc(tl, tr).cpu().numpy()
    pd = cuda.createStereoBM(32, 9).compute(tl, tr).cpu().numpy()
    np.testing.assert_array_equal(flow, pf)
    np.testing.assert_array_equal(disp, pd)
    np.testing.assert_array_equal(fbflow, cuda.FarnebackOpticalFlow.create(numLevels=3).calc(tl, tr).cpu().numpy())
    pk, pdesc = cuda.SURF_CUDA.create(300, 3, 2, False, 0.05).detectWithDescriptors(tl)
    pk = cuda.SURF_CUDA.downloadKeypoints(pk)
    assert nk == pk["x"].shape[0]
    np.testing.assert_array_equal(kps, np.stack([pk["x"], pk["y"], pk["size"], pk["angle"], pk["hessian"]], 1))
    np.testing.assert_array_equal(sdesc, pdesc.cpu().numpy())
    # superres adapters: (u, v) planes of DualTVL1_CUDA == the class's own flow; Farneback_CUDA merged == the class's flow
    sr = np.frombuffer(open(fsr, "rb").read(), np.float32)
    np.testing.assert_array_equal(sr[: 96 * 160].reshape(96, 160), pf[..., 0])
    np.testing.assert_array_equal(sr[96 * 160: 2 * 96 * 160].reshape(96, 160), pf[..., 1])
    np.testing.assert_array_equal(sr[2 * 96 * 160:].reshape(96, 160, 2), fbflow)


def test_samples_compile_against_the_drop_in_headers(tmp_path):
    """samples/optical_flow.cpp follows the reference's sample (cudaoptflow/samples/optical_flow.cpp); it must build with nothing but
    the headers of this repository and fail loudly without a GPU."""
    import torch
    exe = str(tmp_path / "optical_flow")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "samples", "optical_flow.cpp"),
                        "-o", exe, "-L" + LIBDIR, "-lmiflow", "-Wl,-rpath," + LIBDIR], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if not torch.cuda.is_available():
        run = subprocess.run([exe], capture_output=True, text=True)
        assert run.returncode == 3 and "error" in run.stderr
    import ast
    ast.parse(open(os.path.join(ROOT, "samples", "optical_flow.py")).read())
