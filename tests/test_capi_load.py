"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/miflow/c_api.h declares; without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes as C
import os

import pytest

from opencv_contrib_amd import capi


def test_library_built_in_tree():
    assert os.path.exists(capi.LIB_PATH), "run python -m opencv_contrib_amd.build"


def test_exports_every_declared_symbol():
    L = C.CDLL(capi.LIB_PATH)
    declared = capi.declared_symbols()
    assert len(declared) >= 20
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, f"declared in c_api.h but not exported: {missing}"


RELEASE_SWITCHES = {"MIFLOW_BF_W", "MIFLOW_CACHE_GB", "MIFLOW_CACHE_TOTAL_GB", "MIFLOW_FB_FUSE", "MIFLOW_FB_GROUP_MB", "MIFLOW_FB_NARROW", "MIFLOW_FB_PAIR",
                    "MIFLOW_LANES", "MIFLOW_MULTI_RCCL", "MIFLOW_SURF_NMS0", "MIFLOW_SURF_POLY", "MIFLOW_SURF_STAGE_S", "MIFLOW_TB_HIST", "MIFLOW_TB_VERBOSE", "MIFLOW_TILE_MAXPX"}
# strings of the once-per-process warning about variables the library does NOT read: prefixes of names that belong to the
# Python side / bench / tests / build, and the name of the compile-time macro in the message
NOT_SWITCHES = {"MIFLOW_LIB", "MIFLOW_BENCH_", "MIFLOW_SWEEP_", "MIFLOW_BUILD_", "MIFLOW_EXTRA_", "MIFLOW_SLP_", "MIFLOW_EXPERIMENTS", "MIFLOW_"}


def test_release_library_reads_no_experiment_switch():
    """the shipped library does not contain the NAMES of the work-skipping (MIFLOW_X_SKIP), result-changing
    (MIFLOW_TB_P16) or A/B-loser tuning variables -- they exist under -DMIFLOW_EXPERIMENTS only -- and the variables it does read are
    exactly the documented set, each of which has a digest / behaviour test in the GPU suite."""
    import re
    blob = open(capi.LIB_PATH, "rb").read()
    names = {m.decode() for m in re.findall(rb"MIFLOW_[A-Z0-9_]+", blob)}
    assert "MIFLOW_X_SKIP" not in names and "MIFLOW_TB_P16" not in names
    assert names - NOT_SWITCHES == RELEASE_SWITCHES, sorted((names - NOT_SWITCHES) ^ RELEASE_SWITCHES)


def test_binding_covers_header():
    L = capi.lib()
    for s in capi.declared_symbols():
        assert getattr(L, s).argtypes is not None, f"capi.py does not bind {s}"


def test_default_params_match_reference():
    # cv::cuda::OpticalFlowDual_TVL1::create defaults, cudaoptflow.hpp:375-385
    p = capi.TVL1Params()
    capi.lib().mi_tvl1_default_params(C.byref(p))
    assert (p.tau, p.lambda_, p.theta, p.nscales, p.warps) == (0.25, 0.15, 0.3, 5, 5)
    assert (p.epsilon, p.iterations, p.scale_step, p.gamma, p.use_initial_flow) == (0.01, 300, 0.8, 0.0, 0)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = capi.lib()
    assert L.mi_device_count() == 0
    p = capi.TVL1Params()
    L.mi_tvl1_default_params(C.byref(p))
    h = C.c_void_p()
    rc = L.mi_tvl1_create(C.byref(p), C.byref(h))
    assert rc == -7 and b"no CPU fallback" in L.mi_last_error()
    from opencv_contrib_amd import cuda
    with pytest.raises(capi.MiError):
        cuda.OpticalFlowDual_TVL1.create()
    # every other class of the library refuses just as loudly (there is no CPU path anywhere behind the C-ABI)
    for make in (lambda: cuda.createStereoBM(64, 15), lambda: cuda.FarnebackOpticalFlow.create(), lambda: cuda.SURF_CUDA.create(400),
                 lambda: cuda.createStereoSGM(), lambda: cuda.createDisparityBilateralFilter(), lambda: cuda.createBFMatcher(),
                 lambda: cuda.DensePyrLKOpticalFlow.create(), lambda: cuda.SparsePyrLKOpticalFlow.create()):
        with pytest.raises(capi.MiError) as ei:
            make()
        assert "no CPU fallback" in str(ei.value) or ei.value.args[0] == -7
    # ... and so does the plan introspection (the plan depends on the device's SIMD count)
    k, r = C.c_int(-1), C.c_int(-1)
    assert L.mi_tvl1_query_plan(1920, 1080, 16, 10, C.byref(k), C.byref(r)) == -7 and (k.value, r.value) == (-1, -1)



def test_bad_params_rejected_before_device():
    L = capi.lib()
    p = capi.TVL1Params()
    L.mi_tvl1_default_params(C.byref(p))
    p.nscales = 0  # CV_Assert( nscales_ > 0 ), cudaoptflow/src/tvl1flow.cpp:191
    h = C.c_void_p()
    assert L.mi_tvl1_create(C.byref(p), C.byref(h)) == -1


def test_every_entry_point_survives_null_and_zero_arguments():
    """Robustness of the C boundary: every MI_API function called with NULL pointers and zero scalars returns (an error code, or
    nothing for the void ones) instead of crashing.  Runs in a child process so that a crash is a test failure, not a dead runner."""
    import subprocess
    import sys
    code = (
        "import ctypes as C\n"
        "from opencv_contrib_amd import capi\n"
        "L = capi.lib()\n"
        "n = 0\n"
        "for name in capi.declared_symbols():\n"
        "    fn = getattr(L, name)\n"
        "    vals = []\n"
        "    for a in (fn.argtypes or []):\n"
        "        if a in (C.c_int, C.c_longlong, C.c_size_t, C.c_uint): vals.append(0)\n"
        "        elif a in (C.c_float, C.c_double): vals.append(0.0)\n"
        "        else: vals.append(None)\n"
        "    r = fn(*vals)\n"
        "    if fn.restype is C.c_int and name not in ('mi_device_count', 'mi_surf_descriptor_size', 'mi_tvl1_multi_device_count'):\n"
        "        assert r != 0 or name in ('mi_stream_synchronize', 'mi_set_device', 'mi_release_cached_memory'), (name, r)\n"
        "    n += 1\n"
        "print('CALLED', n)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300)
    assert r.returncode == 0 and "CALLED" in r.stdout, (r.returncode, r.stderr[-1500:])
