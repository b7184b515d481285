"""Builds libmiflow.so (HIP kernels + C-ABI) for gfx950 in-tree with hipcc.

`python -m opencv_contrib_amd.build` or build.build().  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmiflow.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


# files compiled WITH the SLP vectoriser (v_pk_*_f32 packed math); everything else keeps -fno-slp-vectorize
SLP_FILES = set(filter(None, os.environ.get("MIFLOW_SLP_FILES", "").split(",")))
VARIANT = os.environ.get("MIFLOW_BUILD_VARIANT", "")   # suffix of the object dir / library name for A/B builds
EXTRA = os.environ.get("MIFLOW_EXTRA_FLAGS", "").split()   # e.g. -DTB_EXPERIMENT=1 for an A/B variant (tuning only)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, src + (("." + VARIANT) if VARIANT else "") + ".o")
    path = os.path.join(CSRC, src)
    deps = [path] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + \
           [os.path.join(ROOT, "include", "miflow", "c_api.h")]
    if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
        return obj
    flags = [f for f in FLAGS if not (f == "-fno-slp-vectorize" and src in SLP_FILES)] + EXTRA
    cmd = [HIPCC] + flags + ["-x", "hip", "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    lib = LIB if not VARIANT else LIB.replace(".so", "_" + VARIANT + ".so")
    if force or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("built", lib)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
