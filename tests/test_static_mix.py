"""The static instruction counts bench.py's roofline objects rest on (profiles/static_mix_*.json) are those of the CURRENT kernel
sources: recompiled here with hipcc (cross-compiles without a GPU) and compared.  Also the launch-plan mirrors bench.py keeps for
its executed-work accounting."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)


def _load(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


def test_tvl1_static_mix_of_record_matches_the_source():
    import static_mix
    for name, jw, lanes_per_px in (("static_mix_tbr.json", 2, 256.0 / 236.0), ("static_mix_tbr_jw0.json", 0, 64.0 / 44.0)):
        rec = _load(name)
        now = static_mix.mix(10, 1, 0, 4, 2, 0, jw)
        assert now["loop_instructions"] == rec["loop_instructions"], f"regenerate profiles/{name} (tools/static_mix.py 10 1 0 4 2 0 {jw})"
        for k, v in rec["per_stage_and_pixel"].items():
            assert abs(now["per_stage_and_pixel"][k] - v) < 1e-9, k
        slots = sum(rec["per_stage_and_pixel"][k] for k in ("valu_plain", "dpp", "cndmask")) + 4.0 * rec["per_stage_and_pixel"]["transcendental"]
        assert 45.0 < slots < 65.0   # bench.py: issue slots per pixel-iteration
        assert 60.0 < slots * lanes_per_px < 85.0   # ... per OWNED pixel-iteration: the joined waves execute fewer (65.6 vs 78.9)


def test_stereobm_static_mix_of_record_matches_the_source():
    import static_mix
    rec = _load("static_mix_sbm.json")
    now = static_mix.mix_sbm(7, 0, 1)   # the LDS-transposed winner-take-all is the kernel of record since round 4
    for k in ("row_loop_valu", "warmup_row_valu", "tile_output_columns"):
        assert now[k] == rec[k], f"{k}: regenerate profiles/static_mix_sbm.json (tools/static_mix.py sbm)"
    assert rec["warmup_row_valu"] < rec["row_loop_valu"] / 3


def test_bench_launch_plan_mirrors():
    import bench
    # TV-L1 band heights of the streaming kernel (mi_tvl1_query_plan is the authority on a GPU box; this mirror is the fallback):
    # 8 / 16 pairs per lane at 1080p -> 8 / 4 bands; never fewer than 4 bands when waves are plentiful
    assert bench.tbr_band_rows(1920, 1080, 8, jw=0) == 135
    assert bench.tbr_band_rows(1920, 1080, 16, jw=0) == 270
    assert bench.tbr_band_rows(1920, 1080, 32, jw=0) == 270
    # joined waves (the default): what MIFLOW_TB_VERBOSE printed on the GPU box (r03x) for 16 pairs per lane at the five 1080p levels
    for (w_, h_), rows in {(1920, 1080): 216, (1536, 864): 144, (1229, 691): 87, (983, 553): 70, (786, 442): 45}.items():
        assert bench.tbr_band_rows(w_, h_, 16, jw=2) == rows, (w_, h_)
    # StereoBM rows per band: one 1080p / 128-disparity pair -> 16, a batch -> the 48-row cap
    assert bench.sbm_band_rows(1080, 1920, 128, 7, 1) == 16
    assert bench.sbm_band_rows(1080, 1920, 128, 7, 8) == 48
    assert 12.0 < bench.sbm_valu_per_pxd() < 20.0
    assert 0.1 < bench.sbm_warmup_ratio() < 0.3


def test_blocked_kernels_keep_their_occupancy():
    """Register budget of the kernels of record (hipcc -Rpass-analysis=kernel-resource-usage, cross-compiled here): the joined-wave T = 10
    kernel must stay within 128 VGPRs (4 waves per SIMD: r03w / r04f -- the barrier form gave its gain back at 3) without spilling, the
    speculative kernels may spill SGPRs (they do: DESIGN section 7) but never VGPRs, and the SURF det / trace kernel with its LDS
    tile and the descriptor kernel must not spill either."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, "opencv_contrib_amd", "csrc")

    def usage(src):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-std=c++17", "-ffp-contract=off",
                            "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-Rpass-analysis=kernel-resource-usage", "-c",
                            os.path.join(csrc, src), "-o", os.devnull], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out = {}
        name = None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                out[name] = {}
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
            if m and name:
                out[name][m.group(1).strip()] = int(m.group(2))
        return out

    tbr = usage("tvl1_tbr_kernels.hip")
    # template arguments <T, PPL, PZ, WPS, PF, MODE, JW, NG, P16, FW, GAM>.  The kernel of record since round 4 forms |grad|^2 itself (NG):
    # ...ELi2ELb1ELb0ELi0ELb0EE (FW = 0: the warp as its own launch; GAM = 0: no illumination channel); the one that reads the plane stays
    # for the stage API
    rec = [v for k, v in tbr.items() if "k_iterate_tbrILi10ELi1ELb" in k and (k.endswith("ELi4ELi2ELi0ELi2ELb1ELb0ELi0ELb0EEEvNS0_6TbArgsE") or
                                                                               k.endswith("ELi4ELi2ELi0ELi2ELb0ELb0ELi0ELb0EEEvNS0_6TbArgsE"))]
    assert len(rec) == 4   # {no |grad|^2 plane, plane} x {first pass of a warp (p = 0), the others}
    for v in rec:
        assert v["VGPRs"] <= 128 and v["VGPRs Spill"] == 0 and v["SGPRs Spill"] == 0 and v["Occupancy"] >= 4, v
    # round 6, gamma != 0 (GAM = 1): the T = 10 pass with ONE prefetched row must keep three waves per SIMD without scratch (163 VGPRs;
    # with two rows it is 168 + 12 spilled dwords), the T = 5 pass four, and no GAM kernel of the joined form may spill VGPRs
    gam10 = [v for k, v in tbr.items() if re.search(r"k_iterate_tbrILi10ELi1ELb[01]ELi3ELi1ELi0ELi2ELb1ELb0ELi0ELb1EEEvNS0_6TbArgsE$", k)]
    gam5 = [v for k, v in tbr.items() if re.search(r"k_iterate_tbrILi5ELi1ELb[01]ELi4ELi2ELi0ELi2ELb1ELb0ELi0ELb1EEEvNS0_6TbArgsE$", k)]
    assert len(gam10) == 2 and len(gam5) == 2
    for v in gam10:
        assert v["VGPRs"] <= 168 and v["VGPRs Spill"] == 0 and v["Occupancy"] >= 3, v
    for v in gam5:
        assert v["VGPRs"] <= 128 and v["VGPRs Spill"] == 0 and v["Occupancy"] >= 4, v
    for k, v in tbr.items():
        if re.search(r"ELi2ELb[01]ELb[01]ELi[012]ELb[01]EEEvNS0_6TbArgsE$", k):      # every joined-wave instantiation (fixed work and speculative steps)
            assert v.get("VGPRs Spill", 0) == 0, k
    # (the fused-warp instantiations -- FW = 1, 2 -- are compiled into the experiments build only since round 6)
    surf = usage("surf_kernels.hip")
    for k, v in surf.items():
        if "k_det_trace_all" in k or "k_descriptors" in k or "k_nms_flag_all" in k:
            assert v["VGPRs Spill"] == 0 and v["SGPRs Spill"] == 0, (k, v)
    assert any("k_det_trace_all" in k for k in surf)
