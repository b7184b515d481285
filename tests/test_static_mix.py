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
    now = static_mix.mix_sbm(7, 0)
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
    assert 15.0 < bench.sbm_valu_per_pxd() < 20.0
    assert 0.1 < bench.sbm_warmup_ratio() < 0.3
