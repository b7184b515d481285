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
    rec = _load("static_mix_tbr.json")
    now = static_mix.mix(10, 1, 0, 4, 2, 0)
    assert now["loop_instructions"] == rec["loop_instructions"], "regenerate profiles/static_mix_tbr.json (tools/static_mix.py)"
    for k, v in rec["per_stage_and_pixel"].items():
        assert abs(now["per_stage_and_pixel"][k] - v) < 1e-9, k
    slots = sum(rec["per_stage_and_pixel"][k] for k in ("valu_plain", "dpp", "cndmask")) + 4.0 * rec["per_stage_and_pixel"]["transcendental"]
    assert 45.0 < slots < 65.0   # bench.py: issue slots per pixel-iteration


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
    assert bench.tbr_band_rows(1920, 1080, 8) == 135
    assert bench.tbr_band_rows(1920, 1080, 16) == 270
    assert bench.tbr_band_rows(1920, 1080, 32) == 270
    # StereoBM rows per band: one 1080p / 128-disparity pair -> 16, a batch -> the 48-row cap
    assert bench.sbm_band_rows(1080, 1920, 128, 7, 1) == 16
    assert bench.sbm_band_rows(1080, 1920, 128, 7, 8) == 48
    assert 15.0 < bench.sbm_valu_per_pxd() < 20.0
    assert 0.1 < bench.sbm_warmup_ratio() < 0.3
