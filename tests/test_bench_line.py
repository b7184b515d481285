"""The bench line the driver parses: one compact strict-JSON line of at most 4 KB as the LAST (and only) stdout line, the long
form in bench_full.json (the 21.9 KB line of round 4 did not reach a driver record)."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "full", "env")
RECORDS = [os.path.join(ROOT, "profiles", d, "bench.json") for d in ("r11z", "r12z")]


def _last_record():
    """The newest tracked bench record; the tests that need one SKIP where profiles/ did not travel with the tree."""
    have = [p for p in RECORDS if os.path.exists(p)]
    if not have:
        pytest.skip("no tracked bench record under profiles/ (measurement files absent)")
    return json.load(open(have[-1]))


def _strict(line):
    def bad(x):
        raise ValueError("non-finite constant " + x)
    return json.loads(line, parse_constant=bad)


@pytest.mark.parametrize("path", [p for p in RECORDS if os.path.exists(p)])
def test_compact_line_of_a_real_record(path):
    out = json.load(open(path))
    line = bench.compact_line(out, "bench_full.json")
    assert "\n" not in line and len(line.encode()) < bench.COMPACT_LIMIT <= 4096
    c = _strict(line)
    for k in REQUIRED:
        assert k in c, k
    assert c["value"] == pytest.approx(out["value"], rel=1e-5)
    assert c["ms_per_step"] == pytest.approx(out["ms_per_step"], rel=1e-5)
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us"):
        assert k in c["roofline"], k
    assert c["roofline"]["frac"] == pytest.approx(out["roofline"]["frac"], rel=1e-5)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    assert c["config"]["workload"] and c["config"]["iterations"] == out["config"]["iterations"]
    assert c["epe_vs_cpu_ref_px"] == pytest.approx(out["epe_vs_cpu_ref_px"], rel=1e-5)
    # every string short enough that the driver's own 120-character clipping leaves it whole
    def walk(x):
        if isinstance(x, str):
            assert len(x) <= 120, x
        elif isinstance(x, dict):
            for v in x.values():
                walk(v)
        elif isinstance(x, list):
            for v in x:
                walk(v)
    walk(c)


def test_compact_line_survives_a_hostile_record(monkeypatch):
    out = _last_record()
    out["variants"] = {("variant_%03d_" % i) + "x" * 80: {"pairs_per_s": float(i)} for i in range(400)}
    out["secondary"]["surf_4k_thr400"]["roofline"]["bound"] = "y" * 5000
    out["roofline"]["kernel"] = "k" * 10000
    out["roofline"]["frac"] = float("nan")
    out["cpu_baseline"]["sample"] = "s" * 10000
    out["epe_vs_cpu_ref_px"] = float("inf")
    monkeypatch.setenv("MIFLOW_TB_JW", "2")
    line = bench.compact_line(out, "bench_full.json")
    assert len(line.encode()) < 4096
    c = _strict(line)
    assert "variants_pairs_per_s" in c.get("dropped", [])
    assert c["roofline"]["frac"] is None and c["env"] == {"MIFLOW_TB_JW": "2"}
    for k in REQUIRED:
        assert k in c, k


def test_emit_prints_exactly_one_stdout_line_and_writes_the_long_form(tmp_path, monkeypatch):
    out = _last_record()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    so, se = io.StringIO(), io.StringIO()
    with redirect_stdout(so), redirect_stderr(se):
        bench.emit(out)
    lines = so.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0].encode()) < 4096
    c = _strict(lines[0])
    assert c["full"] == "bench_full.json"
    full = json.load(open(tmp_path / "bench_full.json"))
    assert full["variants"].keys() == out["variants"].keys() and "secondary" in full


def test_bench_refuses_work_skipping_switches():
    import subprocess
    for var in ("MIFLOW_X_SKIP", "MIFLOW_TB_P16"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], env=dict(os.environ, **{var: "1"}),
                           capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and var in r.stderr and r.stdout.strip() == ""
