"""bench.py's ONE stdout line must stay parsable by the driver (it keeps an 8 KB tail): the headline is built by make_headline() from the
full result, everything else goes to bench_extras.json.  Round 5's line had grown to 23 KB and the driver recorded `parsed: null`."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full_line():
    # the largest line the bench ever produced (round 5, 23 KB): every extra present
    with open(os.path.join(ROOT, "profiles", "r05", "bench.json")) as fh:
        line = json.load(fh)
    line["roofline"].setdefault("frac_profile", 0.18)
    line["roofline"].setdefault("profile_source", "profiles/r05/c1_kernel_stats.csv")
    line["roofline"].setdefault("profile_commit", "c4f38bc 2026-09-30T00:17:00+00:00")
    return line


def test_headline_is_compact_and_complete():
    line = _full_line()
    assert len(json.dumps(line)) > 20000   # the fixture really is the oversized line
    head = bench.make_headline(line)
    text = json.dumps(head, separators=(",", ":"))
    assert len(text) < bench.HEADLINE_MAX_BYTES < 6000
    assert "\n" not in text
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in head, k
    assert head["config"]["workload"].startswith("C1")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us"):
        assert k in head["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in head["cpu_baseline"], k
    # nothing nested beyond the three objects + the second metric
    for k, v in head.items():
        if isinstance(v, dict):
            assert k in ("config", "roofline", "cpu_baseline", "timing", "frame_tracking_ms_per_frame_1280x1024"), k
    assert set(head["frame_tracking_ms_per_frame_1280x1024"]) == {"5_levels", "4_levels"}


def test_headline_refuses_to_grow():
    line = _full_line()
    line["config"]["workload"] = "x" * 5000
    with pytest.raises(RuntimeError):
        bench.make_headline(line)


def test_emit_writes_extras_file(tmp_path, monkeypatch, capsys):
    line = _full_line()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    head, path = bench.emit(line)
    assert head["extras_file"] == "bench_extras.json"
    with open(path) as fh:
        full = json.load(fh)
    assert "tick_sequence" in full and "roofline_large_fullres" in full
    out = capsys.readouterr()
    assert out.out == ""          # stdout belongs to the headline alone; the full result goes to stderr
    assert "tick_sequence" in out.err
