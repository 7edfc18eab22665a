"""bench.py's last stdout line must be a compact record the driver parses (round 5's 21 KB line was not: BENCH_r05.json parsed null).
A recorded full record -- the round-5 default line, committed under profiles/ -- goes through the same formatter bench.py prints with."""
import contextlib
import copy
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_line  # noqa: E402

RECORDED = [f for f in ("r06_bench_full.json", "r05_bench_default.json") if os.path.exists(os.path.join(ROOT, "profiles", f))]


def recorded(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.load(f)


def check_line(text):
    assert "\n" not in text
    assert len(text) < 4096
    assert text.count('"metric"') == 1
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    return d


@pytest.mark.parametrize("name", RECORDED)
def test_recorded_default_line_becomes_a_compact_headline(name, tmp_path):
    full = recorded(name)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        bench_line.emit(full, str(tmp_path / "bench_full.json"))
    lines = out.getvalue().splitlines()
    assert len(lines) == 1  # stdout carries the headline and nothing else
    d = check_line(lines[0])
    assert 0.0 < d["roofline"]["frac"] <= 1.0 and d["roofline"]["bound"] == "hbm" and "traffic" in d["roofline"]
    assert d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["sample"]
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and d["bit_match"] == full["bit_match"]
    assert d["summary"]["ms_per_frame"] == full["summary"]["ms_per_frame"]
    with open(tmp_path / "bench_full.json") as f:  # ... and the full record is in the file, whole
        assert json.load(f) == full


def test_headline_of_an_eight_rank_record_still_fits():
    full = copy.deepcopy(recorded(RECORDED[0]))
    full["n_gpus"] = 8
    full["config"]["sharding"] = {"ranks": 8, "rccl_ranks": 8, "rccl_ranks_seen": 8, "backend": "torch.distributed nccl (RCCL)", "hiz_exchange": "levels >= 2 broadcast, lower levels built by every rank from its own depth copy",
                                  "hiz_broadcast_bytes_per_frame": 5592404, "per_rank_ms_per_frame": [0.612345] * 8, "per_rank_visible": [1234567] * 8,
                                  "scene": {"one_scene": True, "assignment": "interleaved blocks of 64 instances", "scene_mesh_instances": 100000}, "hiz_exchange_ab": {"top": {"ms_per_frame": 0.6}, "whole": {"ms_per_frame": 0.7}}}
    full["summary"]["sharding"] = {"assignment": "interleaved blocks of 64 instances", "per_rank_visible": [1234567] * 8, "per_rank_ms_per_frame": [0.612345] * 8,
                                   "other_assignment": {"assignment": "contiguous instance ranges", "value": 1.234e11, "ms_per_frame": 0.8123, "per_rank_visible": [7654321] * 8}}
    full["summary"]["native_comm_ab"] = {"ms_per_frame": 0.62, "outputs_match_main_line": True, "rccl_ranks_seen": 8}
    d = check_line(json.dumps(bench_line.headline(full)))
    assert d["config"]["sharding"]["assignment"] == "interleaved blocks of 64 instances" and d["config"]["sharding"]["rccl_ranks_seen"] == 8
    assert len(d["config"]["sharding"]["per_rank_visible"]) == 8


def test_nested_records_never_reach_the_headline():
    full = copy.deepcopy(recorded(RECORDED[0]))
    full["summary"]["configs0"] = dict(full.get("configs0") or {"metric": "entities/s", "value": 1.0})  # a whole record smuggled into the summary
    full["summary"]["blob"] = {"text": "x" * 6000}
    d = check_line(json.dumps(bench_line.headline(full)))
    assert "blob" not in d["summary"]  # dropped to fit
    assert d["roofline"] is not None and d["cpu_baseline"] is not None  # ... before anything the driver needs


def test_records_without_a_summary_get_one():
    rec = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 0.5, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "config": {"workload": "w", "per_frame_us": {"a": 1.0}}, "roofline": None, "cpu_baseline": None,
           "batched": {"value": 2.0, "note": "n" * 500, "stage_frac": 0.6}, "kernels": {"k": {"avg_us": 1.0}}}
    d = check_line(json.dumps(bench_line.headline(rec)))
    assert d["summary"]["batched"] == {"value": 2.0, "stage_frac": 0.6}
