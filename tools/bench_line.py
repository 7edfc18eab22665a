"""The last stdout line of bench.py: a compact headline record, <= 4 KB, that the driver parses.

Round 5's line was one ~21 KB JSON object nesting five more complete records (each with its own "metric" / "value" / "n_gpus") and the
driver could not parse it (BENCH_r05.json: parsed null).  Since round 6 the FULL record goes to a file (gpurun_out/bench_full.json, or
$OXC_BENCH_FULL) and stdout carries exactly one line: headline(full).  Rules kept by construction and checked by tests/test_bench_line.py:
  * the contract keys of the driver (metric ... config) + bit_match + roofline + cpu_baseline + summary, nothing else at the top level;
  * "metric" occurs exactly once in the text; no nested object is a record of its own;
  * strings are cut to a stated length, lists to 8 entries; if the line were still >= 4096 bytes the optional parts of `summary` go first.
"""
import json
import os
import sys

MAX_LINE = 4096
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "traffic_profile_is_of_this_device_code", "algorithmic_bytes_per_launch",
                 "kernel_avg_us", "kernel_avg_us_rocprof", "launches_averaged", "measured_stream_read_GBps", "frac_of_measured_stream_read")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "single_thread_value", "matches_single_thread")
SHARD_KEYS = ("ranks", "rccl_ranks", "rccl_ranks_seen", "backend", "assignment", "hiz_exchange", "hiz_broadcast_bytes_per_frame", "per_rank_visible", "per_rank_ms_per_frame",
              "debug_backend_not_a_measurement")


def _cut(v, n):
    if isinstance(v, str) and len(v) > n:
        return v[: n - 3] + "..."
    return v


def _scalars(d, keys=None, strlen=160):
    """The scalar (and short list) members of d, strings cut; nested objects are not carried."""
    out = {}
    for k, v in (d or {}).items():
        if keys is not None and k not in keys:
            continue
        if isinstance(v, dict):
            continue
        if isinstance(v, list):
            if len(v) > 8 or any(isinstance(x, (dict, list)) for x in v):
                continue
        out[k] = _cut(v, strlen)
    return out


def _strip_metric(o):
    """No nested object may look like a record of its own: drop any "metric" key below the top level."""
    if isinstance(o, dict):
        return {k: _strip_metric(v) for k, v in o.items() if k != "metric"}
    if isinstance(o, list):
        return [_strip_metric(v) for v in o]
    return o


def _auto_summary(full: dict) -> dict:
    """A record without a summary of its own (the other workloads): its remaining scalar members, and of each nested object the numbers and flags."""
    handled = set(CONTRACT) | {"config", "roofline", "cpu_baseline", "bit_match", "hiz_bit_match", "summary", "full_record"}
    sm = {}
    for k, v in full.items():
        if k in handled:
            continue
        if isinstance(v, dict):
            inner = {a: b for a, b in v.items() if isinstance(b, (int, float, bool)) or b is None}
            if inner:
                sm[k] = inner
        elif not isinstance(v, list):
            sm[k] = _cut(v, 120)
    return sm


def headline(full: dict) -> dict:
    h = {k: full.get(k) for k in CONTRACT}
    cfg = full.get("config") or {}
    c = _scalars(cfg, strlen=320)
    sh = cfg.get("sharding")
    if isinstance(sh, dict):
        s = _scalars(sh, SHARD_KEYS)
        scene = sh.get("scene") or {}
        s["assignment"] = scene.get("assignment") or ("an independent scene per rank" if scene.get("one_scene") is False else None)
        c["sharding"] = s
    elif sh is not None:
        c["sharding"] = sh
    h["config"] = c
    for k in ("bit_match", "hiz_bit_match"):
        if k in full:
            h[k] = full[k]
    h["roofline"] = _scalars(full["roofline"], ROOFLINE_KEYS, 200) if isinstance(full.get("roofline"), dict) else None
    h["cpu_baseline"] = _scalars(full["cpu_baseline"], CPU_KEYS, 260) if isinstance(full.get("cpu_baseline"), dict) else None
    h["summary"] = _strip_metric(full.get("summary") or _auto_summary(full))
    h["full_record"] = full.get("full_record")
    # shrink until it fits: optional parts of the summary first, then the long strings
    order = ["configs0", "real_geometry", "configs4", "configs1", "tris124", "pairs124", "native_comm_ab", "sharding"]
    order += sorted((k for k in h["summary"] if k not in order), key=lambda k: -len(json.dumps(h["summary"][k])))
    while len(json.dumps(h)) >= MAX_LINE and order:
        h["summary"].pop(order.pop(0), None)
    if len(json.dumps(h)) >= MAX_LINE:
        for part in ("roofline", "cpu_baseline", "config"):
            if isinstance(h.get(part), dict):
                h[part] = {k: _cut(v, 80) for k, v in h[part].items()}
    return h


def emit(full: dict, path: str = "") -> str:
    """Write the full record to a file, print the compact headline as the (only) stdout line; returns the printed text."""
    path = path or os.environ.get("OXC_BENCH_FULL", "")
    if not path:
        root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(root, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f)
            f.write("\n")
        full = dict(full)
        full["full_record"] = os.path.relpath(path, os.getcwd())
    except OSError as err:  # a read-only tree must not cost the run its line
        print(f"[bench] full record not written: {err}", file=sys.stderr, flush=True)
    text = json.dumps(headline(full))
    assert len(text) < MAX_LINE and text.count('"metric"') == 1, (len(text), text.count('"metric"'))
    sys.stdout.flush()
    print(text, flush=True)
    return text
