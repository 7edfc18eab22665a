#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel-trace stats + PMC counter_collection CSVs) under gpurun_out/
into small committed summaries under profiles/.

  python tools/summarize_profiles.py <tag> --stats gpurun_out/prof/x_kernel_stats.csv --pmc gpurun_out/pmc_a
"""
import argparse
import collections
import csv
import json
import os

KEEP = ("oxc::",)


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name.split("(")[0]


def kernel_stats(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if any(k in r["Name"] for k in KEEP):
                rows.append({"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3),
                             "min_us": round(float(r["MinNs"]) / 1e3, 3), "max_us": round(float(r["MaxNs"]) / 1e3, 3),
                             "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3)})
    return rows


def pmc(dirpath):
    out = collections.defaultdict(dict)
    for sub in sorted(os.listdir(dirpath)):
        p = os.path.join(dirpath, sub, "p_counter_collection.csv")
        if not os.path.exists(p):
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(p) as f:
            for r in csv.DictReader(f):
                if any(k in r["Kernel_Name"] for k in KEEP):
                    agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                out[k][c] = {"avg_per_launch": round(sum(v) / len(v), 1), "launches": len(v)}
    return out


def kernel_source_sha16():
    """Identifies the device code a profile belongs to: sha256 over the cull path's device sources (bench.py compares it with the tree it runs from)."""
    import hashlib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("oxcull_kernels.hip", "oxcull_kernels.hpp", "oxcull_device.hpp", "oxcull_types.hpp"):
        h.update(open(os.path.join(root, "oxylus_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--stats")
    ap.add_argument("--pmc")
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    doc = {"tag": a.tag, "note": a.note, "kernel_source_sha16": kernel_source_sha16()}
    if a.stats:
        doc["kernel_trace_stats"] = kernel_stats(a.stats)
        doc["kernel_trace_source"] = a.stats
    if a.pmc:
        p = pmc(a.pmc)
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB-ish units of 1024 B... reported raw here plus
        # the guide's correction: bytes = value * 1024, and FETCH_SIZE reads half of a wide coalesced stream -> x2.
        for k, cs in p.items():
            if "FETCH_SIZE" in cs:
                cs["hbm_read_bytes_corrected"] = round(cs["FETCH_SIZE"]["avg_per_launch"] * 1024 * 2)
            if "WRITE_SIZE" in cs:
                cs["hbm_write_bytes"] = round(cs["WRITE_SIZE"]["avg_per_launch"] * 1024)
        doc["pmc"] = p
        doc["pmc_source"] = a.pmc
    os.makedirs("profiles", exist_ok=True)
    path = os.path.join("profiles", a.tag + ".json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
