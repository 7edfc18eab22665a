#!/usr/bin/env python3
"""tools/isa_summary.py [out.json] -- per-kernel resource usage of liboxcull's device code, read from the code-object
metadata hipcc emits (no GPU needed): VGPRs / SGPRs / spills / LDS / scratch, the waves per SIMD the VGPR count allows
on gfx950 (512 VGPRs per SIMD lane, allocation granule 8, at most 8 waves), and a static instruction mix."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "oxylus_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def waves_per_simd(vgprs):
    alloc = max(8, -(-vgprs // 8) * 8)
    return min(8, 512 // alloc)


def summarize(src):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *FLAGS, "--cuda-device-only", "-S", src, "-o", asm], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    kernels = {}
    # metadata: one YAML map per kernel under amdhsa.kernels
    for block in text.split("  - .agpr_count:")[1:]:
        def field(name, cast=int, default=0):
            m = re.search(r"\.%s:\s+(\S+)" % re.escape(name), block)
            return cast(m.group(1)) if m else default
        name = field("name", str, "")
        if not name:
            continue
        kernels[name] = {"vgprs": field("vgpr_count"), "sgprs": field("sgpr_count"), "sgpr_spills": field("sgpr_spill_count"),
                         "vgpr_spills": field("vgpr_spill_count"), "lds_bytes": field("group_segment_fixed_size"),
                         "scratch_bytes": field("private_segment_fixed_size"), "kernarg_bytes": field("kernarg_segment_size"),
                         "max_flat_workgroup_size": field("max_flat_workgroup_size")}
    # static instruction mix per kernel body
    for name in kernels:
        m = re.search(r"^%s:.*?^\s+s_endpgm" % re.escape(name), text, re.S | re.M)
        body = m.group(0) if m else ""
        ops = re.findall(r"^\s+([a-z][a-z0-9_]+)", body, re.M)
        mix = {"valu": 0, "valu_packed_f32": 0, "salu": 0, "smem": 0, "vmem_load": 0, "vmem_store": 0, "lds": 0, "atomic": 0}
        for op in ops:
            if op.startswith("v_pk_") and "f32" in op:
                mix["valu_packed_f32"] += 1
            if op.startswith("v_"):
                mix["valu"] += 1
            elif op.startswith("s_load") or op.startswith("s_buffer_load"):
                mix["smem"] += 1
            elif op.startswith("s_"):
                mix["salu"] += 1
            elif "atomic" in op:
                mix["atomic"] += 1
            elif op.startswith("global_load") or op.startswith("flat_load") or op.startswith("buffer_load"):
                mix["vmem_load"] += 1
            elif op.startswith("global_store") or op.startswith("flat_store") or op.startswith("buffer_store"):
                mix["vmem_store"] += 1
            elif op.startswith("ds_"):
                mix["lds"] += 1
        kernels[name]["static_instructions"] = mix
        kernels[name]["waves_per_simd_by_vgprs"] = waves_per_simd(kernels[name]["vgprs"])
    pretty = demangle(list(kernels))
    return {pretty[k]: v for k, v in sorted(kernels.items(), key=lambda kv: pretty[kv[0]])}


def main():
    import argparse

    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("out", nargs="?", default="", help="write the summary as JSON to this path (default: print only)")
    args = ap.parse_args()
    out = {"note": "hipcc --offload-arch=gfx950 %s --cuda-device-only -S; code-object metadata + static instruction counts "
                   "(both branches of every conditional are counted: NOT a dynamic mix)" % " ".join(f for f in FLAGS if not f.startswith("-I")),
           "files": {}}
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip"):
            out["files"][f] = summarize(os.path.join(CSRC, f))
    text = json.dumps(out, indent=1, sort_keys=True)
    if args.out:
        open(args.out, "w").write(text + "\n")
    for f, ks in out["files"].items():
        for k, v in ks.items():
            print("%-22s %-70s vgpr %3d  sgpr %3d  spills %2d/%d  lds %6d  waves/SIMD %d" % (f, k[:70], v["vgprs"], v["sgprs"], v["sgpr_spills"], v["vgpr_spills"],
                                                                                   v["lds_bytes"], v["waves_per_simd_by_vgprs"]))


if __name__ == "__main__":
    main()
