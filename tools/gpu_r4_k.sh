#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { tag=$1; shift; ( timeout 1200 python bench.py "$@" > gpurun_out/r4k_$tag.json 2> gpurun_out/r4k_$tag.err; echo "$tag rc=$? bytes=$(wc -c < gpurun_out/r4k_$tag.json)"; grep -E "fault|Error|error" gpurun_out/r4k_$tag.err | head -3 ); }
run real --steps 4 --warmup 1 --no-configs1 --no-configs4 --no-tris124 --no-cpu-baseline --no-scheduling-ab
run cpu --steps 4 --warmup 1 --no-configs1 --no-configs4 --no-tris124 --no-real-geometry --no-scheduling-ab
