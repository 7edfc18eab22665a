#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { tag=$1; shift; ( timeout 900 python bench.py "$@" > gpurun_out/r4j_$tag.json 2> gpurun_out/r4j_$tag.err; echo "$tag rc=$? bytes=$(wc -c < gpurun_out/r4j_$tag.json)"; grep -E "fault|Error|error" gpurun_out/r4j_$tag.err | head -3 ); }
run main --steps 4 --warmup 1 --no-configs1 --no-configs4 --no-real-geometry --no-tris124 --no-cpu-baseline
run tris124 --steps 4 --warmup 1 --no-configs1 --no-configs4 --no-real-geometry --no-cpu-baseline --no-scheduling-ab
run configs1 --workload config2 --steps 4 --warmup 1 --no-cpu-baseline
run configs4 --workload config5 --no-cpu-baseline
