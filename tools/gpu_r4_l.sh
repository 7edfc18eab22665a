#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { tag=$1; shift; ( timeout 1200 python bench.py "$@" > gpurun_out/r4l_$tag.json 2> gpurun_out/r4l_$tag.err; echo "$tag rc=$?"; python -c "
import json,sys
d=json.loads(open('gpurun_out/r4l_$tag.json').read().strip().splitlines()[-1]); print(d['bit_match'], d['bit_match_detail'], d['counts'])" ); }
C="--steps 2 --warmup 1 --no-configs1 --no-configs4 --no-tris124 --no-real-geometry --no-cpu-baseline --no-scheduling-ab --tris 124"
run w_u1_250 $C --unordered-output 1 --cpu-prefix 250
run w_u0_250 $C --unordered-output 0 --cpu-prefix 250
run w_u1_1000 $C --unordered-output 1 --cpu-prefix 1000
