#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time OXC_BENCH_TRACE=1 timeout 900 python bench.py ) > gpurun_out/r4n_bench.json 2> gpurun_out/r4n_bench.err
grep -E "bench\]|fault|real|Error" gpurun_out/r4n_bench.err | tail -24; wc -c gpurun_out/r4n_bench.json
