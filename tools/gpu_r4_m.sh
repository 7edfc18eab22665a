#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python bench.py --steps 6 --warmup 2 ) > gpurun_out/r4m_bench.json 2> gpurun_out/r4m_bench.err
grep -E "bench\]|fault|real|Error" gpurun_out/r4m_bench.err | tail -12; wc -c gpurun_out/r4m_bench.json
