#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 ) > gpurun_out/r4q_tests.log 2>&1
for i in 1 2; do
( time OXC_BENCH_TRACE=1 timeout 900 python bench.py ) > gpurun_out/r4q_bench$i.json 2> gpurun_out/r4q_bench$i.err
grep -E "fault|real|Error" gpurun_out/r4q_bench$i.err | tail -3; wc -c gpurun_out/r4q_bench$i.json
done
cat gpurun_out/r4q_tests.log
