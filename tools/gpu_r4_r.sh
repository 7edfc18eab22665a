#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
( time timeout 900 python bench.py --workload config2 --no-cpu-baseline ) > gpurun_out/r4r_c2_$i.json 2> gpurun_out/r4r_c2_$i.err
echo "run $i: bytes=$(wc -c < gpurun_out/r4r_c2_$i.json)"; grep -E "fault|real|Error" gpurun_out/r4r_c2_$i.err | tail -2
done
