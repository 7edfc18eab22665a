#!/bin/bash
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python bench.py > gpurun_out/r5f_bench.json 2> gpurun_out/r5f_bench.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"; grep -E "bench\]|Error|error" gpurun_out/r5f_bench.err | tail -12; tail -c 2200 gpurun_out/r5f_bench.json
