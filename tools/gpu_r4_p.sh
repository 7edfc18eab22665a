#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=oxylus_amd/liboxcull.so
V=oxylus_amd/variants
( timeout 900 python -m pytest tests/test_gpu_share.py tests/test_gpu_fuzz.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 ) > gpurun_out/r4p_tests.log 2>&1
( timeout 600 python tools/kbench.py --frames 60 --out gpurun_out/r4p_kbench.json --libs "run0=$V/liboxcull_run0.so@SHARE=1@UNORD=1,run2=$L@SHARE=1@UNORD=1,run4=$V/liboxcull_run4.so@SHARE=1@UNORD=1,run1=$V/liboxcull_run1.so@SHARE=1@UNORD=1,run2w4=$V/liboxcull_run2w4.so@SHARE=1@UNORD=1,run4w4=$V/liboxcull_run4w4.so@SHARE=1@UNORD=1,run0b=$V/liboxcull_run0.so@SHARE=1@UNORD=1" 2>&1 | tail -12 ) > gpurun_out/r4p_kbench.log 2>&1
cat gpurun_out/r4p_tests.log gpurun_out/r4p_kbench.log
