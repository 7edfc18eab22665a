#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=oxylus_amd/liboxcull.so
V=oxylus_amd/variants
( timeout 600 python tools/kbench.py --frames 60 --out gpurun_out/r4g_kbench.json --libs "new=$L@SHARE=1,async=$L@SHARE=1@ASYNC=1,asyncu1=$L@SHARE=1@ASYNC=1@UNORD=1,newu1=$L@SHARE=1@UNORD=1,async_t4=$L@SHARE=1@ASYNC=1@TUNE1=4,async_m4=$L@SHARE=1@ASYNC=1@TUNE0=4,new2=$L@SHARE=1" 2>&1 | tail -12 ) > gpurun_out/r4g_kbench.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_unordered.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py tests/test_gpu_round2.py -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 ) > gpurun_out/r4g_tests.log 2>&1
cat gpurun_out/r4g_kbench.log gpurun_out/r4g_tests.log
