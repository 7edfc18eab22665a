#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=oxylus_amd/liboxcull.so
V=oxylus_amd/variants
( timeout 600 python tools/kbench.py --frames 60 --out gpurun_out/r4f_kbench.json --libs "old=$V/liboxcull_old.so@SHARE=1,new=$L@SHARE=1,wident0=$V/liboxcull_wident0.so@SHARE=1,run2048=$V/liboxcull_run2048.so@SHARE=1,run512=$V/liboxcull_run512.so@SHARE=1,newu1=$L@SHARE=1@UNORD=1,old2=$V/liboxcull_old.so@SHARE=1" 2>&1 | tail -12 ) > gpurun_out/r4f_kbench.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_unordered.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py -x -q 2>&1 | tail -4 ) > gpurun_out/r4f_tests.log 2>&1
cat gpurun_out/r4f_kbench.log gpurun_out/r4f_tests.log
