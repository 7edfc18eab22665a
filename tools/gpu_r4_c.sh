#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=oxylus_amd/liboxcull.so
V=oxylus_amd/variants
( timeout 600 python tools/kbench.py --frames 60 --out gpurun_out/r4c_kbench.json --libs "base=$L@SHARE=1,u1=$L@SHARE=1@UNORD=1,u1fs128=$V/liboxcull_fs128.so@SHARE=1@UNORD=1,u1fs64=$V/liboxcull_fs64.so@SHARE=1@UNORD=1,base2=$L@SHARE=1" 2>&1 | tail -12 ) > gpurun_out/r4c_kbench.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_unordered.py -x -q 2>&1 | tail -3 ) > gpurun_out/r4c_tests.log 2>&1
OXC_LIB_PATH=$V/liboxcull_fs64.so timeout 600 python -m pytest tests/test_gpu_unordered.py -x -q 2>&1 | tail -3 >> gpurun_out/r4c_tests.log
cat gpurun_out/r4c_kbench.log gpurun_out/r4c_tests.log
