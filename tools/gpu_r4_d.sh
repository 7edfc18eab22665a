#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=oxylus_amd/liboxcull.so
V=oxylus_amd/variants
( timeout 600 python tools/kbench.py --frames 60 --out gpurun_out/r4d_kbench.json --libs "base=$L@SHARE=1,hiznt=$V/liboxcull_hiznt.so@SHARE=1,u1=$L@SHARE=1@UNORD=1,u1hiznt=$V/liboxcull_hiznt.so@SHARE=1@UNORD=1,base2=$L@SHARE=1" 2>&1 | tail -12 ) > gpurun_out/r4d_kbench.log 2>&1
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r4d_tests.log 2>&1
cat gpurun_out/r4d_kbench.log gpurun_out/r4d_tests.log
