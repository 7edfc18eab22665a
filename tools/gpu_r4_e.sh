#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=oxylus_amd/liboxcull.so
V=oxylus_amd/variants
( timeout 600 python tools/kbench.py --frames 60 --out gpurun_out/r4e_kbench.json --libs "base=$L@SHARE=1,hiznt0=$V/liboxcull_hiznt0.so@SHARE=1,hiznt2=$V/liboxcull_hiznt2.so@SHARE=1,emitnt=$V/liboxcull_emitnt.so@SHARE=1,u1=$L@SHARE=1@UNORD=1,base2=$L@SHARE=1" 2>&1 | tail -12 ) > gpurun_out/r4e_kbench.log 2>&1
cat gpurun_out/r4e_kbench.log
