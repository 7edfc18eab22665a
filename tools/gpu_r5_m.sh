#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -3
S="@SHARE=1@UNORD=1"
L=oxylus_amd/liboxcull.so
timeout 900 python tools/kbench.py --frames 80 --out gpurun_out/r5m_kbench.json --libs "r4static=oxylus_amd/variants/liboxcull_dyn0.so$S,dynamic=$L$S,r4staticb=oxylus_amd/variants/liboxcull_dyn0.so$S,dynamicb=$L$S,ordered=$L@SHARE=1@UNORD=0,defaults=$L@UNORD=0" 2>&1 | tail -7 > gpurun_out/r5m_kbench.txt; cut -c1-260 gpurun_out/r5m_kbench.txt
