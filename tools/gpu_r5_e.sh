#!/bin/bash
mkdir -p gpurun_out
S="@SHARE=1@UNORD=1"
L=oxylus_amd/liboxcull.so
T=oxylus_amd/variants/liboxcull_tk.so
timeout 900 python tools/kbench.py --frames 80 --out gpurun_out/r5e_kbench.json --libs \
"sel=$L$S,tksel=$T$S,r4=$L$S@TUNE4=0,tk=$T$S@TUNE4=0,selb=$L$S,tkselb=$T$S,tksel16=$T$S@TUNE3=16" 2>&1 | tail -20 > gpurun_out/r5e_kbench.txt; cat gpurun_out/r5e_kbench.txt
