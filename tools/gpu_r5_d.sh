#!/bin/bash
mkdir -p gpurun_out
S="@SHARE=1@UNORD=1"
L=oxylus_amd/liboxcull.so
timeout 900 python tools/kbench.py --frames 80 --out gpurun_out/r5d_kbench.json --libs \
"sel8=$L$S,sel10=$L$S@TUNE3=10,sel12=$L$S@TUNE3=12,sel16=$L$S@TUNE3=16,sel24=$L$S@TUNE3=24,sel64=$L$S@TUNE3=64,r4_8=$L$S@TUNE4=0,r4_16=$L$S@TUNE4=0@TUNE3=16,sel8b=$L$S,sel12b=$L$S@TUNE3=12,sel16b=$L$S@TUNE3=16,sel24b=$L$S@TUNE3=24" 2>&1 | tail -20 > gpurun_out/r5d_kbench.txt; cat gpurun_out/r5d_kbench.txt
