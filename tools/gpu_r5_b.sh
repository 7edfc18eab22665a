#!/bin/bash
# round 5, GPU run B: the select form on the round-4 kernel structure, ticket fixes of the new structure
mkdir -p gpurun_out
V=oxylus_amd/variants
S="@SHARE=1@UNORD=1"
timeout 900 python tools/kbench.py --frames 80 --out gpurun_out/r5b_kbench.json --libs \
"r4=$V/liboxcull_r4.so$S@TUNE4=0,r4sel=$V/liboxcull_r4.so$S,r4s64=$V/liboxcull_r4s64.so$S@TUNE4=0,r4s64sel=$V/liboxcull_r4s64.so$S,f2=oxylus_amd/liboxcull.so$S@TUNE4=0,f2sel=oxylus_amd/liboxcull.so$S,f2run=$V/liboxcull_f2run.so$S@TUNE4=0,f2runsel=$V/liboxcull_f2run.so$S,f2sr1=$V/liboxcull_f2sr1.so$S@TUNE4=0,f2s64=$V/liboxcull_f2s64.so$S@TUNE4=0,f2s64sel=$V/liboxcull_f2s64.so$S,f2ov0=$V/liboxcull_f2ov0.so$S@TUNE4=0,r4b=$V/liboxcull_r4.so$S@TUNE4=0,r4selb=$V/liboxcull_r4.so$S" 2>&1 | tail -20 > gpurun_out/r5b_kbench.txt; cat gpurun_out/r5b_kbench.txt
