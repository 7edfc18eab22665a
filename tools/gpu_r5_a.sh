#!/bin/bash
# round 5, GPU run A: parity of the new fused triangle kernel (tickets + select) and an A/B of the variants on the configs[2] frame
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_unordered.py tests/test_gpu_share.py tests/test_gpu_async.py tests/test_gpu_fuzz.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5a_pytest.txt; cat gpurun_out/r5a_pytest.txt
V=oxylus_amd/variants
timeout 900 python tools/kbench.py --frames 60 --out gpurun_out/r5a_kbench.json --libs \
"r4=$V/liboxcull_r4.so@SHARE=1@UNORD=1@TUNE4=0,f2=oxylus_amd/liboxcull.so@SHARE=1@UNORD=1@TUNE4=0,f2sel=oxylus_amd/liboxcull.so@SHARE=1@UNORD=1,ov0=$V/liboxcull_ov0.so@SHARE=1@UNORD=1@TUNE4=0,ov0sel=$V/liboxcull_ov0.so@SHARE=1@UNORD=1,s64=$V/liboxcull_s64.so@SHARE=1@UNORD=1@TUNE4=0,s64sel=$V/liboxcull_s64.so@SHARE=1@UNORD=1,lean5=$V/liboxcull_lean5.so@SHARE=1@UNORD=1,lean6=$V/liboxcull_lean6.so@SHARE=1@UNORD=1,r4b=$V/liboxcull_r4.so@SHARE=1@UNORD=1@TUNE4=0,f2selb=oxylus_amd/liboxcull.so@SHARE=1@UNORD=1" 2>&1 | tail -20 > gpurun_out/r5a_kbench.txt; cat gpurun_out/r5a_kbench.txt
