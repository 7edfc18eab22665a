#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_unordered.py tests/test_gpu_share.py tests/test_gpu_fuzz.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r5c_pytest.txt; cat gpurun_out/r5c_pytest.txt
S="@SHARE=1@UNORD=1"
timeout 900 python tools/kbench.py --frames 80 --out gpurun_out/r5c_kbench.json --libs \
"r4=oxylus_amd/liboxcull.so$S@TUNE4=0,sel=oxylus_amd/liboxcull.so$S,r4b=oxylus_amd/liboxcull.so$S@TUNE4=0,selb=oxylus_amd/liboxcull.so$S,ord=oxylus_amd/liboxcull.so@SHARE=1@UNORD=0" 2>&1 | tail -20 > gpurun_out/r5c_kbench.txt; cat gpurun_out/r5c_kbench.txt
