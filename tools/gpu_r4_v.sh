#!/bin/bash
V=oxylus_amd/variants
timeout 280 python tools/kbench.py --tris 124 --libs "base=oxylus_amd/liboxcull.so@SHARE=1@UNORD=1,ww5=$V/liboxcull_ww5.so@SHARE=1@UNORD=1,ww6=$V/liboxcull_ww6.so@SHARE=1@UNORD=1" --frames 60 2>&1 | grep -v "^W\|rocprof" | tail -12
python -m pytest tests/test_gpu_unordered.py tests/test_gpu_round2.py -q -k "wide or Wide or 124 or tris" 2>&1 | grep -E "passed|failed"
