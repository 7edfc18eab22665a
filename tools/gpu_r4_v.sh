#!/bin/bash
V=oxylus_amd/variants
timeout 280 python tools/kbench.py --libs "nopf=$V/liboxcull_nopf.so@SHARE=1@UNORD=1,pf=oxylus_amd/liboxcull.so@SHARE=1@UNORD=1,pfnt=$V/liboxcull_pfnt.so@SHARE=1@UNORD=1,nopf2=$V/liboxcull_nopf.so@SHARE=1@UNORD=1,pf2=oxylus_amd/liboxcull.so@SHARE=1@UNORD=1" --frames 60 2>&1 | grep -v "^W\|rocprof" | tail -6
python -m pytest tests/test_gpu_share.py tests/test_gpu_round2.py tests/test_gpu_unordered.py -q -x 2>&1 | grep -E "passed|failed|rror" | tail -3
