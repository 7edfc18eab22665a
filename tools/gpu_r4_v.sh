#!/bin/bash
V=oxylus_amd/variants
timeout 280 python tools/kbench.py --libs "base=oxylus_amd/liboxcull.so@SHARE=1@UNORD=1,no_occlusion=$V/liboxcull_abl1.so@SHARE=1@UNORD=1,no_taps=$V/liboxcull_abl2.so@SHARE=1@UNORD=1,no_frustum=$V/liboxcull_abl3.so@SHARE=1@UNORD=1" --frames 60 2>&1 | grep -v "^W\|rocprof" | tail -5
