#!/bin/bash
V=oxylus_amd/variants
timeout 280 python tools/kbench.py --libs "base=oxylus_amd/liboxcull.so@SHARE=1@UNORD=1,slw4=$V/liboxcull_slw4.so@SHARE=1@UNORD=1,slw3=$V/liboxcull_slw3.so@SHARE=1@UNORD=1" --frames 60 2>&1 | grep -v "^W\|rocprof" | tail -12
