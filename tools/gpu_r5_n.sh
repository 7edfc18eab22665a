#!/bin/bash
S="@SHARE=1@UNORD=1"
L=oxylus_amd/liboxcull.so
V=oxylus_amd/variants
timeout 900 python tools/kbench.py --frames 60 --libs "base=$L$S,no5=$V/liboxcull_no5.so$S,no6=$V/liboxcull_no6.so$S,no8=$V/liboxcull_no8.so$S" 2>&1 | tail -4 | cut -c1-140
