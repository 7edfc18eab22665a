#!/bin/bash
S="@SHARE=1@UNORD=1"
L=oxylus_amd/liboxcull.so
V=oxylus_amd/variants
timeout 900 python tools/kbench.py --frames 80 --libs "base=$L$S,em2=$V/liboxcull_em2.so$S,baseb=$L$S,em2b=$V/liboxcull_em2.so$S" 2>&1 | tail -4 | cut -c1-300
