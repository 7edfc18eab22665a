#!/bin/bash
S="@SHARE=1@UNORD=1"
L=oxylus_amd/liboxcull.so
V=oxylus_amd/variants
timeout 900 python tools/kbench.py --frames 80 --libs "base=$L$S,es=$V/liboxcull_es.so$S,tc4=$V/liboxcull_tc4.so$S,tc64=$V/liboxcull_tc64.so$S,es64=$V/liboxcull_es64.so$S,baseb=$L$S,esb=$V/liboxcull_es.so$S,tc64b=$V/liboxcull_tc64.so$S" 2>&1 | tail -8 | cut -c1-250
