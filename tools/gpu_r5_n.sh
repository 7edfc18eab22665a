#!/bin/bash
S="@SHARE=1@UNORD=1"
L=oxylus_amd/liboxcull.so
python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 900 python tools/kbench.py --frames 60 --libs "new=$L$S,ord=$L@SHARE=1,defaults=$L,newb=$L$S,ordb=$L@SHARE=1" 2>&1 | tail -5 | cut -c1-300
timeout 900 python tools/kbench.py --tris 124 --frames 60 --libs "new=$L$S,ord=$L@SHARE=1" 2>&1 | tail -2 | cut -c1-300
