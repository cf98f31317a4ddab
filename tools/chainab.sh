#!/bin/bash
# per-layer chain (inverse direction) A/B over variant libraries: D = 32 and 64
cd /root/repo
V=1
for D in 32 64; do echo "== D=$D"; TNF_FUSION=$V timeout -k 10 500 python tools/abrun.py "$1" 2 $D; done
