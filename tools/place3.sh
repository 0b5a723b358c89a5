#!/bin/bash
run() { echo -n "place $1: "; SB_PLACE=$1 SB_QUICK=1 ./tools/scatter_bench 28 2>&1 | grep -E "^carry" | sed 's/carry bits.*scatter \([0-9.]*\) ms.*/\1 ms/'; }
M=1048576; G=1073741824
for d in $((G)) $((G+4096)) $((G+65536)) $((G+2*M)) $((G+16*M)) $((G+64*M)) $((G-64*M)) $((G-2*M)) $((G+256*M)) $((3*G)) $((3*G+2*M)) $((3*G+64*M)) $((7*G)); do run "0,0,$d"; done
echo "--- src distance (vals - keys = 1 GiB + d1), dst at fast distance"
for d in 0 $((2*M)) $((G)) $((G+2*M)) $((3*G)); do run "$d,0,$G"; done
