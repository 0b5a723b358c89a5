#!/bin/bash
# DMA variant line ("carry dma bits 8 threads 1024 kpt 12") of several builds, alternating runs on one box
R=$1; shift
declare -A RES
for i in $(seq $R); do
  for b in "$@"; do
    out=$(SB_DMA=1 $b 28 2>&1)
    t=$(echo "$out" | grep -E "^carry dma bits 8 threads 1024 kpt 12" | head -1 | sed 's/.*scatter \([0-9.]*\) ms.*/\1/')
    t0=$(echo "$out" | grep -E "^carry bits 8 threads 1024 kpt 12" | head -1 | sed 's/.*scatter \([0-9.]*\) ms.*/\1/')
    RES[$b]="${RES[$b]} $t0/$t"
  done
done
for b in "$@"; do echo "$b (plain/dma): ${RES[$b]}"; done
