#!/bin/bash
# A/B of two scatter_bench builds on the same box: tools/ab.sh binA binB [reps] -> scatter ms of the last variant line
A=$1; B=$2; R=${3:-4}
for i in $(seq $R); do
  a=$($A 28 2>&1 | grep "scatter" | tail -1 | sed 's/.*scatter \([0-9.]*\) ms.*/\1/')
  b=$($B 28 2>&1 | grep "scatter" | tail -1 | sed 's/.*scatter \([0-9.]*\) ms.*/\1/')
  echo "A $a  B $b"
done
