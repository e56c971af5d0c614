#!/bin/bash
# usage (GPU box, repo root): tools/ab_small.sh <libA.so> <libB.so> ...  -- the small-batch numbers (tools/small_bench.py)
# of several builds on the SAME box, each build three times in turn.
R=$PWD
LIB=$R/genz-tokenize_amd/genz_tokenize/libgenz_tokenize_hip.so
cp $LIB /tmp/orig.so
for rep in 1 2 3; do
  for so in "$@"; do
    cp $R/$so $LIB
    echo "== $(basename $so .so) $rep"
    SMALL_ONLY=1 timeout -k 10 120 python3 $R/tools/small_bench.py || { cp /tmp/orig.so $LIB; exit 1; }
  done
done
cp /tmp/orig.so $LIB
