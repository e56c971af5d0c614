#!/bin/bash
# usage (GPU box, repo root): tools/ab_plain.sh <libA.so> <libB.so> ...  -- hipEvent step time without a profiler attached
R=$PWD
LIB=$R/genz-tokenize_amd/genz_tokenize/libgenz_tokenize_hip.so
cp $LIB /tmp/orig.so
for rep in 1 2 3; do
  for so in "$@"; do
    cp $R/$so $LIB
    echo "== $(basename $so .so)_$rep: $(timeout -k 10 300 python3 $R/tools/prof_run.py ${NDOCS:-1000000} 8 2>&1 | grep '^docs' | sed 's/.*kernel ms//')"
  done
done
cp /tmp/orig.so $LIB
