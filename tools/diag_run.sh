#!/bin/bash
# usage (GPU box, repo root): tools/diag_run.sh [n_docs]   -- copies the diagnostic build (build_ab/libgz_diag.so) over the
# in-tree library FOR THIS RUN ONLY, prints the in-kernel stamp sums, restores the library
set -o pipefail
R=$PWD
LIB=$R/genz-tokenize_amd/genz_tokenize/libgenz_tokenize_hip.so
cp $LIB /tmp/orig.so && cp $R/build_ab/libgz_diag.so $LIB
timeout -k 10 300 python3 tools/prof_diag.py ${1:-1000000} 3; rc=$?
cp /tmp/orig.so $LIB
exit $rc
