#!/bin/bash
# usage (GPU box, repo root): tools/diag_run.sh [n_docs]   -- runs the in-kernel stamp report on the diagnostic build
# (build_ab/libgz_diag.so, `make -C genz-tokenize_amd/csrc diag`) through GZ_LIBRARY: the product library is never touched
set -o pipefail
R=$PWD
export GZ_LIBRARY=$R/build_ab/libgz_diag.so
timeout -k 10 300 python3 tools/prof_diag.py ${1:-1000000} 3
