#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_check.sh <tag>
# quick parity subset + kernel trace of 5 steady-state steps of the 1M-document workload
set -o pipefail
tag=${1:-x}
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/tests_$tag.log 2>&1 || { tail -30 gpurun_out/tests_$tag.log; exit 1; }; tail -2 gpurun_out/tests_$tag.log; fi
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t --output-format csv -- python3 $R/tools/prof_run.py 1000000 5 > $R/gpurun_out/prof_$tag.log 2>&1 || { tail -20 $R/gpurun_out/prof_$tag.log; exit 1; }
grep "^docs" $R/gpurun_out/prof_$tag.log
python3 $R/tools/trace_summary.py /tmp/prof_$tag | tee $R/gpurun_out/summary_$tag.txt
