#!/bin/bash
# usage (GPU box, repo root): tools/r06_more.sh <tag>  -- merge-loop phase stamps (diagnostic build), rounds per wave of the row kernel (timing build), new tests
set -o pipefail
tag=${1:-x}
mkdir -p gpurun_out
D=$PWD/build_ab/libgz_diag.so; A=$PWD/build_ab/libgz_ablate.so
(make -C genz-tokenize_amd/csrc diag && make -C genz-tokenize_amd/csrc ablate) > gpurun_out/more_$tag.make 2>&1 || { tail -20 gpurun_out/more_$tag.make; exit 1; }
echo "== new / changed tests on the product build"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense_host_path or csr_host_path or encode_batch_large or emits_its_exchange or compact_block or allocation_failure or pinned_array" > gpurun_out/tests_$tag.log 2>&1 || { tail -60 gpurun_out/tests_$tag.log; exit 1; }
tail -3 gpurun_out/tests_$tag.log
echo "== merge kernel, cycles per phase: tables on, then every word through the merge loop"
GZ_LIBRARY=$D timeout -k 10 300 python3 tools/prof_diag.py 1000000 3 > gpurun_out/merge_stamps_$tag.txt 2>&1 || { tail -20 gpurun_out/merge_stamps_$tag.txt; exit 1; }
NO_WORD_TABLE=1 GZ_LIBRARY=$D timeout -k 10 300 python3 tools/prof_diag.py 1000000 3 >> gpurun_out/merge_stamps_$tag.txt 2>&1 || { tail -20 gpurun_out/merge_stamps_$tag.txt; exit 1; }
cat gpurun_out/merge_stamps_$tag.txt | cut -c1-400
echo "== row kernel: documents per wave (8 = one round per wave, the product's)"
for k in 8 16 8 16 4 8 24 8; do
  echo -n "rows_dpw=$k  "; GZ_LIBRARY=$A GZ_TEST_SWITCHES="rows_dpw=$k" timeout -k 10 200 python3 tools/prof_run.py 1000000 6 | cut -c1-200
done > gpurun_out/rows_dpw_$tag.txt 2>&1
cat gpurun_out/rows_dpw_$tag.txt
