#!/bin/bash
# usage (GPU box, repo root): tools/r06_experiments.sh <tag>   -- the round's timing experiments, one after the other
set -o pipefail
tag=${1:-x}
mkdir -p gpurun_out
D=$PWD/build_ab/libgz_diag.so
A=$PWD/build_ab/libgz_ablate.so
(make -C genz-tokenize_amd/csrc diag && make -C genz-tokenize_amd/csrc ablate) > gpurun_out/exp_$tag.make 2>&1 || { tail -20 gpurun_out/exp_$tag.make; exit 1; }
echo "== self-classifying word kernel (ceiling of the fusion), on the timing-experiment build (no assertions, no stamps)"
GZ_LIBRARY=$A timeout -k 10 300 python3 tools/r06_fused_classify_try.py 1250000 6 100 > gpurun_out/fused_$tag.txt 2>&1 || { tail -20 gpurun_out/fused_$tag.txt; exit 1; }
tail -5 gpurun_out/fused_$tag.txt
echo "== row kernel, cycles per phase"
GZ_LIBRARY=$D timeout -k 10 300 python3 tools/prof_diag.py 1000000 3 > gpurun_out/rows1_stamps_$tag.txt 2>&1 || { tail -20 gpurun_out/rows1_stamps_$tag.txt; exit 1; }
GZ_LIBRARY=$D GZ_TEST_SWITCHES="rows_dbg=64" timeout -k 10 300 python3 tools/prof_diag.py 1000000 3 >> gpurun_out/rows1_stamps_$tag.txt 2>&1 || { tail -20 gpurun_out/rows1_stamps_$tag.txt; exit 1; }
cat gpurun_out/rows1_stamps_$tag.txt
echo "== encode_batch, where the time goes"
timeout -k 10 400 python3 tools/t_encode_batch.py > gpurun_out/encode_batch_$tag.txt 2>&1 || { tail -20 gpurun_out/encode_batch_$tag.txt; exit 1; }
cat gpurun_out/encode_batch_$tag.txt
echo "== size exchange of the exchange step over gloo, on this host"
(nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null) > gpurun_out/size_xchg_$tag.txt
# (CPU processes only: 8 ranks that can see the GPU would trip the box's process guard -- they are given no device)
HIP_VISIBLE_DEVICES= ROCR_VISIBLE_DEVICES= CUDA_VISIBLE_DEVICES= timeout -k 10 300 python -m pytest tests/test_distributed.py -k size_exchange -s -q >> gpurun_out/size_xchg_$tag.txt 2>&1 || { tail -20 gpurun_out/size_xchg_$tag.txt; exit 1; }
grep "gloo all_gather" gpurun_out/size_xchg_$tag.txt
