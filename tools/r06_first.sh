#!/bin/bash
# usage (GPU box, repo root): tools/r06_first.sh <tag>
# first-touch rates of the host, the GPU suite on the product build, the bench line
set -o pipefail
tag=${1:-r6a}
mkdir -p gpurun_out
gcc -O2 -pthread -o tools/micro/hostfill tools/micro/hostfill.c && tools/micro/hostfill 2 > gpurun_out/hostfill_$tag.txt 2>&1
tail -30 gpurun_out/hostfill_$tag.txt
nproc; grep -c processor /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/tests_$tag.log 2>&1 || { tail -60 gpurun_out/tests_$tag.log; exit 1; }
tail -3 gpurun_out/tests_$tag.log
timeout -k 10 600 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -30 gpurun_out/bench_$tag.err; exit 1; }
python3 tools/show_bench.py gpurun_out/bench_$tag.json 2>/dev/null | head -60 || head -c 3000 gpurun_out/bench_$tag.json
