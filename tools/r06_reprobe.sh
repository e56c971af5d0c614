#!/bin/bash
# usage (GPU box, repo root): tools/r06_reprobe.sh <tag>  -- timing build: the merge loop that never probes a pair twice (ablate bit 1: WRONG results,
# a lower bound of the probing's cost) against the exact one, alternating, on the three workloads the round-5 verdict's targets are quoted on
set -o pipefail
tag=${1:-x}
mkdir -p gpurun_out
A=$PWD/build_ab/libgz_ablate.so
make -C genz-tokenize_amd/csrc ablate > gpurun_out/reprobe_$tag.make 2>&1 || { tail -20 gpurun_out/reprobe_$tag.make; exit 1; }
{
echo "== every word through the merge loop (merge_loop_only: 1 M documents, whole-word tables off)"
for a in 0 2 0 2; do echo -n "ablate=$a  "; NO_WORD_TABLE=1 GZ_LIBRARY=$A GZ_TEST_SWITCHES="ablate=$a" timeout -k 10 200 python3 tools/prof_run.py 1000000 5 | cut -c1-220; done
echo "== 20 % of the words replaced by random letters (oov_sensitivity.rate_0.20: 200 k documents)"
for a in 0 2 0 2; do echo -n "ablate=$a  "; TYPOS=0.2 GZ_LIBRARY=$A GZ_TEST_SWITCHES="ablate=$a" timeout -k 10 200 python3 tools/prof_run.py 200000 7 | cut -c1-220; done
echo "== one shard of the headline (1.25 M documents, seed 100, tables on)"
for a in 0 2 0 2; do echo -n "ablate=$a  "; SEED=100 GZ_LIBRARY=$A GZ_TEST_SWITCHES="ablate=$a" timeout -k 10 200 python3 tools/prof_run.py 1250000 6 4 | cut -c1-220; done
} > gpurun_out/reprobe_$tag.txt 2>&1
cat gpurun_out/reprobe_$tag.txt
