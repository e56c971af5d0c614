#!/bin/bash
# usage (GPU box, repo root): tools/r06_ids_ab.sh <tag>  -- the merge kernel's id phase (all loads, then the stores) against the build before, same box,
# traced, alternating: the headline shard, the roofline workload with the whole-word tables off, 20 % typos; then parity of what it touches
set -o pipefail
tag=${1:-x}
mkdir -p gpurun_out
{
echo "== one shard of the headline (1.25 M documents, seed 100, cfg 4)"; NDOCS=1250000 CFG=4 SEED=100 REPS=2 tools/ab.sh build_ab/r6_before.so build_ab/r6_ids.so
echo "== every word through the merge loop (1 M documents)"; NO_WORD_TABLE=1 REPS=2 tools/ab.sh build_ab/r6_before.so build_ab/r6_ids.so
echo "== 20 % of the words replaced by random letters (200 k documents)"; TYPOS=0.2 NDOCS=200000 REPS=2 tools/ab.sh build_ab/r6_before.so build_ab/r6_ids.so
} > gpurun_out/ids_ab_$tag.txt 2>&1 || { tail -30 gpurun_out/ids_ab_$tag.txt; exit 1; }
cat gpurun_out/ids_ab_$tag.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "g1_cases or g3_random or g5_reference or noisy or long_and_huge or random_tables_fuzz or not_utf8 or alternative_kernels or overflow_buckets or cfg5_full" > gpurun_out/tests_$tag.log 2>&1 || { tail -60 gpurun_out/tests_$tag.log; exit 1; }
tail -3 gpurun_out/tests_$tag.log
tools/r06_reprobe.sh $tag
