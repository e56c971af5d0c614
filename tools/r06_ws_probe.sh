#!/bin/bash
# (the kernel and its switch rows_ws exist in commits d395452 and 43d6d23 only: the experiment lost, see profiles/r06_ab_variants.txt series 6 -- check one of them out to run this)
# usage (GPU box, repo root): tools/r06_ws_probe.sh <tag>  -- timing build: which side bounds gz_rows1ws_kernel?  (whole-launch times; the text side is constant)
set -o pipefail
tag=${1:-x}
mkdir -p gpurun_out
A=$PWD/build_ab/libgz_ablate.so
make -C genz-tokenize_amd/csrc ablate > gpurun_out/wsprobe_$tag.make 2>&1 || { tail -20 gpurun_out/wsprobe_$tag.make; exit 1; }
run() { echo -n "$1  "; SEED=100 GZ_LIBRARY=$A GZ_TEST_SWITCHES="$2" timeout -k 10 200 python3 tools/prof_run.py 1250000 6 4 | sed 's/.*kernel ms//' | cut -c1-120; }
{
for rep in 1 2; do
run "old kernel                                         " "rows_ws=0"
run "ws kernel                                          " "rows_ws=2"
run "ws, consumer stores nothing (rows_dbg=6)           " "rows_ws=2,rows_dbg=6"
run "ws, producers load nothing (rows_dbg=17)           " "rows_ws=2,rows_dbg=17"
run "old, loads + walk only (rows_dbg=6)                " "rows_ws=0,rows_dbg=6"
run "old, stores only (rows_dbg=17)                     " "rows_ws=0,rows_dbg=17"
done
} > gpurun_out/wsprobe_$tag.txt 2>&1
cat gpurun_out/wsprobe_$tag.txt
