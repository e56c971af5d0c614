#!/bin/bash
# usage (GPU box, repo root): tools/r06_rows_probe.sh <tag>  -- timing build: what bounds gz_rows1_kernel?  Whole-launch times (the text side is the same
# in every line: differences are the row kernel's) for: the product's kernel; stores only (rows_dbg bit 0: no word records -> every row bos, eos, padding;
# bit 4: no piece pass); stores only with many rounds per wave (rows_dpw: a wave that never waits for a load keeps storing); and the plain fill of the same bytes.
set -o pipefail
tag=${1:-x}
mkdir -p gpurun_out
A=$PWD/build_ab/libgz_ablate.so
make -C genz-tokenize_amd/csrc ablate > gpurun_out/rowsprobe_$tag.make 2>&1 || { tail -20 gpurun_out/rowsprobe_$tag.make; exit 1; }
run() { echo -n "$1  "; SEED=100 GZ_LIBRARY=$A GZ_TEST_SWITCHES="$2" timeout -k 10 200 python3 tools/prof_run.py 1250000 6 4 | sed 's/.*kernel ms//' | cut -c1-120; }
{
for rep in 1 2; do
run "product kernel (rows_dbg=0, 8 docs per wave)      " "rows_dbg=0"
run "stores only, 8 docs per wave (rows_dbg=17)          " "rows_dbg=17"
run "stores only, 16 docs per wave                       " "rows_dbg=17,rows_dpw=16"
run "stores only, 64 docs per wave                       " "rows_dbg=17,rows_dpw=64"
run "no stores at all (rows_dbg=6): loads + walk only    " "rows_dbg=6"
run "ids stores only (rows_dbg=2: no mask stores)        " "rows_dbg=2"
done
if [ -x tools/membw ] || hipcc --offload-arch=gfx950 -O3 -o tools/membw tools/membw.hip 2>/dev/null; then echo "== tools/membw"; timeout -k 10 120 tools/membw 2>&1 | tail -12; fi
} > gpurun_out/rowsprobe_$tag.txt 2>&1
cat gpurun_out/rowsprobe_$tag.txt
