#!/bin/bash
# (the kernel and its switch rows_ws exist in commits d395452 and 43d6d23 only: the experiment lost, see profiles/r06_ab_variants.txt series 6 -- check one of them out to run this)
# usage (GPU box, repo root): tools/r06_ws.sh <tag>  -- gz_rows1ws_kernel (loads and stores of a round in different waves): parity with the switch forced on
# for every batch size, then the launch time against gz_rows1_kernel, alternating on this box (product build, switch rows_ws)
set -o pipefail
tag=${1:-x}
mkdir -p gpurun_out
GZ_TEST_SWITCHES="rows_ws=2" timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "g1_cases or g3_random or g5_reference or cfg4_shard or noisy_corpus or large_noisy or long_and_huge or extreme_batch or not_utf8 or deterministic or chained_device or device_entry_points or small_kernel_shapes or small_path_random or dense_host_path or encode_emits" > gpurun_out/tests_ws_$tag.log 2>&1 || { tail -60 gpurun_out/tests_ws_$tag.log; exit 1; }
tail -3 gpurun_out/tests_ws_$tag.log
run() { echo -n "$1  "; SEED=${4:-100} GZ_TEST_SWITCHES="$2" timeout -k 10 200 python3 tools/prof_run.py ${3:-1250000} 7 ${5:-4} | sed 's/.*kernel ms//' | cut -c1-140; }
{
echo "== one shard of the headline (1.25 M documents)"
for rep in 1 2 3; do run "rows_ws=0 (gz_rows1_kernel)  " "rows_ws=0"; run "rows_ws=2 (gz_rows1ws_kernel)" "rows_ws=2"; done
echo "== BASELINE configs[2] (1 M documents)"
for rep in 1 2; do run "rows_ws=0" "rows_ws=0" 1000000 "" 3; run "rows_ws=2" "rows_ws=2" 1000000 "" 3; done
echo "== 200 k documents"
for rep in 1 2; do run "rows_ws=0" "rows_ws=0" 200000 "" 3; run "rows_ws=2" "rows_ws=2" 200000 "" 3; done
echo "== 50 k documents"
for rep in 1 2; do run "rows_ws=0" "rows_ws=0" 50000 "" 3; run "rows_ws=2" "rows_ws=2" 50000 "" 3; done
} > gpurun_out/ws_ab_$tag.txt 2>&1
cat gpurun_out/ws_ab_$tag.txt
