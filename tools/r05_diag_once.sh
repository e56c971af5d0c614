#!/bin/bash
# usage (GPU box, repo root): tools/r05_diag_once.sh <library under build_ab/> <tag>
# ONE run of the selection of tests/test_gpu_parity.py::test_index_assertions_on_a_poisoned_workspace on a diagnostic build
# (index assertions + poisoned workspace), whole stdout / stderr kept under gpurun_out/.
SEL="g1_cases or g3_random or g4_loader or cfg2_10k or cfg3_20k or noisy_corpus or noisy_pairs or long_and_huge or random_tables_fuzz or extreme_batch or not_utf8"
GZ_LIBRARY=$PWD/build_ab/$1 GZ_TEST_SWITCHES="small=0,scan_multi=0,diag_poison=1,brk_side=0" GZ_TABLE_CACHE=off AMD_LOG_LEVEL=1 \
  timeout -k 10 500 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "$SEL" \
  > gpurun_out/diag_$2.out 2> gpurun_out/diag_$2.err
rc=$?
echo "diag $2: rc $rc"; tail -3 gpurun_out/diag_$2.out; head -c 1500 gpurun_out/diag_$2.err
exit $rc
