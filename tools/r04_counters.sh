#!/bin/bash
# usage (GPU box, repo root): tools/r04_counters.sh <tag>
# The evidence the round-3 verdict asked for, on whatever build is in the tree:
#   (1) SQ / TCP / TCC counters of the pipeline's kernels on BASELINE configs[2] (1 M documents), one --pmc pass per group
#       (no trace domains in a pass), summarised per kernel;
#   (2) BASELINE configs[4] (custom tables, 50 k documents of <= 4 000 characters, max_len 1024): kernel trace + FETCH_SIZE / WRITE_SIZE;
#   (3) the same for configs[2] with the whole-word tables off (every word through the merge loop);
#   (4) TA counters LAST (a TA pass once crashed the profiler: nothing follows it).
set -o pipefail
tag=$1
R=$PWD
O=$R/gpurun_out/r04_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {   # pass <name> <script + args> -- <counters>
  local name=$1; shift
  local cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
  rm -rf $O/pmc_$name
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 "${cmd[@]}" > $O/pmc_$name.log 2>&1
}
trace() {  # trace <name> <script + args>
  local name=$1; shift
  rm -rf /tmp/tr_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/tr_$name -o t --output-format csv -- python3 "$@" > $O/trace_$name.log 2>&1 || return 1
  cp $(ls /tmp/tr_$name/*kernel_stats.csv /tmp/tr_$name/*/*kernel_stats.csv 2>/dev/null | head -1) $O/${name}_kernel_stats.csv
  python3 $R/tools/trace_summary.py /tmp/tr_$name > $O/${name}_steady_state.txt
  grep '^docs' $O/trace_$name.log
  tail -1 $O/${name}_steady_state.txt
}
P=$R/tools/prof_run.py
# (1)
pass sq1 $P 1000000 2 -- SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES || exit 1
pass sq2 $P 1000000 2 -- SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
pass sq3 $P 1000000 2 -- SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH || exit 1
pass tcp $P 1000000 2 -- TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum || exit 1
pass tcc $P 1000000 2 -- TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum || exit 1
python3 $R/tools/pmc_summary.py $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 $O/pmc_tcp $O/pmc_tcc > $O/pmc_sq_counters.txt
echo "(1) counters done"; echo "(1)" > $O/progress.txt
# kernel durations of the same build and workload (for cycles = duration x clock)
trace cfg3 $P 1000000 5 || exit 1
# (2)
trace cfg5 $R/tools/prof_cfg5.py 5 || exit 1
pass cfg5_fetch $R/tools/prof_cfg5.py 2 -- FETCH_SIZE || exit 1
pass cfg5_write $R/tools/prof_cfg5.py 2 -- WRITE_SIZE || exit 1
( cd $R && python3 tools/pmc_traffic.py $O/pmc_cfg5_fetch $O/pmc_cfg5_write $O/pmc_traffic_cfg5.json "" "" "BASELINE configs[4]: custom tables, 50 000 documents of <= 4 000 characters, max_len 1024 (tools/prof_cfg5.py 2)" )
echo "(2)" >> $O/progress.txt
# (3)
export NO_WORD_TABLE=1
trace mergeonly $P 1000000 3 || exit 1
pass mo_fetch $P 1000000 2 -- FETCH_SIZE || exit 1
pass mo_write $P 1000000 2 -- WRITE_SIZE || exit 1
unset NO_WORD_TABLE
( cd $R && python3 tools/pmc_traffic.py $O/pmc_mo_fetch $O/pmc_mo_write $O/pmc_traffic_mergeonly.json "" "" "BASELINE configs[2] with the whole-word tables off: 1 M documents, every word through the merge loop (NO_WORD_TABLE=1 tools/prof_run.py 1000000 2)" )
echo "(3)" >> $O/progress.txt
# (4) last: TA
pass ta $P 1000000 2 -- TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum && python3 $R/tools/pmc_summary.py $O/pmc_ta > $O/pmc_ta_counters.txt
echo "(4) rc $?" >> $O/progress.txt
# keep what is small: the per-kernel CSVs of the passes are large
find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O | tail -1
