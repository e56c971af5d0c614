#!/bin/bash
# usage (GPU box, repo root): tools/final_set.sh <tag> [round, default r05]   -- everything profiles/<round>_* holds for one build:
#   (1) kernel stats of a bench.py run (rocprofv3 --kernel-trace --stats); (2) steady-state per-kernel times + the timeline of one
#   launch on BASELINE configs[2]; (3) HBM traffic (separate --pmc passes, no trace domains) of configs[2], of one shard of the
#   headline job, of BASELINE configs[4] and of configs[2] with the whole-word tables off; (4) SQ / TCP / TCC counters of the
#   pipeline's kernels; (5) kernel stats of configs[4] and of the no-table mode; (6) table load and small-call latencies;
#   (7) the default bench line.
set -o pipefail
tag=$1
ROUND=${2:-r06}
R=$PWD
O=$R/gpurun_out/final_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PIPE="$(cd $R && python3 -c 'import bench; print(bench.PIPELINE)')"
pass() {   # pass <name> <script + args> -- <counters>
  local name=$1; shift
  local cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
  rm -rf $O/pmc_$name
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 "${cmd[@]}" > $O/pmc_$name.log 2>&1
}
trace() {  # trace <name> <script + args>
  local name=$1; shift
  rm -rf /tmp/tr_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/tr_$name -o t --output-format csv -- python3 "$@" > $O/trace_$name.log 2>&1 || return 1
  cp $(ls /tmp/tr_$name/*kernel_stats.csv /tmp/tr_$name/*/*kernel_stats.csv 2>/dev/null | head -1) $O/${name}_kernel_stats.csv
  python3 $R/tools/trace_summary.py /tmp/tr_$name > $O/${name}_steady_state.txt
  tail -1 $O/${name}_steady_state.txt
}
P=$R/tools/prof_run.py
# (1)
trace bench $R/bench.py --steps 5 --warmup 1 --no-secondary --no-verify || exit 1
grep '^{' $O/trace_bench.log | tail -1 > $O/rocprof_bench_line.json
echo "(1)" > $O/progress.txt
# (2)
trace cfg3 $P 1000000 5 || exit 1
python3 $R/tools/trace_timeline.py /tmp/tr_cfg3 > $O/launch_timeline.txt 2>&1
echo "(2)" >> $O/progress.txt
# (3)
pass fetch $P 1000000 2 -- FETCH_SIZE || exit 1
pass write $P 1000000 2 -- WRITE_SIZE || exit 1
( cd $R && python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "$PIPE" )
export SEED=100
pass sfetch $P 1250000 2 -- FETCH_SIZE || exit 1
pass swrite $P 1250000 2 -- WRITE_SIZE || exit 1
unset SEED
( cd $R && python3 tools/pmc_traffic.py $O/pmc_sfetch $O/pmc_swrite $O/pmc_traffic_shard.json "$PIPE" "" "BASELINE configs[3], shard 0: 1.25 M documents (seed 100), 358 MB, max_len 256 (SEED=100 tools/prof_run.py 1250000 2)" )
pass c5fetch $R/tools/prof_cfg5.py 2 -- FETCH_SIZE || exit 1
pass c5write $R/tools/prof_cfg5.py 2 -- WRITE_SIZE || exit 1
( cd $R && python3 tools/pmc_traffic.py $O/pmc_c5fetch $O/pmc_c5write $O/pmc_traffic_cfg4.json "$PIPE" "" "BASELINE configs[4]: custom tables, 50 000 documents of <= 4 000 characters, max_len 1024 (tools/prof_cfg5.py 2)" )
export NO_WORD_TABLE=1
pass mofetch $P 1000000 2 -- FETCH_SIZE || exit 1
pass mowrite $P 1000000 2 -- WRITE_SIZE || exit 1
( cd $R && python3 tools/pmc_traffic.py $O/pmc_mofetch $O/pmc_mowrite $O/pmc_traffic_mergeonly.json "$PIPE" "" "BASELINE configs[2] with the whole-word tables off: 1 M documents, every word through the merge loop (NO_WORD_TABLE=1 tools/prof_run.py 1000000 2)" )
trace mergeonly $P 1000000 3 || exit 1
unset NO_WORD_TABLE
echo "(3)" >> $O/progress.txt
# (4)
pass sq1 $P 1000000 2 -- SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES || exit 1
pass sq2 $P 1000000 2 -- SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
pass sq3 $P 1000000 2 -- SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH || exit 1
pass tcp $P 1000000 2 -- TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum || exit 1
pass tcc $P 1000000 2 -- TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum || exit 1
python3 $R/tools/pmc_summary.py $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 $O/pmc_tcp $O/pmc_tcc > $O/pmc_sq_counters.txt
echo "(4)" >> $O/progress.txt
# (5)
trace cfg4 $R/tools/prof_cfg5.py 5 || exit 1
echo "(5)" >> $O/progress.txt
# (6)
( cd $R && python3 tools/t_load.py > $O/table_load.txt 2>&1; python3 tools/small_bench.py > $O/small_calls.txt 2>&1; tools/pp_trace.sh > $O/prepass_kernels.txt 2>&1 )
echo "(6)" >> $O/progress.txt
# (7) (the traffic files go where bench.py looks for them: this run's line then carries them)
cp $O/pmc_traffic.json $R/profiles/${ROUND}_pmc_traffic.json; cp $O/pmc_traffic_shard.json $R/profiles/${ROUND}_pmc_traffic_shard.json; cp $O/pmc_traffic_cfg4.json $R/profiles/${ROUND}_pmc_traffic_cfg4.json
( cd $R && timeout -k 10 900 python3 bench.py > $O/bench.json.log 2> $O/bench.err ) || { tail -5 $O/bench.err; exit 1; }
tail -c 400 $O/bench.json.log
find $O -name "*counter_collection.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O | tail -1
