#!/bin/bash
# usage (GPU box, repo root): tools/pmc_traffic_only.sh [round, default r05]   -- step (3) of tools/final_set.sh alone: the four HBM traffic
# files profiles/<round>_pmc_traffic{,_shard,_cfg4,_mergeonly}.json re-taken on the current kernel sources (bench.py quotes them only for
# the sources they were taken on), written under gpurun_out/traffic/ for tools/copy_final.sh-style copying
set -o pipefail
ROUND=${1:-r05}
R=$PWD
O=$R/gpurun_out/traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PIPE="$(cd $R && python3 -c 'import bench; print(bench.PIPELINE)')"
pass() { local name=$1; shift; local cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
  rm -rf $O/pmc_$name; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 "${cmd[@]}" > $O/pmc_$name.log 2>&1; }
P=$R/tools/prof_run.py
pass fetch $P 1000000 2 -- FETCH_SIZE || exit 1
pass write $P 1000000 2 -- WRITE_SIZE || exit 1
( cd $R && python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/${ROUND}_pmc_traffic.json "$PIPE" )
export SEED=100
pass sfetch $P 1250000 2 -- FETCH_SIZE || exit 1
pass swrite $P 1250000 2 -- WRITE_SIZE || exit 1
unset SEED
( cd $R && python3 tools/pmc_traffic.py $O/pmc_sfetch $O/pmc_swrite $O/${ROUND}_pmc_traffic_shard.json "$PIPE" "" "BASELINE configs[3], shard 0: 1.25 M documents (seed 100), 358 MB, max_len 256 (SEED=100 tools/prof_run.py 1250000 2)" )
pass c5fetch $R/tools/prof_cfg5.py 2 -- FETCH_SIZE || exit 1
pass c5write $R/tools/prof_cfg5.py 2 -- WRITE_SIZE || exit 1
( cd $R && python3 tools/pmc_traffic.py $O/pmc_c5fetch $O/pmc_c5write $O/${ROUND}_pmc_traffic_cfg4.json "$PIPE" "" "BASELINE configs[4]: custom tables, 50 000 documents of <= 4 000 characters, max_len 1024 (tools/prof_cfg5.py 2)" )
export NO_WORD_TABLE=1
pass mofetch $P 1000000 2 -- FETCH_SIZE || exit 1
pass mowrite $P 1000000 2 -- WRITE_SIZE || exit 1
unset NO_WORD_TABLE
( cd $R && python3 tools/pmc_traffic.py $O/pmc_mofetch $O/pmc_mowrite $O/${ROUND}_pmc_traffic_mergeonly.json "$PIPE" "" "BASELINE configs[2] with the whole-word tables off: 1 M documents, every word through the merge loop (NO_WORD_TABLE=1 tools/prof_run.py 1000000 2)" )
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
ls $O/*.json
