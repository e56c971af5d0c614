#!/bin/bash
# usage (GPU box, repo root): tools/final_profile.sh <tag>   -- everything profiles/ holds for one build:
#   kernel stats of a bench.py run (rocprofv3 --kernel-trace --stats), steady-state per-kernel times of the roofline
#   workload, PMC traffic (separate --pmc passes, no trace domains), the default bench.py line
set -o pipefail
tag=$1
R=$PWD
O=$R/gpurun_out/final_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp_$tag /tmp/fs_$tag
# (1) the bench command under the profiler: per-kernel averages over the timed launches
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/fp_$tag -o b --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-secondary --no-verify > $O/rocprof_bench_stdout.log 2>$O/rocprof_bench_stderr.log || { tail -20 $O/rocprof_bench_stderr.log; exit 1; }
cp $(ls /tmp/fp_$tag/*kernel_stats.csv /tmp/fp_$tag/*/*kernel_stats.csv 2>/dev/null | head -1) $O/bench_kernel_stats.csv
echo "(1) done" > $O/progress.txt
# (2) steady-state per-kernel durations of one launch of the pipeline on BASELINE configs[2]
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/fs_$tag -o t --output-format csv -- python3 $R/tools/prof_run.py 1000000 5 > $O/prof_run.log 2>&1 || { tail -20 $O/prof_run.log; exit 1; }
python3 $R/tools/trace_summary.py /tmp/fs_$tag > $O/steady_state_per_kernel.txt
tail -3 $O/steady_state_per_kernel.txt
echo "(2) done" >> $O/progress.txt
cd $R
# (3) HBM traffic, one counter per pass
tools/pmc.sh fin_$tag 1000000 "FETCH_SIZE" "WRITE_SIZE" || exit 1
python3 tools/pmc_traffic.py gpurun_out/pmc_fin_${tag}_1 gpurun_out/pmc_fin_${tag}_2 $O/pmc_traffic.json "$(python3 -c 'import bench; print(bench.PIPELINE)')"
# (3b) the same for one shard of the headline job (BASELINE configs[3]: 1.25 M documents, seed 100 = shard 0)
SEED=100 tools/pmc.sh finS_$tag 1250000 "FETCH_SIZE" "WRITE_SIZE" || exit 1
python3 tools/pmc_traffic.py gpurun_out/pmc_finS_${tag}_1 gpurun_out/pmc_finS_${tag}_2 $O/pmc_traffic_shard.json "$(python3 -c 'import bench; print(bench.PIPELINE)')" "" "BASELINE configs[3], shard 0: 1.25 M documents (seed 100), 358 MB, max_len 256 (SEED=100 tools/prof_run.py 1250000 2)"
echo "(3) done" >> $O/progress.txt
# (4) the default bench line
timeout -k 10 700 python3 bench.py > $O/bench.json.log 2>$O/bench.err || { tail -5 $O/bench.err; exit 1; }
tail -c 300 $O/bench.json.log
