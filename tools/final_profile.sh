#!/bin/bash
# usage (GPU box, repo root): tools/final_profile.sh <tag>   -- everything profiles/ holds for one build
set -o pipefail
tag=$1
R=$PWD
O=$R/gpurun_out/final_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/fp_$tag -o b --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-merge-only > $O/rocprof_bench_stdout.log 2>$O/rocprof_bench_stderr.log || { tail -20 $O/rocprof_bench_stderr.log; exit 1; }
cp $(ls /tmp/fp_$tag/*kernel_stats.csv /tmp/fp_$tag/*/*kernel_stats.csv 2>/dev/null | head -1) $O/bench_kernel_stats.csv
python3 $R/tools/trace_summary.py /tmp/fp_$tag > $O/steady_state_per_kernel.txt
tail -3 $O/steady_state_per_kernel.txt
cd $R
tools/pmc.sh fin_$tag 1000000 "FETCH_SIZE" "WRITE_SIZE" || exit 1
python3 tools/pmc_traffic.py gpurun_out/pmc_fin_${tag}_1 gpurun_out/pmc_fin_${tag}_2 $O/pmc_traffic.json
timeout -k 10 400 python3 bench.py > $O/bench.json.log 2>$O/bench.err || { tail -5 $O/bench.err; exit 1; }
tail -c 400 $O/bench.json.log
