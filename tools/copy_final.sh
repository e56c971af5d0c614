#!/bin/bash
# usage (build container, repo root): tools/copy_final.sh <tag> [round, default r05]   -- gpurun_out/final_<tag>/* (tools/final_set.sh) -> profiles/<round>_*
set -e
O=gpurun_out/final_$1
ROUND=${2:-r06}
cp $O/bench.json.log profiles/${ROUND}_bench.json.log
cp $O/bench_kernel_stats.csv profiles/${ROUND}_bench_kernel_stats.csv
cp $O/bench_steady_state.txt profiles/${ROUND}_bench_steady_state.txt
cp $O/rocprof_bench_line.json profiles/${ROUND}_rocprof_bench_line.json
cp $O/cfg3_steady_state.txt profiles/${ROUND}_steady_state_per_kernel.txt
cp $O/launch_timeline.txt profiles/${ROUND}_launch_timeline.txt
cp $O/cfg4_kernel_stats.csv profiles/${ROUND}_cfg4_kernel_stats.csv
cp $O/cfg4_steady_state.txt profiles/${ROUND}_cfg4_steady_state.txt
cp $O/mergeonly_kernel_stats.csv profiles/${ROUND}_mergeonly_kernel_stats.csv
cp $O/mergeonly_steady_state.txt profiles/${ROUND}_mergeonly_steady_state.txt
cp $O/pmc_sq_counters.txt profiles/${ROUND}_pmc_sq_counters.txt
for f in pmc_traffic pmc_traffic_shard pmc_traffic_cfg4 pmc_traffic_mergeonly; do cp $O/$f.json profiles/${ROUND}_$f.json; done
cp $O/small_calls.txt profiles/${ROUND}_small_calls.txt
cp $O/table_load.txt profiles/${ROUND}_table_load.txt
cp $O/prepass_kernels.txt profiles/${ROUND}_prepass_kernels.txt
python3 tools/show_bench.py profiles/${ROUND}_bench.json.log | head -1
python3 tools/show_bench.py profiles/${ROUND}_rocprof_bench_line.json | head -1
