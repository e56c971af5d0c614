#!/bin/bash
# usage (build container, repo root): tools/copy_final.sh <tag>   -- gpurun_out/final_<tag>/* (tools/r04_final.sh) -> profiles/r04_*
set -e
O=gpurun_out/final_$1
cp $O/bench.json.log profiles/r04_bench.json.log
cp $O/bench_kernel_stats.csv profiles/r04_bench_kernel_stats.csv
cp $O/bench_steady_state.txt profiles/r04_bench_steady_state.txt
cp $O/rocprof_bench_line.json profiles/r04_rocprof_bench_line.json
cp $O/cfg3_steady_state.txt profiles/r04_steady_state_per_kernel.txt
cp $O/launch_timeline.txt profiles/r04_launch_timeline.txt
cp $O/cfg4_kernel_stats.csv profiles/r04_cfg4_kernel_stats.csv
cp $O/cfg4_steady_state.txt profiles/r04_cfg4_steady_state.txt
cp $O/mergeonly_kernel_stats.csv profiles/r04_mergeonly_kernel_stats.csv
cp $O/mergeonly_steady_state.txt profiles/r04_mergeonly_steady_state.txt
cp $O/pmc_sq_counters.txt profiles/r04_pmc_sq_counters.txt
for f in pmc_traffic pmc_traffic_shard pmc_traffic_cfg4 pmc_traffic_mergeonly; do cp $O/$f.json profiles/r04_$f.json; done
cp $O/small_calls.txt profiles/r04_small_calls.txt
cp $O/table_load.txt profiles/r04_table_load.txt
cp $O/prepass_kernels.txt profiles/r04_prepass_kernels.txt
python3 tools/show_bench.py profiles/r04_bench.json.log | head -1
python3 tools/show_bench.py profiles/r04_rocprof_bench_line.json | head -1
