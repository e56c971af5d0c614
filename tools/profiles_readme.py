#!/usr/bin/env python3
"""Writes profiles/README.md: one section per round, ONE line per file.  Lines come from DESCRIPTIONS below (exact names first, then
patterns); run from the repo root after adding files: `python3 tools/profiles_readme.py` (it fails when a file has no line)."""
import os
import re
import sys

EXACT = {
    "r06_ab_variants.txt": "the round's record of everything tried (kept or not): 0 the sporadic abort (guard allocator, backtrace, exception barrier, no caller memory in HIP copies), 1 dense host path, 2 fusion ceiling, 3 no-re-probe ceiling, 4 row-kernel stamps, 5 multi-GPU",
    "r06_guard_granule_runs.txt": "diagnostic build, guard-granule allocator (`tools/r06_guard_once.sh`): whole GPU suite in mode 1 and mode 2, pipeline / merge-kernel selections: every run green, no fault",
    "r06_guard_address_reuse.txt": "`tools/r06_fresh_bisect.py`: golden single calls under hipMalloc with filled / exact allocations (right) and under the guard allocator with and without address reuse: wrong rows only when a freed range is handed out again",
    "r06_vmm_interior_copies.txt": "`tools/micro/vmm_copy.hip`: H2D / D2D / D2H copies and memsets at interior pointers of virtual-memory mappings: all right (the guard allocator's wrong rows were address reuse, not this)",
    "r06_docw0_behind_word_kernel.txt": "launcher variant: `gz_docw0_kernel` behind the word kernel instead of beside it (where a self-classifying word kernel would need it), traced, three alternations: text side + 10 us on average",
    "r06_encode_batch_breakdown.txt": "`tools/t_encode_batch.py`, 1 M documents: `encode_batch(list of str)` 70-115 ms = packing 51-53 + `gz_encode_batch` 23-31; by threads, by hints, `dense_csr=0` 92 ms",
    "r06_host_first_touch.txt": "`tools/micro/hostfill.c` on the GPU box's host: first-touch rate of a fresh 2 GiB mapping by 1-32 threads, plain / MADV_HUGEPAGE / MADV_POPULATE_WRITE",
    "r06_fused_classify_ceiling.txt": "`tools/r06_fused_classify_try.py` on the timing build: a word kernel that classifies its own tiles (no classification kernel, no scan, counts for free): 37 us of a 1.29 ms launch; rows equal",
    "r06_merge_no_reprobe_ceiling.txt": "`tools/r06_reprobe.sh` on the timing build: a merge loop that never probes a pair twice (wrong results) vs the exact one: tables off -11 %, 20 % typos 117 -> 129-132 GB/s, headline shard ~ -25 us",
    "r06_merge_id_phase_ab.txt": "`tools/r06_ids_ab.sh`: the merge kernel's id loads all before the stores vs the build before, traced, alternating, three workloads: nothing outside the box band; not kept",
    "r06_phase_stamps.txt": "`tools/prof_diag.py` on the diagnostic build: cycles per phase of `gz_miss2_kernel` (tables on / off) and of a round of `gz_rows1_kernel` (one wave in 61 reporting; with and without the wait for the stores)",
    "r06_rows1_docs_per_wave.txt": "timing build, switch `rows_dpw`: 4 / 8 / 16 / 24 documents per wave of the row kernel, alternating: 8 (one round per wave) stays best",
    "r06_rows1_what_bounds_it.txt": "`tools/r06_rows_probe.sh` on the timing build: the row kernel with stores only / loads + walk only / ids only / more documents per wave, and the plain fill (`tools/membw`) on the same box",
    "r06_rows1_waves_per_workgroup.txt": "build parameter `GZ_ROWS_WPB` = 1 / 2 / 4 / 8 waves per workgroup of the row kernel, traced, alternating: 4 stays",
    "r06_rows1_scalar_first_indices_ab.txt": "the row kernel's first-word indices by scalar loads (s_load_dword) instead of one vector load, traced, three alternations: 2-3 % slower; not kept",
    "r06_rows_ws_ab.txt": "`tools/r06_ws.sh` (commit d395452): the row kernel with producer and consumer waves against `gz_rows1_kernel`, four batch sizes, alternating: slower everywhere",
    "r06_rows_ws_probe.txt": "`tools/r06_ws_probe.sh`, first form (7 producers + 1 consumer): each side alone -- the consumer side is the slow one",
    "r06_rows_ws_probe_v2.txt": "... second form (6 + 2, no store in a producer, indices one group ahead): each side near the old kernel's, together nearly the sum",
    "r06_side_stream_tail_variants.txt": "launcher variants of the side stream's tail (no `gz_long_kernel`, wide kernel on 1 024 / 4 096 workgroups, without its LDS table copy), traced, alternating: nothing outside the spread",
    "r06_size_exchange_gloo.txt": "the exchange step's size exchange (one gloo all-gather of an int64) ON THE GPU BOX'S HOST: 85 us at world 2, 506 us at world 8",
    "r05_size_exchange_gloo.txt": "the same test in the BUILD CONTAINER (8 cores): 465 us at world 2, 1 932 us at world 8 -- a figure of that container, not of a node",
}
PATTERNS = [
    (r"r\d\d(_v\d)?_bench\.json\.log$", "the default `python bench.py` line of the round's (version's) final build"),
    (r"r\d\d(_v\d)?_final_bench\.json\.log$", "the default `python bench.py` line of the round's last commit"),
    (r"r\d\d(_v\d)?_bench_kernel_stats\.csv$", "`rocprofv3 --kernel-trace --stats -- python3 bench.py ...`: per-kernel totals of that bench run"),
    (r"r\d\d(_v\d)?_rocprof_bench_stdout\.log$", "stdout of the traced bench run (its JSON line: the hipEvents figure beside the tracer's)"),
    (r"r\d\d_rocprof_bench_line\.json$", "the JSON line bench.py printed UNDER the tracer (kernel_ms_avg by hipEvents to set beside the kernel stats)"),
    (r"r\d\d_bench_steady_state\.txt$", "the traced bench run, last launches of every kernel (`tools/trace_summary.py`): per-shard steady state"),
    (r"r\d\d(_v\d)?_steady_state_per_kernel\.txt$", "BASELINE configs[2] (1 M documents) traced: last launches of every kernel, their sum against hipEvents"),
    (r"r\d\d_launch_timeline\.txt$", "one launch of the pipeline as a timeline: start, duration and gap of every kernel, both streams"),
    (r"r\d\d(_v\d)?_pmc_traffic\.(json|txt)$", "HBM traffic of one launch on BASELINE configs[2]: `--pmc FETCH_SIZE` / `WRITE_SIZE` in separate passes, per kernel"),
    (r"r\d\d_pmc_traffic_shard\.json$", "... of one 1.25 M-document shard of the headline job (what `roofline.traffic` quotes)"),
    (r"r\d\d_pmc_traffic_cfg4(_base)?\.json$", "... of BASELINE configs[4] (custom tables, long documents)"),
    (r"r\d\d_pmc_traffic_mergeonly(_base)?\.json$", "... of configs[2] with the whole-word tables off"),
    (r"r\d\d_pmc_sq_counters.*\.txt$", "SQ / TCP / TCC counters of the pipeline's kernels (separate `--pmc` passes, `tools/pmc_summary.py`)"),
    (r"r\d\d_pmc_l1_l2_latency\.txt$", "TCP / TCC request and latency counters of the table-bound kernels"),
    (r"r\d\d_cfg4_kernel_stats(_base)?\.csv$", "BASELINE configs[4] traced: per-kernel totals"),
    (r"r\d\d_cfg4_steady_state(_base)?\.txt$", "BASELINE configs[4] traced: steady state per kernel"),
    (r"r\d\d_cfg3_steady_state_base\.txt$", "configs[2] traced on the round's starting sources (the base of its A/Bs)"),
    (r"r\d\d_mergeonly_kernel_stats(_base)?\.csv$", "configs[2] with the whole-word tables off, traced: per-kernel totals"),
    (r"r\d\d_mergeonly_steady_state(_base)?\.txt$", "... steady state per kernel"),
    (r"r\d\d_ab_variants\.txt$", "the round's record of every variant and investigation: what, where the raw output is, numbers, kept or not"),
    (r"r\d\d_ablation_.*\.txt$", "timing-switch ablations of that kernel (diagnostic build)"),
    (r"r\d\d_table_load\.txt$", "`Tokenize()` / `gz_load_tables` phase times with and without the table cache (`tools/t_load.py`)"),
    (r"r\d\d_small_calls.*\.txt$", "latency of single `__call__` / `encode` calls (`tools/small_bench.py`), the variant named in the file name"),
    (r"r\d\d_prepass_kernels\.txt$", "the text pre-pass traced: per-kernel times of the five filters / of `remove_html` alone"),
    (r"r\d\d_python_to_device_(before|after)\.txt$", "`tools/t_todevice.py`: where `encode_to_device(list of 1 M str)` spends its time, before / after the pinned arena"),
    (r"r\d\d_unpadded_rows\.txt$", "rows without padding: raw area + finalize vs count, scan, write once -- same-box alternation"),
    (r"r\d\d_rows_apart_experiment\.txt$", "the row kernel beside the next launch's text side: every schedule tried, with kernel timelines"),
    (r"r\d\d_lds_access_costs\.txt$", "`tools/micro/ldsrate.hip`: LDS pipe cycles per wave instruction, aligned vs misaligned, by width"),
    (r"r\d\d_valu_issue_rates\.txt$", "`tools/micro/mulrate.hip`: cycles per vector instruction by kind"),
    (r"r\d\d_bench_forced_exchange\.json$", "`bench.py --force-exchange` on the whole job with one rank: what the exchange step costs the kernels"),
    (r"r\d\d_diag_index_assertions\.txt$", "the diagnostic build's index assertions + poisoned workspace on the abort's selection: green"),
    (r"r\d\d_two_ranks_one_gpu\.txt$", "two ranks on one GPU over the gloo stand-in transport: the exchange rehearsed"),
    (r"r\d\d_pmc_sq_counters_first_hot_build\.txt$", "counters of the first perfect-hash word / merge kernels"),
]


def describe(name):
    if name in EXACT:
        return EXACT[name]
    for pat, text in PATTERNS:
        if re.match(pat, name):
            return text
    return None


def main():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    files = sorted(f for f in os.listdir(root) if f != "README.md")
    rounds = {}
    missing = []
    for f in files:
        m = re.match(r"r(\d\d)_", f)
        rounds.setdefault(int(m.group(1)) if m else 0, []).append(f)
        if describe(f) is None:
            missing.append(f)
    if missing:
        sys.exit("profiles_readme.py: no line for: " + ", ".join(missing))
    out = ["# profiles/", "",
           "The rocprofv3 summaries, counter passes, bench lines and experiment records the numbers in DESIGN.md / bench.py come from: one section",
           "per round (newest first), ONE line per file.  `rNN_ab_variants.txt` is a round's record of everything that was tried, kept or not, and",
           "names the tool behind every file.  Boxes differ by +- 4 % on the write-bound row kernel: A/Bs are alternated on ONE box; `_base` files",
           "are the same measurement on the round's starting sources.  `vN` (round 1) = the N-th kernel generation of that round.", ""]
    for r in sorted(rounds, reverse=True):
        out += ["## round %d" % r, ""]
        out += ["- `%s` -- %s" % (f, describe(f)) for f in rounds[r]]
        out.append("")
    open(os.path.join(root, "README.md"), "w").write("\n".join(out))
    print("profiles/README.md: %d files, %d rounds" % (len(files), len(rounds)))


if __name__ == "__main__":
    main()
