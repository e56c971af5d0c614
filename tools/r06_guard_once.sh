#!/bin/bash
# usage (GPU box, repo root): tools/r06_guard_once.sh <mode 1|2> <tag> [pytest -k selection]
# ONE run of the GPU suite (or a selection) on the DIAGNOSTIC build with the guard-granule allocator (switch diag_guard): every
# device buffer -- workspace, tables, the caller's own gz_device_alloc buffers -- is its own mapping between unmapped granules
# with no slack behind it, mode 1: the buffer ENDS at its mapping's last byte; mode 2: it STARTS at the first.  A kernel that
# touches one byte outside a buffer faults at once (the runtime's message names the address; AMD_LOG_LEVEL=1 the kernel), instead
# of on the day the allocator's layout says so.  Whole stdout / stderr kept under gpurun_out/.  Tests that load the diagnostic or
# a variant library themselves (children with their own GZ_LIBRARY / switches) are left out: one process, one build.
mode=${1:-1}; tag=${2:-g$mode}
SEL=${3:-"not index_assertions and not scan_time_out and not no_gpu_means and not bench_ and not multirank and not rccl and not table_cache"}
mkdir -p gpurun_out
make -C genz-tokenize_amd/csrc diag > gpurun_out/guard_$tag.make 2>&1 || { tail -20 gpurun_out/guard_$tag.make; exit 1; }
GZ_LIBRARY=$PWD/build_ab/libgz_diag.so GZ_TEST_SWITCHES="diag_guard=$mode${EXTRA_SWITCHES:+,$EXTRA_SWITCHES}" GZ_TABLE_CACHE=off AMD_LOG_LEVEL=1 \
  timeout -k 10 1000 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "$SEL" \
  > gpurun_out/guard_$tag.out 2> gpurun_out/guard_$tag.err
rc=$?
echo "guard mode $mode ($tag): rc $rc"; tail -5 gpurun_out/guard_$tag.out; grep -n "gpu test\]" gpurun_out/guard_$tag.err | tail -2
grep -n -i "fault\|sigtrace\|abort\|error" gpurun_out/guard_$tag.err | head -40
exit $rc
