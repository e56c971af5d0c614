#!/bin/bash
# two ranks on ONE device: does RCCL take it?
R=$PWD; D=$(mktemp -d)
export GZ_CHILD_DEVICE=0 NCCL_DEBUG=WARN HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 5 120 python3 tests/gather_child.py 0 2 $D > $D/r0.log 2>&1 &
P0=$!
timeout -k 5 120 python3 tests/gather_child.py 1 2 $D > $D/r1.log 2>&1 &
P1=$!
wait $P0; echo "rank0 rc=$?"; wait $P1; echo "rank1 rc=$?"
tail -5 $D/r0.log; tail -5 $D/r1.log; ls $D
