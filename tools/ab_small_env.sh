#!/bin/bash
# usage (GPU box, repo root): tools/ab_small_env.sh "VAR=a" "VAR=b" ...  -- tools/small_bench.py (configs[1] line) under
# several environments on the SAME box, three times in turn
for rep in 1 2 3; do
  for e in "$@"; do
    echo "== $e $rep"
    env $e SMALL_ONLY=1 timeout -k 10 120 python3 tools/small_bench.py || exit 1
  done
done
