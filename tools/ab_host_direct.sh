for rep in 1 2; do for hd in 0 4096; do echo "== host_direct=$hd"; GZ_TEST_SWITCHES=host_direct=$hd python tools/small_bench.py 2>&1 | tail -6; done; done
