#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06c2; mkdir -p $O
timeout 600 python tools/diag_r06.py c > $O/diag_c.txt 2>&1; echo "diag rc $?" >> $O/diag_c.txt
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider "tests/test_engine_gpu.py::test_merged_pass_survives_a_range_verdict_that_flips_at_this_step" "tests/test_kernels_gpu.py::test_kmajor_backward_gemms_are_fp32_equivalent" "tests/test_scripts_gpu.py::test_ddp_two_ranks_optimizer_rides_in_the_exchange_gloo" "tests/test_scripts_gpu.py::test_bench_multi_rank_control_flow" > $O/retest.txt 2>&1; echo "retest rc $?" >> $O/retest.txt
grep "^C" $O/diag_c.txt | tail -60; tail -15 $O/retest.txt
