#!/bin/bash
# kernel sequence of a one-stream step (what runs between forward and backward), the new GELU test
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r06c6; mkdir -p $O
timeout 300 python -m pytest -q -m gpu -p no:cacheprovider tests/test_kernels_gpu.py -k "gelu_epilogues" > $O/gelu_test.txt 2>&1; tail -3 $O/gelu_test.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/seq && mkdir -p /tmp/seq
timeout 600 rocprofv3 --kernel-trace -d /tmp/seq -o t -- python $R/bench.py --single-stream --steps 3 --warmup 2 --cpu-baseline skip --no-roofline --no-exact-f32 --no-second > /tmp/seq/bench.log 2>&1
DB=$(find /tmp/seq -name "*.db" | head -1)
python $R/tools/rocpd_stats.py --sequence $DB 0.22 > $O/sequence_1s.txt 2>&1
wc -l $O/sequence_1s.txt; grep '^{' /tmp/seq/bench.log | tail -1 | cut -c1-160
