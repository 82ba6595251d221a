#!/bin/bash
O=gpurun_out/r04h; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o k -- python $R/bench.py --single-stream --steps 3 --warmup 1 --cpu-baseline skip --no-exact-f32 --no-second --no-roofline > $R/$O/k.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_k -name '*.db' | head -1) > $R/$O/kernel_stats.txt 2>&1
cd $R
head -60 $O/kernel_stats.txt | cut -c1-170
python -m pytest tests/test_engine_gpu.py -x -q -k "full_size and pretrained_like" -s > $O/t_full.log 2>&1; echo "full rc=$?" > $O/rc.txt; grep -E "gradients|ReLU|passed|failed" $O/t_full.log | tail
python -m pytest tests/test_scripts_gpu.py -x -q -k "ddp_exchange or c_abi" -s > $O/t_scripts.log 2>&1; echo "scripts rc=$?" >> $O/rc.txt; grep -E "abi_smoke|CONTROL|STALE|passed|failed|skipped" $O/t_scripts.log | tail
cat $O/rc.txt
