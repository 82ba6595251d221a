#!/bin/bash
# Chip-occupancy timeline of the timed (two-stream) configuration at a given per-GPU batch (run through gpurun):
#   gpurun --timeout 900 -- 'bash tools/timeline.sh r06 2'      -> gpurun_out/<tag>_timeline_b<batch>.txt
# rocprofv3 --kernel-trace of `bench.py --batch B`, then tools/rocpd_stats.py --timeline over the last half of the trace.
TAG=${1:-r06}; BATCH=${2:-4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_timeline_b${BATCH}.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$TAG && mkdir -p /tmp/tl_$TAG
timeout 600 rocprofv3 --kernel-trace -d /tmp/tl_$TAG -o t -- python $R/bench.py --batch $BATCH --steps 6 --warmup 2 --cpu-baseline skip \
    --no-roofline --no-exact-f32 --no-second > /tmp/tl_$TAG/bench.log 2>&1
grep '^{' /tmp/tl_$TAG/bench.log | tail -1 | cut -c1-200 > $OUT
DB=$(find /tmp/tl_$TAG -name "*.db" | head -1); python $R/tools/rocpd_stats.py --timeline $DB >> $OUT 2>&1; python $R/tools/rocpd_stats.py --light $DB >> $OUT 2>&1; python $R/tools/rocpd_stats.py --gaps $DB 0.25 100 >> $OUT 2>&1
head -40 $OUT
