#!/bin/bash
# round 6 final artefacts: profile passes, secondary configurations, soak, timelines
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/profile_round.sh r06_final > gpurun_out/r06_final_profile.log 2>&1
bash tools/bench_configs.sh > gpurun_out/r06_configs.txt 2>&1
timeout 300 python tools/soak.py > gpurun_out/r06_soak.txt 2>&1
bash tools/timeline.sh r06 4 > /dev/null 2>&1
bash tools/timeline.sh r06 2 > /dev/null 2>&1
timeout 300 python tools/gemm_shapes.py --steps 3 > gpurun_out/r06_shapes_1s.txt 2>/dev/null
tail -3 gpurun_out/r06_final_profile.log; cat gpurun_out/r06_configs.txt; tail -4 gpurun_out/r06_soak.txt; head -12 gpurun_out/r06_timeline_b4.txt | cut -c1-200
