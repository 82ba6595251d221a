#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O
python bench.py --cpu-baseline skip --no-second --no-exact-f32 --no-roofline --steps 10 > $O/bench.json 2> $O/bench.log
DUPL_WGRAD_STREAM=1 python bench.py --cpu-baseline skip --no-second --no-exact-f32 --no-roofline --steps 10 > $O/bench_side.json 2> $O/bench_side.log
DUPL_WGRAD_STREAM=1 GPU_MAX_HW_QUEUES=8 python bench.py --cpu-baseline skip --no-second --no-exact-f32 --no-roofline --steps 10 > $O/bench_side8.json 2> $O/bench_side8.log
python bench.py --cpu-baseline skip --no-second --no-exact-f32 --no-roofline --steps 10 --batch 2 > $O/bench_b2.json 2> $O/bench_b2.log
DUPL_WGRAD_STREAM=1 python bench.py --cpu-baseline skip --no-second --no-exact-f32 --no-roofline --steps 10 --batch 2 > $O/bench_b2_side.json 2> $O/bench_b2_side.log
python - <<'PY'
import json
for f in ("bench", "bench_side", "bench_side8", "bench_b2", "bench_b2_side"):
    try:
        d = json.load(open(f"gpurun_out/r04l/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d["config"]["loss"])
    except Exception as e:
        print(f, "unreadable", e)
PY
python -m pytest tests/test_engine_gpu.py -x -q -k "full_size and f16x3 and (voc_B_bs4 or coco_B2_bs2)" -s > $O/t_full.log 2>&1; echo "full rc=$?" > $O/rc.txt; grep -E "gradients|ReLU|passed|failed" $O/t_full.log | tail -8
