#!/bin/bash
O=gpurun_out/r04k; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "grouped or kmajor" -s > $O/t_group.log 2>&1; echo "group rc=$?" > $O/rc.txt; grep -E "grouped wgrad|passed|failed|Error" $O/t_group.log | tail -16
python -m pytest tests/test_engine_gpu.py -x -q -k "tiny or deterministic or run_to_run" > $O/t_tiny.log 2>&1; echo "tiny rc=$?" >> $O/rc.txt; tail -3 $O/t_tiny.log
python bench.py --cpu-baseline skip --no-second --no-exact-f32 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" >> $O/rc.txt
DUPL_WGRAD_GROUP=0 python bench.py --cpu-baseline skip --no-second --no-exact-f32 > $O/bench_nogroup.json 2> $O/bench_nogroup.log
python bench.py --cpu-baseline skip --no-second --no-exact-f32 --batch 2 > $O/bench_b2.json 2> $O/bench_b2.log
DUPL_DETERMINISTIC=1 python bench.py --cpu-baseline skip --no-second --no-exact-f32 --no-roofline > $O/bench_det.json 2> $O/bench_det.log
cat $O/rc.txt
python - <<'PY'
import json
for f in ("bench", "bench_nogroup", "bench_b2", "bench_det"):
    try:
        d = json.load(open(f"gpurun_out/r04k/{f}.json")); r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r and r["frac"], r and {k: (v["ms_per_step"], v["frac"]) for k, v in r["single_stream"]["families"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
