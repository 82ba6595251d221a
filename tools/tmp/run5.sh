#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q > $O/t_kernels.log 2>&1; echo "kernels rc=$?" > $O/rc.txt; tail -3 $O/t_kernels.log
python -m pytest tests/test_engine_gpu.py -x -q -k "tiny or ragged or cam_with_grad" > $O/t_tiny.log 2>&1; echo "tiny rc=$?" >> $O/rc.txt; tail -3 $O/t_tiny.log
python -m pytest tests/test_engine_gpu.py -x -q -k "full_size and (voc_B_bs4 or voc_C) and f16x3" -s > $O/t_full.log 2>&1; echo "full rc=$?" >> $O/rc.txt; grep -E "gradients:|ReLU|passed|failed" $O/t_full.log | tail -20
python bench.py --cpu-baseline skip --no-second --no-exact-f32 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" >> $O/rc.txt
DUPL_KM_BWD=0 python bench.py --cpu-baseline skip --no-second --no-exact-f32 > $O/bench_old.json 2> $O/bench_old.log; echo "bench_old rc=$?" >> $O/rc.txt
python bench.py --cpu-baseline skip --no-second --no-exact-f32 --batch 2 > $O/bench_b2.json 2> $O/bench_b2.log; echo "bench_b2 rc=$?" >> $O/rc.txt
cat $O/rc.txt
python - <<'PY'
import json
for f in ("bench", "bench_old", "bench_b2"):
    try:
        d = json.load(open(f"gpurun_out/r04e/{f}.json"))
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], {k: (v["ms_per_step"], v["frac"]) for k, v in r["single_stream"]["families"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
