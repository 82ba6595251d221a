#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt; tail -15 $O/tests.log
python bench.py --cpu-baseline skip --no-exact-f32 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" >> $O/rc.txt
python bench.py --cpu-baseline skip --no-second --no-exact-f32 --batch 2 > $O/bench_b2.json 2> $O/bench_b2.log; echo "bench_b2 rc=$?" >> $O/rc.txt
DUPL_SK_DGRAD=0 python bench.py --cpu-baseline skip --no-second --no-exact-f32 --batch 2 > $O/bench_b2_nosk.json 2> $O/bench_b2_nosk.log
DUPL_SK_DGRAD=0 python bench.py --cpu-baseline skip --no-second --no-exact-f32 > $O/bench_nosk.json 2> $O/bench_nosk.log
cat $O/rc.txt
python - <<'PY'
import json
for f in ("bench", "bench_nosk", "bench_b2", "bench_b2_nosk"):
    try:
        d = json.load(open(f"gpurun_out/r04g/{f}.json"))
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], (d.get("second_config") or {}).get("value"), {k: (v["ms_per_step"], v["frac"]) for k, v in r["single_stream"]["families"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
