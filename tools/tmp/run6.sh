#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q > $O/t_kernels.log 2>&1; echo "kernels rc=$?" > $O/rc.txt; tail -3 $O/t_kernels.log
S="15696x3072x768,15696x768x3072,15696x2304x768,15696x768x768,6280x3072x768,6280x768x3072"
{
for e in 0 1 2 3; do
  echo "# format 1, epilogue $e, two streams"
  tools/gemm16_bench -f -t 8,12 -s $S -e $e -2 -w 150
done
echo "# k-major dgrad with the gelu' epilogue is not in this tool; the step bench below covers it"
} > $O/epi_bench.txt 2>&1
cat $O/epi_bench.txt
python -m pytest tests/test_engine_gpu.py -x -q -k "tiny or vitb_forward or vitb_f16x3" > $O/t_tiny.log 2>&1; echo "tiny rc=$?" >> $O/rc.txt; tail -3 $O/t_tiny.log
python bench.py --cpu-baseline skip --no-second --no-exact-f32 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt
python - <<'PY'
import json
for f in ("bench",):
    try:
        d = json.load(open(f"gpurun_out/r04f/{f}.json"))
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], {k: (v["ms_per_step"], v["frac"]) for k, v in r["single_stream"]["families"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
