#!/bin/bash
# round-4 scratch run: k-major kernels -- unit tests, stand-alone rates, tiny + full engine parity, bench
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "kmajor" > $O/t_km.log 2>&1; echo "km rc=$?" > $O/rc.txt; tail -3 $O/t_km.log
DG="3140x768x3072,3140x3072x768,3140x768x2304,3140x768x768"
WG="3072x768x3168,768x3072x3168,2304x768x3168,768x768x3168"
{
echo "# dgrad: old (format 0, B = W^T planes, heuristic tile) vs k-major (format 1, B = W planes k-major)"
tools/gemm16_bench -t 0 -s $DG -w 150
tools/gemm16_bench -t 0 -s $DG -w 150 -f -L nk
echo "# the same, two streams"
tools/gemm16_bench -t 0 -s $DG -w 150 -2
tools/gemm16_bench -t 0 -s $DG -w 150 -f -L nk -2
echo "# wgrad (accumulate): old (transposed planes) vs k-major"
tools/gemm16_bench -t 0 -s $WG -w 150 -e 4
tools/gemm16_bench -t 0 -s $WG -w 150 -e 4 -f -L kk
echo "# the same, two streams"
tools/gemm16_bench -t 0 -s $WG -w 150 -e 4 -2
tools/gemm16_bench -t 0 -s $WG -w 150 -e 4 -f -L kk -2
} > $O/km_bench.txt 2>&1
cat $O/km_bench.txt
python -m pytest tests/test_kernels_gpu.py -x -q > $O/t_kernels.log 2>&1; echo "kernels rc=$?" >> $O/rc.txt; tail -3 $O/t_kernels.log
python -m pytest tests/test_engine_gpu.py -x -q -k "tiny or ragged or cam_with_grad" > $O/t_tiny.log 2>&1; echo "tiny rc=$?" >> $O/rc.txt; tail -3 $O/t_tiny.log
python -m pytest tests/test_engine_gpu.py -x -q -k "full_size and (voc_B_bs4 or voc_B_bs2) and f16x3" -s > $O/t_full.log 2>&1; echo "full rc=$?" >> $O/rc.txt; grep -E "gradients:|ReLU|passed|failed|grad " $O/t_full.log | tail -20
python bench.py --cpu-baseline skip --no-second > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" >> $O/rc.txt
DUPL_KM_BWD=0 python bench.py --cpu-baseline skip --no-second --no-exact-f32 > $O/bench_old.json 2> $O/bench_old.log; echo "bench_old rc=$?" >> $O/rc.txt
cat $O/rc.txt
python - <<'PY'
import json
for f in ("bench", "bench_old"):
    try:
        d = json.load(open(f"gpurun_out/r04b/{f}.json"))
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], {k: (v["ms_per_step"], v["frac"]) for k, v in r["single_stream"]["families"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
