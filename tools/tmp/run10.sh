#!/bin/bash
O=gpurun_out/r04j; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q > $O/t_kernels.log 2>&1; echo "kernels rc=$?" > $O/rc.txt; tail -3 $O/t_kernels.log
python -m pytest tests/test_engine_gpu.py -x -q -k "tiny or vitb or deterministic or run_to_run" > $O/t_tiny.log 2>&1; echo "tiny rc=$?" >> $O/rc.txt; tail -3 $O/t_tiny.log
tools/attn16_bench -w 200 > $O/attn16.txt 2>&1; cat $O/attn16.txt
python tools/op_bench.py ln_bwd > $O/ln_bwd.txt 2>&1; tail -6 $O/ln_bwd.txt
python bench.py --cpu-baseline skip --no-second --no-exact-f32 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" >> $O/rc.txt
B=tools/gemm16_bench
DG=3140x768x3072,3140x3072x768,3140x768x2304,3140x768x768,1570x768x3072,1570x3072x768,1570x768x2304
WG=3072x768x3168,768x3072x3168,2304x768x3168,768x768x3168,3072x768x1600,768x3072x1600,2304x768x1600
{ echo "# round 4, k-major single-accumulator backward GEMMs (-f -L nk: dgrad, B = the forward's W planes read k-major; -L kk -e 4: wgrad, dy and x planes k-major, stream-K) vs the transposed-planes path they replace (format 0, heuristic tile), one stream";
  echo "# dgrad, transposed-planes path"; $B -s $DG -t 0 -w 150;
  echo "# dgrad, k-major"; $B -s $DG -t 0 -w 150 -f -L nk;
  echo "# dgrad with a linear epilogue, k-major stream-K into a zero-filled dx (-e 4)"; $B -s $DG -t 0 -w 150 -f -L nk -e 4;
  echo "# wgrad, transposed-planes path"; $B -s $WG -t 0 -w 150 -e 4;
  echo "# wgrad, k-major"; $B -s $WG -t 0 -w 150 -e 4 -f -L kk;
  echo "# the same five, two streams";
  $B -s $DG -t 0 -w 150 -2; $B -s $DG -t 0 -w 150 -f -L nk -2; $B -s $DG -t 0 -w 150 -f -L nk -e 4 -2; $B -s $WG -t 0 -w 150 -e 4 -2; $B -s $WG -t 0 -w 150 -e 4 -f -L kk -2; } > $O/km_tiles.txt 2>&1
cat $O/rc.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04j/bench.json")); r = d["roofline"]
print("bench", d["value"], d["ms_per_step"], "frac", r["frac"], {k: (v["ms_per_step"], v["frac"]) for k, v in r["single_stream"]["families"].items()})
PY
