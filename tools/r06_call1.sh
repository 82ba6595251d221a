#!/bin/bash
# round 6, call 1: full suite without -x, diagnostics, shapes table, tile 8 vs 16
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06c1; mkdir -p $O
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider > $O/suite.txt 2>&1; echo "suite rc $?" >> $O/suite.txt
timeout 900 python tools/diag_r06.py ab > $O/diag.txt 2>&1; echo "diag rc $?" >> $O/diag.txt
timeout 300 python tools/gemm_shapes.py --steps 3 > $O/shapes_1s.txt 2> $O/shapes_1s.err
timeout 300 python tools/gemm_shapes.py --steps 3 --dual > $O/shapes_2s.txt 2> $O/shapes_2s.err
for e in 1 2 3; do
  timeout 300 tools/gemm16_bench -f -t 8,16 -e $e -w 150 -s fwd > $O/tile16_e$e.txt 2>&1
done
timeout 300 tools/gemm16_bench -f -t 8,16 -e 3 -w 150 -2 -s fwd > $O/tile16_e3_two.txt 2>&1
tail -5 $O/suite.txt; tail -30 $O/diag.txt; cat $O/tile16_e3.txt
