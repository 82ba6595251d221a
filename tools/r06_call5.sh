cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06c5; mkdir -p $O
timeout 300 python tools/gemm_shapes.py --steps 3 --dual > $O/shapes_2s.txt 2> $O/shapes_2s.err
timeout 300 python tools/gemm_shapes.py --steps 3 > $O/shapes_1s.txt 2> $O/shapes_1s.err
B="python bench.py --no-second --no-exact-f32 --cpu-baseline skip --no-roofline --steps 30 --warmup 5"
for off in 0 200000 1000000 3000000 0 10000000; do
  DUPL_STREAM_OFFSET=$off $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('offset $off', d['value'], d['ms_per_step'])" >> $O/offset.txt
done
head -8 $O/shapes_2s.txt; head -6 $O/shapes_1s.txt; cat $O/offset.txt
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_kernels_gpu.py -k "gelu or format1 or kmajor" > $O/gelu_tests.txt 2>&1; tail -3 $O/gelu_tests.txt
