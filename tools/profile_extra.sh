#!/bin/bash
# Side measurements of a round (run through gpurun from the repo root; everything lands in gpurun_out/<tag>/ and is then copied to
# profiles/ by hand):   gpurun --timeout 1500 -- 'bash tools/profile_extra.sh r03'
#   <tag>_gemm16_tiles.txt    sustained TF/s-eq of every tile variant of dupl_gemm_f16x3 on the step's shapes, one and two streams
#   <tag>_gemm16_phases.txt   per-block s_memtime breakdown (prologue / k-loop / epilogue) and in-block clock of the ring kernel
#   <tag>_power.txt           socket power + sclk (rocm-smi) under: pure f16 MFMA probe, the GEMM tiles, GEMM minus MFMA, LDS only
#   <tag>_attn16.txt          dupl_attention_fwd16 stand-alone + per-phase cycles of wave 0
#   <tag>_configs.txt         bench.py on the secondary configurations
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/$TAG
mkdir -p $OUT
B=tools/gemm16_bench
{ echo "# tools/gemm16_bench -s all -t 3,5,6,10 -w 200 (one stream)"; $B -s all -t 3,5,6,10 -w 200;
  echo "# the same with -2 (every launch on two streams at a time: the two students)"; $B -s all -t 3,5,6,10 -w 200 -2;
  echo "# epilogue 1 (bias -> planes), 2 (bias + GELU + stored pre-activation -> planes), 3 (bias + residual -> fp32), two streams";
  for e in 1 2 3; do $B -s fwd -t 5,10 -w 150 -2 -e $e; done;
  W=3072x768x3168,768x3072x3168,2304x768x3168,768x768x3168,3072x768x1600,768x3072x1600,2304x768x1600,768x768x1600
  echo "# epilogue 4 (ACCUM: the weight gradients at 4 / 2 img per GPU; 3 / 5: split-K grids, 11: stream-K form of the persistent kernel), one stream"; $B -s $W -t 3,5,11,0 -w 150 -e 4;
  echo "# the same, two streams"; $B -s $W -t 3,5,11,0 -w 150 -2 -e 4;
  echo "# format 1 operand planes (-f: A * 2^3, B * 2^9, unscaled lo; one accumulator set): 8 = 256 x 256, 12 = 256 x 128, 14 = persistent 256 x 128, 0 = the launcher's choice; two streams";
  $B -s fwd -t 8,12,14,0 -w 150 -2 -f; $B -s 1576x3072x768,1576x768x3072,1576x768x768 -t 8,12,14,0 -w 150 -2 -f;
  echo "# the same, one stream"; $B -s fwd -t 8,12,0 -w 150 -f;
  echo "# format 1, epilogues 1 / 2 / 3 (planes out in format 1), two streams"; for e in 1 2 3; do $B -s fwd -t 8,12 -w 150 -2 -f -e $e; done;
  DG=3140x768x3072,3140x3072x768,3140x768x2304,3140x768x768,1570x768x3072,1570x3072x768,1570x768x2304
  WG=3072x768x3168,768x3072x3168,2304x768x3168,768x768x3168,3072x768x1600,768x3072x1600,2304x768x1600
  echo "# round 4, k-major single-accumulator backward GEMMs (-f -L nk: dgrad, B = the forward's W planes read k-major; -L kk -e 4: wgrad, dy and x planes k-major, stream-K) vs the transposed-planes path they replace (format 0, heuristic tile), one stream";
  echo "# dgrad, transposed-planes path"; $B -s $DG -t 0 -w 150;
  echo "# dgrad, k-major"; $B -s $DG -t 0 -w 150 -f -L nk;
  echo "# dgrad with a linear epilogue, k-major stream-K into a zero-filled dx (-e 4)"; $B -s $DG -t 0 -w 150 -f -L nk -e 4;
  echo "# wgrad, transposed-planes path"; $B -s $WG -t 0 -w 150 -e 4;
  echo "# wgrad, k-major"; $B -s $WG -t 0 -w 150 -e 4 -f -L kk;
  echo "# the same five, two streams";
  $B -s $DG -t 0 -w 150 -2; $B -s $DG -t 0 -w 150 -f -L nk -2; $B -s $DG -t 0 -w 150 -f -L nk -e 4 -2; $B -s $WG -t 0 -w 150 -e 4 -2; $B -s $WG -t 0 -w 150 -e 4 -f -L kk -2; } > $OUT/${TAG}_gemm16_tiles.txt 2>&1
if [ "${WITH_ABL:-0}" = 1 ]; then
{ echo "# LD_LIBRARY_PATH=tools/abl/16 (G16_ABL=16: s_memtime stamps of wave 0 per block): prologue / k-loop / epilogue cycles";
  LD_LIBRARY_PATH=tools/abl/16 $B -s 15696x3072x768,15696x768x3072,6280x3072x768,3140x3072x768 -t 6,7 -d -p; } > $OUT/${TAG}_gemm16_phases.txt 2>&1
fi
S=15696x3072x768
{ echo "# rocm-smi samples (sclk, socket W) while the command runs; idle first"; rocm-smi --showpower --showclocks | grep -E "Power \(W\)|sclk";
  tools/power_probe.sh "pure f16 MFMA probe (registers only, random operands, 2 waves / SIMD)" $B -P 4000000 -s 129x128x32 -t 5 -c;
  for t in 5 6 10; do tools/power_probe.sh "dupl_gemm_f16x3 tile $t, $S" $B -s $S -t $t -n 8000; done;
  for t in 8 12; do tools/power_probe.sh "dupl_gemm_f16x3 format 1 tile $t, $S" $B -s $S -t $t -n 8000 -f; done;
  tools/power_probe.sh "k-major wgrad 3072x768x3168 (stream-K)" $B -s 3072x768x3168 -t 0 -n 20000 -e 4 -f -L kk;
  if [ "${WITH_ABL:-0}" = 1 ]; then
  LD_LIBRARY_PATH=tools/abl/18 tools/power_probe.sh "tile 6 without the MFMAs (DMA + LDS reads + epilogue)" $B -s $S -t 6 -n 12000;
  LD_LIBRARY_PATH=tools/abl/19 tools/power_probe.sh "tile 6 without MFMAs and DMA (LDS reads + barriers + epilogue)" $B -s $S -t 6 -n 20000; fi; } > $OUT/${TAG}_power.txt 2>&1
{ tools/attn16_bench -w 200; if [ "${WITH_ABL:-0}" = 1 ]; then echo "# ATT_ABL=16 build: cycles per key tile of wave 0"; LD_LIBRARY_PATH=tools/abl/att16 tools/attn16_bench -d -w 50; fi; } > $OUT/${TAG}_attn16.txt 2>&1
bash tools/bench_configs.sh > $OUT/${TAG}_configs.txt 2>&1
ls -la $OUT
