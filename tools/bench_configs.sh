#!/bin/bash
# Secondary configurations of bench.py on one GPU (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/bench_configs.sh > gpurun_out/r02_configs.txt'
# one line per configuration: img/s, ms/step, the workload string, GEMM mode, streams.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
B="python bench.py --cpu-baseline skip --no-roofline --no-exact-f32"
line() {   # name, then bench flags / env assignments via env
  local name=$1; shift
  "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(f\"$name: {d['value']} img/s {d['ms_per_step']} ms/step | {c['workload']} | forward_gemm {c['forward_gemm']} | streams {c['student_streams']} | deterministic {c.get('deterministic')} | shared pass {c['shared_scale1_encoder_pass']}\")"
}
line voc_b4 $B
line voc_b2 $B --batch 2
line voc_b8 $B --batch 8
line voc_b16 $B --batch 16 --steps 3 --warmup 1
line voc_A $B --n-iter 500
line voc_C $B --n-iter 9000
line coco_b2 $B --dataset coco --batch 2
line coco_b8 $B --dataset coco --batch 8
line coco_b2_vit21k $B --dataset coco --batch 2 --backbone vit_base_patch16_224
line voc_b4_single $B --single-stream
line voc_b4_noshare $B --no-share-encoder
line voc_b4_f32mode env DUPL_GEMM=f32 $B
line voc_b4_deterministic env DUPL_DETERMINISTIC=1 $B
line voc_b2_deterministic env DUPL_DETERMINISTIC=1 $B --batch 2
