#!/bin/bash
# Regenerate the round's profile summaries on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- "GIT_HEAD=$(git rev-parse HEAD) bash tools/profile_round.sh r02_final"
# Passes (all of `python bench.py --single-stream --steps 3 --warmup 1 --cpu-baseline skip`, so that per-kernel
# durations are not inflated by a co-running stream):
#   1. rocprofv3 --kernel-trace --stats            -> profiles/<tag>_kernel_stats.txt   (tools/rocpd_stats.py)
#   2. rocprofv3 --pmc SQ_* / GRBM_GUI_ACTIVE      -> profiles/<tag>_pmc_sq.txt         (tools/rocpd_pmc.py)
#   3. rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes) -> profiles/<tag>_pmc_hbm.txt
# plus the un-profiled default bench line -> profiles/<tag>_bench.json and the line measured under pass 1.
# PMC passes never combine with sys/hip/hsa traces (gpurun refuses that combination).
set -u
TAG=${1:-r02_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --single-stream --steps 3 --warmup 1 --cpu-baseline skip --no-exact-f32 --no-second"
rm -rf /tmp/prof_$TAG && mkdir -p /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/k -o k -- $BENCH > $OUT/k.log 2>&1
grep '^{' $OUT/k.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
python $R/tools/rocpd_stats.py $(find /tmp/prof_$TAG/k -name '*.db' | head -1) > $OUT/${TAG}_kernel_stats.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT \
    -d /tmp/prof_$TAG/a -o a -- $BENCH --no-roofline > $OUT/a.log 2>&1 || \
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT \
    -d /tmp/prof_$TAG/a -o a -- $BENCH --no-roofline > $OUT/a.log 2>&1
python $R/tools/rocpd_pmc.py $(find /tmp/prof_$TAG/a -name '*.db' | head -1) > $OUT/${TAG}_pmc_sq.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_$TAG/b -o b -- $BENCH --no-roofline > $OUT/b.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_$TAG/c -o c -- $BENCH --no-roofline > $OUT/c.log 2>&1
# the summary is tagged with the digest of the kernel sources it was measured on: bench.py quotes roofline.traffic from it
# only while that digest equals the build's (a changed kernel + a forgotten re-profile gives traffic = null, not a stale number)
SHA=$(cd $R && python -c "from dupl_amd.build import source_digest; print(source_digest())")
{ echo "# tag: $TAG"; echo "# steps: 4"; echo "# csrc_sha256: $SHA"; echo "# git_head: ${GIT_HEAD:-unknown}"; echo "# gemm_mode: ${DUPL_GEMM:-f16x3}"; echo "# workload: python bench.py --single-stream --steps 3 --warmup 1 (VOC 448^2, 4 img/GPU, phase B)";
  python $R/tools/rocpd_pmc.py $(find /tmp/prof_$TAG/b -name '*.db' | head -1) $(find /tmp/prof_$TAG/c -name '*.db' | head -1); } > $OUT/${TAG}_pmc_hbm.txt 2>&1
rm -rf /tmp/prof_$TAG
# the un-profiled default bench line goes last: its roofline.traffic is read from the PMC summary just produced
cp $OUT/${TAG}_pmc_hbm.txt $R/profiles/${TAG}_pmc_hbm.txt
( cd $R && timeout 600 python bench.py --pmc-profile profiles/${TAG}_pmc_hbm.txt 2> $OUT/bench.log | tail -1 > $OUT/${TAG}_bench.json )
ls -la $OUT
head -12 $OUT/${TAG}_kernel_stats.txt | cut -c1-160
