#!/bin/bash
O=gpurun_out/r04c; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "kmajor" > $O/t_km.log 2>&1; echo "km rc=$?" > $O/rc.txt; tail -3 $O/t_km.log
DG="3140x768x3072,3140x3072x768,3140x768x2304,3140x768x768"
WG="3072x768x3168,768x3072x3168,2304x768x3168,768x768x3168"
{
echo "# dgrad k-major, one / two streams"
tools/gemm16_bench -t 0 -s $DG -w 150 -f -L nk
tools/gemm16_bench -t 0 -s $DG -w 150 -f -L nk -2
echo "# wgrad k-major, one / two streams"
tools/gemm16_bench -t 0 -s $WG -w 150 -e 4 -f -L kk
tools/gemm16_bench -t 0 -s $WG -w 150 -e 4 -f -L kk -2
echo "# format 1 k-contiguous persistent 256x128 (tile 14) on the dgrad shapes, for reference (same kernel structure, b128 reads)"
tools/gemm16_bench -t 14,12 -s $DG -w 150 -f
tools/gemm16_bench -t 14,12 -s $DG -w 150 -f -2
} > $O/km_bench.txt 2>&1
cat $O/km_bench.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in nk kk; do
  E=0; S=$DG; [ $L = kk ] && E=4 && S=$WG
  rm -rf /tmp/pmc_$L
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_$L -o p -- $R/tools/gemm16_bench -t 0 -s $S -n 5 -e $E -f -L $L > $R/$O/pmc_$L.log 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/pmc_$L -name '*.db' | head -1) > $R/$O/pmc_$L.txt 2>&1
done
rm -rf /tmp/pmc_ref
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_ref -o p -- $R/tools/gemm16_bench -t 14 -s $DG -n 5 -f > $R/$O/pmc_ref.log 2>&1
python $R/tools/rocpd_pmc.py $(find /tmp/pmc_ref -name '*.db' | head -1) > $R/$O/pmc_ref.txt 2>&1
cd $R
head -8 $O/pmc_nk.txt | cut -c1-260; head -8 $O/pmc_kk.txt | cut -c1-260; head -8 $O/pmc_ref.txt | cut -c1-260
cat $O/rc.txt
