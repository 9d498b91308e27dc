#!/bin/bash
# SQ counters of the stream kernels on the headline batch and on the HaplotypeCaller-shaped resident batch
R=$PWD
O=$R/gpurun_out/pmc_hc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"; do
  tag=$(echo $g | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $g --output-format csv -d $O/c3_$tag -o p -- python3 $R/tools/prof_run.py 4096 3 > $O/c3_$tag.log 2>&1 || { echo "c3 pass $tag failed"; tail -3 $O/c3_$tag.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc $g --output-format csv -d $O/hc_$tag -o p -- python3 $R/tools/hc_resident_only.py 2000 2 > $O/hc_$tag.log 2>&1 || { echo "hc pass $tag failed"; tail -3 $O/hc_$tag.log; exit 1; }
done
cd $R
python3 tools/pmc_summary.py $(find $O -name "p_counter_collection.csv" | sort) | grep "fwd_stream\|csv" > $O/summary.txt
cat $O/summary.txt
