#!/bin/bash
# Round profile set: kernel-trace stats of the bench command, then one --pmc pass per counter group over
# tools/prof_run.py 4096 4 (4096 x 64, 151/600, four resident passes).  Run from the repository root on the GPU box.
R=$PWD
O=$R/gpurun_out/r3_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --no-configs --no-cpu-baseline --no-end-to-end > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || exit 1
for g in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $g | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $g --output-format csv -d $O/pmc_$tag -o p -- python3 $R/tools/prof_run.py 4096 4 > $O/pmc_$tag.log 2>&1 || { echo "pmc pass $tag failed"; tail -3 $O/pmc_$tag.log; exit 1; }
done
cd $R
python3 tools/pmc_summary.py $(find $O -name "p_counter_collection.csv" | sort) > $O/pmc_summary.txt
cat $O/pmc_summary.txt | grep -v "prepare_batch\|finish_stream\|fwd_pairs"
find $O -name "bench_kernel_stats.csv" | head -1 | xargs head -8
tail -c 1500 $O/bench_under_rocprof.json
# the other workloads' kernel statistics: the long-read / indel stress batch and the HaplotypeCaller-shaped call
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o c5 -- python3 $R/tools/config_call.py c5 6 > $O/c5_under_rocprof.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/hc -o hc -- python3 $R/tools/hc_call.py 2000 4 > $O/hc_under_rocprof.txt 2>&1 || exit 1
cd $R
python3 tools/trace_gaps.py $(dirname $(find $O/hc -name "hc_kernel_trace.csv" | head -1)) hc 300 > $O/hc_gaps.txt
find $O/c5 -name "c5_kernel_stats.csv" | head -1 | xargs head -8
find $O/hc -name "hc_kernel_stats.csv" | head -1 | xargs head -12
