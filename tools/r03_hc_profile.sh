#!/bin/bash
# HaplotypeCaller-shaped call: wall time, then a kernel trace of the same command.  Run from the repository root on the GPU box.
R=$PWD
O=$R/gpurun_out/r3_hc
mkdir -p $O
timeout -k 10 120 python3 tools/hc_call.py 2000 10 || exit 1
timeout -k 10 120 python3 tools/hc_call.py 2000 10 timing=1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace -o hc -- python3 $R/tools/hc_call.py 2000 4 > $O/hc_under_rocprof.txt 2>&1 || exit 1
cd $R
f=$(find $O/trace -name "hc_kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $(dirname $f) hc 300
find $O/trace -name "hc_kernel_stats.csv" | head -1 | xargs head -14
