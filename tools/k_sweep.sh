#!/bin/bash
# fp32 stream kernel time of the large-batch workload at forced rows-per-lane values (row-slot rate = groups x 64 x K / time)
for k in 8 10 11 12 14 16; do
  timeout -k 10 120 python tools/prof_run.py 4096 6 $k || exit 1
done
