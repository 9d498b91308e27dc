#!/bin/bash
# thread_bench over caller counts and calling modes (see tools/thread_bench.cpp); run on the GPU box from the repository root
set -e
B=tools/thread_bench_bin
for args in "1 0" "1 -2" "1 -3" "1 -4" "2 -2" "4 0" "4 1" "4 -2" "4 -3" "8 1" "8 -2" "16 1" "16 -2" "32 1"; do
  timeout -k 10 120 $B 2000 $args
done
