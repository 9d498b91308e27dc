#!/bin/bash
# Builds exp_build/stress_asan: tools/stress_driver.cpp + the engine's host code under AddressSanitizer and
# UndefinedBehaviorSanitizer (host side only; the device code is the shipped objects), and
# exp_build/stress: the same without sanitizers.
set -e
cd "$(dirname "$0")/.."
mkdir -p exp_build
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
C=gatk_amd/csrc
make -s -C $C
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC $SAN -x hip -c $C/pairhmm_api.cpp -o exp_build/api_asan.o
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC $SAN -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $C/pairhmm_async.cpp -o exp_build/async_asan.o
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC $SAN -x c++ -c $C/pairhmm_tables.cpp -o exp_build/tables_asan.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Iinclude $SAN -c $C/sw_kernels.hip -o exp_build/sw_asan.o
/opt/rocm/bin/hipcc -O1 -g -std=c++17 $SAN -x c++ -c tools/stress_driver.cpp -o exp_build/stress_asan.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $SAN -pthread -o exp_build/stress_asan exp_build/stress_asan.o exp_build/api_asan.o exp_build/async_asan.o exp_build/tables_asan.o exp_build/sw_asan.o $C/pairhmm_kernels.o
g++ -O2 -std=c++17 -pthread tools/stress_driver.cpp -Lgatk_amd -lgatk_pairhmm_hip -Wl,-rpath,'$ORIGIN/../gatk_amd' -o exp_build/stress
rm -f exp_build/*_asan.o
ls -la exp_build/stress_asan exp_build/stress
