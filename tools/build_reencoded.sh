#!/bin/bash
# Builds exp_build/perm/lib_<tag>.so from the device assembly after a sed script has been applied to it.
# usage: build_reencoded.sh TAG 'sed script'
set -e
cd "$(dirname "$0")/.."
TAG=$1
LLVM=/opt/rocm/lib/llvm/bin
D=exp_build/perm
mkdir -p $D
FLAGS="-O3 -std=c++17 -fPIC -fno-slp-vectorize"
if [ ! -f $D/dev.s ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS --cuda-device-only -S gatk_amd/csrc/pairhmm_kernels.hip -o $D/dev.s
fi
sed -E "$2" $D/dev.s > $D/dev_$TAG.s
$LLVM/clang -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $D/dev_$TAG.s -o $D/dev_$TAG.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared $D/dev_$TAG.o -o $D/dev_$TAG.out
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$D/dev_$TAG.out -output=$D/k_$TAG.hipfb
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $D/k_$TAG.hipfb -c gatk_amd/csrc/pairhmm_kernels.hip -o $D/k_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $D/lib_$TAG.so $D/k_$TAG.o gatk_amd/csrc/pairhmm_api.o gatk_amd/csrc/pairhmm_tables.o gatk_amd/csrc/sw_kernels.o 2>/dev/null
rm -f $D/dev_$TAG.o $D/dev_$TAG.out $D/k_$TAG.hipfb $D/k_$TAG.o $D/dev_$TAG.s
ls -la $D/lib_$TAG.so
