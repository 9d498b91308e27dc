#!/bin/bash
# Register, spill and LDS figures of the kernels in a built library whose names match a pattern.  usage: kernel_regs.sh LIB PATTERN
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co 2>/dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/co | awk -v pat="$2" '
/\.agpr_count:/ {agpr=$NF} /\.name:/ {name=$NF} /\.vgpr_count:/ {v=$NF} /\.vgpr_spill_count:/ {sp=$NF} /\.sgpr_count:/ {sg=$NF}
/\.private_segment_fixed_size:/ {pr=$NF} /\.group_segment_fixed_size:/ {lds=$NF}
/\.wavefront_size:/ { if (name ~ pat) printf "%-60s vgpr %3d agpr %3d sgpr %3d spills %3d scratch %4d\n", name, v, agpr, sg, sp, pr }'
rm -rf $T
