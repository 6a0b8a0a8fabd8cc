#!/bin/bash
# usage: bash profiles/kernel_resources.sh <file.o> ...   -> per kernel: LDS bytes, name, scratch bytes, spilled SGPRs, VGPRs, spilled VGPRs (code-object notes of the gfx950 bundle)
for f in "$@"; do
t=$(mktemp -d); objcopy -O binary --only-section=.hip_fatbin "$f" $t/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$t/fat.bin --output=$t/k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $t/k.co | grep -E "^\s+\.name:|\.vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|\.group_segment_fixed" | awk '{printf "%s ", $0} /vgpr_spill/ {print ""}' | sed 's/  */ /g' | cut -c1-260
rm -rf $t
done
