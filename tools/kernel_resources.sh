#!/usr/bin/env bash
# registers / scratch / LDS of every kernel in a built engine library: tools/kernel_resources.sh [library]
set -e
lib=${1:-protocol_amd/libpm_engine.so}
LLVM=/opt/rocm/lib/llvm/bin
d=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section=.hip_fatbin=$d/fat "$lib" /dev/null
$LLVM/clang-offload-bundler --type=o --unbundle --input=$d/fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$d/dev.co
$LLVM/llvm-readelf --notes $d/dev.co | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.private_segment_fixed_size:/{p=$2} /\.group_segment_fixed_size:/{g=$2} /\.vgpr_spill_count:/{sp=$2; printf "%-60s vgpr %3s sgpr %3s scratch %5s lds %6s spills %s\n", n, v, s, p, g, sp}' | sort
rm -rf $d
