#!/bin/bash
# kernel_regs.sh [object] [filter]: VGPRs, spills and LDS of the kernels in a built object (default: the probe kernels)
set -e
obj=${1:-$(dirname "$0")/../trio_binning_amd/csrc/build/tbk_kernels.o}
B=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fatbin "$obj"
$B/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fatbin --output=$tmp/dev.co
$B/llvm-readelf --notes $tmp/dev.co | awk '/\.group_segment_fixed_size:/{lds=$2} /^ +\.name:/{name=$2} /\.sgpr_spill_count:/{ss=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{print name, "vgpr", v, "sgpr_spill", ss, "vgpr_spill", $2, "lds", lds}' | grep "${2:-probe}"
rm -rf $tmp
