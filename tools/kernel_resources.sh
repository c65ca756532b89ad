#!/bin/bash
# Register / LDS / scratch use of the kernels in a compiled unit (object file or libazg_hip.so), from the code object's metadata notes.
#   tools/kernel_resources.sh <file.o|.so> [grep -E pattern on the mangled name]
set -e
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1" /dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | awk '
  /\.group_segment_fixed_size:/{lds=$2} /\.name:/{name=$2} /\.private_segment_fixed_size:/{scr=$2}
  /\.sgpr_count:/{sg=$2} /\.sgpr_spill_count:/{ss=$2} /\.vgpr_count:/{vg=$2}
  /\.vgpr_spill_count:/{vs=$2; printf "vgpr %-4s sgpr %-4s sgpr_spill %-4s vgpr_spill %-4s scratch %-6s lds %-7s %s\n", vg, sg, ss, vs, scr, lds, name}' \
  | (if [ -n "$2" ]; then grep -E "$2"; else cat; fi) | c++filt | cut -c1-200
rm -rf $T
