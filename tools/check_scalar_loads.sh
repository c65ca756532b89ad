#!/bin/bash
# Audit: which kernels of the forest unit would read GLOBAL memory through the scalar data cache?  Compiles azg.hip twice (default
# flags / -amdgpu-scalarize-global-loads=false) and lists the kernels whose s_load count differs -- every such kernel reads some
# wave-uniform global word with an s_load.  Mutable forest memory must not show up here (forest.hip.h ld_agent_u32).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
for v in sc ns; do
  mkdir -p $T/$v && cd $T/$v
  EXTRA=""; [ $v = ns ] && EXTRA="-mllvm -amdgpu-scalarize-global-loads=false"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $EXTRA -I$ROOT/include --save-temps \
      -c $ROOT/alpha-zero-general_amd/csrc/azg.hip -o azg.o 2>/dev/null
  awk '/^_ZN3azg[^:]*:/{name=$1} /^\ts_load_dword/{c[name]++} END{for(k in c) print c[k], k}' azg-hip-amdgcn-amd-amdhsa-gfx950.s | sort -k2 > $T/$v.cnt
done
echo "s_loads default / no-scalarize / kernel (only kernels that differ):"
join -1 2 -2 2 $T/sc.cnt $T/ns.cnt | awk '$2>$3{print $2, $3, $1}' | c++filt | cut -c1-160
rm -rf $T
