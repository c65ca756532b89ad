#!/bin/bash
# Register / LDS / scratch use of the kernels in a compiled unit (object file or libazg_hip.so: every code object bundled into it), from the
# code objects' metadata notes (the same reader as tests/test_kernel_resources.py).
#   tools/kernel_resources.sh <file.o|.so> [grep -E pattern on the demangled name]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
python - "$1" "$2" <<PY
import sys, re
sys.path.insert(0, '$R/tests')
from test_kernel_resources import kernel_notes
pat = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else None
for name, v in sorted(kernel_notes(sys.argv[1]).items()):
    if pat and not re.search(pat, name):
        continue
    print('vgpr %-4d sgpr %-4d sgpr_spill %-4d vgpr_spill %-4d scratch %-6d lds %-7d %s' % (v['vgpr'], v['sgpr'], v['sgpr_spill'], v['vgpr_spill'], v['scratch'], v['lds'], name[:160]))
PY
