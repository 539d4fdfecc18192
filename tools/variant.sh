#!/bin/bash
# experiment build of the library with only the kernels of the BASELINE configs (-DPINN_ONLY_BASELINE): tools/variant.sh <name> [extra hipcc flags...]
# -> gpurun_variants/lib_<name>.so (git-ignored, travels to the GPU box); compare with tools/kbench.py cfg2 <libs...>
NAME=$1; shift
python - "$NAME" "$@" <<'PY'
import sys
from pydens_amd.csrc import build
name, flags = sys.argv[1], sys.argv[2:]
print(build.build(force=True, extra_flags=['-DPINN_ONLY_BASELINE', '-DPINN_DEBUG_ABI', *flags], out=f'/root/repo/gpurun_variants/lib_{name}.so', widths=tuple(int(w) for w in __import__('os').environ.get('VARIANT_WIDTHS', '64,128,256').split(','))))
PY
