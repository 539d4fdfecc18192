""" experiment build of the library with ALL kernels of width 64 (other widths stubbed): python tools/variant64.py <name> [extra hipcc flags...] -> gpurun_variants/lib_<name>.so """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pydens_amd.csrc import build
name, flags = sys.argv[1], sys.argv[2:]
print(build.build(force=True, extra_flags=list(flags), out=f'/root/repo/gpurun_variants/lib_{name}.so', widths=(64,)))
