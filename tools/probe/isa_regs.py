#!/usr/bin/env python3
"""Register use and spill counts of every kernel in a -save-temps .s file whose mangled name contains the pattern.
    python tools/probe/isa_regs.py /tmp/sbe_mixture_mfma-hip-amdgcn-amd-amdhsa-gfx950.s tuple_mfma"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
md = s[s.index("amdhsa.kernels"):]
for blk in md.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if pat not in name:
        continue
    get = lambda k: re.search(k + r":\s+(\d+)", blk).group(1)
    print(f"{name[:110]:110s} agpr {blk.split()[0]:>3s} vgpr {get(r'.vgpr_count'):>3s} spill {get(r'.vgpr_spill_count'):>3s} "
          f"sgpr {get(r'.sgpr_count'):>3s} sgpr_spill {get(r'.sgpr_spill_count'):>3s} lds {get(r'.group_segment_fixed_size')}")
