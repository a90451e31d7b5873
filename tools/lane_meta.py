#!/usr/bin/env python3
"""Scratch / registers / spills of every kernel in an AMDGPU assembly file's metadata (default /tmp/q/lane.s)."""
import re, sys
text = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/q/lane.s").read()
meta = text[text.index("amdhsa.kernels:"):]
for block in meta.split("  - .agpr_count:")[1:]:
    get = lambda key: re.search(r"\." + key + r":\s+(\S+)", block).group(1)
    print(f"{get('name'):20s} scratch {get('private_segment_fixed_size'):>5s} B  vgpr {get('vgpr_count'):>4s} spills {get('vgpr_spill_count'):>4s}  sgpr spills {get('sgpr_spill_count'):>3s}  lds {get('group_segment_fixed_size')}")
