#!/usr/bin/env python3
"""Register / spill / LDS figures of every kernel in an AMDGPU assembly file (hipcc -save-temps).
usage: kernel_regs.py file.s [name-filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    d = dict(re.findall(r"\.(\w+):\s+(\S+)", "  - .agpr_count:" + blk.split("\n  - .agpr_count:")[0]))
    name = d.get("name", "?")
    if flt in name:
        print("%-90s vgpr %s agpr %s sgpr %s spill %s scratch %s lds %s" % (name[:90], d.get("vgpr_count"), d.get("agpr_count"),
              d.get("sgpr_count"), d.get("vgpr_spill_count"), d.get("private_segment_fixed_size"), d.get("group_segment_fixed_size")))
