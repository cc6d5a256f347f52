#!/usr/bin/env python
"""Register / LDS / scratch footprint of every kernel in libia3.so's object files (from the code-object metadata).
usage: python scripts/kernel_regs.py [object ...]   (default: imageanalysis3_amd/csrc/build/*.o)"""
import glob, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "imageanalysis3_amd/csrc/build/*.o")))
for o in objs:
    with tempfile.TemporaryDirectory() as td:
        co, fat = os.path.join(td, "k.co"), os.path.join(td, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", o, fat], capture_output=True)
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:90]
        print("%-14s vgpr %3s agpr %3s sgpr %3s spill %3s lds %6s scratch %5s  %s" % (
            os.path.basename(o), g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"),
            g("group_segment_fixed_size"), g("private_segment_fixed_size"), name))
