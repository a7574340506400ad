#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of the built objects (AMDGPU metadata notes): tools/kernel_regs.py [object stems...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "dkt_stereo_amd", "lib", "obj")
LLVM = "/opt/rocm/lib/llvm/bin"
stems = sys.argv[1:] or ["conv_c8", "gru_c8"]
tmp = tempfile.mkdtemp(prefix="dkt_regs_")
try:
    for stem in stems:
        src = stem if os.path.exists(stem) else os.path.join(OBJ, stem + ".o")
        local = os.path.join(tmp, os.path.basename(src))
        shutil.copy(src, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if not f.startswith(os.path.basename(src) + ".") or "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], stdout=subprocess.PIPE, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                get = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
                name = subprocess.run(["c++filt", get("name")], stdout=subprocess.PIPE, text=True).stdout.strip()
                print("%-70s vgpr %3s agpr %3s spill %3s scratch %5s lds %6s" % (re.sub(r"\(.*", "", name)[:70], get("vgpr_count"), blk.split()[0],
                                                                              get("vgpr_spill_count"), get("private_segment_fixed_size"), get("group_segment_fixed_size")))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
