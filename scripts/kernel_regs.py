#!/usr/bin/env python3
"""VGPRs, scratch bytes and instruction counts of the level kernels from the ISA dump (make -C ray-tracing_amd/csrc asm).
Scratch in a level kernel's hot loops costs whole percents of a frame (30-50 B per lane cost S500 10 %): look at this
after every change to kernels.h.  usage: kernel_regs.py [file.s] [substring ...]"""
import re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".s") else os.path.join(ROOT, "ray-tracing_amd", "csrc", "cftrt.gfx950.s")
keys = [a for a in sys.argv[1:] if not a.endswith(".s")] or ["ILi4ELi1ELi0ELb1", "ILi4ELi1ELi4ELb0", "ILi2ELi1ELi4ELb0", "ILi2ELi2ELi7ELb",
                                                             "ILi2ELi2ELi3ELb1", "ILi4ELi1ELi2ELb", "ILi4ELi1ELi6ELb", "finish", "shadow"]
s = open(path).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, flags=re.S):
    name, body = m.group(1), m.group(2)
    if not any(k in name for k in keys):
        continue
    g = lambda k: re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body).group(1)
    i = s.index(name + ":"); j = s.index(".end_amdhsa_kernel", i)
    text = s[i:j]
    n = sum(1 for l in text.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";")))
    spills = len(re.findall(r"scratch_(?:load|store)", text))
    print(f"{name[:52]:52s} vgpr {g('next_free_vgpr'):>3s} scratch {g('private_segment_fixed_size'):>4s} B  lds {g('group_segment_fixed_size'):>6s}  instrs {n:6d}  scratch ops {spills}")
