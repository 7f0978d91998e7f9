"""Print VGPR / AGPR / scratch / LDS of every kernel in a hipcc -S device listing (optionally filtered).
usage: python tools/isa_regs.py file.s [substring]"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_%s (\S+)" % k, body) or [None, "?"])[1]
    dm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dm = dm.replace("(anonymous namespace)::", "").split("(")[0]
    if flt in dm:
        print(f"{dm:70s} vgpr {g('next_free_vgpr'):>4} accum_off {g('accum_offset'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}")
