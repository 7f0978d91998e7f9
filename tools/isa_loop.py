"""Order of memory / MFMA / wait instructions in a kernel's innermost loop, runs compressed.
usage: python tools/isa_loop.py file.s <substring of the demangled kernel name> [max lines]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
want = sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 400
for m in re.finditer(r"\.amdhsa_kernel (\S+)", txt):
    sym = m.group(1)
    dm = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    if want not in dm:
        continue
    b = txt[txt.index("\n" + sym + ":"):m.start()].split("\n")
    heads = [i for i, l in enumerate(b) if "Inner Loop Header" in l]
    if not heads:
        continue
    print(dm.split("(")[0])
    out = []
    for l in b[heads[0]:heads[0] + 3000]:
        t = l.strip().split(";")[0].strip()
        if re.match(r"(s_waitcnt|s_barrier|buffer_load|buffer_store|global_|v_mfma|ds_read|ds_write|s_cbranch|v_fma_mix|s_setprio)", t):
            w = t.split()
            out.append(w[0] + (" " + w[1] if w[0] in ("s_waitcnt", "s_cbranch_scc0", "s_cbranch_scc1") else ""))
        if t.startswith("s_cbranch_scc") and len(out) > 40:
            break
    prev, c, n = None, 0, 0
    for o in out + [None]:
        if o == prev:
            c += 1
        else:
            if prev and n < lim:
                print(f"  {c:3d} x {prev}"); n += 1
            prev, c = o, 1
    break
