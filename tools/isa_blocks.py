"""Instruction mix per basic block of one kernel in a hipcc -S listing (laboratory tool).
usage: python tools/isa_blocks.py <file.s> <substring of the mangled kernel name> [min instructions per block]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
start = [i for i, l in enumerate(s) if l.startswith("_Z") and pat in l.split(":")[0] and ":" in l][0]
end = [i for i in range(start, len(s)) if "s_endpgm" in s[i]][0]
print(s[start])
blocks, cur, lab = [], [], "entry"
for l in s[start + 1:end]:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append((lab, cur))
        cur, lab = [], l
    else:
        cur.append(l.strip())
blocks.append((lab, cur))
for lab, b in blocks:
    ins = [x.split()[0] for x in b if x and not x.startswith(";") and not x.startswith(".")]
    if len(ins) < min_n:
        continue
    c = Counter()
    for x in ins:
        k = ("mfma" if "mfma" in x else "ds_read" if x.startswith("ds_read") else "ds_write" if x.startswith("ds_write")
             else "vmem_ld" if "load" in x and (x.startswith("global") or x.startswith("buffer") or x.startswith("scratch"))
             else "vmem_st" if "store" in x else "waitcnt" if x == "s_waitcnt" else "barrier" if x == "s_barrier"
             else "branch" if x.startswith("s_cbranch") or x == "s_branch" else "salu" if x.startswith("s_") else "valu")
        c[k] += 1
    # longest run of MFMAs without another vector instruction between them
    run = best = 0
    for x in ins:
        if "mfma" in x:
            run += 1
            best = max(best, run)
        elif x.startswith("v_") or x.startswith("ds_") or x.startswith("global") or x.startswith("buffer"):
            run = 0
    print(lab, len(ins), dict(c), "longest bare MFMA run", best)

if len(sys.argv) > 4:  # pattern of one block: M mfma, v valu, r ds_read, w ds_write, L load, S store, B barrier, . waitcnt
    want = sys.argv[4]
    for lab, b in blocks:
        if not lab.startswith(want):
            continue
        out = ""
        for x in b:
            if not x or x.startswith(";") or x.startswith("."):
                continue
            x = x.split()[0]
            out += ("M" if "mfma" in x else "r" if x.startswith("ds_read") else "w" if x.startswith("ds_write") else
                    "L" if "load" in x else "S" if "store" in x else "." if x == "s_waitcnt" else "B" if x == "s_barrier" else
                    "" if x.startswith("s_") else "v")
        print(out)
