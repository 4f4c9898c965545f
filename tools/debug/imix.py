"""Instruction mix per kernel from a --save-temps gfx950 .s file: VALU / MFMA / LDS / VMEM / nop wait states."""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
ends = [i for i, l in enumerate(lines) if "s_endpgm" in l]
names = [l.split(":")[0] for l in lines if re.match(r"^_Z\w+:", l)]
start = 0
for k, e in enumerate(ends):
    body = lines[start:e]; start = e
    if pat and pat not in names[k]: continue
    c = dict(valu=0, cvt=0, mfma=0, lds=0, vmem=0, salu=0, nop=0, barrier=0)
    for l in body:
        t = l.strip().split(" ")[0] if l.strip() else ""
        if t.startswith("v_mfma"): c["mfma"] += 1
        elif t.startswith("v_cvt"): c["cvt"] += 1; c["valu"] += 1
        elif t.startswith("v_"): c["valu"] += 1
        elif t.startswith("ds_"): c["lds"] += 1
        elif t.startswith(("global_", "scratch_", "buffer_")): c["vmem"] += 1
        elif t == "s_nop": c["nop"] += int(l.strip().split()[1]) + 1
        elif t == "s_barrier": c["barrier"] += 1
        elif t.startswith("s_"): c["salu"] += 1
    print("%-50s %s" % (names[k][:50], c))
