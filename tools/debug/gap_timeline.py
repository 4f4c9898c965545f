"""Timeline of a rocprofv3 --kernel-trace run: per kernel, mean duration and mean idle gap to the NEXT dispatch (start of the
next kernel minus end of this one), over the sampling loop.   python tools/debug/gap_timeline.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(({"name": r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])}
               for r in csv.DictReader(open(f))), key=lambda r: r["s"])
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for a, b in zip(rows[:-1], rows[1:]):
    g = b["s"] - a["e"]
    if g > 50_000:            # host-side pause (between passes), not part of the loop
        continue
    key = a["name"] + " -> " + b["name"].split("<")[0]
    dur[key] += a["e"] - a["s"]; gap[key] += g; cnt[key] += 1
tot_d = tot_g = 0.0
for k in sorted(cnt, key=lambda k: -cnt[k])[:14]:
    print(f"{k:80s} n {cnt[k]:5d}  dur {dur[k] / cnt[k] / 1e3:7.2f} us   gap {gap[k] / cnt[k] / 1e3:6.2f} us")
    tot_d += dur[k]; tot_g += gap[k]
print(f"sum of durations {tot_d / 1e6:.2f} ms, sum of gaps {tot_g / 1e6:.2f} ms")
