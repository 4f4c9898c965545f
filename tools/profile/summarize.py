"""Condense rocprofv3 output (run_profiles.sh) into the small files committed under profiles/.

profiles/<round>_<tag>_kernel_stats.csv   per-kernel calls / total / average / percentage (from --kernel-trace --stats)
profiles/<round>_<tag>_pmc_traffic.json   per-kernel mean FETCH_SIZE / WRITE_SIZE per launch, in bytes, with the gfx950
                                          correction of MI355X_MICROARCH.md "HBM": FETCH_SIZE is reported in KiB and counts
                                          128-B requests as 64 B (doubled here); WRITE_SIZE in KiB is exact.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROUND = os.environ.get("GRAFT_ROUND", "r05")


def short(name):
    name = name.split("(")[0].strip()
    if name.startswith("void "):
        name = name[5:]
    name = name.split("<")[0]
    return name.split("::")[-1].strip()


def kernel_stats(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = []
    if f:
        for r in csv.DictReader(open(f[0])):
            rows.append(r)
    return rows


def pmc_mean(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main(out, tag):
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    rows = kernel_stats(os.path.join(out, "stats"))
    if rows:
        keep = [k for k in rows[0].keys()]
        with open(os.path.join(prof, f"{ROUND}_{tag}_kernel_stats.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=keep)
            w.writeheader()
            for r in rows:
                w.writerow(r)
    fetch = pmc_mean(os.path.join(out, "fetch"), "FETCH_SIZE")
    write = pmc_mean(os.path.join(out, "write"), "WRITE_SIZE")
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        fk, n = fetch.get(k, (0.0, 0))
        wk, _ = write.get(k, (0.0, 0))
        traffic[k] = {"launches": n, "fetch_size_kib_raw": fk, "write_size_kib_raw": wk,
                      "fetch_bytes": 2.0 * fk * 1024.0, "write_bytes": wk * 1024.0,
                      "hbm_bytes": 2.0 * fk * 1024.0 + wk * 1024.0}
    meta = {"command": "python3 bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-secondary",
            "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE KiB x1024",
            "kernels": traffic}
    with open(os.path.join(prof, f"{ROUND}_{tag}_pmc_traffic.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    for name in ("stats.json", "fetch.json", "write.json"):
        p = os.path.join(out, name)
        if os.path.exists(p) and name == "stats.json":
            line = open(p).read().strip().splitlines()
            if line:
                open(os.path.join(prof, f"{ROUND}_{tag}_bench_under_rocprof.json"), "w").write(line[-1] + "\n")
    print(json.dumps({k: round(v["hbm_bytes"] / 1e6, 2) for k, v in traffic.items()}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "final")
