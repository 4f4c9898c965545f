"""Condense the per-regime counter passes of tools/profile/run_regimes.sh into profiles/<PROFILE_TAG>_regimes.json, the file
bench.py reads for `roofline.regimes[*].mfma_busy` and `.l2_request_bytes`.

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 shader engines * 1024 SIMDs): the share of the launch during which a
SIMD's matrix pipe is busy, averaged over the chip (SQ_BUSY_CYCLES sums the 32 shader engines; MI355X_MICROARCH.md: SQ_VALU_MFMA_
BUSY_CYCLES counts cycles).  l2_request_bytes = TCC_REQ_sum * 128.  Per launch, layer-1 instance of the edge-update kernel (the
one that also runs the W_B stages; the layer-0 instance is listed next to it)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    if f:
        for r in csv.DictReader(open(f[0])):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    return {k: {c: v / cnt[(k, c)] for c, v in cs.items()} | {"launches": max(cnt[(k, c)] for c in cs)} for k, cs in agg.items()}


def main(out):
    tag = os.environ.get("PROFILE_TAG", "r05_v1")
    regimes = {}
    for w, key in (("t1124", "t1124"), ("s1500", "s1500"), ("c5share", "c5")):
        sq, tcc = per_kernel(os.path.join(out, "sq_" + w)), per_kernel(os.path.join(out, "tcc_" + w))
        edge = sorted(k for k in sq if "edge_update" in k)
        if not edge:
            continue
        # layer 1 (ST0 = false) is the instance with the W_B stages: "<..., false, true...>" / the mixed launch "<false, true...>"
        l1 = [k for k in edge if "false, true" in k] or edge
        k = l1[0]
        e = {"kernel": k, "launches": sq[k]["launches"],
             "mfma_busy": sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq[k]["SQ_BUSY_CYCLES"] / 32.0 * 1024.0),
             "sq": {c: round(v) for c, v in sq[k].items()}}
        if k in tcc:
            e["l2_request_bytes"] = 128.0 * tcc[k]["TCC_REQ_sum"]
            e["l2_hit_rate"] = tcc[k]["TCC_HIT_sum"] / max(tcc[k]["TCC_HIT_sum"] + tcc[k]["TCC_MISS_sum"], 1.0)
        others = {}
        for k2 in edge:
            if k2 != k:
                others[k2] = {"mfma_busy": sq[k2]["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq[k2]["SQ_BUSY_CYCLES"] / 32.0 * 1024.0),
                              "l2_request_bytes": 128.0 * tcc[k2]["TCC_REQ_sum"] if k2 in tcc else None}
        e["other_instances"] = others
        regimes[key] = e
    path = os.path.join(ROOT, "profiles", tag + "_regimes.json")
    json.dump({"command": "rocprofv3 --pmc <SQ set | TCC set> -- python3 bench.py --workload W --steps 1 --warmup 0 --cpu-steps 0 --no-secondary --no-roofline "
                          "--diffusion-steps 6 (HIP_FORCE_DEV_KERNARG=1 exported first); tools/profile/run_regimes.sh",
               "regimes": regimes}, open(path, "w"), indent=1)
    print(json.dumps({k: {"mfma_busy": round(v["mfma_busy"], 3), "l2_MB": round(v.get("l2_request_bytes", 0) / 1e6, 1)} for k, v in regimes.items()}))


if __name__ == "__main__":
    main(sys.argv[1])
