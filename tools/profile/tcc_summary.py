"""Condense a TCC counter pass (tools/profile/run_tcc.sh <workload>) into profiles/<PROFILE_TAG>_tcc_<workload>.json:
128-byte L2 requests, hits and misses per launch of the hot kernels.   python3 tools/profile/tcc_summary.py <out dir> <workload>"""
import csv, glob, collections, json, os, sys
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        if "edge_update" in k or "node_message" in k or "node_update" in k:
            d = {c: v / cnt[(k, c)] for c, v in agg[k].items()}
            hit = d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1)
            print(k, {c: round(v) for c, v in d.items()}, "L2 hit rate %.3f" % hit)
            out[k.replace("void ", "")] = {"launches": cnt[(k, "TCC_REQ_sum")], "l2_requests": d.get("TCC_REQ_sum", 0), "l2_hits": d.get("TCC_HIT_sum", 0),
                                           "l2_misses": d.get("TCC_MISS_sum", 0), "l2_hit_rate": hit,
                                           "l2_request_bytes": 128.0 * d.get("TCC_REQ_sum", 0)}
tag = os.environ.get("PROFILE_TAG", "r05_v1")
json.dump({"command": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- python3 bench.py --workload %s --steps 1 --warmup 1 --cpu-steps 0 --no-secondary" % sys.argv[2],
           "request_size_bytes": 128, "kernels": out}, open(os.path.join("profiles", "%s_tcc_%s.json" % (tag, sys.argv[2])), "w"), indent=1)
