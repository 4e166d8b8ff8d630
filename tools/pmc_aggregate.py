"""Dev tool: average per-dispatch PMC values per kernel from rocprofv3 --pmc passes -> CSV (+ JSON for the fused kernel).
usage: pmc_aggregate.py <out_prefix> <pass_dir> [<pass_dir> ...]"""
import collections, csv, glob, json, sys

out, dirs = sys.argv[1], sys.argv[2:]
val = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for d in dirs:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            val[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
ctrs = sorted({c for v in val.values() for c in v})
rows = []
for k, v in val.items():
    n = max(cnt[(k, c)] for c in v)
    rows.append([k[:70], n] + [round(v[c] / cnt[(k, c)], 1) if c in v else "" for c in ctrs])
rows.sort(key=lambda r: -max(x for x in r[2:] if x != ""))
with open(out + "_pmc_per_kernel.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches"] + [c + "_avg" for c in ctrs])
    w.writerows(rows[:16])
import os
want = os.environ.get("LGNN_PMC_KERNEL", "spmm_gram256_kernel")
wl = os.environ.get("LGNN_PMC_WORKLOAD", "arxiv")
# the dominant kernel = the template instance of `want` with the largest summed FETCH_SIZE (or dispatch count)
cands = [k for k in val if want in k]
cands.sort(key=lambda k: -(val[k].get("FETCH_SIZE", 0.0) + val[k].get("GRBM_GUI_ACTIVE", 0.0)))
for k in cands[:1]:
    v = val[k]
    if True:
        a = {c: v[c] / cnt[(k, c)] for c in v}
        j = {"kernel": k[:120], "workload": wl,
             "command": f"rocprofv3 --pmc <ctrs> -- python3 bench.py --workload {wl} --steps 1 --warmup 0 --no-cpu-baseline (one pass per counter group)",
             "dispatches": max(cnt[(k, c)] for c in v), "planes_per_launch": float(os.environ.get("LGNN_PLANES_PER_LAUNCH", "40"))}
        j.update({c + "_avg": a[c] for c in a})
        # every template instance of the kernel with its own averages (arxiv: <false> = the full batches' launches,
        # <true> = the short last batch's, which walks a node list)
        j["instances"] = [{"kernel": kk[:120], "dispatches": max(cnt[(kk, c)] for c in val[kk]),
                           **{c + "_avg": val[kk][c] / cnt[(kk, c)] for c in val[kk]}} for kk in cands]
        if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
            j["note"] = ("gfx950: FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md HBM section) -> "
                         "traffic = (2*FETCH_SIZE + WRITE_SIZE) KB; FETCH_SIZE counts Infinity-Cache hits too")
            j["traffic_bytes_per_launch"] = (2 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024
        json.dump(j, open(out + ("_pmc_fused.json" if wl == "arxiv" else "_pmc_dominant.json"), "w"), indent=1)
print(open(out + "_pmc_per_kernel.csv").read()[:1500])
