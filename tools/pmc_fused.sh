#!/bin/bash
# dev: PMC counters of the fused kernel on the sweep tool.  usage: tools/pmc_fused.sh "<ctrs>" [debug]
cd /tmp && export TMPDIR=/tmp
export LGNN_FUSED_DEBUG=${2:-0}
rm -rf /tmp/pmcf
timeout -k 10 300 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d /tmp/pmcf -- python3 $GRAFT_REPO_ROOT/tools/sweep_fused.py 169343 > /tmp/pmcf.log 2>&1
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmcf/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "spmm_gram256" in r["Kernel_Name"]:
            agg["fused"][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg["fused"]["dur_us"].append((float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3)
for k,v in agg.items():
    print(k, {c:(round(sum(x)/len(x),1)) for c,x in v.items()}, "n=",len(v["dur_us"]))
PY
