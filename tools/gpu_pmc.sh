#!/bin/bash
# usage: tools_gpu_pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]  -- per-kernel PMC sums (bench, 1 step)
tag=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
n=0
for ctrs in "$@"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n.log 2>&1
  python3 - <<PY
import csv,glob,collections
fs=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n/*/*counter_collection.csv")
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:50]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        cnt[(k,r["Counter_Name"])]+=1
for k,v in sorted(agg.items(), key=lambda kv:-max(kv[1].values()))[:8]:
    print(k.ljust(50), {c:(round(x/cnt[(k,c)],1), cnt[(k,c)]) for c,x in v.items()})
PY
done
