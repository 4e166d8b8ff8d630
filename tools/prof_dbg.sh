#!/bin/bash
# usage (inside gpurun): tools/prof_dbg.sh <tag>   -- kernel stats of bench.py under the current LGNN_* debug environment
tag=$1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print("$tag", r["Name"][:60].ljust(60), r["Calls"].rjust(5), str(round(float(r["AverageNs"])/1e3,1)).rjust(9))
PY
