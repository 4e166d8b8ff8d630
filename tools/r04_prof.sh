#!/bin/bash
# usage (inside gpurun): tools/r04_prof.sh <tag> <workload> [steps]  -- rocprofv3 kernel stats of bench.py -> gpurun_out/<tag>_<workload>_kernel_stats.csv
tag=$1; wl=$2; steps=${3:-5}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_${wl}_prof -- python3 $R/bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline > $O/${tag}_${wl}_prof.log 2>&1 || { tail -5 $O/${tag}_${wl}_prof.log; exit 1; }
cp $O/${tag}_${wl}_prof/*/*kernel_stats.csv $O/${tag}_${wl}_kernel_stats.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/${tag}_${wl}_kernel_stats.csv")))[:16]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), str(round(float(r["TotalDurationNs"])/1e6,3)).rjust(9), str(round(float(r["AverageNs"])/1e3,1)).rjust(9), r["Percentage"])
PY
