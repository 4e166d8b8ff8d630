#!/bin/bash
# usage: tools_gpu_cycle.sh <tag>   -- GPU tests (kfac subset), bench, rocprof kernel stats
tag=$1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu > gpurun_out/tests_$tag.log 2>&1
tail -4 gpurun_out/tests_$tag.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$tag.log 2>&1
tail -1 gpurun_out/bench_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value',round(d['value']),'ms/step',round(d['ms_per_step'],1),'roofline',d['roofline'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(r["Name"][:58].ljust(58), r["Calls"].rjust(5), str(round(float(r["TotalDurationNs"])/1e6,2)).rjust(9), str(round(float(r["AverageNs"])/1e3,1)).rjust(9), r["Percentage"])
PY
