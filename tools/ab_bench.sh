#!/bin/bash
# usage (inside gpurun): tools/ab_bench.sh <variant> [<variant> ...]   -- bench.py per lib_<variant> build ("lib" = default)
mkdir -p gpurun_out
for v in "$@"; do
  d=lib_$v; [ "$v" = lib ] && d=lib
  LGNN_LIB_DIR=$d timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1 || { echo "$v FAILED"; tail -5 gpurun_out/ab_$v.log; exit 1; }
  tail -1 gpurun_out/ab_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'value',round(d['value']),'ms/step',round(d['ms_per_step'],1),'fused ms',round(d['roofline']['avg_launch_ms'],3))"
done
