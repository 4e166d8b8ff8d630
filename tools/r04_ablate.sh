#!/bin/bash
# usage (inside gpurun): tools/r04_ablate.sh <tag> <debug values...>  -- arxiv bench per LGNN_FUSED_DEBUG value (timing only)
tag=$1; shift
mkdir -p gpurun_out
for d in "$@"; do
  LGNN_FUSED_DEBUG=$d timeout -k 10 300 python bench.py --workload arxiv --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_dbg$d.log 2>&1 || { tail -5 gpurun_out/${tag}_dbg$d.log; exit 1; }
  tail -1 gpurun_out/${tag}_dbg$d.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug $d ms/step',round(d['ms_per_step'],2),'fused avg ms',round(d['roofline']['avg_launch_ms'],3))"
done
