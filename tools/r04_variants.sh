#!/bin/bash
# usage (inside gpurun): tools/r04_variants.sh <tag> <debug> <lib dirs...>  -- arxiv bench per library variant at one LGNN_FUSED_DEBUG
tag=$1; dbg=$2; shift; shift
mkdir -p gpurun_out
for v in "$@"; do
  LGNN_LIB_DIR=$v LGNN_FUSED_DEBUG=$dbg timeout -k 10 300 python bench.py --workload arxiv --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_$v.log 2>&1 || { tail -5 gpurun_out/${tag}_$v.log; exit 1; }
  tail -1 gpurun_out/${tag}_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v debug $dbg ms/step',round(d['ms_per_step'],2),'fused avg ms',round(d['roofline']['avg_launch_ms'],3))"
done
