#!/bin/bash
# dev: arxiv_sage (and arxiv) fit time against the number of workgroups of paths_fused_kernel: fewer than one per CU leaves
# CUs to the side stream's decomposition kernels
for wl in arxiv_sage arxiv; do
  for n in 256 252 248 240; do
    LGNN_FUSED_WGS=$n python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', $n, 'fit', round(d['ms_per_step'], 2), 'median', round(d['ms_per_step_median'], 2), 'accumulate', round(d['accumulate_ms'], 2), 'kernel', round(d['roofline']['avg_launch_ms'], 3))"
  done
done
