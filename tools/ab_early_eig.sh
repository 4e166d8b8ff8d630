#!/bin/bash
# dev: arxiv_sage fit with / without the early decomposition of the 512 x 512 input factor on the side stream
for e in 0 1; do
  if [ $e = 1 ]; then export LGNN_NO_EARLY_EIG=1; fi
  python bench.py --workload arxiv_sage --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('no_early_eig=$e', 'fit', round(d['ms_per_step'], 2), 'median', round(d['ms_per_step_median'], 2), 'accumulate', round(d['accumulate_ms'], 2), 'kernel', round(d['roofline']['avg_launch_ms'], 3))"
done
