#!/bin/bash
# usage (inside gpurun): tools/r04_cora.sh <tag> [LGNN_DIAG_SLAB values...]  -- diag tests, then the Cora bench per slab length
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_diag_small.py tests/test_gpu_baseline_shapes.py -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -3 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
for sl in default "$@"; do
  if [ "$sl" = default ]; then unset LGNN_DIAG_SLAB; else export LGNN_DIAG_SLAB=$sl; fi
  timeout -k 10 300 python bench.py --workload cora --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_cora_$sl.log 2>&1 || { tail -5 gpurun_out/${tag}_cora_$sl.log; exit 1; }
  tail -1 gpurun_out/${tag}_cora_$sl.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('slab $sl ms/step',round(d['ms_per_step'],4),'median',round(d['ms_per_step_median'],4),'kernel ms',round(d['roofline']['avg_launch_ms'],4))"
done
unset LGNN_DIAG_SLAB
LGNN_DIAG_ATOMIC=1 timeout -k 10 300 python bench.py --workload cora --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-3 kernel ms/step',round(d['ms_per_step'],4),'median',round(d['ms_per_step_median'],4),'kernel ms',round(d['roofline']['avg_launch_ms'],4))"
