#!/bin/bash
# usage (inside gpurun): tools/r04_cycle.sh <tag> [pytest files...]  -- path-route tests, arxiv + arxiv_sage bench, DEV phase report
tag=$1; shift
tests=${@:-tests/test_gpu_paths.py}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest $tests -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -5 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
for wl in arxiv arxiv_sage; do
  timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_${wl}_bench.log 2>&1 || { tail -5 gpurun_out/${tag}_${wl}_bench.log; exit 1; }
  tail -1 gpurun_out/${tag}_${wl}_bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'roofline',d['roofline'])"
done
LGNN_LIB_DIR=lib_dev LGNN_PHASE_REPORT=1 timeout -k 10 300 python tools/phase_report.py arxiv 2 > gpurun_out/${tag}_phase.log 2>&1
tail -12 gpurun_out/${tag}_phase.log
