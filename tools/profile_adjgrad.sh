#!/bin/bash
# dev: wall time + rocprofv3 kernel stats of KronLaplace.neg_marglik_adj_grad at the arxiv shape (run through gpurun).
# usage: tools/profile_adjgrad.sh <tag>   -> gpurun_out/<tag>_adjgrad_{gcn,sage}.log, ..._kernel_stats.csv
set -e
tag=${1:-r03}
root=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
for kind in gcn sage; do
  timeout -k 10 300 python3 tools/time_adjgrad.py $kind > gpurun_out/${tag}_adjgrad_${kind}.log 2>&1
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_adjgrad_${kind}_prof -o run -- python3 $root/tools/time_adjgrad.py $kind > $root/gpurun_out/${tag}_adjgrad_${kind}_prof.log 2>&1)
  cp $(ls -t gpurun_out/${tag}_adjgrad_${kind}_prof/*/*kernel_stats.csv gpurun_out/${tag}_adjgrad_${kind}_prof/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${tag}_adjgrad_${kind}_kernel_stats.csv
  tail -3 gpurun_out/${tag}_adjgrad_${kind}.log
done
