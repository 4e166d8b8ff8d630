#!/bin/bash
# usage (inside gpurun): tools/bench_all.sh <tag> -- the bench line (with CPU baseline) of every workload -> gpurun_out/<tag>_<workload>_bench.json
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for wl in arxiv cora products arxiv_sage arxiv_powerlaw; do
  timeout -k 10 900 python $R/bench.py --workload $wl --steps 10 --warmup 2 > $O/${tag}_${wl}_bench.log 2>&1 || { tail -3 $O/${tag}_${wl}_bench.log; exit 1; }
  tail -1 $O/${tag}_${wl}_bench.log > $O/${tag}_${wl}_bench.json
  cut -c1-220 $O/${tag}_${wl}_bench.json
done
