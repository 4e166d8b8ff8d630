#!/bin/bash
# usage (inside gpurun): tools/gpu_profile_r04.sh <tag> <workload> [steps]
# Everything profiles/<tag>_<workload>_* is made of: the bench line with CPU baseline, the rocprofv3 kernel stats of the
# same command, and the PMC passes (one rocprofv3 --pmc run per counter group; never combined with traces).
tag=$1; wl=$2; steps=${3:-10}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; P=${tag}_${wl}
case $wl in
  arxiv) kern=paths_fused_kernel; units=40 ;;
  arxiv_powerlaw) kern=spmm_gram256_kernel; units=40 ;;   # hub-heavy graph: full batches keep the class planes (paths_pay)
  arxiv_sage) kern=paths_fused_kernel; units=40 ;;   # round 3: one-hop path route
  cora) kern=diag_first_layer_tile_kernel; units=1299 ;;
  products) kern=gram_mem_kernel; units=1128 ;;
esac
mkdir -p $O
timeout -k 10 900 python $R/bench.py --workload $wl --steps $steps --warmup 2 > $O/${P}_bench.log 2>&1 || { tail -5 $O/${P}_bench.log; exit 1; }
tail -1 $O/${P}_bench.log > $O/${P}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_prof -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $O/${P}_prof.log 2>&1 || exit 1
cp $O/${P}_prof/*/*kernel_stats.csv $O/${P}_kernel_stats.csv
n=0; dirs=""
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  n=$((n+1))
  timeout -k 10 600 rocprofv3 --pmc $ctrs --output-format csv -d $O/${P}_pmc_$n -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline > $O/${P}_pmc_$n.log 2>&1 || { echo "pmc pass $n failed"; tail -3 $O/${P}_pmc_$n.log; }
  dirs="$dirs $O/${P}_pmc_$n"
done
LGNN_PMC_KERNEL=$kern LGNN_PMC_WORKLOAD=$wl LGNN_PLANES_PER_LAUNCH=$units python3 $R/tools/pmc_aggregate.py $O/$P $dirs
cat $O/${P}_bench.json
