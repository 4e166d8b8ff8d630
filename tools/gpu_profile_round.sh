#!/bin/bash
# usage (inside gpurun): tools/gpu_profile_round.sh <tag>  -- everything profiles/<tag>_* is made of: bench line with CPU
# baseline, rocprofv3 kernel stats, PMC passes (one rocprofv3 --pmc run per counter group; never combined with traces)
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
timeout -k 10 400 python $R/bench.py --steps 3 --warmup 1 > $O/${tag}_bench.log 2>&1 || { tail -5 $O/${tag}_bench.log; exit 1; }
tail -1 $O/${tag}_bench.log > $O/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${tag}_prof.log 2>&1 || exit 1
cp $O/${tag}_prof/*/*kernel_stats.csv $O/${tag}_kernel_stats.csv
n=0; dirs=""
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $O/${tag}_pmc_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/${tag}_pmc_$n.log 2>&1 || { echo "pmc pass $n failed"; tail -3 $O/${tag}_pmc_$n.log; }
  dirs="$dirs $O/${tag}_pmc_$n"
done
python3 $R/tools/pmc_aggregate.py $O/$tag $dirs
cat $O/${tag}_bench.json
