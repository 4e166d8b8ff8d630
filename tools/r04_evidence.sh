#!/bin/bash
# usage (inside gpurun): tools/r04_evidence.sh -- the round's side logs under gpurun_out/ (copied to profiles/ afterwards)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 300 python $R/tools/time_eigh.py 2>&1 | grep -v amdgpu.ids > $O/r04_time_eigh.log || exit 1
for a in "cora diag" "cora kron" "webkb diag" "webkb kron" "arxiv diag"; do
  timeout -k 10 300 python $R/tools/time_structure_step.py $a 2>&1 | grep -v amdgpu.ids >> $O/r04_structure_step.log || exit 1
done
timeout -k 10 300 python $R/bench.py --workload arxiv --emulate-world 8 --no-cpu-baseline 2>&1 | tail -1 > $O/r04_emulate_world8.json || exit 1
timeout -k 10 300 python $R/bench.py --workload cora --no-fit-graph 2>&1 | tail -1 > $O/r04_cora_nograph_bench.json || exit 1
timeout -k 10 600 python $R/bench.py --workload products --structure kron --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > $O/r04_products_kron_bench.json || exit 1
cat $O/r04_time_eigh.log | tail -6; cat $O/r04_structure_step.log | cut -c1-200; cut -c1-300 $O/r04_emulate_world8.json
