#!/bin/bash
# usage (inside gpurun): tools/pmc_cora.sh <tag> -- SQ-level counter passes for the Cora-shaped diag fit (one rocprofv3 --pmc run per group)
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
n=0
for ctrs in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$n -- python3 $R/bench.py --workload cora --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc_$n.log 2>&1 || { echo "pass $n failed"; tail -3 /tmp/pmc_$n.log; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-60:]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$O/${tag}_cora_sq_counters.txt', 'w') as out:
    for k, c in agg.items():
        if 'diag' not in k and 'gemm_nt' not in k: continue
        out.write(k + '\n')
        for name, v in sorted(c.items()):
            out.write(f'   {name:28s} n={len(v):3d} avg={sum(v)/len(v):14.1f}\n')
print(open('$O/${tag}_cora_sq_counters.txt').read())
PY
