#!/bin/bash
# PMC profile of the SMC trace kernel (run on the GPU box): counters in separate passes, csv output.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
S=${1:-256}; N=${2:-100}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" ; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-include-regex smc_trace --output-format csv -d $OUT/$tag -- python $GRAFT_REPO_ROOT/scripts/explore.py $S $N > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'smc_trace' in r.get('Kernel_Name',''):
            agg[r['Counter_Name']] += float(r['Counter_Value'])
for k in sorted(agg): print(f"{k:28s} {agg[k]:.4g}")
PY
