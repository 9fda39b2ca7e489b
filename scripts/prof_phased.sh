#!/bin/bash
# PMC profile of the phased SMC kernels (run on the GPU box): counters in separate passes, csv output.
# usage: bash scripts/prof_phased.sh [size] [nseeds]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_phased
rm -rf $OUT && mkdir -p $OUT
S=${1:-512}; N=${2:-600}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" ; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-include-regex "ph_sample|ph_sums" --output-format csv -d $OUT/$tag -- python $ROOT/scripts/sweep_batches.py $S $N 128:200:1024 > $OUT/$tag.log 2>&1 || tail -3 $OUT/$tag.log
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = 'ph_sample' if 'ph_sample' in r.get('Kernel_Name','') else ('ph_sums' if 'ph_sums' in r.get('Kernel_Name','') else None)
        if k: agg[(k, r['Counter_Name'])] += float(r['Counter_Value'])
for k in sorted(agg): print(f"{k[0]:10s} {k[1]:28s} {agg[k]:.5g}")
PY
