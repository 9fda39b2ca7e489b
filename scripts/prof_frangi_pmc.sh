#!/bin/bash
# PMC breakdown of the Frangi kernels (run on the GPU box): where do hessian_tile and the Gaussian passes spend their cycles -- VALU issue,
# LDS, memory?  Counters in separate passes (rocprofv3 --pmc only; never combined with a trace), one Frangi pass over one stack each.
# usage: bash scripts/prof_frangi_pmc.sh <tag> [size]   -> gpurun_out/frangi_pmc_<tag>/summary.txt
set -e
TAG=$1; SIZE=${2:-1024}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/frangi_pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" ; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-include-regex "gauss|hessian|eigen_queue|j8_kernel" --output-format csv -d $OUT/$tag -- python $ROOT/scripts/frangi_bench.py $SIZE 1 > $OUT/$tag.log 2>&1 || tail -3 $OUT/$tag.log
done
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(float)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(gauss_[a-z0-9_]+?_tILi[0-9]+E|gauss_[a-z0-9_]+|hessian_tile|eigen_queue|j8_kernel)', r.get('Kernel_Name', ''))
        if m: agg[(m.group(1), r['Counter_Name'])] += float(r['Counter_Value'])
with open("$OUT/summary.txt", "w") as o:
    o.write("# bash scripts/prof_frangi_pmc.sh $TAG $SIZE: rocprofv3 --pmc <group> (three groups, separate passes), ONE Frangi pass over one $SIZE^3 stack, scales {2,4,6}; sums over all launches of a kernel\n")
    for k in sorted(agg): o.write(f"{k[0]:28s} {k[1]:28s} {agg[k]:.5g}\n")
print(open("$OUT/summary.txt").read())
PY
