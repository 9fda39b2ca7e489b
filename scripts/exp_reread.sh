#!/bin/bash
# VERDICT r03 item 1, the cost side: the sampling kernel re-reading and adding up every item it has stored (no hand-over), against the
# product.  Needs the variant libraries (make -C pnr_amd/csrc variant NAME=reread1 DEFS="-DPNR_EXPERIMENT_HOOKS -DPNR_EXP_REREAD=1",
# NAME=reread2 ... =2).  usage: bash scripts/exp_reread.sh   -> gpurun_out/exp_reread.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp_reread.txt
: > $OUT
run() { # label, library, options
  PNR_LIB_DIAG=$2 PNR_BENCH_OPTS=$3 timeout -k 10 300 python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']; n = d['smc_launches_per_step']
print('%-34s step %7.1f ms  tracing %7.1f ms  per launch: ph_sample %.4f  ph_sums %.4f  predict+update %.4f ms  (%d launches, %d iterations, %d nodes)' % ('$1', d['ms_per_step'], d['stages_ms']['trace_replay_gather_ms'], k['smc'] / n, k['smc_sums'] / n, (k['smc_predict'] + k['smc_update']) / n, n, d['counts']['iters'], d['counts']['nodes']))" >> $OUT
}
for g in 2 1; do
  run "product, groups=$g" "" "groups=$g"
  run "re-read, nt stores, groups=$g" $ROOT/pnr_amd/libpnr_hip_reread1.so "groups=$g"
  run "re-read, plain stores, groups=$g" $ROOT/pnr_amd/libpnr_hip_reread2.so "groups=$g"
done
# where the re-read is served from: L2 hits / misses of the sampling kernel (one trace group, one bench step each)
cd /tmp && export TMPDIR=/tmp
for v in "" reread1 reread2; do
  lib=""; [ -n "$v" ] && lib=$ROOT/pnr_amd/libpnr_hip_$v.so
  rm -rf /tmp/rr_$v
  PNR_LIB_DIAG=$lib PNR_BENCH_OPTS=groups=1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-include-regex "ph_sample|ph_sums" --output-format csv -d /tmp/rr_$v -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2>&1
  python3 - <<PY >> $OUT
import csv, glob, collections
a = collections.defaultdict(float)
for f in glob.glob("/tmp/rr_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = 'ph_sample' if 'ph_sample' in r['Kernel_Name'] else 'ph_sums'
        a[(k, r['Counter_Name'])] += float(r['Counter_Value'])
for k in ('ph_sample', 'ph_sums'):
    h, m, q = a[(k, 'TCC_HIT_sum')], a[(k, 'TCC_MISS_sum')], a[(k, 'TCC_REQ_sum')]
    print('PMC %-10s %-8s TCC_REQ %.4g  TCC_HIT %.4g  TCC_MISS %.4g  hit rate %.3f' % ('${v:-product}', k, q, h, m, h / max(h + m, 1)))
PY
done
cat $OUT
