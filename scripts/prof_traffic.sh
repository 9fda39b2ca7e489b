#!/bin/bash
# HBM traffic of the SMC and Frangi kernels from PMC counters (separate passes: FETCH_SIZE takes 3 TCC slots,
# WRITE_SIZE 2), calibrated on a known byte count in the same access pattern (MI355X_MICROARCH.md, HBM).
#   PNR_BENCH_OPTS=groups=1 bash scripts/prof_traffic.sh [bench.py arguments]   -> gpurun_out/traffic/traffic.json
# (one trace group: no two kernels overlap while the counters run).  The JSON carries the workload it was taken on and the hash of the
# kernel sources (pnr_amd.lib.kernel_source_hash): copy it to profiles/rNN_traffic_1024_s2000.json -- bench.py quotes it only while
# the sources it runs have that hash.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
[ -x $ROOT/scripts/probes/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $ROOT/scripts/probes/fetch_calib $ROOT/scripts/probes/fetch_calib.hip
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/traffic
rm -rf $OUT && mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/calib_$ctr -- $ROOT/scripts/probes/fetch_calib > $OUT/calib_$ctr.log 2>&1
  rocprofv3 --pmc $ctr --kernel-include-regex "smc_trace|ph_predict|ph_cube|ph_sample|ph_sums|ph_update|hessian_tile|eigen_queue|gauss|j8_kernel|layer_maxima" --output-format csv -d $OUT/bench_$ctr -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" > $OUT/bench_$ctr.json 2> $OUT/bench_$ctr.err
done
python - <<PY
import csv, glob, collections, json
def load(pat):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            import re
            m = re.search(r'(smc_trace|ph_predict|ph_cube|ph_sample|ph_sums|ph_update|hessian_tile|eigen_queue|gauss_xy_u8|gauss_x_u8|gauss_axis_t|gauss_axis|j8_kernel|layer_maxima|rd|wr|fillBuffer)', r['Kernel_Name'])
            k = m.group(1) if m else r['Kernel_Name'][:30]
            agg[(k, r['Counter_Name'])][0] += float(r['Counter_Value']); agg[(k, r['Counter_Name'])][1] += 1
    return agg
cal = {}
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    a = load("$OUT/calib_%s/**/*counter_collection.csv" % ctr)
    for (k, c), (v, n) in a.items():
        print('calib', k, c, 'KB-units', v, 'launches', n)
        cal[(k.strip()[-2:], c)] = v
GiB = 2.0 ** 30
f_rd = GiB / (cal[('rd', 'FETCH_SIZE')] * 1024) if ('rd', 'FETCH_SIZE') in cal else None
f_wr = GiB / (cal[('wr', 'WRITE_SIZE')] * 1024) if ('wr', 'WRITE_SIZE') in cal else None
print('calibration: true bytes / (counter*1024): read', f_rd, 'write', f_wr)
out = {'calibration': {'read': f_rd, 'write': f_wr}}
for ctr, f in (('FETCH_SIZE', f_rd), ('WRITE_SIZE', f_wr)):
    a = load("$OUT/bench_%s/**/*counter_collection.csv" % ctr)
    for (k, c), (v, n) in sorted(a.items()):
        b = v * 1024 * (f or 1.0)
        print('%-42s %-11s launches %4d  bytes(calibrated) %.4g  per launch %.4g' % (k, c, n, b, b / n))
        out.setdefault(k.strip(), {})[c] = {'launches': n, 'bytes': b, 'bytes_per_launch': b / n}
import os, sys
sys.path.insert(0, "$ROOT")
from pnr_amd import lib as pl
bj = None
for ln in open("$OUT/bench_FETCH_SIZE.json"):
    if ln.startswith("{"):
        bj = json.loads(ln)
out["workload"] = {"command": "PNR_BENCH_OPTS=%s python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra " % os.environ.get("PNR_BENCH_OPTS", "") + "$*",
                   "kernel_source_hash": pl.kernel_source_hash(), "driver": "phased",
                   "config": (bj or {}).get("config"), "smc_iterations": (bj or {}).get("counts", {}).get("iters"),
                   "smc_steps": (bj or {}).get("smc_launches_per_step"), "nodes": (bj or {}).get("counts", {}).get("nodes"),
                   "note": "counters collected under rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in separate passes), one bench step each"}
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(out["workload"]))
PY
