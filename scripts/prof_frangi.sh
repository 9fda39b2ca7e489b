#!/bin/bash
# Frangi kernel group under rocprofv3 (run on the GPU box through gpurun): per-kernel times (--kernel-trace --stats) and, in separate
# passes, the HBM bytes each kernel fetches / writes (FETCH_SIZE, WRITE_SIZE; calibrated on scripts/probes/fetch_calib as the guide's
# HBM section prescribes).   usage: bash scripts/prof_frangi.sh <tag> [size]   -> gpurun_out/frangi_<tag>/summary.txt
set -e
TAG=$1; SIZE=${2:-1024}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
[ -x $ROOT/scripts/probes/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $ROOT/scripts/probes/fetch_calib $ROOT/scripts/probes/fetch_calib.hip
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/frangi_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/scripts/frangi_bench.py $SIZE > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/calib_$ctr -- $ROOT/scripts/probes/fetch_calib > $OUT/calib_$ctr.log 2>&1
  rocprofv3 --pmc $ctr --kernel-include-regex "gauss|hessian|eigen_queue|j8_kernel|vdir" --output-format csv -d $OUT/pmc_$ctr -- python $ROOT/scripts/frangi_bench.py $SIZE 1 > $OUT/pmc_$ctr.log 2>&1
done
python - <<PY
import csv, glob, collections, re
OUT = "$OUT"
rows = []
for f in glob.glob(OUT + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
def load(pat):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r'(gauss_\w+?(?=E|I)|hessian_tile|eigen_queue|j8_kernel|vdir_points|\brd\b|\bwr\b)', r['Kernel_Name'])
            k = m.group(1) if m else r['Kernel_Name'][:40]
            agg[(k, r['Counter_Name'])][0] += float(r['Counter_Value']); agg[(k, r['Counter_Name'])][1] += 1
    return agg
cal = {}
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    for (k, c), (v, n) in load(OUT + "/calib_%s/**/*counter_collection.csv" % ctr).items():
        cal[(k.strip()[-2:], c)] = v
GiB = 2.0 ** 30
f_rd = GiB / (cal[('rd', 'FETCH_SIZE')] * 1024) if ('rd', 'FETCH_SIZE') in cal else 1.0
f_wr = GiB / (cal[('wr', 'WRITE_SIZE')] * 1024) if ('wr', 'WRITE_SIZE') in cal else 1.0
with open(OUT + "/summary.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python scripts/frangi_bench.py $SIZE   (3 Frangi passes over one $SIZE^3 stack, scales {2,4,6})\n")
    o.write("%-72s %8s %12s %12s %8s\n" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for r in rows:
        o.write("%-72s %8s %12.3f %12.3f %8s\n" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
    o.write("\n# HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, ONE Frangi pass); calibration true/(counter*1024): read %.4f write %.4f\n" % (f_rd, f_wr))
    for ctr, f in (('FETCH_SIZE', f_rd), ('WRITE_SIZE', f_wr)):
        for (k, c), (v, n) in sorted(load(OUT + "/pmc_%s/**/*counter_collection.csv" % ctr).items()):
            b = v * 1024 * f
            o.write("%-30s %-11s launches %4d  bytes %.4g  per launch %.4g\n" % (k, c, n, b, b / n))
    o.write("\n# script output\n" + open(OUT + "/trace.log").read()[-1500:])
print(open(OUT + "/summary.txt").read())
PY
