#!/bin/bash
# per-kernel times of one Frangi pass (rocprofv3 --kernel-trace --stats), our kernels only.  usage: bash scripts/prof_frangi_kernels.sh [size]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/frk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/frk -- python $ROOT/scripts/frangi_bench.py ${1:-1024} 2 > /tmp/frk.log 2>&1 || tail -5 /tmp/frk.log
python3 - <<'PY'
import csv, glob, re
for f in glob.glob("/tmp/frk/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(r"gauss|hessian|eigen_queue|j8_kernel|layer_|vdir", r["Name"]):
            nm = re.sub(r"\(.*", "", r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))
            print("%-28s calls %4s  avg %8.3f ms  total %8.3f ms" % (nm, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
