# A/B of library builds on the two bench workloads, alternating on one box:
#   bash scripts/ab_libs.sh 2 "" _variant ...   (library suffixes: pnr_amd/libpnr_hip<suffix>.so; first argument = repetitions)
R=${GRAFT_REPO_ROOT:-$(pwd)}
reps=$1; shift
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    for seeds in 2000 0; do
      PNR_LIB_DIAG=$R/pnr_amd/libpnr_hip$v.so PNR_BENCH_OPTS=trace_timing=1 python bench.py --seeds $seeds --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/abl.json 2> gpurun_out/abl.err || { tail -3 gpurun_out/abl.err; exit 1; }
      python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/abl.json") if l.startswith("{")][-1])
print("lib'$v' seeds=$seeds", "ms/step %.1f"%d["ms_per_step"], "trace %.1f"%d["stages_ms"]["trace_replay_gather_ms"], "frangi %.1f"%d["stages_ms"]["frangi_ms"], "iters", d["counts"]["iters"], "nodes", d["counts"]["nodes"], flush=True)
PY
    done
  done
done
