# A/B of scheduler / kernel options on the two bench workloads (2000 sorted seeds; --seeds 0 = the reference's full trace loop):
#   bash scripts/ab_opts.sh "cut_words=1" "cut_words=0" ...   -> one line per option set and workload (pnr_set_option keys; "-" = defaults)
for o in "$@"; do
  oo=$o; [ "$o" = "-" ] && oo="trace_timing=1" || oo="trace_timing=1,$o"
  for seeds in 2000 0; do
    PNR_BENCH_OPTS=$oo python bench.py --seeds $seeds --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/ab_${o}_${seeds}.json 2> gpurun_out/ab_${o}_${seeds}.err || exit 1
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab_${o}_${seeds}.json") if l.startswith("{")][-1])
print("$o seeds=$seeds", "ms/step %.1f"%d["ms_per_step"], "trace %.1f"%d["stages_ms"]["trace_replay_gather_ms"], "iters", d["counts"]["iters"], "nodes", d["counts"]["nodes"], "traces", d["counts"]["traces_used"], flush=True)
PY
  done
done
