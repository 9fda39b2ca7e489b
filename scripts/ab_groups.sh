for rep in 1 2; do
for cfg in "4:-" "8:-" "8:groups=3" "8:groups=3,target=150" "8:groups=4,target=160" "8:groups=3,target=120,poll=3"; do
  q=${cfg%%:*}; o=${cfg#*:}
  oo=$o; [ "$o" = "-" ] && oo="trace_timing=1" || oo="trace_timing=1,$o"
  GPU_MAX_HW_QUEUES=$q PNR_BENCH_OPTS=$oo python bench.py --seeds 2000 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/abg.json 2> gpurun_out/abg.err || { tail -3 gpurun_out/abg.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/abg.json") if l.startswith("{")][-1])
print("queues $q opts $o", "ms/step %.1f"%d["ms_per_step"], "trace %.1f"%d["stages_ms"]["trace_replay_gather_ms"], "iters", d["counts"]["iters"], "nodes", d["counts"]["nodes"], flush=True)
PY
done; done
