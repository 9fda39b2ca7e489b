for o in "" ",small_threads=256,small_max=64" ",small_threads=512,small_max=64" ",small_threads=256,small_max=64,small_x10=20" ",small_threads=512,small_max=64,small_x10=20"; do
for t in 4 12 24 40; do PNR_BENCH_OPTS=groups=1,target=$t,profile_every=1$o python bench.py --seeds 120 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/tiny.json 2>/dev/null; python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/tiny.json") if l.startswith("{")][-1])
n=d["smc_launches_per_step"]; k=d["kernel_ms_per_step"]
print("opts '$o' target $t: trace %.1f ms, %.3f ms per step; sample %.3f sums %.3f nodes %d" % (d["stages_ms"]["trace_replay_gather_ms"], d["stages_ms"]["trace_replay_gather_ms"]/max(n,1), k["smc"]/n, k["smc_sums"]/n, d["counts"]["nodes"]))
PY
done; done
