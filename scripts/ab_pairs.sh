# many alternating pairs of two option sets on the bench workload (2000 seeds) -- the boxes have two modes ~4 % apart, so pairs:
#   bash scripts/ab_pairs.sh 6 "-" "small_threads=0"
n=$1; a=$2; b=$3
for i in $(seq 1 $n); do for o in "$a" "$b"; do
  oo=$o; [ "$o" = "-" ] && oo="trace_timing=0" 
  PNR_BENCH_OPTS=$oo python bench.py --seeds ${SEEDS:-2000} --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/abp.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/abp.json") if l.startswith("{")][-1])
print("$o", "trace %.1f"%d["stages_ms"]["trace_replay_gather_ms"], "iters", d["counts"]["iters"], flush=True)
PY
done; done
