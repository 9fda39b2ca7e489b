for i in 1 2 3 4 5 6; do
  ( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -2 | head -1; sleep 0.25; done ) > gpurun_out/clk_$i.txt &
  SP=$!
  python bench.py --seeds 2000 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/clk.json 2>/dev/null
  kill $SP
  python - <<PY
import json,re
d=json.loads([l for l in open("gpurun_out/clk.json") if l.startswith("{")][-1])
rows=[l.strip() for l in open("gpurun_out/clk_$i.txt") if l.strip()]
print("run $i trace %.1f ms; %d samples; last: %s" % (d["stages_ms"]["trace_replay_gather_ms"], len(rows), rows[-3] if len(rows)>3 else rows))
PY
done
rocm-smi --showclocks --showpower --csv 2>/dev/null | head -3
