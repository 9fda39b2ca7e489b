# usage: sweep_env.sh "VAR=val VAR2=val" "VAR=val" ...   one bench run per argument, results appended to gpurun_out/sweep.log
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],1), 'ph_sample', round(d['roofline']['avg_launch_ms'],4), 'ph_sums', round(d['roofline_sums']['avg_launch_ms'],4), d['counts']['nodes'])" >> gpurun_out/sweep.log || exit 1
done
