# usage: sweep_env.sh "window=512,groups=1" "look_pct=40" ...   one bench run per argument (pnr_set_option keys, passed through
# PNR_BENCH_OPTS), results appended to gpurun_out/sweep.log
for cfg in "$@"; do
  PNR_BENCH_OPTS=$cfg timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],1), 'ph_sample', round(d['roofline']['avg_launch_ms'],4), 'ph_sums', round(d['roofline_sums']['avg_launch_ms'],4), d['counts']['nodes'])" >> gpurun_out/sweep.log || exit 1
done
