# usage: sweep_full_loop.sh "max_split=48" "sums_deep_max=96" ...  one `bench.py --seeds 0` run (the reference's full trace loop) per argument
# (pnr_set_option keys through PNR_BENCH_OPTS; "-" = defaults), one line each appended to gpurun_out/sweep_full.log
for cfg in "$@"; do
  o=$cfg; [ "$cfg" = "-" ] && o=""
  PNR_BENCH_OPTS=$o timeout -k 10 300 python bench.py --seeds 0 --steps 2 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', 'step', round(d['ms_per_step'],1), 'trace', round(d['stages_ms']['trace_replay_gather_ms'],1), 'iters', d['counts']['iters'], 'launches', d['smc_launches_per_step'], 'nodes', d['counts']['nodes'])" >> gpurun_out/sweep_full.log || exit 1
done
