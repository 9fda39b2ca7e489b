cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/phase_$i; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err
  python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
print("run $i trace %.1f ms" % d["stages_ms"]["trace_replay_gather_ms"])
PY
  python $GRAFT_REPO_ROOT/scripts/phase_overlap.py $OUT "run $i"
done
