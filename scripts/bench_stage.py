"""print the stage / kernel-group times of one bench.py line (helper for tuning runs on the GPU box)"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step %.1f" % d["ms_per_step"], {k: round(v, 1) for k, v in d["stages_ms"].items()},
      {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()})
