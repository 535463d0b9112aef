#!/bin/bash
# Does the TFD + FE-residual epoch depend on the bench's flags or on the box?  Three bench runs on ONE box.
cd "$GRAFT_REPO_ROOT"
for tag in drv1 def drv2; do
  if [ $tag = def ]; then a=""; else a="--steps 20 --warmup 5"; fi
  python bench.py $a > gpurun_out/phys_ab_$tag.json 2> gpurun_out/phys_ab_$tag.err
  python - gpurun_out/phys_ab_$tag.json $tag <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); se = r["surrogate_epochs"]
print(sys.argv[2], "ms/step", round(r["ms_per_step"], 5), "frac", round(r["roofline"]["frac"], 4),
      {k: round(v["epoch_s"], 5) for k, v in se.items() if isinstance(v, dict) and "epoch_s" in v})
PY
done
