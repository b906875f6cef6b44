#!/bin/bash
# Dev tool (GPU box, through gpurun): the round's final measurement set for one tag - bench lines (step + trainer) and eager kernel traces
# of cfg2 / cfg4 / cfg5 on the build in the tree.  usage: bash tools/final_set.sh <tag>   -> gpurun_out/<tag>_*
tag=${1:-final}
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root && mkdir -p gpurun_out
for cfg in cfg2 cfg4 cfg5; do
  extra=""; [ $cfg = cfg2 ] || extra="--no-cpu-baseline"
  python bench.py --config $cfg --steps 200 --warmup 20 $extra > gpurun_out/${tag}_${cfg}_step.json 2> gpurun_out/${tag}_${cfg}_step.err || { tail -n 5 gpurun_out/${tag}_${cfg}_step.err; exit 1; }
  python bench.py --config $cfg --path trainer --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${tag}_${cfg}_trainer.json 2> gpurun_out/${tag}_${cfg}_trainer.err || { tail -n 5 gpurun_out/${tag}_${cfg}_trainer.err; exit 1; }
  echo "$cfg bench lines done"
done
for cfg in cfg2 cfg4 cfg5; do
  bash tools/prof_cfg.sh ${tag}_${cfg}_bf16 $cfg 30 || exit 1
  echo "$cfg trace done"
done
cd $root
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for cfg in ("cfg2", "cfg4", "cfg5"):
    for kind in ("step", "trainer"):
        d = json.loads(open(f"gpurun_out/{tag}_{cfg}_{kind}.json").read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        print(cfg, kind, round(d["ms_per_step"], 4), round(d["value"], 1), "frac", round(r.get("frac", 0), 4), "step_frac", round(r.get("step_frac", 0), 4),
              "traffic", r.get("traffic"))
PY
