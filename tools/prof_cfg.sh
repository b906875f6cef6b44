#!/bin/bash
# Dev tool (run on the GPU box through gpurun): kernel-trace profile of bench.py for one config.
# usage: bash tools/prof_cfg.sh <tag> <cfg> [steps]   -> gpurun_out/<tag>_kernels.txt, <tag>_kernel_stats.csv, <tag>_bench.json
tag=$1; cfg=$2; steps=${3:-20}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/bench.py --config $cfg --no-cpu-baseline --steps $steps --warmup 5 > $root/gpurun_out/prof_$tag.log 2>&1 || { tail -5 $root/gpurun_out/prof_$tag.log; exit 1; }
db=$(find $root/gpurun_out/prof_$tag -name "*.db" | head -1)
# the bench runs warmup + steps + instrumented steps: normalise per step by the total number of steps executed
python3 - "$root/gpurun_out/prof_$tag.log" > /tmp/nsteps.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        r = json.loads(l); print(r["steps"] + r["warmup"] + (8 if "cfg5" in r["config"]["name"] else max(2, min(r["steps"], 5))))
PY
n=$(cat /tmp/nsteps.txt)
python3 $root/tools/prof_summary.py $db $n $root/gpurun_out/${tag}_kernel_stats.csv 45 > $root/gpurun_out/${tag}_kernels.txt
grep '"metric"' $root/gpurun_out/prof_$tag.log > $root/gpurun_out/${tag}_bench.json
rm -rf $root/gpurun_out/prof_$tag
