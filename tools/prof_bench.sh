#!/bin/bash
# Dev tool (run on the GPU box through gpurun): kernel-trace profile of bench.py, summary to gpurun_out/<tag>_kernels.txt
# usage: bash tools/prof_bench.sh <tag> [bench args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $root/gpurun_out/prof_$tag.log 2>&1 || exit 1
db=$(find $root/gpurun_out/prof_$tag -name "*.db" | head -1)
python3 $root/tools/prof_summary.py $db 30 $root/gpurun_out/${tag}_kernel_stats.csv 40 > $root/gpurun_out/${tag}_kernels.txt
grep '"metric"' $root/gpurun_out/prof_$tag.log > $root/gpurun_out/${tag}_bench.json
