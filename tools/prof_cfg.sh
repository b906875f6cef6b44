#!/bin/bash
# Dev tool (run on the GPU box through gpurun): kernel-trace profile of bench.py for one config.
# usage: bash tools/prof_cfg.sh <tag> <cfg> [steps] [extra bench args]  -> gpurun_out/<tag>_kernels.txt, <tag>_kernel_stats.csv, <tag>_bench.json
# Eager launches (LASR_BENCH_GRAPH=0: a graph capture adds un-timed warm-up bodies to the trace) and the per-step normalisation
# taken from the trace itself: the number of calls of novograd_update_kernel, which runs exactly once per executed step.
tag=$1; cfg=$2; steps=${3:-20}; shift $(( $# < 3 ? $# : 3 ))
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
LASR_BENCH_GRAPH=0 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/bench.py --config $cfg --no-cpu-baseline --steps $steps --warmup 5 "$@" > $root/gpurun_out/prof_$tag.log 2>&1 || { tail -5 $root/gpurun_out/prof_$tag.log; exit 1; }
db=$(find $root/gpurun_out/prof_$tag -name "*.db" | head -1)
python3 $root/tools/prof_summary.py $db auto:novograd_update_kernel $root/gpurun_out/${tag}_kernel_stats.csv 45 > $root/gpurun_out/${tag}_kernels.txt
grep '"metric"' $root/gpurun_out/prof_$tag.log > $root/gpurun_out/${tag}_bench.json
rm -rf $root/gpurun_out/prof_$tag
