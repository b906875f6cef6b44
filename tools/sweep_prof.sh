#!/bin/bash
# Dev tool (GPU box): kernel-only durations of tools/gemm_sweep.py under rocprofv3.  usage: bash tools/sweep_prof.sh <tag>   (env passes through)
tag=$1
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $root/gpurun_out/sw_$tag -o sw -- python3 $root/tools/gemm_sweep.py 20 > $root/gpurun_out/sw_$tag.log 2>&1 || exit 1
db=$(find $root/gpurun_out/sw_$tag -name "*.db" | head -1)
python3 $root/tools/gemm_sweep_report.py $db 20 > $root/gpurun_out/sw_$tag.txt
