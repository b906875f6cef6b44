#!/bin/bash
# Dev tool (GPU box): kernel-trace profile of the staged / overlapped backward on a 1-rank nccl group
tag=$1
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export LASR_FORCE_OVERLAP=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $root/gpurun_out/prof_$tag.log 2>&1 || exit 1
db=$(find $root/gpurun_out/prof_$tag -name "*.db" | head -1)
python3 $root/tools/prof_summary.py $db 30 $root/gpurun_out/${tag}_kernel_stats.csv 40 > $root/gpurun_out/${tag}_kernels.txt
grep '"metric"' $root/gpurun_out/prof_$tag.log > $root/gpurun_out/${tag}_bench.json
