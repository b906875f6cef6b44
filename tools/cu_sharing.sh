#!/bin/bash
# Dev tool (GPU box): what does sharing CUs with RCCL's channel kernels cost the staged cfg2 step?  ONE GPU, the N > 1 form of the step
# (LASR_FORCE_OVERLAP=1: staged backward, bucket all-reduces on the library's side stream, captured into the hipGraph) over the test
# stand-in for librccl in its CU-holding mode: every bucket's all-reduce = n workgroups that each hold a CU for the bucket's wire time.
# usage: bash tools/cu_sharing.sh <out.json> [steps]   -> one JSON line per n in {0 (no collective kernel), 4, 8, 16, 32, 64}
out=$1; steps=${2:-100}
root=${GRAFT_REPO_ROOT:-$PWD}
: > $out
for buckets in 2 1; do
for n in 0 4 8 16 32 64; do
  LASR_FORCE_OVERLAP=1 LASR_DP_BUCKETS=$buckets LASR_RCCL_PATH=$root/tests/stub_rccl/libstubrccl.so LASR_STUB_HOLD_CUS=$n \
    python $root/bench.py --steps $steps --warmup 10 --no-cpu-baseline 2> /tmp/cu_sharing.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'buckets': $buckets, 'held_cus': $n, 'ms_per_step': d['ms_per_step'], 'hip_graph': d['config']['hip_graph'], 'staged': d['config']['staged_backward'], 'comm': d.get('comm')}))" >> $out || { tail -5 /tmp/cu_sharing.err; exit 1; }
done
done
# the unstaged one-GPU step of the same build, same box
python $root/bench.py --steps $steps --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'buckets': 0, 'held_cus': 0, 'ms_per_step': d['ms_per_step'], 'unstaged_single_gpu_step': True}))" >> $out
cat $out
