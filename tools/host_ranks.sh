#!/bin/bash
# Dev tool (GPU box): the host side of `bench.py --path trainer` under N concurrent ranks SHARING the box's one GPU.
# The ranks are started by bench.py itself (lightning_asr_amd/launch.py: plain command, no outer launcher), meet over gloo
# (LASR_DIST_BACKEND) and exchange gradients through the library's communicator over the test stand-in for librccl.  The GPU is
# shared, so ms/step grows ~N x by construction - what is read off is the per-rank HOST side: CPU time of the enqueuing thread,
# time spent waiting for the ingest ring, ingest threads per rank.  The pool's process guard allows at most 6 processes on the card.
# usage: bash tools/host_ranks.sh <out.txt> <steps> "<N [ENV=val ...]>" ...
out=$1; steps=$2; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}
: > $out
for c in "$@"; do
  set -- $c; n=$1; shift; envs="$*"
  line=$(env LASR_DIST_BACKEND=gloo LASR_RCCL_PATH=$root/tests/stub_rccl/libstubrccl.so LASR_BENCH_CROP=${LASR_BENCH_CROP:-1} $envs \
         python $root/bench.py --path trainer --gpus $n --steps $steps --warmup 15 2>$root/gpurun_out/host_ranks_err.log | tail -1)
  echo "$line" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config']
rows = c.get('host_ms_per_step_by_rank') or [dict(c['host_ms_per_step'], rank=0, graph_steps=c['hip_graph_steps'], eager_steps=c['eager_steps'], ingest_threads=c['ingest_threads'])]
print('ranks=%d [%s] ms/step %.3f (shared GPU)  value %.0f audio-s/s  rung %s' % (d['n_gpus'], '$envs', d['ms_per_step'], d['value'], (d.get('launcher') or {}).get('rung')))
for r in rows:
    print('   rank %d: enqueue cpu %.3f ms/step, enqueue wall %.3f, waiting for ingest %.3f, ingest threads %s, graph/eager steps %s/%s' % (r['rank'], r['enqueue_cpu_time'], r['enqueuing_the_step'], r['waiting_for_ingest'], r['ingest_threads'], r['graph_steps'], r['eager_steps']))
" | tee -a $out
done
