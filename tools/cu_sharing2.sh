#!/bin/bash
# Dev tool (GPU box): follow-up of tools/cu_sharing.sh - the same staged cfg2 step over the CU-holding stand-in, varying what the
# design can change: the GEMM tile width (LASR_GEMM_FORCE_NARROW), the LDS a channel workgroup claims (can a BN / depthwise workgroup
# still sit beside it?), and the wire bandwidth (a channel cap makes the window longer).
# usage: bash tools/cu_sharing2.sh <out.json> [steps]
out=$1; steps=${2:-100}
root=${GRAFT_REPO_ROOT:-$PWD}
: > $out
run() {  # label, env...
  label=$1; shift
  env LASR_FORCE_OVERLAP=1 LASR_DP_BUCKETS=2 LASR_RCCL_PATH=$root/tests/stub_rccl/libstubrccl.so "$@" \
    python $root/bench.py --steps $steps --warmup 10 --no-cpu-baseline 2> /tmp/cu2.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d.get('comm') or {}; print(json.dumps({'case': '$label', 'ms_per_step': d['ms_per_step'], 'exposed_wait_us': c.get('exposed_wait_us_per_step'), 'buckets': [(round(b['mb'],1), round(b['allreduce_us'],1)) for b in c.get('buckets', [])]}))" >> $out || { tail -5 /tmp/cu2.err; exit 1; }
}
run "hold0" LASR_STUB_HOLD_CUS=0
run "hold8_lds96" LASR_STUB_HOLD_CUS=8
run "hold8_lds96_narrow" LASR_STUB_HOLD_CUS=8 LASR_GEMM_FORCE_NARROW=1
run "hold0_narrow" LASR_STUB_HOLD_CUS=0 LASR_GEMM_FORCE_NARROW=1
run "hold8_lds16" LASR_STUB_HOLD_CUS=8 LASR_STUB_HOLD_LDS_KB=16
run "hold2_lds96" LASR_STUB_HOLD_CUS=2
run "hold8_lds96_wire40" LASR_STUB_HOLD_CUS=8 LASR_STUB_WIRE_GBS=40
run "hold8_lds96_wire170" LASR_STUB_HOLD_CUS=8 LASR_STUB_WIRE_GBS=170
run "hold32_lds96_wire170" LASR_STUB_HOLD_CUS=32 LASR_STUB_WIRE_GBS=170
cat $out
