#!/bin/bash
# Dev tool (GPU box, through gpurun): HBM traffic of the dominant kernel class (the 1x1-conv GEMMs) from two rocprofv3 --pmc
# passes of bench.py (FETCH_SIZE, WRITE_SIZE; separate passes, no trace domains - MI355X_MICROARCH.md, HBM section), written to
# profiles/gemm_traffic_<cfg>_<dtype>.json together with the source hash of the kernels it was measured on: bench.py quotes
# roofline.traffic only when that hash equals the running build's.
# usage: bash tools/pmc_pass.sh <cfg> [dtype]
cfg=${1:-cfg2}; dtype=${2:-bf16}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $root/gpurun_out/pmc_${cfg}_$c
  LASR_BENCH_GRAPH=0 rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/pmc_${cfg}_$c -o pmc -- python3 $root/bench.py --config $cfg --dtype $dtype --no-cpu-baseline --steps 6 --warmup 3 \
      > $root/gpurun_out/pmc_${cfg}_$c.log 2>&1 || { tail -5 $root/gpurun_out/pmc_${cfg}_$c.log; exit 1; }
done
mkdir -p $root/profiles
python3 $root/tools/pmc_traffic.py $root/gpurun_out/pmc_${cfg}_FETCH_SIZE $root/gpurun_out/pmc_${cfg}_WRITE_SIZE gemm_bf16 $root/gpurun_out/gemm_traffic_${cfg}_${dtype}.json $cfg $root/gpurun_out/step_traffic_${cfg}_${dtype}.json
rm -rf $root/gpurun_out/pmc_${cfg}_FETCH_SIZE $root/gpurun_out/pmc_${cfg}_WRITE_SIZE
