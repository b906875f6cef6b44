#!/bin/bash
# Dev tool (GPU box): LDS / VALU / wait counters per kernel for one bench config (separate --pmc passes, no trace domains),
# summarised by tools/pmc_summary.py into gpurun_out/<tag>_pmc.json (copy to profiles/).   usage: bash tools/pmc_lds.sh <tag> <cfg>
tag=$1; cfg=${2:-cfg2}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
dirs=""
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  d=$root/gpurun_out/pmcl_${tag}_$i
  rm -rf $d
  LASR_BENCH_GRAPH=0 rocprofv3 --pmc $set --output-format csv -d $d -o pmc -- python3 $root/bench.py --config $cfg --no-cpu-baseline --steps 4 --warmup 2 > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  dirs="$dirs $d"
  i=$((i+1))
done
python3 $root/tools/pmc_summary.py $dirs $root/gpurun_out/${tag}_pmc.json
for d in $dirs; do rm -rf $d; done
