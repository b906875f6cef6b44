#!/bin/bash
# Dev tool (GPU box): SQ counters of the 256 x 256-tile GEMM (tools/one_gemm.py: forward, dgrad and weight-gradient launch of a
# 16 032 x 512 x 512 layer): where the waves' cycles go (MI355X_MICROARCH.md, PMC slots: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES).
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/gemm_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 \
  --output-format csv -d $root/gpurun_out/gemm_sq -o sq -- python3 $root/tools/one_gemm.py > $root/gpurun_out/gemm_sq.log 2>&1 || { tail -5 $root/gpurun_out/gemm_sq.log; exit 1; }
python3 - $root/gpurun_out/gemm_sq <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in acc.items():
    if "gemm" not in k: continue
    w = c["SQ_WAVE_CYCLES"] or 1
    print("%s  (%d launches)" % (k, n[k]))
    print("   of the waves' cycles: parked (s_waitcnt / barrier) %.1f %%, issue stall %.1f %% (of which LDS issue %.1f %%), issuing %.1f %%"
          % (100 * c["SQ_WAIT_ANY"] / w, 100 * c["SQ_WAIT_INST_ANY"] / w, 100 * c["SQ_WAIT_INST_LDS"] / w, 100 * c["SQ_ACTIVE_INST_ANY"] / w))
    print("   MFMA pipe busy cycles / SQ busy cycles: %.3f   (raw: %s)" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c["SQ_BUSY_CYCLES"], 1), {a: int(b) for a, b in c.items()}))
PY
rm -rf $root/gpurun_out/gemm_sq
