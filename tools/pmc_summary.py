"""Per-kernel sums of rocprofv3 --pmc counters (counter_collection.csv) -> JSON.
    python tools/pmc_summary.py <dir> [<dir> ...] <out.json>"""
import csv, glob, json, sys
from collections import defaultdict
out = sys.argv[-1]
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:-1]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
res = {}
for k in acc:
    if not k.startswith("lasr::"): continue
    e = {c: acc[k][c] / max(calls[k][c], 1) for c in acc[k]}
    e["launches"] = max(calls[k].values())
    # (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE are kept raw: their per-SE / per-XCC aggregation in this CSV is not
    #  documented well enough to quote a utilisation from them)
    if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        e["lds_conflict_pct_of_lds_cycles"] = 100.0 * e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
    res[k] = e
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["launches"])[:14]:
    print("%-48s launches %4d  lds conflicts %5.1f %% of LDS cycles" % (k[:48], e["launches"], e.get("lds_conflict_pct_of_lds_cycles", float("nan"))))
