"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-launch HBM traffic of the
dominant kernel class.  gfx950 corrections per MI355X_MICROARCH.md §HBM: counters are in KiB;
FETCH_SIZE reads exactly half of a wide (16 B/lane) coalesced stream, so it is doubled; WRITE_SIZE is exact.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <kernel-substring> <out.json> [config]
The output records the source hash of the kernels (bench.source_id): bench.py quotes it as roofline.traffic only on the same build.
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(d, counter, sub):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and sub in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n


def whole_step(fd, wd, step_marker="novograd_update_kernel"):
    """every kernel of the run: (2 x FETCH_SIZE + WRITE_SIZE) KiB summed per kernel name and normalised per training step - the
    number of dispatches of `step_marker`, which runs exactly once per executed step (warm-up and roofline steps included, so the
    normalisation is over all of them: every step moves the same bytes).  The 2x FETCH correction is calibrated for wide (16 B per
    lane) coalesced loads only (MI355X_MICROARCH.md, HBM): the step's streaming kernels all load that way; the few narrow-load
    kernels (lattice, LSTM, small SE kernels) move < 2 % of the bytes.  Infinity-Cache hits are counted by these counters too (same
    section): this is traffic at the L2's memory side, an upper bound of the HBM bytes."""
    def table(d, counter):
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
        t = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].split("(")[0]
                e = t.setdefault(k, [0.0, 0])
                e[0] += float(r["Counter_Value"]); e[1] += 1
        return t
    ft, wt = table(fd, "FETCH_SIZE"), table(wd, "WRITE_SIZE")
    steps_f = sum(v[1] for k, v in ft.items() if step_marker in k)
    steps_w = sum(v[1] for k, v in wt.items() if step_marker in k)
    rows = []
    for k in sorted(set(ft) | set(wt)):
        f_ = ft.get(k, [0.0, 0]); w_ = wt.get(k, [0.0, 0])
        rows.append({"kernel": k, "launches_per_step": f_[1] / max(steps_f, 1), "fetch_mb_per_step": 2.0 * f_[0] * 1024 / 1e6 / max(steps_f, 1),
                     "write_mb_per_step": w_[0] * 1024 / 1e6 / max(steps_w, 1)})
    rows.sort(key=lambda r: -(r["fetch_mb_per_step"] + r["write_mb_per_step"]))
    tot_f = sum(r["fetch_mb_per_step"] for r in rows); tot_w = sum(r["write_mb_per_step"] for r in rows)
    return {"steps_in_fetch_pass": steps_f, "steps_in_write_pass": steps_w, "fetch_mb_per_step": tot_f, "write_mb_per_step": tot_w,
            "traffic_mb_per_step": tot_f + tot_w, "kernels": rows[:40]}


def main():
    fd, wd, sub, out = sys.argv[1:5]
    fetch, nf = per_kernel(fd, "FETCH_SIZE", sub)
    write, nw = per_kernel(wd, "WRITE_SIZE", sub)
    import bench
    res = {"kernel": sub, "config": sys.argv[5] if len(sys.argv) > 5 else None, "source_id": bench.source_id(), "launches_fetch_pass": nf, "launches_write_pass": nw,
           "fetch_kib_raw_per_launch": fetch / max(nf, 1), "write_kib_per_launch": write / max(nw, 1),
           "hbm_bytes_per_launch": (2.0 * fetch / max(nf, 1) + write / max(nw, 1)) * 1024.0,
           "note": "FETCH_SIZE doubled (gfx950 wide-load correction), KiB -> bytes"}
    json.dump(res, open(out, "w"), indent=1)
    print(res)
    if len(sys.argv) > 6:           # whole-step table -> <step_out.json> (bench.py: roofline.step_traffic)
        st = whole_step(fd, wd)
        st.update({"config": res["config"], "source_id": res["source_id"],
                   "note": "all kernels of the step: 2 x FETCH_SIZE + WRITE_SIZE (KiB -> MB), memory-side requests of the L2s (Infinity-Cache hits included)"})
        json.dump(st, open(sys.argv[6], "w"), indent=1)
        print({k: v for k, v in st.items() if k != "kernels"})


if __name__ == "__main__":
    main()
