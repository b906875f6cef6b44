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


if __name__ == "__main__":
    main()
