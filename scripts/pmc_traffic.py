#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, CSV output) into the
per-launch traffic record bench.py reports next to a timing: profiles/<name>.json with the hash
of the kernel sources the measured library was built from (bench.py only uses a record whose
hash matches the sources beside it).

  python scripts/pmc_traffic.py <fetch_dir> <write_dir> <kernel-name-substring>[|<second>...] <out.json> \
         --sources kgat_spmm.hip,kgat_common.h --workload "..." --command "..." [--algorithmic BYTES]
"""
import argparse
import csv
import glob
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(directory, counter, needles):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under %s" % directory
    out = {n: [] for n in needles}
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                for n in needles:
                    if n in row["Kernel_Name"]:
                        out[n].append(float(row["Counter_Value"]))
    return {n: (float(np.median(v)) if v else None, len(v)) for n, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("kernels")
    ap.add_argument("out")
    ap.add_argument("--sources", required=True)
    ap.add_argument("--workload", default="")
    ap.add_argument("--command", default="")
    ap.add_argument("--algorithmic", type=int, default=None)
    a = ap.parse_args()
    needles = a.kernels.split("|")
    fetch = per_launch(a.fetch_dir, "FETCH_SIZE", needles)
    write = per_launch(a.write_dir, "WRITE_SIZE", needles)
    h = hashlib.sha256()
    for nm in a.sources.split(","):
        with open(os.path.join(ROOT, "dgl-kgat_amd", "csrc", nm), "rb") as fh:
            h.update(fh.read())
    rec = {"kernels": needles, "workload": a.workload, "command": a.command,
           "kernel_source_sha16": h.hexdigest()[:16], "kernel_sources": a.sources.split(","),
           "FETCH_SIZE_KB_per_launch_raw": {n: fetch[n][0] for n in needles},
           "WRITE_SIZE_KB_per_launch": {n: write[n][0] for n in needles},
           "launches_seen": {n: [fetch[n][1], write[n][1]] for n in needles},
           "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for 16-B-per-lane loads -> doubled "
                         "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; counters are the L2's fabric-side "
                         "requests (Infinity-Cache hits included): an upper bound on HBM bytes"}
    total = 0.0
    for n in needles:
        total += 2.0 * (fetch[n][0] or 0.0) * 1024 + (write[n][0] or 0.0) * 1024
    rec["traffic_bytes_per_launch"] = int(total)
    if a.algorithmic:
        rec["algorithmic_bytes_per_launch"] = a.algorithmic
        rec["traffic_over_algorithmic"] = round(total / a.algorithmic, 4)
    with open(a.out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
