#!/usr/bin/env python3
"""Developer tool: per-kernel averages of the counters in rocprofv3 --pmc output directories.

  python scripts/pmc_summary.py <kernel name regex> <dir> [<dir> ...]
"""
import csv
import glob
import re
import sys
from collections import defaultdict

pat = re.compile(sys.argv[1])
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not pat.search(k):
                continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = re.sub(r"\(.*", "", k)[:60]
        for (disp, cname), v in per_dispatch.items():
            acc[names[disp]][cname].append(v)
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s avg %.4e over %d dispatches" % (c, sum(v) / len(v), len(v)))
