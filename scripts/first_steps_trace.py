"""Which kernels make the first ~15 steps after the setup step slower than steady state (task 6e)?
Reads a rocprofv3 kernel trace (CSV) of `bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg --no-train-leg`,
cuts it into steps at the attention launch, and prints per step the wall time (start of its attention launch to the
start of the next one), the sum of its kernels' durations, and per kernel name the mean duration over step ranges."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)


def short(name):
    for key in ("att_fold_fused", "softmax_local", "softmax_fix", "softmax", "gather4", "spmm_merge2", "spmm_finish", "bi_interaction",
                "bi_mul", "bi_kernel", "readout"):
        if key in name:
            tail = name.split("<", 1)[1].split(">")[0] if "<" in name and key in ("spmm_merge2", "bi_interaction", "bi_mul", "bi_kernel") else ""
            return key + ("<" + tail[:14] + ">" if tail else "")
    return name.split("(")[0][-40:]


starts = [i for i, k in enumerate(ks) if "att_fold_fused" in k[2]]
steps = []
for a, b in zip(starts, starts[1:] + [len(ks)]):
    seg = ks[a:b]
    # a step ends with its last propagation kernel: cut trailing non-step kernels (fills, copies between loops)
    steps.append(seg)
print("%d steps found" % len(steps))
wall = [(steps[i + 1][0][0] - steps[i][0][0]) / 1e3 for i in range(len(steps) - 1)]
busy = [sum(e - s for s, e, _ in seg) / 1e3 for seg in steps]
print("step : wall us (start to next start) / busy us (sum of kernel durations)")
for i in range(min(32, len(wall))):
    print("  %3d : %8.1f / %7.1f%s" % (i, wall[i], busy[i], "   <- setup / loop boundary" if wall[i] > 2000 else ""))
ranges = [("steps 1-5 (warm-up)", 1, 6), ("steps 6-15", 6, 16), ("steps 16-25", 16, 26), ("last 20 steps", len(steps) - 21, len(steps) - 1)]
table = defaultdict(dict)
for label, lo, hi in ranges:
    acc = defaultdict(list)
    for seg in steps[lo:hi]:
        per = defaultdict(float)
        for s, e, n in seg:
            per[short(n)] += (e - s) / 1e3
        for k, v in per.items():
            acc[k].append(v)
    for k, v in acc.items():
        table[k][label] = sum(v) / len(v)
print("\nper kernel, mean us per step:")
print("  %-34s" % "kernel" + "".join("%22s" % r[0] for r in ranges))
for k in sorted(table, key=lambda k: -max(table[k].values())):
    if max(table[k].values()) < 1.0:
        continue
    print("  %-34s" % k + "".join("%22.1f" % table[k].get(r[0], 0.0) for r in ranges))
tot = {r[0]: sum(table[k].get(r[0], 0.0) for k in table) for r in ranges}
print("  %-34s" % "sum" + "".join("%22.1f" % tot[r[0]] for r in ranges))
