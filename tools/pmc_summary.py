"""Summarise rocprofv3 --pmc counter_collection.csv files: per (counter, kernel, grid size) dispatch count and mean value.
The grid size separates the launches of one kernel on different problem sizes (2^20 / 2^24 MSM, 2^20 .. 2^26 sumchecks)."""
import csv
import sys
from collections import defaultdict

print("counter(KB per dispatch; separate rocprofv3 --pmc passes of bench.py),kernel,grid_size,dispatches,avg_value_KB")
for path in sys.argv[1:]:
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if not ("zk::" in name):
            continue
        k = (r["Counter_Name"], name, r.get("Grid_Size", "?"))
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    for (c, name, grid), (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{c},{name},{grid},{n},{tot / n:.1f}")
