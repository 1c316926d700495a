"""Summarise rocprofv3 --pmc counter_collection.csv files: per (counter, kernel) dispatch count and mean value."""
import csv
import sys
from collections import defaultdict

print("counter(KB per dispatch; separate rocprofv3 --pmc passes; bench.py --steps 3 2^20),kernel,dispatches,avg_value_KB")
for path in sys.argv[1:]:
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if not ("zk::" in name):
            continue
        k = (r["Counter_Name"], name)
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    for (c, name), (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{c},{name},{n},{tot / n:.1f}")
