"""pretty-print a rocprofv3 *_kernel_stats.csv"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:46]:46s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
