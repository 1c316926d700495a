"""aggregate ZKHIP_TRACE_MSM=1 lines (stderr of a run) by batch signature"""
import ast, sys
from collections import defaultdict
agg = defaultdict(lambda: [0, [0.0] * 6])
for line in open(sys.argv[1]):
    if not line.startswith("zkhip-msm"):
        continue
    a, b = line[len("zkhip-msm "):].split("] [")
    lens = ast.literal_eval(a + "]"); t = ast.literal_eval("[" + b)
    key = (len(lens), sum(lens), max(lens))
    agg[key][0] += 1
    for i in range(6):
        agg[key][1][i] += t[i]
tot = [0.0] * 6
print("count  items  total_pts    max_len | ms: sort accum fixup reduce host total")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1][5]):
    print(f"{c:5d} {k[0]:6d} {k[1]:10d} {k[2]:10d} | " + " ".join(f"{x:8.2f}" for x in t))
    tot = [a + b for a, b in zip(tot, t)]
print("all" + " " * 33 + "| " + " ".join(f"{x:8.2f}" for x in tot))
