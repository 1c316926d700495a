"""
Extracts the party exchanges of the reference's OWN sample run -- hack/run-hyperplonk/output.txt, the leader's log of a real
128-party run (l = 16, 2^12 constraints: 264-byte blocks of 8 Fr = 4M / (l N_p), 488-byte opens of 10 commitments = log2(2^14 / 16)) --
into tests/golden/ref_log_n12_l16.json: every `Comm: from A to B, <bytes>B` line with the stack of timer labels around it.
The byte counts are the sizes of party 0's ark-serialize (compressed) messages, i.e. the SHAPES of what every collaborative
primitive hands to the network; tests/test_reference_log.py replays the primitives at the same parameters and compares.
(`Comm: from leader to all, 128B` lines carry the party count, not a size, in the version that produced the log: kept, unused.)

    python tests/golden/make_ref_log_fixture.py        (needs /root/reference; the JSON is committed)
"""
import json
import os
import re

SRC = "/root/reference/hack/run-hyperplonk/output.txt"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_log_n12_l16.json")

stack, entries = [], []
for ln, line in enumerate(open(SRC, encoding="utf-8"), 1):
    m = re.match(r"^(·*)Start:\s+(.*?) \(thread", line)
    if not m:
        continue
    depth, label = len(m.group(1)), m.group(2)
    del stack[depth:]
    c = re.match(r"Comm: from (\w+) to (\w+), (\d+)B", label)
    if c:
        entries.append([ln, " > ".join(stack[-2:]), c.group(1), c.group(2), int(c.group(3))])  # [line, enclosing timers, from, to, bytes]
    else:
        stack.append(label)
tot = re.findall(r"^Comm: \((\d+), (\d+)\)", open(SRC, encoding="utf-8").read(), flags=re.M)
json.dump({"source": "hack/run-hyperplonk/output.txt (LBruyne/Scalable-Collaborative-zkSNARK)", "n": 12, "l": 16, "parties": 128,
           "comm_totals_up_down": [int(x) for x in tot[-1]] if tot else None, "columns": ["line", "enclosing timers", "from", "to", "bytes"], "exchanges": entries}, open(OUT, "w"), separators=(",", ":"))
print(len(entries), "exchanges ->", OUT)
