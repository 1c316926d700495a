#!/usr/bin/env python3
"""
Where the time of ONE proof goes on the GPU: python tools/e2e_timeline.py <kernel_trace.csv> [proofs_to_skip=1]
(trace: rocprofv3 --kernel-trace of `host/bin/hyperplonk --l 1 --n 20 --reps R`; see tools/profile_timeline.sh).

The bucket accumulation (k_accum_tiles) is the throughput engine of a proof; everything else either runs beside it or is
exposed.  For every proof (a run of kernels between two host gaps that contains one k_batch_div) this prints
  span                      first kernel -> last kernel
  accumulate (union)        time during which at least one k_accum_tiles is running
  exposed                   span - accumulate, split by what IS running then: sort family / fix-up + reduction / sumcheck
                            family + element-wise / nothing (idle)
  sum of durations          per family (durations overlap: a kernel that shares the chip runs longer)
  mixed additions           sum over the accumulation launches of their sorted entries (grid x tile length is not in the trace:
                            taken from the proof's item list, n = 20: 342.5 M) -> additions/s inside the accumulate union
"""
import csv, sys
from collections import defaultdict

FAMILIES = (("accum", ("k_accum_tiles",)), ("sort", ("k_digits", "k_part_")), ("reduce", ("k_fixup", "k_halve", "k_finish")),
            ("sumcheck", ("k_pass", "k_local", "k_fold", "k_plain", "k_tree", "k_fr_", "k_batch_div", "k_open")), ("copy", ("__amd_rocclr",)))


def family(name):
    for f, keys in FAMILIES:
        if any(k in name for k in keys):
            return f
    return "other"


def union(iv):
    iv = sorted(iv)
    out, cs, ce = [], None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            out.append((cs, ce))
            cs, ce = s, e
    if cs is not None:
        out.append((cs, ce))
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def subtract(a, b):
    """intervals of a not covered by b (both unions)"""
    out, j = [], 0
    for s, e in a:
        cur = s
        while j < len(b) and b[j][1] <= cur:
            j += 1
        k = j
        while k < len(b) and b[k][0] < e:
            if b[k][0] > cur:
                out.append((cur, b[k][0]))
            cur = max(cur, b[k][1])
            k += 1
        if cur < e:
            out.append((cur, e))
    return out


def main():
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(sys.argv[1]))]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows.sort()
    # a proof = the kernels around ONE k_batch_div (step 2.e runs it once per proof), cut at the longest kernel-free gaps between
    # consecutive anchors (the host work between two repetitions: transcript assembly, digest) and at >= 5 ms gaps at both ends
    anchors = [s for s, _, n in rows if "k_batch_div" in n]
    gaps, end = [], None  # (gap length, time the gap starts, time it ends)
    for s, e, n in rows:
        if end is not None and s > end:
            gaps.append((s - end, end, s))
        end = e if end is None else max(end, e)
    cuts = []
    for a0, a1 in zip(anchors, anchors[1:]):
        between = [g for g in gaps if a0 < g[1] and g[2] < a1]
        if between:
            cuts.append(max(between))
    bounds = []
    for i, a0 in enumerate(anchors):
        lo = cuts[i - 1][2] if i > 0 else max([g[2] for g in gaps if g[2] <= a0 and g[0] > 5_000_000] or [rows[0][0]])
        hi = cuts[i][1] if i < len(cuts) else min([g[1] for g in gaps if g[1] >= a0 and g[0] > 5_000_000] or [max(e for _, e, _ in rows)])
        bounds.append((lo, hi))
    segs = [[r for r in rows if lo <= r[0] and r[1] <= hi] for lo, hi in bounds]
    proofs = [sg for sg in segs if sum("k_accum_tiles" in n for _, _, n in sg) > 10]
    print(f"{len(rows)} dispatches, {len(segs)} segments, {len(proofs)} proofs found; skipping the first {skip}")
    for pi, sg in enumerate(proofs[skip:]):
        t0, t1 = min(s for s, _, _ in sg), max(e for _, e, _ in sg)
        fam = defaultdict(list)
        for s, e, n in sg:
            fam[family(n)].append((s, e))
        acc = union(fam["accum"])
        whole = [(t0, t1)]
        exposed = subtract(whole, acc)
        rest = exposed
        parts = {}
        for f in ("reduce", "sort", "sumcheck", "copy", "other"):
            u = union(fam[f])
            covered = length(rest) - length(subtract(rest, u))
            parts[f] = covered
            rest = subtract(rest, u)
        ms = lambda x: x / 1e6
        print(f"proof {pi}: span {ms(t1 - t0):.2f} ms; accumulate running {ms(length(acc)):.2f} ms in {len(acc)} stretches; exposed {ms(length(exposed)):.2f} ms = "
              + ", ".join(f"{f} {ms(v):.2f}" for f, v in parts.items()) + f", idle {ms(length(rest)):.2f}")
        print("   sum of durations (ms): " + ", ".join(f"{f} {ms(sum(e - s for s, e in fam[f])):.2f} ({len(fam[f])})" for f, _ in FAMILIES if fam[f]))
        # the longest exposed stretches and what ran in them
        ex = sorted(exposed, key=lambda iv: iv[0] - iv[1])[:6]
        for s, e in sorted(ex):
            names = defaultdict(int)
            for ks, ke, n in sg:
                ov = min(ke, e) - max(ks, s)
                if ov > 0:
                    names[n.replace("zk::", "")[:24]] += ov
            top = sorted(names.items(), key=lambda kv: -kv[1])[:4]
            print(f"   exposed {ms(e - s):6.2f} ms at +{ms(s - t0):6.2f}: " + ", ".join(f"{n} {ms(v):.2f}" for n, v in top))


if __name__ == "__main__":
    main()
