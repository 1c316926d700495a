#!/usr/bin/env python3
"""
Generates tests/golden/vectors.json from oracle/pyoracle.py (pure big-int restatement of the
reference).  The reference itself (Rust/arkworks) cannot run here, so these are oracle-generated
vectors: they pin the C oracle, the GPU path and future refactors to ONE set of values; they do
not pin the oracle to arkworks (see "parity unpinned" in oracle/pyoracle.py).

    python tests/golden/make_golden.py      # rewrites vectors.json (deterministic, seeded)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyoracle as po  # noqa: E402

h = lambda x: hex(x)
pt = lambda P: None if P is None else [hex(P[0]), hex(P[1])]


def main():
    rng = po.SplitMix64(0x5CA1AB1E)
    out = {"_comment": "oracle-generated (pyoracle.py, seed 0x5CA1AB1E); canonical (non-Montgomery) integers in hex"}
    n = 5
    f, g, ch = rng.fr_vec(1 << n), rng.fr_vec(1 << n), rng.fr_vec(n)
    out["sumcheck"] = {"table": [h(x) for x in f], "challenge": [h(x) for x in ch], "result": [[h(a), h(b)] for a, b in po.sumcheck(f, ch)]}
    out["sumcheck_product"] = {
        "f": [h(x) for x in f], "g": [h(x) for x in g], "challenge": [h(x) for x in ch],
        "result": [[h(a), h(b), h(c)] for a, b, c in po.sumcheck_product(f, g, ch)],
    }
    out["fix_variable"] = {"table": [h(x) for x in f], "points": [h(x) for x in ch[:3]], "result": [h(x) for x in po.fix_variable(f, ch[:3])]}
    cur, qs = list(f), []
    for i in range(n):
        hh = len(cur) // 2
        qs.append([h((cur[j + hh] - cur[j]) % po.R_MOD) for j in range(hh)])
        cur = po.fold(cur, ch[i])
    out["open_quotients"] = {"table": [h(x) for x in f], "point": [h(x) for x in ch], "q": qs, "value": h(cur[0])}
    x = rng.fr_vec(16)
    out["product_tree"] = {"x": [h(v) for v in x], "tree": [h(v) for v in po.product_tree(x)]}
    out["acc_product_reference_kat"] = {"x": [1, 2, 3, 4], "vx0": [1, 3, 2, 24], "vx1": [2, 4, 12, 0], "v1x": [2, 12, 24, 0],
                                        "source": "dist-primitive/src/dacc_product.rs:450-466"}
    m = 40
    bases = po.g1_bases(m, 77)
    sc = rng.fr_vec(m)
    sc[0], sc[1], sc[2] = 0, 1, po.R_MOD - 1
    bases[5] = bases[4]
    bases[7] = None
    out["msm_g1"] = {"bases": [pt(P) for P in bases], "scalars": [h(s) for s in sc], "result": pt(po.g1_msm(bases, sc))}
    for l in (1, 2):
        pp = po.PackedSharingParams(l)
        sec = rng.fr_vec(l)
        out[f"pss_l{l}"] = {
            "secrets": [h(s) for s in sec],
            "pack_from_public": [h(s) for s in pp.pack_from_public(sec)],
            "pack_single_of_first": [h(s) for s in pp.pack_single(sec[0])],
            "unpack2_of_squares": [h(s) for s in pp.unpack2([v * v % po.R_MOD for v in pp.pack_from_public(sec)])],
        }
    pp = po.PackedSharingParams(1)
    pb = [[po.g1_bases(4, 200 + p)] for p in range(8)]
    ps = [[rng.fr_vec(4)] for _ in range(8)]
    res = po.d_msm_all(pb, ps, pp)
    out["d_msm_l1"] = {"seed_bases": [200 + p for p in range(8)], "scalars": [[h(s) for s in ps[p][0]] for p in range(8)],
                       "shares": [pt(res[p][0]) for p in range(8)]}
    with open(os.path.join(HERE, "vectors.json"), "w") as fo:
        json.dump(out, fo, indent=0)
    print("wrote vectors.json", os.path.getsize(os.path.join(HERE, "vectors.json")), "bytes")


if __name__ == "__main__":
    main()
