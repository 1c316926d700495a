#!/usr/bin/env python3
"""
Generates csrc/fp_mul_gen.cuh: the FIPS Montgomery multiplication for Fr (8 limbs) and Fq (12
limbs) as one inline-asm statement per accumulation run.

Why generated asm instead of one statement per multiply-accumulate:
  * hipcc pads every asm statement whose outputs feed the next VALU op with an `s_nop`; with one
    statement per MAC that is 288 pads per Fq multiplication (measured: -12% at 2 waves/SIMD,
    -30% at 1 wave/SIMD, tools/ubench_mac.hip);
  * gfx950 needs 2 wait states between a VALU writing an SGPR/VCC (the carry-out of
    v_mad_u64_u32) and a VALU reading it as carry-in (hipcc's own code pads `s_nop 1` there).
    Inside a statement nothing is padded for us, so carries rotate through three SGPR pairs and
    each `v_addc_co_u32` is issued two instructions after its `v_mad_u64_u32`: hazard-safe
    without idle slots.

    python tools/gen_fp_mul.py > scalable-collaborative-zksnark_amd/csrc/fp_mul_gen.cuh
"""


def run(pairs):
    """pairs: list of (src0, src1) operand names for consecutive MACs into (lo, hi).  Returns asm lines."""
    n = len(pairs)
    lines = []
    mad = lambda i: f"v_mad_u64_u32 %[lo], %[c{i % 3}], {pairs[i][0]}, {pairs[i][1]}, %[lo]"
    addc = lambda i: f"v_addc_co_u32 %[hi], vcc, 0, %[hi], %[c{i % 3}]"
    if n == 1:
        return [mad(0), "s_nop 1", addc(0)]
    if n == 2:
        return [mad(0), mad(1), "s_nop 0", addc(0), addc(1)]
    for i in range(n):
        lines.append(mad(i))
        if i >= 2:
            lines.append(addc(i - 2))
    lines.append(addc(n - 2))
    lines.append(addc(n - 1))
    return lines


def stmt(pairs, ins):
    """one asm statement; ins: dict name -> (constraint, C expression)"""
    body = "\\n\\t".join(run(pairs))
    in_ops = ", ".join(f'[{k}] "{c}"({e})' for k, (c, e) in ins.items())
    return (f'    asm("{body}"\n        : [lo] "+v"(lo), [hi] "+v"(hi), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)\n'
            f'        : {in_ops}\n        : "vcc");\n')


def gen(cfg, N):
    out = []
    out.append(f"template <>\n__device__ __forceinline__ Fp<{cfg}> fp_mul<{cfg}>(const Fp<{cfg}>& a, const Fp<{cfg}>& b) {{\n")
    out.append(f"    u64 lo = 0, c0, c1, c2;\n    u32 hi = 0;\n    u32 m[{N}];\n    Fp<{cfg}> t;\n")
    for k in range(2 * N - 1):
        js = [j for j in range(N) if 0 <= k - j < N]
        # part A: a_j * b_{k-j}
        pairs = [(f"%[a{j}]", f"%[b{k - j}]") for j in js]
        ins = {}
        for j in js:
            ins[f"a{j}"] = ("v", f"a.l[{j}]")
        for j in js:
            ins[f"b{k - j}"] = ("v", f"b.l[{k - j}]")
        out.append(f"    // column {k}\n")
        out.append(stmt(pairs, ins))
        # part B: m_j * P_{k-j}, j < min(k, N) (P limbs are wave-uniform: SGPR operands)
        jb = [j for j in js if j < k and j < N] if k < N else js
        if k < N:
            jb = [j for j in range(k)]
        if jb:
            pairs = [(f"%[p{k - j}]", f"%[m{j}]") for j in jb]
            ins = {}
            for j in jb:
                ins[f"m{j}"] = ("v", f"m[{j}]")
            for j in jb:
                ins[f"p{k - j}"] = ("s", f"{cfg}::P({k - j})")
            out.append(stmt(pairs, ins))
        if k < N:
            out.append(f"    m[{k}] = (u32)lo * {cfg}::INV;\n")
            out.append(stmt([("%[p0]", "%[mk]")], {"mk": ("v", f"m[{k}]"), "p0": ("s", f"{cfg}::P(0)")}))
        else:
            out.append(f"    t.l[{k - N}] = (u32)lo;\n")
        out.append("    lo = (lo >> 32) | ((u64)hi << 32);\n    hi = 0;\n")
    out.append(f"    t.l[{N - 1}] = (u32)lo;\n    return fp_reduce_once<{cfg}>(t);\n}}\n\n")
    return "".join(out)


def gen_mac_wide(cfg, N, uniform_b=False):
    """acc (2N + 1 limbs) += a * b as integers, no reduction: the lazily reduced running sums of the product sumcheck
    (sum of f_lo g_lo over many pairs, reduced once per call).  Column k starts from acc[k] plus the carry of column k-1
    (one more mad with the constant 1: it cannot carry), runs the same 96-bit accumulate as fp_mul and leaves its low
    word in acc[k]."""
    out = []
    name = "fp_mac_wide_s" if uniform_b else "fp_mac_wide"  # _s: b is wave-uniform (a kernel argument): its limbs are SGPR operands
    out.append(f"__device__ __forceinline__ void {name}(u32 (&acc)[{2 * N + 1}], const Fp<{cfg}>& a, const Fp<{cfg}>& b) {{\n")
    out.append("    u64 lo = 0, c0, c1, c2;\n    u32 hi = 0;\n")
    for k in range(2 * N - 1):
        js = [j for j in range(N) if 0 <= k - j < N]
        pairs = [(f"%[a{j}]", f"%[b{k - j}]") for j in js]
        ins = {"acc": ("v", f"acc[{k}]")}
        for j in js:
            ins[f"a{j}"] = ("v", f"a.l[{j}]")
        for j in js:
            ins[f"b{k - j}"] = ("s" if uniform_b else "v", f"b.l[{k - j}]")
        body = "\\n\\t".join(["v_mad_u64_u32 %[lo], %[c2], %[acc], 1, %[lo]"] + run(pairs))
        in_ops = ", ".join(f'[{kk}] "{c}"({e})' for kk, (c, e) in ins.items())
        out.append(f"    // column {k}\n")
        out.append(f'    asm("{body}"\n        : [lo] "+v"(lo), [hi] "+v"(hi), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)\n'
                   f'        : {in_ops}\n        : "vcc");\n')
        out.append(f"    acc[{k}] = (u32)lo;\n    lo = (lo >> 32) | ((u64)hi << 32);\n    hi = 0;\n")
    out.append(f"    lo += acc[{2 * N - 1}];\n    acc[{2 * N - 1}] = (u32)lo;\n    acc[{2 * N}] += (u32)(lo >> 32);\n}}\n\n")
    return "".join(out)


def gen_redc_wide(cfg, N):
    """t (2N + 1 limbs, < 2^(32 N) * 2^4 * p) -> (t + m p) / 2^(32 N) as N + 1 limbs (< t / 2^(32 N) + p): the reduction half
    of fp_mul on a running sum of products (fp_mac_wide).  The caller subtracts multiples of p."""
    out = []
    out.append(f"__device__ __forceinline__ void fp_redc_wide(u32 (&r)[{N + 1}], const u32 (&t)[{2 * N + 1}]) {{\n")
    out.append(f"    u64 lo = 0, c0, c1, c2;\n    u32 hi = 0;\n    u32 m[{N}];\n")
    for k in range(2 * N - 1):
        jb = list(range(k)) if k < N else [j for j in range(N) if 0 <= k - j < N and k - j >= 1]
        out.append(f"    // column {k}\n")
        pairs = [(f"%[p{k - j}]", f"%[m{j}]") for j in jb]
        ins = {"acc": ("v", f"t[{k}]")}
        for j in jb:
            ins[f"m{j}"] = ("v", f"m[{j}]")
        for j in jb:
            ins[f"p{k - j}"] = ("s", f"{cfg}::P({k - j})")
        body = "\\n\\t".join(["v_mad_u64_u32 %[lo], %[c2], %[acc], 1, %[lo]"] + (run(pairs) if pairs else []))
        in_ops = ", ".join(f'[{kk}] "{c}"({e})' for kk, (c, e) in ins.items())
        out.append(f'    asm("{body}"\n        : [lo] "+v"(lo), [hi] "+v"(hi), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)\n'
                   f'        : {in_ops}\n        : "vcc");\n')
        if k < N:
            out.append(f"    m[{k}] = (u32)lo * {cfg}::INV;\n")
            out.append(stmt([("%[p0]", "%[mk]")], {"mk": ("v", f"m[{k}]"), "p0": ("s", f"{cfg}::P(0)")}))
        else:
            out.append(f"    r[{k - N}] = (u32)lo;\n")
        out.append("    lo = (lo >> 32) | ((u64)hi << 32);\n    hi = 0;\n")
    out.append(f"    lo += t[{2 * N - 1}];\n    r[{N - 1}] = (u32)lo;\n    r[{N}] = (u32)(lo >> 32) + t[{2 * N}];\n}}\n\n")
    return "".join(out)


print("// GENERATED by tools/gen_fp_mul.py -- do not edit.  See that file for the rationale.")
print("#pragma once\n// included from fp.cuh (inside namespace zk), after the generic fp_mul template\n")
print(gen("FrCfg", 8))
print(gen("FqCfg", 12))
print(gen_mac_wide("FrCfg", 8))
print(gen_mac_wide("FrCfg", 8, uniform_b=True))
print(gen_redc_wide("FrCfg", 8))
