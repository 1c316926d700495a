"""
The C++ host mirror of `dist-primitive` (scalable-collaborative-zksnark_amd/host/zkhost, header-only, over the C ABI).

The reference's host is Rust and no Rust toolchain exists in this image; the mirror is the compiled host side a user of the
crate would switch to: the reference's function names, arguments, output shapes and ordering, every loop body on the GPU.
tests/native/host_mirror.cpp drives every function of it once on inputs written here and writes record files back:

  * CPU (no GPU): its field arithmetic, PackedSharingParams (matrices, maps, transform tables), pss2ss / degree_reduce /
    d_unpack* over the thread net, the leader rounds of the collaborative sumchecks, merge / transpose / sub_index -- against
    the oracle's independent restatement (python big-ints) and the reference's own KATs;
  * GPU: every collaborative primitive, all 8 l parties as threads with one ctx each (and party 0 on the leader-echo net),
    against the Python host layer on the same inputs -- which the other GPU tests pin to the oracle -- bit for bit.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")
R = po.R_MOD


def _need_gxx():
    import shutil

    if shutil.which("g++") is None:
        pytest.skip("no g++ on this box: the C++ host is not built (libzkhip.so and the Python host do not depend on it)")


def _build():
    _need_gxx()
    subprocess.check_call(["make", "-C", NATIVE, "-s", "host_mirror"])
    return os.path.join(NATIVE, "host_mirror")


def _write(path, records):
    with open(path, "wb") as f:
        for (name, party), payload in records.items():
            b = np.ascontiguousarray(payload).tobytes()
            f.write(name.encode().ljust(24, b"\0") + struct.pack("<QQ", party, len(b)) + b)


def _read(path):
    out, raw, off = {}, open(path, "rb").read(), 0
    while off < len(raw):
        name = raw[off : off + 24].split(b"\0")[0].decode()
        party, n = struct.unpack_from("<QQ", raw, off + 24)
        out[(name, party)] = raw[off + 40 : off + 40 + n]
        off += 40 + n
    return out


def _mont(xs):
    """python ints -> the Montgomery limbs a Vec<Fr> holds"""
    return np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)


def _ints(b):
    a = np.frombuffer(b, dtype="<u8").reshape(-1, 4)
    return [po.fr_from_mont_limbs(x) for x in a]


def _run(mode, records, tmp_path, timeout=600):
    exe = _build()
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _write(fin, records)
    r = subprocess.run([exe, mode, fin, fout], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    return r, (_read(fout) if r.returncode == 0 else None)


# ---------------------------------------------------------------------------------------
# CPU: the host-only part against the oracle
# ---------------------------------------------------------------------------------------
def test_cpp_host_arithmetic_pss_and_leader_rounds_match_the_oracle(tmp_path):
    rng = po.SplitMix64(4100)
    a, b = rng.fr_vec(64), rng.fr_vec(64)
    a[5] = 0
    ls = [1, 2, 4, 8]
    # points for the wire formats: k G (python big-int group law), infinity, and a point whose y is the smaller root
    from helpers import pt_mont
    import zkhip.serialize as ser
    from zkhip.pss import PackedSharingParams as HostPP

    pts = [po.g1_mul(po.G1_GEN, k) for k in (1, 2, 3, 0x1234567, R - 1)] + [None]
    one_q = np.array(po.fq_to_mont_limbs(1), dtype=np.uint64)
    jac = np.stack([np.concatenate([pt_mont(P), one_q if P is not None else np.zeros(6, dtype=np.uint64)]) for P in pts])
    share_dir = tmp_path / "shares"
    share_dir.mkdir()
    r, out = _run("host", {("a", 0): _mont(a), ("b", 0): _mont(b), ("ls", 0): np.array(ls, dtype=np.uint64), ("points", 0): jac,
                           ("dir", 0): np.frombuffer(str(share_dir).encode(), dtype=np.uint8)}, tmp_path)
    assert r.returncode == 0, r.stderr
    # wire formats against zkhip.serialize (python big-ints) -- the C++ side has already checked its own round trips
    assert out[("fr_vec_serialize", 0)] == ser.fr_vec_serialize(a)
    assert out[("g1_compressed", 0)] == b"".join(ser.g1_serialize_compressed(P) for P in pts)
    assert out[("g1_uncompressed", 0)] == b"".join(ser.g1_serialize_uncompressed(P) for P in pts)
    # the published compressed encoding of the BLS12-381 G1 generator (zcash / IETF pairing-friendly-curves draft): an anchor from outside this repo
    assert out[("g1_compressed", 0)][:48].hex() == "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    assert out[("g1_vec_compressed", 0)] == ser.g1_vec_serialize_compressed(pts) and len(out[("g1_vec_compressed", 0)]) == 8 + 48 * len(pts)
    assert {bool(ser.g1_serialize_compressed(P)[0] & 0x20) for P in pts if P} == {True, False}  # both roots occur
    want_dir = tmp_path / "shares_py"
    want_dir.mkdir()
    ser.delegator_write(str(want_dir), a, HostPP(2))  # examples/delegator.rs:71-95: `delegator` + 16 worker files
    names = sorted(os.listdir(want_dir))
    assert names == sorted(os.listdir(share_dir)) and len(names) == 17
    for nm in names:
        assert open(share_dir / nm, "rb").read() == open(want_dir / nm, "rb").read(), nm
    g = lambda name, party=0: _ints(out[(name, party)])
    assert g("add") == [(x + y) % R for x, y in zip(a, b)]
    assert g("sub") == [(x - y) % R for x, y in zip(a, b)]
    assert g("mul") == [x * y % R for x, y in zip(a, b)]
    assert g("inv") == [pow(x, -1, R) if x else 0 for x in a]
    # to_canonical: the limbs ARE the integer
    assert [int.from_bytes(out[("canonical", 0)][32 * i : 32 * i + 32], "little") for i in range(64)] == a
    assert g("root_of_unity") == [0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B]  # SURVEY.md 8 "PSS exact semantics"
    # the synthetic-table generator of both hosts is the oracle's SplitMix64 (sequential in C++, vectorised in numpy: chunk boundary at 2^16 candidates)
    from zkhip.field import limbs_to_int, splitmix_fr

    raw = out[("splitmix_fr", 0)]  # (the generator's integers ARE the limb patterns: the Montgomery form of a uniform element)
    assert [int.from_bytes(raw[32 * i : 32 * i + 32], "little") for i in range(300)] == po.SplitMix64(0x5CA1AB1E + 100001).fr_vec(300)
    seq = po.SplitMix64(31337).fr_vec(73000)
    assert [limbs_to_int(x) for x in splitmix_fr(73000, 31337)] == seq
    for l in ls:
        pp = po.PackedSharingParams(l)
        n = pp.n
        flat = lambda m: [x for row in m for x in row]
        pack, unpack, unpack2 = pp.pack_matrix(), pp.unpack_matrix(), pp.unpack2_matrix()
        assert g("pack_matrix", l) == flat(pack)
        assert g("unpack_matrix", l) == flat(unpack)
        assert g("unpack2_matrix", l) == flat(unpack2)
        assert g("pack_from_public", l) == pp.pack_from_public(a[:l])
        assert g("pack_single", l) == pp.pack_single(a[0])
        assert g("unpack", l) == pp.unpack(b[:n])
        assert g("unpack2", l) == pp.unpack2(b[:n])
        lam = [sum(unpack2[j][i] for j in range(l)) % R for i in range(n)]
        c_last = sum(pack[n - 1][j] for j in range(l)) % R
        assert g("dmsm_coeffs", l) == [c_last * x % R for x in lam]
        assert g("degree_reduce_row", l) == [sum(pack[1][j] * unpack2[j][i] for j in range(l)) % R for i in range(n)]
        # the exchanges that need no device, all parties as threads
        assert g("pss2ss", l) == flat(po.pss2ss_all(b[:n], pp))
        assert g("degree_reduce", l) == pp.pack_from_public(pp.unpack2(b[:n]))
        assert g("d_unpack_0", l) == [pp.unpack(b[:n])[0]] * n
        assert g("d_unpack", l) == pp.unpack(b[:n])  # only party 2 holds it: the others contribute empty vectors
        assert g("d_unpack2", l) == pp.unpack2(b[:n])
        # pss2ss on the no-`comm` fake: n copies of the leader's share
        assert g("pss2ss_echo", l) == po.pss2ss_all([b[0]] * n, pp)[0]
        # five exchanges of one Fr each: 32 (n - 1) bytes up and down per exchange and party (MPCNet::get_comm accounting)
        comm = np.frombuffer(out[("comm", l)], dtype="<u8").reshape(n, 2)
        assert (comm == 5 * 32 * (n - 1)).all()
        # the transform tables are the ones the Python host hands to zk_fr_ntt_map
        from zkhip.pss import PackedSharingParams as HostPP

        hp = HostPP(l)
        for name, key in (("ntt_winv", "winv"), ("ntt_w", "w"), ("ntt_scale", "scale")):
            want = np.concatenate([hp.ntt_tables(k)[key] for k in ("pack", "unpack", "unpack2")])
            assert out[(name, l)] == want.tobytes(), (name, l)

    # the leader rounds (dsumcheck.rs:226-283,:440-507) on host vectors
    f, gg, trip = a[:8], b[:8], []
    for i in range(3):
        t, f, gg = _round_product(f, gg, a[8 + i])
        trip += list(t)
    assert g("round_product") == trip
    v, pairs = a[:8], []
    for i in range(3):
        h = len(v) // 2
        pairs += [sum(v[:h]) % R, sum(v[h:]) % R]
        v = [(v[j] * (1 - b[8 + i]) + v[j + h] * b[8 + i]) % R for j in range(h)]
    assert g("round_plain") == pairs
    assert g("round_last") == [f[0], gg[0], v[0]]
    assert g("merge") == po.merge([a[7 * q : 7 * q + 7] for q in range(3)])
    si = np.frombuffer(out[("sub_index", 0)], dtype="<u8").reshape(-1, 2)
    assert [tuple(int(x) for x in row) for row in si] == [po.sub_index(i) for i in range(1, 16)]
    assert (si[3:7] == [[0, 1], [2, 3], [4, 5], [6, 7]]).all() and (si[7:9] == [[0, 1], [2, 3]]).all()  # dacc_product.rs:442-448
    assert np.frombuffer(out[("transpose", 0)], dtype="<u8").tolist() == [1, 4, 2, 5, 3, 6]  # operator.rs:42-49


def _round_product(f, g, r):
    h = len(f) // 2
    t = (sum(f[j] * g[j] for j in range(h)) % R, sum(f[j + h] * g[j + h] for j in range(h)) % R,
         sum((2 * f[j + h] - f[j]) * (2 * g[j + h] - g[j]) for j in range(h)) % R)
    fold = lambda v: [(v[j] * (1 - r) + v[j + h] * r) % R for j in range(h)]
    return t, fold(f), fold(g)


def test_cpp_host_refuses_to_compute_without_a_gpu(tmp_path):
    import zkhip

    if zkhip.lib().zk_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu tests")
    r, _ = _run("gpu", {("params", 0): np.array([1, 6, 0], dtype=np.uint64)}, tmp_path)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr, (r.returncode, r.stderr)


# ---------------------------------------------------------------------------------------
# GPU: every collaborative primitive, C++ host == Python host, bit for bit
# ---------------------------------------------------------------------------------------
def _python_party(be, net, pp, m, tabs, chal, point, rec):
    """the sequence of tests/native/host_mirror.cpp run_party through zkhip.dist_primitive"""
    import zkhip.dist_primitive as dp

    p, M, P = net.party_id, 1 << m, pp.n
    logP, logl = P.bit_length() - 1, pp.l.bit_length() - 1
    put = lambda name, *arrs: rec.__setitem__((name, p), rec.get((name, p), b"") + b"".join(np.ascontiguousarray(a, dtype=np.uint64).tobytes() for a in arrs))
    fh, gh = tabs["f"][p], tabs["g"][p]
    f, g, h0, h1, h2 = (be.to_device(tabs[k][p]) for k in ("f", "g", "h0", "h1", "h2"))
    gd = dp.PolynomialCommitmentCub.new_random(be, m + logP, P, 3).mature()
    gc = dp.PolynomialCommitmentCub.new_single(be, m + logl, pp, 5).mature()
    share = fh[0]

    put("sumcheck", dp.sumcheck(be, f, M, chal))
    put("sumcheck_product", dp.sumcheck_product(be, f, g, M, chal))
    put("pss2ss", dp.pss2ss(share, pp, net))
    put("c_sumcheck", dp.c_sumcheck(be, f, M, chal, pp, net))
    put("c_sumcheck_product", dp.c_sumcheck_product(be, f, g, M, chal, pp, net))
    put("d_sumcheck", dp.d_sumcheck(be, f, M, chal, net))
    put("d_sumcheck_product", dp.d_sumcheck_product(be, f, g, M, chal, net))

    bases = [gc[m + logl], gc[m - 1 + logl]]
    put("d_msm", dp.d_msm(be, bases, [f, g], [M, M // 2], pp, net))
    put("d_msm_unscaled", dp.d_msm(be, bases, [f, g], [M, M // 2], pp, net, prescale=False))
    put("commit", dp.commit(be, gd, f, M))
    put("open", *dp.open_(be, gd, f, M, point))
    put("d_commit", dp.d_commit(be, gd, f, M, net))
    put("d_open", *dp.d_open(be, gd, f, M, point, net))
    put("c_commit", dp.c_commit(be, gc, [f, g.at(32 * (M // 2))], [M, M // 2], pp, net))
    put("c_open", *dp.c_open(be, gc, f, M, point, pp, net))

    tree = dp.acc_product(be, f, M)
    put("acc_product", tree.download((2 * M, 4)))
    ev, od = be.fr_deinterleave(tree, M)
    put("acc_product_views", ev.download((M, 4)), od.download((M, 4)), tree.download((M, 4), offset=32 * M))
    sub, top = dp.d_acc_product(be, g, M, net)
    put("d_acc_product", sub.download((2 * M, 4)))
    put("d_acc_product_top", top if top is not None else np.zeros((0, 4), dtype=np.uint64))
    sub, top = dp.c_acc_product(be, g, M, pp, net)
    put("c_acc_product", sub.download((2 * M, 4)))
    put("c_acc_product_top", top if top is not None else np.zeros((0, 4), dtype=np.uint64))

    put("fix_variable", dp.fix_variable(be, f, M, point[:3]).download((M >> 3, 4)))
    fv = dp.d_fix_variable(be, f, M, point[: m + logl], pp, net)  # more points than local variables once l > 1: a host value
    put("d_fix_variable", fv if isinstance(fv, np.ndarray) else fv.download((1, 4)))
    put("d_fix_variable_short", dp.d_fix_variable(be, f, M, point[:2], pp, net).download((M >> 2, 4)))

    few = gh[:5]
    put("degree_reduce", dp.degree_reduce(share, pp, net))
    put("degree_reduce_many", dp.degree_reduce_many(few, pp, net, be=be))
    put("d_unpack_0", dp.d_unpack_0(share, pp, net))
    put("d_unpack", dp.d_unpack(share, P - 1, pp, net))
    put("d_unpack2", dp.d_unpack2(share, 1 % P, pp, net))
    put("d_unpack2_many", dp.d_unpack2_many(few, 0, pp, net, be=be))

    cub = dp.PolynomialCommitmentCub.new(be, chal[: m + logl])
    packed = cub.to_packed(be, pp, p).mature()
    put("structured_commit", dp.commit(be, cub.mature(), f, M))
    put("structured_c_commit", dp.c_commit(be, packed, [f], [M], pp, net))
    put("structured_c_open", *dp.c_open(be, packed, f, M, point, pp, net))
    from zkhip import sharding as sh

    put("sharded_msm", sh.sharded_msm(be, gd[m], f, M, net))
    put("shard_sc", sh.sharded_sumcheck(be, f, M, chal, net))
    put("shard_sc_product", sh.sharded_sumcheck_product(be, f, g, M, chal, net))
    if p == 0 and not getattr(net, "echo", False):
        full = lambda k: be.to_device(np.ascontiguousarray(np.stack(tabs[k], axis=1).reshape(-1, 4)))  # full[q + P i] = tabs[q][i]
        put("mono_sc", dp.sumcheck(be, full("f"), P * M, chal))
        put("mono_sc_product", dp.sumcheck_product(be, full("f"), full("g"), P * M, chal))
    for buf, cnt in dp.c_acc_product_and_share(be, f, g, h0, h1, h2, M, pp, net):
        put("c_acc_share_len", np.array([cnt], dtype=np.uint64))
        put("c_acc_product_and_share", buf.download((cnt, 4)))
    put("comm", np.array([net.upload, net.download], dtype=np.uint64))


def _case(l, m, echo, seed):
    from zkhip.field import random_fr
    from zkhip.pss import PackedSharingParams

    pp = PackedSharingParams(l)
    M = 1 << m
    tabs = {k: [random_fr(M, seed + 100 * i + p) for p in range(pp.n)] for i, k in enumerate(("f", "g", "h0", "h1", "h2"))}
    nvar = m + (pp.n.bit_length() - 1) + 4
    chal, point = random_fr(nvar, seed + 7), random_fr(nvar, seed + 8)
    records = {("params", 0): np.array([l, m, int(echo)], dtype=np.uint64), ("chal", 0): chal, ("point", 0): point}
    for k, per_party in tabs.items():
        for p, t in enumerate(per_party):
            if not echo or p == 0:
                records[(k, p)] = t
    return pp, tabs, chal, point, records


@pytest.mark.gpu
@pytest.mark.parametrize("l,m,echo", [(1, 10, False), (2, 9, False), (1, 11, True), (4, 8, True), (8, 9, True)])  # l = 8: 64 parties, the transform (zk_fr_ntt_map) branches
def test_cpp_host_equals_python_host_on_the_gpu(tmp_path, l, m, echo):
    import zkhip
    from zkhip.net import LeaderEchoNet, LocalTestNet

    pp, tabs, chal, point, records = _case(l, m, echo, 5200 + 17 * l + m)
    # a delegator share directory (examples/delegator.rs) for the file -> device -> file round trip of the C++ host
    import zkhip.serialize as ser

    share_dir = tmp_path / "shares"
    share_dir.mkdir()
    witness = po.SplitMix64(99).fr_vec(64)
    ser.delegator_write(str(share_dir), witness, pp)
    records[("dir", 0)] = np.frombuffer(str(share_dir).encode(), dtype=np.uint8)
    r, got = _run("gpu", records, tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    shares3 = ser.delegator_share(witness, pp)[3]
    assert _ints(got.pop(("share_file_on_device", 0))) == shares3  # Montgomery limbs in HBM == the file's values
    assert open(share_dir / "worker_3.copy", "rb").read() == open(share_dir / "worker_3", "rb").read()

    want = {}
    if echo:
        be = zkhip.Ctx(0)
        _python_party(be, LeaderEchoNet(pp.n), pp, m, tabs, chal, point, want)
    else:
        def party(net):
            be = zkhip.Ctx(0)  # one ctx per party thread (calls on one ctx are not concurrent)
            rec = {}
            _python_party(be, net, pp, m, tabs, chal, point, rec)
            return rec

        for rec in LocalTestNet.simulate_network_round(pp.n, party):
            want.update(rec)
    assert set(got) == set(want), (sorted(set(got) ^ set(want)))
    bad = [k for k in sorted(want) if got[k] != want[k]]
    assert not bad, bad
    if not echo:  # the sharded transcripts ARE the monolithic ones (every rank holds them)
        for p in range(pp.n):
            assert got[("shard_sc", p)] == got[("mono_sc", 0)] and got[("shard_sc_product", p)] == got[("mono_sc_product", 0)]
    # sanity of what was compared: a d_ result reaches the leader only, c_ results reach everyone
    assert len(want[("d_sumcheck_product", 0)]) == 96 * (m + pp.n.bit_length() - 1)
    if not echo:
        assert len(want[("d_sumcheck_product", 1)]) == 0 and len(want[("c_open", pp.n - 1)]) == 32 + 144 * (m + pp.l.bit_length() - 1)


# ---------------------------------------------------------------------------------------
# GPU: the protocol drivers of the C++ host (zkhost/hyperplonk.hpp) against zkhip.hyperplonk, transcript by transcript
# ---------------------------------------------------------------------------------------
def _flat_transcript(res):
    (gate_proofs, gate_commitments), (w_proofs, w_commits, w_opens) = res
    b = lambda *arrs: b"".join(np.ascontiguousarray(a, dtype=np.uint64).tobytes() for a in arrs)
    return {
        "gate_proofs": b(*gate_proofs),
        "gate_commitments": b"".join(b(c, v, prf) for c, (v, prf) in gate_commitments),
        "wiring_proofs": b(*w_proofs),
        "wiring_commits": b(*w_commits),
        "wiring_opens": b"".join(b(v, prf) for v, prf in w_opens),
    }


@pytest.mark.gpu
@pytest.mark.parametrize("which,l,n,echo", [("dhyperplonk", 1, 8, False), ("dhyperplonk", 1, 12, True), ("data_parallel", 1, 7, False), ("dhyperplonk", 2, 7, False),
                                            ("dpermcheck", 1, 8, False), ("cpermcheck", 1, 7, False), ("cpermcheck", 2, 7, True), ("cpermcheck", 8, 11, True),
                                            ("dhyperplonk", 8, 11, True)])  # l = 8: 64 parties (transform branches), leader echo
def test_cpp_protocol_drivers_equal_the_python_drivers(tmp_path, which, l, n, echo):
    import zkhip
    from zkhip.field import random_fr
    from zkhip.hyperplonk import PackedProvingParameters, cpermcheck, dhyperplonk, dpermcheck
    from zkhip.net import LeaderEchoNet, LocalTestNet
    from zkhip.pss import PackedSharingParams

    pp = PackedSharingParams(l)
    M, npar, chal_seed = 1 << n, pp.n, 4242
    code = {"dhyperplonk": 0, "data_parallel": 1, "dpermcheck": 2, "cpermcheck": 3}[which]
    records = {("params", 0): np.array([l, n, int(echo), code], dtype=np.uint64), ("seeds", 0): np.array([100 + p for p in range(npar)], dtype=np.uint64)}
    want = {}

    def party(net):
        p = net.party_id
        be = zkhip.Ctx(0)
        seed, run_seed = 100 + p, 200 + p
        pk = PackedProvingParameters.new(n, pp, be, seed=seed, chal_seed=chal_seed)
        rec = {(name, p): pk.tables[name].download((pk.lens[name], 4)) for name in pk.tables if not name.endswith("_evals")}
        # the per-run random data the Python drivers draw from their `seed` ("Jump from sky", dhyperplonk.rs:187-190, and :603)
        rec[("local_s_p", p)] = random_fr(4 * M // npar, run_seed * 31 + 1)
        rec[("local_s_l", p)] = random_fr(4 * M // npar // l, run_seed * 31 + 2)
        rec[("eq_top", p)] = random_fr(pp.n, run_seed * 31 + 3)
        rec[("s_data_parallel", p)] = random_fr(4 * M // l, run_seed * 31 + 4)
        if which == "cpermcheck":
            for i, name in enumerate(("mask", "unmask0", "unmask1", "unmask2")):
                rec[(name, p)] = random_fr(4 * (M // l), run_seed * 977 + 50 + i)
        if p == 0:
            rec[("chal", 0)] = np.concatenate([pk.challenge, pk.challenge_r1, pk.challenge_r2, pk.alpha[None], pk.beta[None], pk.gamma[None]])
        if which == "cpermcheck":
            res = cpermcheck(n, pk, pp, be, net, seed=run_seed)[0]
            res = (([], []), res)
        elif which == "dpermcheck":
            res = (([], []), dpermcheck(n, pk, pp, be, net, seed=run_seed)[0])
        else:
            res = dhyperplonk(n, pk, pp, be, net, seed=run_seed, data_parallel=which == "data_parallel")[0]
        tr = {(k, p): v for k, v in _flat_transcript(res).items()}
        tr[("comm", p)] = struct.pack("<QQ", net.upload, net.download)  # MPCNet::get_comm accounting of the whole run
        return rec, tr

    outs = [party(LeaderEchoNet(npar))] if echo else LocalTestNet.simulate_network_round(npar, party)
    for rec, tr in outs:
        records.update(rec)
        want.update(tr)
    r, got = _run("proof", records, tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    bad = [k for k in sorted(want) if got[k] != want[k]]
    assert not bad, bad
    assert len(want[("wiring_proofs", 0)]) > 0 and len(want[("wiring_opens", 0)]) > 0


# ---------------------------------------------------------------------------------------
# the example binary (host/examples/hyperplonk.cpp: the reference's hyperplonk/examples/*.rs in one program)
# ---------------------------------------------------------------------------------------
def _example():
    _need_gxx()
    host = os.path.join(ROOT, "scalable-collaborative-zksnark_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    return os.path.join(host, "bin", "hyperplonk")


def test_cpp_example_builds_and_refuses_to_run_without_a_gpu():
    exe = _example()
    import zkhip

    if zkhip.lib().zk_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu test")
    r = subprocess.run([exe, "--l", "1", "--n", "8"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("args,label,comm", [
    (["--l", "1", "--n", "12"], "Distributed HyperPlonk", "(959224, 959224)"),  # (the drivers' byte counts are compared with the Python drivers' in the parity test above)
    (["--l", "1", "--n", "10", "--mode", "threads"], "Distributed HyperPlonk", None),
    (["--l", "2", "--n", "9", "--mode", "threads", "--which", "cpermcheck"], "Collaborative Permcheck", None),
    (["--l", "1", "--n", "9", "--which", "dpermcheck"], "Distributed Permcheck", None),
    (["--l", "1", "--n", "9", "--which", "data-parallel", "--no-tables"], "Distributed HyperPlonk", None),
])
def test_cpp_example_on_the_gpu(args, label, comm):
    exe = _example()
    r = subprocess.run([exe] + args + ["--reps", "2"], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    ends = [l for l in r.stdout.splitlines() if l.startswith("  End: " + label)]
    assert len(ends) == 2 and all(float(l.split()[-2]) > 0 for l in ends), r.stdout
    comms = [l for l in r.stdout.splitlines() if l.startswith("Comm: ")]
    assert len(comms) == 2 and comms[0] == comms[1]  # the same exchanges every proof
    if comm:
        assert comms[0] == "Comm: " + comm


@pytest.mark.gpu
def test_cpp_example_rccl_mode_needs_one_gpu_per_party():
    import zkhip

    if zkhip.lib().zk_device_count() >= 8:
        pytest.skip("8 GPUs visible: the mode would run")
    r = subprocess.run([_example(), "--l", "1", "--n", "8", "--mode", "rccl"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "one GPU per party" in r.stderr


@pytest.mark.gpu
def test_cpp_rccl_net_and_zk_d_msm_on_a_world_of_one():
    """RcclNet (zk_comm_unique_id / zk_comm_init / zk_allgather / zk_alltoall) and d_msm as ONE zk_d_msm call, from the C++ host"""
    r = subprocess.run([_build(), "rccl1"], capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "rccl world of one" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("which,n", [("dhyperplonk", 20), ("data-parallel", 16), ("dpermcheck", 18)])
def test_cpp_host_full_size_proof_has_the_python_hosts_digest(which, n):
    """
    BASELINE configs[3] (n = 20, l = 1) from the compiled host: the same SplitMix64 parameter set built by both hosts
    (PackedProvingParameters::make / .new_splitmix), leader mode, SHA-256 over the whole transcript.  The Python driver's
    transcripts at this size are pinned by the anchored checks of tests/test_gpu_e2e_fullsize.py.
    """
    import hashlib

    import zkhip
    from zkhip.hyperplonk import PackedProvingParameters, dhyperplonk, dpermcheck
    from zkhip.net import LeaderEchoNet
    from zkhip.pss import PackedSharingParams

    r = subprocess.run([_example(), "--l", "1", "--n", str(n), "--which", which, "--reps", "2", "--digest"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    digests = [l.split()[-1] for l in r.stdout.splitlines() if l.startswith("transcript sha256")]
    assert len(digests) == 2 and digests[0] == digests[1]  # (and the proof is reproducible run to run)

    pp, be = PackedSharingParams(1), zkhip.Ctx(0)
    pk = PackedProvingParameters.new_splitmix(n, pp, be, seed=100, chal_seed=4242)
    net = LeaderEchoNet(8)
    if which == "dpermcheck":
        res = (([], []), dpermcheck(n, pk, pp, be, net)[0])
    else:
        res = dhyperplonk(n, pk, pp, be, net, data_parallel=which == "data-parallel")[0]
    t = _flat_transcript(res)
    want = hashlib.sha256(b"".join(t[k] for k in ("gate_proofs", "gate_commitments", "wiring_proofs", "wiring_commits", "wiring_opens"))).hexdigest()
    assert digests[0] == want


# ---------------------------------------------------------------------------------------
# `hyperplonk --check`: the compiled host verifies its own proofs (zkhost/verify.hpp) -- no Python, no oracle in the loop
# ---------------------------------------------------------------------------------------
def _check_lines(r):
    return [l for l in r.stdout.splitlines() if l.startswith("check: party ")]


@pytest.mark.gpu
@pytest.mark.parametrize("args,parties,transcripts", [
    (["--l", "1", "--n", "12"], 1, 6 + 1 + 3 + 3 * 9 + 3),                          # leader echo: every chain of the proof
    (["--l", "1", "--n", "10", "--mode", "threads"], 8, 6 + 1 + 3 + 3 * 7 + 3),     # 8 party threads: the leader's d_ chains use all parties' values
    (["--l", "2", "--n", "9", "--mode", "threads"], 16, 6 + 1 + 3 + 3 * 5 + 3),     # l = 2: the c_ tails have one clear round
    (["--l", "1", "--n", "9", "--which", "dpermcheck"], 1, 1 + 3 + 3 * 6 + 3),
    (["--l", "2", "--n", "9", "--mode", "threads", "--which", "cpermcheck"], 16, 6),
    (["--l", "1", "--n", "9", "--which", "data-parallel", "--no-tables"], 1, 6 + 1 + 3 + 3 * 6 + 3),
])
def test_cpp_example_checks_its_own_proofs(args, parties, transcripts):
    r = subprocess.run([_example()] + args + ["--reps", "2", "--check"], capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-1500:]
    lines = _check_lines(r)
    assert len(lines) == parties and all(" ok -- anchored" in l and "flipped limb rejected as" in l for l in lines), r.stdout[-2500:]
    leader = [l for l in lines if l.startswith("check: party 0 ")]
    assert len(leader) == 1 and f", {transcripts} transcripts pinned at both ends" in leader[0], leader
    if "cpermcheck" in args:  # the pipelined driver (one MSM pass, one kernel batch, repeated opens once) against single calls
        assert all(", 5 recomputed commits / opens" in l for l in lines), lines


@pytest.mark.gpu
def test_cpp_cpermcheck_pipelined_equals_the_call_by_call_form():
    """ZKHOST_CPERM_SERIAL=1 is the reference's order (10 c_commit, 12 c_open, 6 c_sumcheck_product one after the other): same digest and
    the same Comm totals as the default schedule, leader mode and 16 party threads"""
    for args in (["--l", "1", "--n", "12"], ["--l", "2", "--n", "9", "--mode", "threads"]):
        seen = []
        for serial in ("0", "1"):
            r = subprocess.run([_example()] + args + ["--which", "cpermcheck", "--reps", "1", "--digest", "--check"], capture_output=True, text=True, timeout=900,
                               env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ZKHOST_CPERM_SERIAL=serial))
            assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-1500:]
            seen.append(sorted({l for l in r.stdout.splitlines() if l.startswith("transcript sha256") or l.startswith("Comm:")}))
        assert seen[0] == seen[1] and len(seen[0]) == 2, (args, seen)


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--l", "1", "--n", "12"], ["--l", "1", "--n", "10", "--mode", "threads"]])
def test_cpp_example_rejects_a_flipped_limb(args):
    """--tamper: one limb of one t2 of the transcript under test is flipped; the self-check must fail (exit code 3) on every party"""
    r = subprocess.run([_example()] + args + ["--reps", "1", "--tamper"], capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 3, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    lines = _check_lines(r)
    assert lines and all("FAILED [gate[3]]" in l for l in lines), lines
    assert "the self-check failed" in r.stderr


@pytest.mark.gpu
def test_cpp_example_checks_the_n20_proof_with_8_party_threads():
    """BASELINE configs[3] per-party work, the real 8-party exchanges (threads sharing GPU 0), verified by the compiled host itself"""
    r = subprocess.run([_example(), "--l", "1", "--n", "20", "--mode", "threads", "--reps", "1", "--check"], capture_output=True, text=True, timeout=1800,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-1500:]
    lines = _check_lines(r)
    assert len(lines) == 8 and all(" ok -- anchored" in l for l in lines), lines


@pytest.mark.gpu
def test_cpp_example_checks_the_n24_proof():
    """BASELINE configs[4]'s per-party work (leader mode, 2^24 constraints) -- behind a memory guard"""
    import zkhip

    probe = zkhip.Ctx(0)
    free, _total = probe.mem_info()
    probe.close()
    if free < (96 << 30):
        pytest.skip(f"only {free >> 30} GiB of HBM free: the n = 24 run wants ~70 GiB")
    r = subprocess.run([_example(), "--l", "1", "--n", "24", "--reps", "1", "--check"], capture_output=True, text=True, timeout=1800, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-1500:]
    lines = _check_lines(r)
    assert len(lines) == 1 and " ok -- anchored" in lines[0] and f", {6 + 1 + 3 + 3 * 21 + 3} transcripts pinned" in lines[0], lines


# ---------------------------------------------------------------------------------------
# the reference's own unit tests restated for the C++ host (tests/native/host_props.cpp): oracle-free properties
# ---------------------------------------------------------------------------------------
def _props():
    _need_gxx()
    subprocess.check_call(["make", "-C", NATIVE, "-s", "host_props"])
    return os.path.join(NATIVE, "host_props")


def test_cpp_reference_unit_tests_host_part():
    """pss.rs tests, utils/operator.rs:42-49, dacc_product.rs:442-448, dsumcheck.rs:541-588 (the shipped verifiers accept and reject), :591-621 -- no GPU needed"""
    r = subprocess.run([_props(), "host"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 failure(s)" in r.stdout and r.stdout.count(" ok\n") == 7, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_reference_unit_tests_gpu_part():
    """dacc_product.rs:450-466, pss.rs test_group_addition, dmsm.rs:73-138, dsumcheck.rs:541-588,623-859, dpoly_comm.rs:502-581 (no pairing)"""
    r = subprocess.run([_props(), "gpu"], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "0 failure(s)" in r.stdout and r.stdout.count(" ok\n") == 8, r.stdout[-3000:] + r.stderr[-3000:]
