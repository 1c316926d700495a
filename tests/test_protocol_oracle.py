"""
The protocol drivers against an INDEPENDENT statement of the reference's call sequence.

oracle/pyoracle.py `dhyperplonk_all`, `dpermcheck_all`, `cpermcheck_all` restate hyperplonk/src/dhyperplonk.rs:159-571, 573-960,
962-1247, 1249-1385 as one sequential call per reference line over the *_all primitives -- no queues, no batching, no
de-duplication.  The product's drivers (zkhip/hyperplonk.py, host/zkhost/hyperplonk.hpp) run the same calls re-scheduled
(pipelined MSM passes, one kernel batch for steps 2-4, de-duplicated commitments).  Here every party's FULL output of a product
driver is compared with the oracle driver's, position by position:

  * CPU  : the Python host over tests/oracle_backend.py (host logic: which table, which challenge slice, which SRS level, which
           output position), 8 and 16 real parties + the `leader` (no-`comm`) mode;
  * GPU  : the Python host over libzkhip.so and the compiled host (`host/bin/hyperplonk --dump`), same comparisons.

`test_a_swapped_challenge_slice_or_output_position_is_caught` shows the pin has teeth: drivers with one challenge slice or one
output position changed on purpose -- mistakes BOTH hosts could share, which the host-against-host tests cannot see -- fail here.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

import protocol_parity as pq
from oracle_backend import OracleBackend
from zkhip import hyperplonk as hp
from zkhip.net import LeaderEchoNet, LocalTestNet
from zkhip.pss import PackedSharingParams

CHAL = 777


def _host_run(which, n, l, make_backend, comm=True, splitmix=False, seed0=100, run0=200, chal=CHAL):
    """every party's output of the Python host's driver (party threads; `leader` mode: the one party of the echo net)"""
    pp = PackedSharingParams(l)
    make = hp.PackedProvingParameters.new_splitmix if splitmix else hp.PackedProvingParameters.new

    def party(net):
        be = make_backend()
        p = net.party_id
        pk = make(n, pp, be, seed=seed0 + p, chal_seed=chal, window_tables=not isinstance(be, OracleBackend))
        if which == "cpermcheck":
            pq.cperm_masks(pk, be, n, l, (seed0 if splitmix else run0) + p, splitmix)
            return hp.cpermcheck(n, pk, pp, be, net, seed=run0 + p)[0]
        if which == "dpermcheck":
            return hp.dpermcheck(n, pk, pp, be, net, seed=run0 + p)[0]
        return hp.dhyperplonk(n, pk, pp, be, net, seed=run0 + p, data_parallel=which == "data-parallel")[0]

    if not comm:
        return [party(LeaderEchoNet(pp.n))]
    return LocalTestNet.simulate_network_round(pp.n, party)


def _oracle_run(which, n, l, comm=True, splitmix=False, seed0=100, run0=200, chal=CHAL):
    npar = 8 * l
    seeds = [seed0 + p for p in range(npar)]
    runs = [run0 + p for p in range(npar)]
    return pq.oracle_run(which, n, l, seeds, chal, runs, comm=comm, splitmix=splitmix, mask_seeds=seeds if splitmix else runs)


def _compare(which, host, want):
    assert len(host) == len(want)
    for p, (h, w) in enumerate(zip(host, want)):
        (pq.compare_dhyperplonk if which in ("dhyperplonk", "data-parallel") else pq.compare_wiring)(h, w, f"{which} party {p}")


CASES = [("dhyperplonk", 5, 1), ("dhyperplonk", 6, 1), ("data-parallel", 5, 1), ("dpermcheck", 5, 1), ("cpermcheck", 5, 1),
         ("dhyperplonk", 6, 2), ("cpermcheck", 6, 2)]  # (l = 2 needs n >= 6: the leader tree of 16 parties wants SRS level 4, dpoly_comm.rs:239)


@pytest.mark.parametrize("which,n,l", CASES)
def test_python_host_on_the_cpu_backend_equals_the_straight_line_oracle(which, n, l):
    _compare(which, _host_run(which, n, l, OracleBackend), _oracle_run(which, n, l))


@pytest.mark.parametrize("which,n,l", [("dhyperplonk", 5, 1), ("dpermcheck", 5, 1), ("cpermcheck", 5, 1), ("cpermcheck", 6, 2), ("dhyperplonk", 6, 2)])
def test_leader_mode_equals_the_oracles_no_comm_form(which, n, l):
    """`leader` mode (BASELINE configs[0]): the no-`comm` fake of serializing_net.rs:144-264, incl. c_acc_product_and_share's own
    placeholders (dacc_product.rs:194-202)"""
    _compare(which, _host_run(which, n, l, OracleBackend, comm=False), _oracle_run(which, n, l, comm=False))


def test_splitmix_parameter_set_equals_the_oracle():
    """the parameter set the compiled host builds (PackedProvingParameters::make), through the Python host"""
    _compare("dhyperplonk", _host_run("dhyperplonk", 5, 1, OracleBackend, splitmix=True, chal=4242), _oracle_run("dhyperplonk", 5, 1, splitmix=True, chal=4242))


def test_a_swapped_challenge_slice_or_output_position_is_caught(monkeypatch):
    """
    Mistakes a host-against-host comparison cannot see, because both hosts (and both backends of one host) would share them:
    (1) the layered sumchecks / opens of step 2.e.2 given challenge_r2[i + 1..] instead of [i..] (dhyperplonk.rs:427-461),
    (2) two wiring commitments swapped in the output list (:363-380), (3) 2.c run on challenge_r2 instead of challenge_r1 (:304).
    Each is injected into the Python host's driver; the comparison with the oracle driver must fail and name the position.
    """
    want = _oracle_run("dpermcheck", 5, 1)

    # (1) a shifted challenge slice in the layered part
    real_enqueue = hp._wiring_enqueue

    def shifted(n, pk, pp, be, net, *a, **k):
        class PK:  # the parameter set with challenge_r2 shifted by one from entry 1 on (what `[i + 1..]` would read)
            def __getattr__(self, name):
                return getattr(pk, name)

        view = PK()
        r2 = np.array(pk.challenge_r2, copy=True)
        r2[2:] = np.concatenate([pk.challenge_r2[3:], pk.challenge_r2[1:2]])
        view.challenge_r2 = r2
        return real_enqueue(n, view, pp, be, net, *a, **k)

    monkeypatch.setattr(hp, "_wiring_enqueue", shifted)
    with pytest.raises(AssertionError, match=r"wiring_(proofs|opens)\["):
        _compare("dpermcheck", _host_run("dpermcheck", 5, 1, OracleBackend), want)
    monkeypatch.setattr(hp, "_wiring_enqueue", real_enqueue)

    # (2) two commitments swapped in the output
    good = _host_run("dpermcheck", 5, 1, OracleBackend)
    _compare("dpermcheck", good, want)
    swapped = [(pr, list(cm), op) for pr, cm, op in good]
    swapped[0][1][3], swapped[0][1][4] = swapped[0][1][4], swapped[0][1][3]
    with pytest.raises(AssertionError, match=r"party 0: wiring_commits\[3\]"):
        _compare("dpermcheck", swapped, want)

    # (3) the wrong challenge vector for 2.c
    def wrong_2c(n, pk, pp, be, net, *a, **k):
        class PK:
            def __getattr__(self, name):
                return getattr(pk, name)

        view = PK()
        view.challenge_r1 = pk.challenge_r2
        return real_enqueue(n, view, pp, be, net, *a, **k)

    monkeypatch.setattr(hp, "_wiring_enqueue", wrong_2c)
    with pytest.raises(AssertionError, match=r"wiring_proofs\[0\]"):
        _compare("dpermcheck", _host_run("dpermcheck", 5, 1, OracleBackend), want)


# ---------------------------------------------------------------- GPU: the same comparisons through libzkhip.so
GPU_CASES = CASES + [("dpermcheck", 6, 2), ("data-parallel", 6, 2), ("dhyperplonk", 8, 4)]


@pytest.mark.gpu
@pytest.mark.parametrize("which,n,l", GPU_CASES)
def test_python_host_on_the_gpu_equals_the_straight_line_oracle(which, n, l):
    import zkhip

    _compare(which, _host_run(which, n, l, lambda: zkhip.Ctx(0)), _oracle_run(which, n, l))


@pytest.mark.gpu
@pytest.mark.parametrize("which,n,l", [("dhyperplonk", 6, 1), ("cpermcheck", 5, 1), ("cpermcheck", 6, 2)])
def test_python_host_leader_mode_on_the_gpu_equals_the_oracles_no_comm_form(which, n, l):
    import zkhip

    _compare(which, _host_run(which, n, l, lambda: zkhip.Ctx(0), comm=False), _oracle_run(which, n, l, comm=False))


def _read_dump(path):
    """host/examples/hyperplonk.cpp dump_transcript -> the nesting the Python host returns"""
    raw, off = open(path, "rb").read(), [0]

    def u64():
        v = struct.unpack_from("<Q", raw, off[0])[0]
        off[0] += 8
        return v

    def arr(count, width):
        a = np.frombuffer(raw, dtype="<u8", count=count * width, offset=off[0]).astype(np.uint64).reshape(count, width)
        off[0] += 8 * count * width
        return a

    def proofs():
        return [arr(3 * u64(), 4).reshape(-1, 3, 4) for _ in range(u64())]

    def opening():
        v = arr(1, 4)[0]
        return (v, arr(u64(), 18))

    gate = proofs()
    gate_com = [(arr(1, 18)[0], opening()) for _ in range(u64())]
    wp = proofs()
    wc = list(arr(u64(), 18))
    wo = [opening() for _ in range(u64())]
    assert off[0] == len(raw)
    return (gate, gate_com), (wp, wc, wo)


def _cpp_run(tmp_path, which, n, l, mode, env=None):
    import shutil

    if shutil.which("g++") is None:
        pytest.skip("no g++ on this box: the C++ host is not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "scalable-collaborative-zksnark_amd", "host"), "-s"])
    exe = os.path.join(root, "scalable-collaborative-zksnark_amd", "host", "bin", "hyperplonk")
    prefix = str(tmp_path / "t")
    r = subprocess.run([exe, "--l", str(l), "--n", str(n), "--mode", mode, "--which", which, "--reps", "2", "--dump", prefix], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    parties = 8 * l if mode != "leader" else 1
    return [_read_dump(f"{prefix}.party{p}.bin") for p in range(parties)]


@pytest.mark.gpu
@pytest.mark.parametrize("which,n,l,mode", [("dhyperplonk", 5, 1, "threads"), ("dhyperplonk", 6, 1, "threads"), ("data-parallel", 5, 1, "threads"), ("dpermcheck", 5, 1, "threads"),
                                            ("cpermcheck", 5, 1, "threads"), ("dhyperplonk", 6, 2, "threads"), ("cpermcheck", 6, 2, "threads"),
                                            ("dhyperplonk", 6, 1, "leader"), ("cpermcheck", 5, 1, "leader"), ("cpermcheck", 6, 2, "leader"),
                                            # 32 real parties (l = 4 needs n >= 8: the leader tree of 32 parties wants SRS level 5) and the leader of 64
                                            ("dhyperplonk", 8, 4, "threads"), ("cpermcheck", 8, 4, "threads"), ("dhyperplonk", 10, 8, "leader")])
def test_compiled_host_equals_the_straight_line_oracle(tmp_path, which, n, l, mode):
    if (which, n, l) in (("cpermcheck", 8, 4), ("dhyperplonk", 10, 8)) and os.environ.get("ZK_SLOW_TESTS") != "1":
        pytest.skip("the oracle side of this case takes 70-85 s of python big-ints: ZK_SLOW_TESTS=1 runs it (passed in profiles/r06z_pytest_gpu.txt)")
    """host/bin/hyperplonk (zkhost/hyperplonk.hpp): every party's dumped transcript against the oracle driver on the same SplitMix64
    parameter set (PackedProvingParameters::make = PackedProvingParameters.new_splitmix: seed 100 + p, challenges 4242)"""
    got = _cpp_run(tmp_path, which, n, l, mode)
    want = _oracle_run(which, n, l, comm=mode != "leader", splitmix=True, chal=4242)
    full = which in ("dhyperplonk", "data-parallel")
    for p, (g, w) in enumerate(zip(got, want)):
        if full:
            pq.compare_dhyperplonk(g, w, f"{which} party {p}")
        else:
            assert g[0] == ([], [])
            pq.compare_wiring(g[1], w, f"{which} party {p}")


@pytest.mark.gpu
def test_compiled_host_over_the_rccl_double_equals_the_straight_line_oracle(tmp_path):
    """the same through `--mode rccl` (zkhost::RcclNet: zk_allgather / zk_alltoall / zk_d_msm inside the ctx) over the test double of
    librccl (tests/native/fake_rccl.cpp: 8 ranks sharing the one GPU)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    native = os.path.join(root, "tests", "native")
    subprocess.check_call(["make", "-C", native, "-s", "fake_rccl/librccl.so.1"])
    env = {"LD_LIBRARY_PATH": os.path.join(native, "fake_rccl") + ":" + os.environ.get("LD_LIBRARY_PATH", "")}
    for which in ("dhyperplonk", "cpermcheck"):
        import shutil

        if shutil.which("g++") is None:
            pytest.skip("no g++")
        exe = os.path.join(root, "scalable-collaborative-zksnark_amd", "host", "bin", "hyperplonk")
        prefix = str(tmp_path / which)
        r = subprocess.run([exe, "--l", "1", "--n", "6", "--mode", "rccl", "--share-gpus", "--which", which, "--reps", "1", "--dump", prefix], capture_output=True,
                           text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        want = _oracle_run(which, 6, 1, splitmix=True, chal=4242)
        for p in range(8):
            g = _read_dump(f"{prefix}.party{p}.bin")
            if which == "dhyperplonk":
                pq.compare_dhyperplonk(g, want[p], f"rccl {which} party {p}")
            else:
                pq.compare_wiring(g[1], want[p], f"rccl {which} party {p}")
