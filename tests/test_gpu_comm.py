"""
The party exchanges behind the C ABI (zk_comm_* / zk_allgather / zk_alltoall / zk_gather / zk_scatter /
zk_d_msm, include/zkhip.h) on the GPU box.

  * world size 1: a real RCCL communicator on GPU 0 -- API, layout and the d_msm composite against the oracle;
  * world size min(8, device_count): one process per GPU over RCCL/xGMI, the protocol primitives
    (d_msm, d_sumcheck_product, sharded MSM / sumcheck) against the oracle.  Skips itself below 2 GPUs.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import jac_norm_to_affine, pt_ints, pt_mont, rand_fr, synthetic_bases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cctx():
    """a ctx of its own carrying a world-1 communicator"""
    import zkhip

    c = zkhip.Ctx(0)
    c.comm_init(0, 1, c.comm_unique_id())
    yield c
    c.close()


def test_collectives_world1(cctx):
    assert (cctx.comm_rank, cctx.comm_size) == (0, 1)
    a = rand_fr(37, 5)
    d = cctx.to_device(a)
    for fn in (lambda: cctx.allgather(d, a.nbytes), lambda: cctx.alltoall(d, a.nbytes), lambda: cctx.gather(d, a.nbytes, 0),
               lambda: cctx.scatter(d, a.nbytes, 0)):
        out = fn()
        assert (out.download(a.shape) == a).all()


def test_comm_errors(ctx, cctx):
    import zkhip
    from zkhip._lib import ZK_ERR_COMM, ZK_ERR_INVALID

    d = ctx.to_device(rand_fr(4, 1))
    with pytest.raises(zkhip.ZkError) as e:  # the session ctx has no communicator
        ctx.allgather(d, 128)
    assert e.value.code == ZK_ERR_COMM
    with pytest.raises(zkhip.ZkError) as e:
        cctx.gather(cctx.to_device(rand_fr(4, 1)), 128, 3)
    assert e.value.code == ZK_ERR_INVALID
    with pytest.raises(zkhip.ZkError) as e:  # a second communicator on the same ctx
        cctx.comm_init(0, 1, cctx.comm_unique_id())
    assert e.value.code == ZK_ERR_INVALID


def test_d_msm_world1_against_oracle(cctx, co):
    """one party: out = coeff * MSM(bases, lambda * scalars); batch of two sizes"""
    import pyoracle as po

    lam, coeff = 0x1234567890ABCDEF1122334455667788, 0x0FEDCBA987654321
    outs, exp = None, []
    srs_l, sc_l, lens = [], [], []
    for n, seed in ((300, 3), (1024, 4)):
        bases, _ = synthetic_bases(n, seed)
        sc = rand_fr(n, seed + 10)
        srs_l.append(cctx.srs_register(bases))
        sc_l.append(cctx.to_device(sc))
        lens.append(n)
        plain = pt_ints(co.msm_g1(bases, sc))
        exp.append(po.g1_mul(plain, lam * coeff % po.R_MOD))
    co_limbs = np.array([[(coeff >> (64 * i)) & (2**64 - 1) for i in range(4)]], dtype=np.uint64)
    lam_m = np.array(po.fr_to_mont_limbs(lam), dtype=np.uint64)
    outs = cctx.d_msm(srs_l, sc_l, lens, co_limbs, lam_mont=lam_m)
    for k in range(2):
        assert pt_ints(jac_norm_to_affine(outs[k])) == exp[k]
    # without the pre-scaling
    outs = cctx.d_msm(srs_l, sc_l, lens, co_limbs)
    for k in range(2):
        assert po.g1_mul(pt_ints(jac_norm_to_affine(outs[k])), lam) == exp[k]


WORKER = r"""
import os, sys
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
import pyoracle as po
from helpers import pt_ints, pt_mont, jac_norm_to_affine
import zkhip
from zkhip import dist_primitive as dp, sharding as sh
from zkhip.net import RcclNet, TorchDistNet
from zkhip.pss import PackedSharingParams
rank, world, lrank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(lrank)
dist.init_process_group("nccl", device_id=torch.device("cuda", lrank))
ctx = zkhip.Ctx(lrank)
net = RcclNet.from_torch_dist(ctx)              # the C-ABI communicator (RCCL inside the ctx)
tnet = TorchDistNet(device=torch.device("cuda", lrank))  # torch.distributed's RCCL, same semantics
W, p = net.n_parties, net.party_id
to_m = lambda xs: np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)
ints = lambda a: [po.fr_from_mont_limbs(r) for r in np.asarray(a).reshape(-1, 4)]
rng = po.SplitMix64(4321)          # same stream on every rank: every rank can rebuild all inputs
# raw collectives: every rank's payload is recognisable
mine = np.full((5, 4), p + 1, dtype=np.uint64)
got = net.all_gather(mine)
assert all((got[q] == q + 1).all() for q in range(W))
got = net.all_to_all([np.full((3, 4), 100 * p + q, dtype=np.uint64) for q in range(W)])
assert all((got[q] == 100 * q + p).all() for q in range(W))
d = ctx.to_device(mine)
g = ctx.gather(d, mine.nbytes, W - 1)
if p == W - 1:
    h = g.download((W, 5, 4))
    assert all((h[q] == q + 1).all() for q in range(W))
src = ctx.to_device(np.arange(W * 8, dtype=np.uint64)) if p == 0 else None
r = ctx.scatter(src, 64, 0)
assert (r.download((8,)) == np.arange(8 * p, 8 * p + 8)).all()
# protocol primitives over both nets
n = 6
pf = [rng.fr_vec(1 << n) for _ in range(W)]
pg = [rng.fr_vec(1 << n) for _ in range(W)]
s = W.bit_length() - 1
ch = rng.fr_vec(n + s)
for nt in (net, tnet):
    out = dp.d_sumcheck_product(ctx, ctx.to_device(to_m(pf[p])), ctx.to_device(to_m(pg[p])), 1 << n, to_m(ch), nt)
    if p == 0:
        assert [tuple(ints(t)) for t in out] == po.d_sumcheck_product_all(pf, pg, ch)
    else:
        assert len(out) == 0
    N = 1 << 10
    full_f, full_g, chs = rng.fr_vec(N), rng.fr_vec(N), rng.fr_vec(10)
    got = sh.sharded_sumcheck_product(ctx, ctx.to_device(sh.cyclic_shard(to_m(full_f), p, W)), ctx.to_device(sh.cyclic_shard(to_m(full_g), p, W)), N // W, to_m(chs), nt)
    assert [tuple(ints(t)) for t in got] == po.sumcheck_product(full_f, full_g, chs)
    pts, scs = po.g1_bases(64, 77), rng.fr_vec(64)
    per = 64 // W
    got = sh.sharded_msm(ctx, ctx.srs_register(np.array([pt_mont(P) for P in pts[p*per:(p+1)*per]])), ctx.to_device(to_m(scs[p*per:(p+1)*per])), per, nt)
    assert pt_ints(jac_norm_to_affine(got)) == po.g1_msm(pts, scs)
if W == 8:                           # the l = 1, 8-party d_msm: zk_d_msm (C ABI) and the torch.distributed path
    pp, opp = PackedSharingParams(1), po.PackedSharingParams(1)
    bases = [[po.g1_bases(40, 50 + q)] for q in range(W)]
    scal = [[rng.fr_vec(40)] for _ in range(W)]
    exp = po.d_msm_all(bases, scal, opp)
    srs = ctx.srs_register(np.array([pt_mont(P) for P in bases[p][0]]))
    for nt in (net, tnet):
        got = dp.d_msm(ctx, [srs], [ctx.to_device(to_m(scal[p][0]))], [40], pp, nt)
        assert pt_ints(jac_norm_to_affine(got[0])) == exp[p][0]
dist.barrier()
print("RANK_OK", p)
ctx.close()
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_multi_rank_protocol():
    import zkhip

    ndev = zkhip.lib().zk_device_count()  # (not torch: its bundled HIP/RCCL must not be mapped into this process after ours)
    if ndev < 2:
        pytest.skip(f"{ndev} GPU visible: the multi-rank RCCL test needs at least 2")
    world = 8 if ndev >= 8 else (4 if ndev >= 4 else 2)
    script = "ROOT = %r\n" % ROOT + WORKER
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "--no-python", sys.executable, "-c", script]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("RANK_OK") == world


def test_rccl_worker_script_world1():
    """the same worker under torchrun with ONE rank: keeps the script itself exercised on 1-GPU boxes"""
    script = "ROOT = %r\n" % ROOT + WORKER
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "--no-python", sys.executable, "-c", script]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("RANK_OK") == 1


def _party_body(ctx, net, co):
    """the protocol primitives of WORKER on a ready (ctx, net) pair: one thread per party under zk_comm_init_all"""
    import pyoracle as po
    from zkhip import dist_primitive as dp, sharding as sh
    from zkhip.pss import PackedSharingParams

    W, p = net.n_parties, net.party_id
    to_m = lambda xs: np.array([po.fr_to_mont_limbs(x) for x in xs], dtype=np.uint64).reshape(-1, 4)
    ints = lambda a: [po.fr_from_mont_limbs(r) for r in np.asarray(a).reshape(-1, 4)]
    rng = po.SplitMix64(8642)  # the same stream for every party: each can rebuild all inputs
    mine = np.full((5, 4), p + 1, dtype=np.uint64)
    got = net.all_gather(mine)
    assert all((got[q] == q + 1).all() for q in range(W))
    got = net.all_to_all([np.full((3, 4), 100 * p + q, dtype=np.uint64) for q in range(W)])
    assert all((got[q] == 100 * q + p).all() for q in range(W))
    d = ctx.to_device(mine)
    g = ctx.gather(d, mine.nbytes, W - 1)  # a remote root (serializing_net.rs:41-72)
    if p == W - 1:
        h = g.download((W, 5, 4))
        assert all((h[q] == q + 1).all() for q in range(W))
    src = ctx.to_device(np.arange(W * 8, dtype=np.uint64)) if p == W - 1 else None
    r = ctx.scatter(src, 64, W - 1)  # scatter from a remote root (serializing_net.rs:98-122)
    assert (r.download((8,)) == np.arange(8 * p, 8 * p + 8)).all()
    # HBM -> HBM all-gather (step 2.a of the protocol, dhyperplonk.rs:270-294)
    big = np.full((1 << 12, 4), 7 * p + 3, dtype=np.uint64)
    ag = net.all_gather_device(ctx.to_device(big), big.nbytes).download((W, 1 << 12, 4))
    assert all((ag[q] == 7 * q + 3).all() for q in range(W))
    n = 6
    pf = [rng.fr_vec(1 << n) for _ in range(W)]
    pg = [rng.fr_vec(1 << n) for _ in range(W)]
    s = W.bit_length() - 1
    ch = rng.fr_vec(n + s)
    out = dp.d_sumcheck_product(ctx, ctx.to_device(to_m(pf[p])), ctx.to_device(to_m(pg[p])), 1 << n, to_m(ch), net)
    if p == 0:
        assert [tuple(ints(t)) for t in out] == po.d_sumcheck_product_all(pf, pg, ch)
    else:
        assert len(out) == 0
    pts, scs = po.g1_bases(64, 77), rng.fr_vec(64)
    per = 64 // W
    got = sh.sharded_msm(ctx, ctx.srs_register(np.array([pt_mont(P) for P in pts[p * per : (p + 1) * per]])), ctx.to_device(to_m(scs[p * per : (p + 1) * per])), per, net)
    assert pt_ints(jac_norm_to_affine(got)) == po.g1_msm(pts, scs)
    if W == 8:  # the l = 1, 8-party d_msm through ONE C-ABI call per party (zk_d_msm)
        pp, opp = PackedSharingParams(1), po.PackedSharingParams(1)
        bases = [[po.g1_bases(40, 50 + q)] for q in range(W)]
        scal = [[rng.fr_vec(40)] for _ in range(W)]
        exp = po.d_msm_all(bases, scal, opp)
        srs = ctx.srs_register(np.array([pt_mont(P) for P in bases[p][0]]))
        got = dp.d_msm(ctx, [srs], [ctx.to_device(to_m(scal[p][0]))], [40], pp, net)
        assert pt_ints(jac_norm_to_affine(got[0])) == exp[p][0]
    # error propagation of zk_d_msm: the LAST party asks for more scalars than it has bases; nobody may hang --
    # that party reports its own ZK_ERR_LENGTH, every other party a ZK_ERR_COMM naming it
    import zkhip
    from zkhip._lib import ZK_ERR_COMM, ZK_ERR_LENGTH

    srs = ctx.srs_generate(11 + p, 13, 64)
    sc = ctx.to_device(to_m(rng.fr_vec(64)))
    ones = np.tile(np.array([1, 0, 0, 0], dtype=np.uint64), (W, 1))
    with pytest.raises(zkhip.ZkError) as e:
        ctx.d_msm([srs], [sc], [128 if p == W - 1 else 64], ones)
    assert e.value.code == (ZK_ERR_LENGTH if p == W - 1 else ZK_ERR_COMM), str(e.value)
    ok = ctx.d_msm([srs], [sc], [64], ones)  # the communicator is still usable afterwards
    assert ok.shape == (1, 18)
    return True


def _init_all_world(world, devices, co):
    """`world` parties as threads of this process, party p on devices[p], communicators from zk_comm_init_all; every party runs _party_body"""
    import threading

    import zkhip
    from zkhip.net import RcclNet

    ctxs = [zkhip.Ctx(d) for d in devices]
    try:
        nets = RcclNet.from_init_all(ctxs)
        assert [c.comm_rank for c in ctxs] == list(range(world)) and all(c.comm_size == world for c in ctxs)
        res, errs = [None] * world, []

        def run(p):
            try:
                res[p] = _party_body(ctxs[p], nets[p], co)
            except BaseException as e:  # noqa: BLE001
                errs.append((p, e))

        th = [threading.Thread(target=run, args=(p,)) for p in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=600)
        assert not any(t.is_alive() for t in th), "a party is still blocked in an exchange"
        assert not errs, errs
        assert all(res)
    finally:
        for c in ctxs:
            c.close()


def test_comm_init_all_one_process_party_threads(co):
    """
    zk_comm_init_all (the reference's one-task-per-party model, mpc-net/src/multi.rs:330-352): ONE process, a ctx per
    visible GPU (1 on the builder's boxes, 8 on the real node), one host thread per party running the same protocol
    primitives as the torchrun worker -- exchanges through the communicator in each ctx.
    """
    import zkhip

    ndev = zkhip.lib().zk_device_count()
    world = 8 if ndev >= 8 else (4 if ndev >= 4 else (2 if ndev >= 2 else 1))
    _init_all_world(world, list(range(world)), co)


# ---------------------------------------------------------------------------------------
# world size 8 on a ONE-GPU box: the library's RCCL calls against a test double of librccl.so.1 (tests/native/fake_rccl.cpp:
# the entry points csrc/zk_comm.cpp resolves, with the real semantics, between host threads that share the GPU -- the real RCCL
# refuses two ranks on one device).  Everything of ours around the collectives runs as on the 8-GPU node; the wire does not.
# ---------------------------------------------------------------------------------------
FAKE_DIR = os.path.join(ROOT, "tests", "native", "fake_rccl")


def _fake_env():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "native"), "-s", "fake_rccl/librccl.so.1"])
    return dict(os.environ, LD_LIBRARY_PATH=FAKE_DIR + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0",
                PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scalable-collaborative-zksnark_amd")]))


def test_world8_protocol_primitives_over_the_rccl_test_double():
    """gather / scatter with a remote root, all-to-all, the HBM all-gather, d_sumcheck_product, a sharded MSM, the 8-party zk_d_msm and
    its error propagation -- _party_body at world 8, eight ctxs on GPU 0, in a process of its own (no torch: its bundled RCCL must
    not be the copy the library finds)"""
    code = ("import sys, coracle; import test_gpu_comm as t\n"
            "assert 'torch' not in sys.modules\n"
            "t._init_all_world(8, [0] * 8, coracle)\n"
            "print('WORLD8_OK')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1200, env=_fake_env(), cwd=os.path.join(ROOT, "tests"))
    assert r.returncode == 0 and "WORLD8_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "[fake_rccl] the test double of librccl.so.1 is in use: 8 ranks" in r.stderr


_PY_PROTOCOL = r"""
import hashlib, sys, threading
import numpy as np
import zkhip
from zkhip.hyperplonk import PackedProvingParameters, cpermcheck, dhyperplonk
from zkhip.net import LocalTestNet, RcclNet
from zkhip.pss import PackedSharingParams
from zkhip.field import random_fr
assert 'torch' not in sys.modules
n, which = int(sys.argv[1]), sys.argv[2]
pp = PackedSharingParams(1)

def digest(res):
    h = hashlib.sha256()
    def feed(x):
        if isinstance(x, np.ndarray): h.update(np.ascontiguousarray(x, dtype=np.uint64).tobytes())
        elif isinstance(x, (list, tuple)):
            for e in x: feed(e)
    feed(res)
    return h.hexdigest()

def party(ctx, net):
    pk = PackedProvingParameters.new_splitmix(n, pp, ctx, seed=100 + net.party_id, chal_seed=4242)
    if which == 'cpermcheck':
        res = cpermcheck(n, pk, pp, ctx, net, seed=3 + net.party_id)[0]  # (the masks are drawn from `seed` on first use)
    else:
        res = dhyperplonk(n, pk, pp, ctx, net, seed=7 + net.party_id)[0]
    return digest(res), (net.upload, net.download)

def over(nets_of):
    ctxs = [zkhip.Ctx(0) for _ in range(8)]
    nets = nets_of(ctxs)
    out, errs = [None] * 8, []
    def run(p):
        try: out[p] = party(ctxs[p], nets[p])
        except BaseException as e: errs.append((p, repr(e)))
    th = [threading.Thread(target=run, args=(p,)) for p in range(8)]
    [t.start() for t in th]; [t.join(900) for t in th]
    assert not errs and not any(t.is_alive() for t in th), errs
    for c in ctxs: c.close()
    return out

a = over(RcclNet.from_init_all)
from zkhip.net import _LocalHub
lh = _LocalHub(8)
b = over(lambda ctxs: [LocalTestNet(lh, p) for p in range(8)])
assert [x[0] for x in a] == [x[0] for x in b], 'transcripts over RcclNet differ from the thread net'
assert a[0][1] == b[0][1], ('bytes', a[0][1], b[0][1])
print('PY_WORLD8_OK', a[0][0][:16], a[0][1])
"""


@pytest.mark.parametrize("n,which", [(10, "dhyperplonk"), (9, "cpermcheck")])
def test_python_host_protocol_over_the_rccl_test_double_equals_thread_net(n, which):
    """the Python host's drivers with zkhip.net.RcclNet (zk_comm_init_all, eight ctxs on GPU 0) against the test double: every party's
    transcript and the leader's byte counters equal the run over LocalTestNet -- zk_d_msm, zk_allgather (HBM), zk_alltoall (the
    device-resident c_acc_product_and_share) at world 8"""
    r = subprocess.run([sys.executable, "-c", _PY_PROTOCOL, str(n), which], capture_output=True, text=True, timeout=1500, env=_fake_env(), cwd=os.path.join(ROOT, "tests"))
    assert r.returncode == 0 and "PY_WORLD8_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "[fake_rccl] the test double of librccl.so.1 is in use: 8 ranks" in r.stderr


def _hyperplonk(args, env, timeout=1200):
    host = os.path.join(ROOT, "scalable-collaborative-zksnark_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    return subprocess.run([os.path.join(host, "bin", "hyperplonk")] + args, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("args,parties", [
    (["--l", "1", "--n", "12"], 8),                             # dhyperplonk: zk_d_msm, the HBM all-gather of step 2.a, every typed exchange
    (["--l", "1", "--n", "10", "--which", "cpermcheck"], 8),    # c_acc_product_and_share: zk_alltoall on device buffers
    (["--l", "2", "--n", "10"], 16),                            # 16 parties
])
def test_cpp_host_rccl_mode_over_the_test_double_equals_thread_mode(args, parties):
    """`hyperplonk --mode rccl` (party p = a ctx with an in-ctx communicator, zkhost::RcclNet) on ONE GPU against the test double:
    every party's self-check passes and the transcript digest equals the one of --mode threads (exchanges through host memory)"""
    env = _fake_env()
    r = _hyperplonk(args + ["--mode", "rccl", "--share-gpus", "--reps", "2", "--check", "--digest"], env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert f"[fake_rccl] the test double of librccl.so.1 is in use: {parties} ranks" in r.stderr
    checks = [l for l in r.stdout.splitlines() if l.startswith("check: party ")]
    assert len(checks) == parties and all(" ok -- anchored" in l for l in checks), r.stdout[-3000:]
    dig = {l.split()[-1] for l in r.stdout.splitlines() if l.startswith("transcript sha256")}
    t = _hyperplonk(args + ["--mode", "threads", "--reps", "1", "--digest"], dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert t.returncode == 0, t.stdout[-2000:] + t.stderr[-2000:]
    dig_t = {l.split()[-1] for l in t.stdout.splitlines() if l.startswith("transcript sha256")}
    assert len(dig) == 1 and dig == dig_t, (dig, dig_t)
    comm_r = {l for l in r.stdout.splitlines() if l.startswith("Comm: ")}
    comm_t = {l for l in t.stdout.splitlines() if l.startswith("Comm: ")}
    assert len(comm_r) == 1 and comm_r == comm_t, (comm_r, comm_t)  # the same bytes exchanged


def test_cpp_host_rccl_mode_rejects_a_flipped_limb_on_every_party():
    r = _hyperplonk(["--l", "1", "--n", "10", "--mode", "rccl", "--share-gpus", "--reps", "1", "--tamper"], _fake_env())
    checks = [l for l in r.stdout.splitlines() if l.startswith("check: party ")]
    assert r.returncode == 3 and len(checks) == 8 and all("FAILED [gate[3]]" in l for l in checks), (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


def test_comm_abort_takes_the_peers_out_of_their_collective_over_the_rccl_test_double():
    """
    zk_comm_abort (include/zkhip.h): a party that fails for a reason of its own -- out of memory, a failed kernel -- must not leave the others
    waiting inside an exchange.  Four parties over the test double; three enter an all-gather, the fourth aborts its communicator
    instead: the three return ZK_ERR_COMM (nobody hangs), and the aborted ctx has no communicator left.
    """
    code = r"""
import sys, threading, time
import numpy as np
import zkhip
from zkhip._lib import ZK_ERR_COMM
from zkhip.net import RcclNet
assert 'torch' not in sys.modules
W = 4
ctxs = [zkhip.Ctx(0) for _ in range(W)]
RcclNet.from_init_all(ctxs)
res = [None] * W
def run(p):
    c = ctxs[p]
    if p == W - 1:
        time.sleep(0.5)   # the others are inside the collective by now
        c.comm_abort()
        res[p] = ('aborted', c.comm_size)
        return
    try:
        d = c.to_device(np.full((4, 4), p, dtype=np.uint64))
        c.allgather(d, 128)
        c.sync()
        res[p] = ('returned', 0)
    except zkhip.ZkError as e:
        res[p] = ('error', e.code)
th = [threading.Thread(target=run, args=(p,)) for p in range(W)]
[t.start() for t in th]
[t.join(timeout=120) for t in th]
assert not any(t.is_alive() for t in th), 'a party is still blocked in the exchange'
print('RES', res)
assert res[W - 1][0] == 'aborted'
assert all(r == ('error', ZK_ERR_COMM) for r in res[:W - 1]), res
print('ABORT_OK')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=_fake_env(), cwd=os.path.join(ROOT, "tests"))
    assert r.returncode == 0 and "ABORT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
