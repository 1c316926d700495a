"""
Collaborative HyperPlonk call sequence -- host-side mirror of hyperplonk/src/dhyperplonk.rs
(`PackedProvingParameters::new` :65-156, `dhyperplonk` :159-571, `dhyperplonk_data_parallel`
:573-960).  Like the reference it is a FIXED sequence of dist-primitive calls on synthetic
(random) tables: there is no circuit and no Fiat-Shamir, every challenge is pre-sampled.
All tables and SRS levels are resident in HBM; every primitive runs through libzkhip.so.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from . import dist_primitive as dp
from .field import fr_mont, random_fr, splitmix_fr
from .net import Net
from .pss import PackedSharingParams

ONE_BATCH = os.environ.get("ZKHIP_ONE_BATCH", "1") != "0"  # the sumcheck-family kernels of a proof's steps 2-4 as ONE batch (dp.ScQueue); 0: a batch per call
LATE_COMMIT = os.environ.get("ZKHIP_LATE_COMMIT", "1") != "0"  # one-batch schedule: the commit pass starts after the kernel batch (zkhost: ZKHOST_LATE_COMMIT=2)
TABLE_REC = int(os.environ.get("ZKHIP_TABLE_REC", "0"))  # 0: per level by free memory (see _synthetic_srs); 96 / 128: one G1 window-table record layout for every level (A/B)
CPERM_SERIAL = os.environ.get("ZKHIP_CPERM_SERIAL", "0") == "1"  # cpermcheck call by call as the reference writes it (A/B switch: same transcript)


class Timers:
    """wall-clock sections with the reference's labels (mpc-net/src/utils/timer.rs), leader only"""

    def __init__(self, enabled: bool):
        self.enabled, self.t, self.stack = enabled, {}, []

    def start(self, label):
        self.stack.append((label, time.perf_counter()))

    def end(self):
        label, t0 = self.stack.pop()
        self.t[label] = self.t.get(label, 0.0) + time.perf_counter() - t0


@dataclass
class PackedProvingParameters:
    """hyperplonk/src/dhyperplonk.rs:19-63; every table is a device buffer of Fr, lengths in `n_`"""

    n: int
    tables: Dict[str, object] = field(default_factory=dict)
    lens: Dict[str, int] = field(default_factory=dict)
    challenge: np.ndarray = None
    challenge_r1: np.ndarray = None
    challenge_r2: np.ndarray = None
    alpha: np.ndarray = None
    beta: np.ndarray = None
    gamma: np.ndarray = None
    c_commitment: List = None  # powers_of_g levels 0..n+2 (new_single, dpoly_comm.rs:197-219)
    d_commitment: List = None  # levels 0..n-1          (new_random, dpoly_comm.rs:220-233)

    @staticmethod
    def new(n: int, pp: PackedSharingParams, be, seed: int, chal_seed: int = None, window_tables: bool = True, table_max_log2: int = 25) -> "PackedProvingParameters":
        """
        dhyperplonk.rs:65-156 with a documented seed instead of StdRng::from_entropy().  chal_seed: the
        challenges are public values every party shares (the reference's local mode clones ONE parameter set
        for all parties, mpc-net/src/multi.rs:344); pass the same chal_seed to parties with different table seeds.
        window_tables: build the MSM window table of every SRS level up to 2^table_max_log2 points (zk_srs_precompute: setup work
        like generating the level, 13-16 x its memory; results are bit-identical with and without).  The tables only take
        memory the proof does not need: a level is skipped when building its table would leave less than half of the device
        free (the pass arenas of an n = 24 proof grow to ~110 GB); at n = 24 on a 288-GB MI355X the levels up to 2^24 points get
        theirs, the 2^25 / 2^26-point levels of the c-SRS run table-less (a 2^25-point table is built where the memory allows: n <= 23).
        """
        l, npar = pp.l, pp.n
        M = 1 << n
        pk = PackedProvingParameters(n=n)
        sd = [seed * 1000]

        def rnd(name, length):
            sd[0] += 1
            pk.tables[name] = be.to_device(random_fr(length, sd[0]))
            pk.lens[name] = length

        rnd("V", 4 * M // l)
        zero, one = fr_mont(0), fr_mont(1)
        for name, pts in (("a_evals", (zero, zero)), ("b_evals", (zero, one)), ("c_evals", (one, zero))):
            pk.tables[name] = be.fold(pk.tables["V"], 4 * M // l, np.stack(pts))  # fix_variable(&V, ..) :71-73
            pk.lens[name] = M // l
        for name, length in (
            ("I", M // l), ("I_p", M // npar), ("S1", M // l), ("S2", M // l), ("S1_p", M // npar), ("S2_p", M // npar),
            ("ssigma", 4 * M // l), ("ssigma_p", 4 * M // npar), ("sid", 4 * M // l), ("sid_p", 4 * M // npar),
            ("eq", M // l), ("eq_top_p", 2 * npar), ("eq_r1", 4 * M // l), ("eq_r1_p", 4 * M // npar),
            ("eq_r2", 4 * M // l), ("eq_r2_p", 4 * M // npar),
        ):
            rnd(name, length)
        sd[0] += 1
        ch = random_fr(3 * n + 4 + 3, sd[0] if chal_seed is None else chal_seed)
        pk.challenge, pk.challenge_r1, pk.challenge_r2 = ch[:n], ch[n : 2 * n + 2], ch[2 * n + 2 : 3 * n + 4]
        pk.alpha, pk.beta, pk.gamma = ch[3 * n + 4], ch[3 * n + 5], ch[3 * n + 6]
        pk._synthetic_srs(be, pp, seed, window_tables, table_max_log2)
        return pk

    # the 21 tables of the SplitMix64 parameter set, in generation order (seed + 1, + 2, ..): the 17 of dhyperplonk.rs:65-156 and the
    # per-run random data of the drivers ("Jump from sky" :187-190, the data-parallel s :603), which this form keeps with the tables
    SPLITMIX_LAYOUT = ("V", "I", "I_p", "S1", "S2", "S1_p", "S2_p", "ssigma", "ssigma_p", "sid", "sid_p", "eq", "eq_top_p", "eq_r1", "eq_r1_p",
                       "eq_r2", "eq_r2_p", "local_s_p", "local_s_l", "eq_top", "s_data_parallel")

    @staticmethod
    def new_splitmix(n: int, pp: PackedSharingParams, be, seed: int, chal_seed: int = 0, window_tables: bool = True, table_max_log2: int = 25) -> "PackedProvingParameters":
        """
        The parameter set the C++ host builds (host/zkhost/hyperplonk.hpp `PackedProvingParameters::make`): every table from
        SplitMix64(0x5CA1AB1E + 1000 seed + k) (SURVEY.md 8(d) "Synthetic inputs"), the same challenges and SRS seeds -- so that a
        proof of one host can be compared with the other's bit for bit at any size (tests/test_host_cpp.py).
        """
        l, npar, M = pp.l, pp.n, 1 << n
        lens = {"V": 4 * M // l, "I": M // l, "I_p": M // npar, "S1": M // l, "S2": M // l, "S1_p": M // npar, "S2_p": M // npar, "ssigma": 4 * M // l,
                "ssigma_p": 4 * M // npar, "sid": 4 * M // l, "sid_p": 4 * M // npar, "eq": M // l, "eq_top_p": 2 * npar, "eq_r1": 4 * M // l,
                "eq_r1_p": 4 * M // npar, "eq_r2": 4 * M // l, "eq_r2_p": 4 * M // npar, "local_s_p": 4 * M // npar, "local_s_l": 4 * M // npar // l,
                "eq_top": npar, "s_data_parallel": 4 * M // l}
        pk = PackedProvingParameters(n=n)
        sd = 0x5CA1AB1E + 1000 * seed
        for name in PackedProvingParameters.SPLITMIX_LAYOUT:
            sd += 1
            pk.tables[name] = be.to_device(splitmix_fr(lens[name], sd))
            pk.lens[name] = lens[name]
        zero, one = fr_mont(0), fr_mont(1)
        for name, pts in (("a_evals", (zero, zero)), ("b_evals", (zero, one)), ("c_evals", (one, zero))):
            pk.tables[name] = be.fold(pk.tables["V"], 4 * M // l, np.stack(pts))
            pk.lens[name] = M // l
        ch = splitmix_fr(3 * n + 7, chal_seed if chal_seed else sd + 1)
        pk.challenge, pk.challenge_r1, pk.challenge_r2 = ch[:n], ch[n : 2 * n + 2], ch[2 * n + 2 : 3 * n + 4]
        pk.alpha, pk.beta, pk.gamma = ch[3 * n + 4], ch[3 * n + 5], ch[3 * n + 6]
        pk._synthetic_srs(be, pp, seed, window_tables, table_max_log2)
        return pk

    def _synthetic_srs(pk, be, pp, seed, window_tables, table_max_log2):
        n, l, npar = pk.n, pp.l, pp.n
        # synthetic SRS (random points in the reference as well)
        pk.c_commitment = [be.srs_generate(seed * 7919 + 2 * i + 1, seed * 104729 + 2 * i + 3, max(1, (1 << i) // l)) for i in range(n + 3)]
        pk.d_commitment = [be.srs_generate(seed * 6007 + 2 * i + 5, seed * 15485863 + 2 * i + 7, 1 << i) for i in range(n - (npar.bit_length() - 1) + 3)]
        if window_tables:
            # Largest levels last; each takes what the device can spare at that moment (the rule of zkhost/hyperplonk.hpp finish_setup): 128-B
            # records -- one per cache line, k_accum_tiles -7 .. -10 % for 4/3 of the memory -- while that leaves >= 60 % of the device free,
            # packed 96-B records while that leaves >= 50 % (the pass arenas of an n = 24 proof grow to ~110 GB), no table below that: the level
            # then runs the table-less path.  A GPU shared by several parties fills up and the later / larger levels fall back by themselves.
            for lv in sorted(pk.c_commitment + pk.d_commitment, key=len):
                if hasattr(lv, "precompute") and 64 <= len(lv) <= (1 << table_max_log2):
                    rec = TABLE_REC
                    if hasattr(be, "mem_info"):
                        free, total = be.mem_info()
                        left_after = lambda rec_bytes: (free - 16 * rec_bytes * len(lv)) / total
                        if TABLE_REC in (96, 128):
                            if left_after(TABLE_REC) < 0.5:
                                break
                        elif left_after(128) >= 0.6:
                            rec = 128
                        elif left_after(96) < 0.5:
                            break
                        else:
                            rec = 96
                    try:
                        lv.precompute(0, record_bytes=rec)
                    except Exception as e:
                        if getattr(e, "code", None) != -6:  # ZK_ERR_OOM: keep going without the remaining tables
                            raise
                        break


def _per_run_data(pk, pp, be, net, seed):
    """"Jump from sky" (dhyperplonk.rs:187-190): the per-run random tables -- the parameter set's own when it carries them
    (PackedProvingParameters.new_splitmix), otherwise drawn from `seed`; resident like every other table"""
    T, M, l, npar = pk.tables, 1 << pk.n, pp.l, net.n_parties
    if "local_s_p" in T:
        return T["local_s_p"], T["local_s_l"], T["eq_top"]
    return (be.to_device(random_fr(4 * M // npar, seed * 31 + 1)), be.to_device(random_fr(4 * M // npar // l, seed * 31 + 2)),
            be.to_device(random_fr(pp.n, seed * 31 + 3)))


def _halves(buf, length):
    """(first half, second half) of a device table as (pointer-like, length) pairs"""
    return buf, _at(buf, 32 * (length // 2))


def _done(value):
    """a phase-B / finishing closure for work that has already run (the batch-per-call forms next to the *_sq ones)"""
    return lambda: value


def _at(buf, byte_off):
    return buf.at(byte_off) if hasattr(buf, "at") else buf + byte_off


def _wiring_identity(n, pk, pp, be, net, seed, data_parallel, local_s_p, local_s_l, eq_top):
    """step 2 of dhyperplonk (:262-514) == the body of dpermcheck (:992-1245)"""
    sq = dp.ScQueue(be) if ONE_BATCH else None  # the step's sumcheck-family kernels as one batch
    q, phase_b = _wiring_enqueue(n, pk, pp, be, net, seed, data_parallel, local_s_p, local_s_l, eq_top, sq)
    if sq is not None:
        sq.run()
    finalize = phase_b()
    q.run()
    return finalize()


def _wiring_enqueue(n, pk, pp, be, net, seed, data_parallel, local_s_p, local_s_l, eq_top, sq=None):
    """
    the step up to (not including) its one batched MSM pass.  -> (queue, phase_b); finalize = phase_b(); run (or start / finish) the
    queue, then finalize() performs the exchanges of the MSM results and returns (wiring_proofs, wiring_commits, wiring_opens) in the
    reference's order.
    sq = None: every sumcheck / fold / open-round kernel runs inside the call that owns it (a blocking batch per call), phase_b()
    has nothing left to do.  sq = a dp.ScQueue (the mirror of zkhost/hyperplonk.hpp wiring_enqueue_sq): the calls only ADD their
    kernels; the caller runs the batch, and phase_b() performs the exchanges that need a kernel result (pss2ss of 2.c and of the
    opens of V, the gather of the local open values, the leader's root opens).  Same outputs.
    """
    one = sq is not None
    T, L = pk.tables, pk.lens
    l, npar = pp.l, net.n_parties
    M = 1 << n
    cc, dc = pk.c_commitment, pk.d_commitment
    wiring_proofs, wiring_commits, wiring_opens = [], [], []
    # 2.a (:268-294): every party broadcasts local_s; s = concatenation over parties (an all-gather)
    if data_parallel:
        s_dev = T["s_data_parallel"] if "s_data_parallel" in T else be.to_device(random_fr(4 * M // l, seed * 31 + 4))
    elif hasattr(net, "all_gather_device") and getattr(net, "ctx", None) is be:  # (a foreign ctx's stream is not ordered with ours)
        # RCCL inside the ctx: the 4M/(l N_p) Fr of every party meet in HBM (256 MiB per party at n = 24),
        # nothing crosses PCIe
        s_dev = net.all_gather_device(local_s_l, 32 * (4 * M // npar // l))
    elif getattr(net, "echo", False) and hasattr(be, "copy_d2d"):
        # the no-comm fake hands the leader N_p copies of its own message (dhyperplonk.rs:289-293)
        cnt = 4 * M // npar // l
        net._count(32 * cnt)
        s_dev = be.alloc(32 * cnt * npar)
        for q in range(npar):
            be.copy_d2d(s_dev.at(32 * cnt * q), local_s_l, 32 * cnt)
    else:
        s_dev = be.to_device(np.concatenate(net.all_gather(local_s_l.download((4 * M // npar // l, 4)))))
    # 2.b (commit of local_s) and the d_open of local_s in 2.d are independent of everything below: they ride
    # in the batched passes of :363-407 (same positions in the output lists as in the reference)
    # every MSM of the step is queued and runs in ONE batched pass at its end (dp.MsmQueue); the closures below put
    # the results at the reference's positions.  Exchanges keep the reference's order on every party.
    q = dp.MsmQueue(be)
    if one:
        a_2c = dp.c_sumcheck_product_many_sq(be, sq, [(s_dev, T["V"])], 4 * M // l, pk.challenge_r1, pp, net)  # 2.c
        a_copen = dp.c_open_many_sq(be, sq, q, cc, [T["V"], T["V"]], [4 * M // l] * 2, [pk.challenge_r1, pk.challenge_r2], pp, net)  # 2.d
    else:
        a_2c = _done([dp.c_sumcheck_product(be, s_dev, T["V"], 4 * M // l, pk.challenge_r1, pp, net)])  # 2.c
        # 2.d: the two opens of V are independent -> their q_i commitments share one d_msm
        a_copen = _done(dp.c_open_many_q(be, q, cc, [T["V"], T["V"]], [4 * M // l] * 2, [pk.challenge_r1, pk.challenge_r2], pp, net))
    # 2.e (:322-340)
    hlen = 4 * M // npar
    num = be.fr_axpb(local_s_p, T["sid_p"], pk.alpha, pk.beta, hlen)
    den = be.fr_axpb(T["eq_r1_p"], T["ssigma_p"], pk.alpha, pk.beta, hlen)
    h_p = be.fr_batch_div(num, den, hlen)
    subtree, top = dp.d_acc_product(be, h_p, hlen, net)  # :342
    q.keep.append(subtree)  # v1x and the layer slices below are views into it; the queued MSMs read them after this function returns
    if getattr(be, "sc_trace", None) is not None:
        be.sc_trace.append(("keepalive", subtree, None, 0, np.zeros((0, 4), dtype=np.uint64)))  # traced slices point into it
    v1x = _at(subtree, 32 * hlen)  # tree[N..]
    vx0, vx1 = be.fr_deinterleave(subtree, hlen)  # tree[0::2], tree[1::2]  :344-359
    # :363-380 / :383-407: independent commits / opens
    tabs8 = [T["ssigma_p"], T["sid_p"], h_p, num, den, v1x, vx0, vx1]
    f_dcommit = dp.d_commit_many_q(be, q, dc, [local_s_p] + tabs8, [4 * M // npar] + [hlen] * 8, net)  # 2.b, then :363-380
    lay_tabs, lay_lens, lay_pts = [local_s_p] + tabs8[:5], [4 * M // npar] + [hlen] * 5, [pk.challenge_r2] * 6  # 2.d, then :383-407
    # the 3 + 3 (n - s) d_sumcheck_products of 2.e are independent of each other: one batched local phase, one exchange
    dsp_items = [(den, T["eq_r2_p"], hlen, pk.challenge_r2), (h_p, den, hlen, pk.challenge_r2), (num, T["eq_r2_p"], hlen, pk.challenge_r2)]  # 2.e.1 :411-413
    # 2.e.2 layered sumcheck + opens on halving slices (:417-478)
    sbits = npar.bit_length() - 1
    cur = {"v1x": v1x, "vx0": vx0, "vx1": vx1, "eq": T["eq_r2_p"]}
    clen = hlen // 2  # current_* = first half
    for i in range(1, n - sbits + 1):
        ch = pk.challenge_r2[i:]
        dsp_items += [(cur["eq"], cur["v1x"], clen, ch), (cur["eq"], cur["vx0"], clen, ch), (cur["vx0"], cur["vx1"], clen, ch)]
        for k in ("v1x", "vx0", "vx1"):
            lay_tabs.append(cur[k])
            lay_lens.append(clen)
            lay_pts.append(ch)
        for k in cur:  # current = current[len/2..]
            cur[k] = _at(cur[k], 32 * (clen // 2))
        clen //= 2
    # (local phases now; the exchange and the leader rounds -- host arithmetic -- in finalize(), beside the step's MSM pass)
    f_dsp = dp.d_sumcheck_product_many_sq(be, sq, dsp_items, net) if one else dp.d_sumcheck_product_many_q(be, dsp_items, net)
    # the opens of local_s, of the five tables and of all layers are independent of each other
    a_dopen = dp.d_open_many_sq(be, sq, q, dc, lay_tabs, lay_lens, lay_pts, net) if one else _done(dp.d_open_many_q(be, q, dc, lay_tabs, lay_lens, lay_pts, net))
    f_top_commits, f_top_opens, top_proofs = [], None, []
    if top is not None:  # leader-only tail on the N_p-leaf top tree (:480-511)
        tt = np.asarray(top, dtype=np.uint64).reshape(-1, 4)
        half = len(tt) // 2
        lv1x, lvx0, lvx1 = tt[half:], tt[0::2], tt[1::2]
        chs = pk.challenge_r2[:sbits]
        dv = [be.to_device(np.ascontiguousarray(v)) for v in (lvx0, lvx1, lv1x)]
        f_top_commits = [dp.commit_q(q, dc, d, len(v)) for d, v in zip(dv, (lvx0, lvx1, lv1x))]
        f_top_opens = (dp.open_many_sq(be, sq, q, dc, dv, [len(lvx0), len(lvx1), len(lv1x)], [chs] * 3) if one
                       else dp.open_many_q(be, q, dc, dv, [len(lvx0), len(lvx1), len(lv1x)], [chs] * 3))
        q.keep += dv
        d0, dd1, d1 = dv
        for f_, g_, m_ in ((eq_top, d1, len(lv1x)), (eq_top, d0, len(lvx0)), (d0, dd1, len(lvx0))):
            top_proofs.append(dp.sumcheck_product_sq(be, sq, f_, g_, m_, chs) if one else _done(dp.sumcheck_product(be, f_, g_, m_, chs)))

    def phase_b():
        wiring_proofs.extend(a_2c())  # (one batch: pss2ss of 2.c)
        f_copen = a_copen()           # (one batch: pss2ss of the two opens of V, their d_msm queued)
        f_dopen = a_dopen()           # (one batch: the gather of the local open values, the leader's root opens)

        def finalize():
            wiring_proofs.extend(f_dsp())                  # 2.e: after 2.c, before the leader-tree sumchecks (the reference's order)
            wiring_opens.extend(f_copen())                 # 2.d
            wiring_commits.extend(list(f_dcommit()))       # 2.b, then :363-380
            wiring_opens.extend(f_dopen())
            if top is not None:
                for fc, fo in zip(f_top_commits, f_top_opens()):  # (commit, open) per table, in the reference's order
                    wiring_commits.append(fc())
                    wiring_opens.append(fo)
                wiring_proofs.extend(p() for p in top_proofs)
            return wiring_proofs, wiring_commits, wiring_opens

        return finalize

    return q, phase_b


def dhyperplonk(n: int, pk: PackedProvingParameters, pp: PackedSharingParams, be, net: Net, seed: int = 1, data_parallel: bool = False):
    """
    dhyperplonk.rs:159-571 (data_parallel=True: dhyperplonk_data_parallel :573-960, which differs
    only at step 2.a -- `s` is local random data, no exchange, :603).
    Returns ((gate_identity_proofs, gate_identity_commitments), (wiring_proofs, wiring_commits, wiring_opens)), timers.
    """
    T, L = pk.tables, pk.lens
    l, npar = pp.l, net.n_parties
    M = 1 << n
    tm = Timers(net.is_leader)
    # "Jump from sky" (:187-190)
    local_s_p, local_s_l, eq_top = _per_run_data(pk, pp, be, net, seed)
    net.sync()
    tm.start("Distributed HyperPlonk")

    # Step 1: commit (:198-215)
    tm.start("Commit")
    cc, dc = pk.c_commitment, pk.d_commitment
    com = {}
    names_c, names_d = ("a_evals", "b_evals", "c_evals"), ("I_p", "S1_p", "S2_p")
    q = dp.MsmQueue(be)  # both commit families in one batched MSM pass
    f_c = dp.c_commit_q(be, q, cc, [T[x] for x in names_c], [L[x] for x in names_c], pp, net)
    f_d = dp.d_commit_many_q(be, q, dc, [T[x] for x in names_d], [L[x] for x in names_d], net)
    # The MSM pass of a step is STARTED here and collected one step later (MsmQueue.start / finish): nothing consumes a
    # commitment before the proof is assembled (challenges are pre-sampled, :159-186), and the host part of a step -- the
    # exchanges of its results and the 8-term point combinations -- then runs while the GPU works on the next step's pass.
    # The timer labels keep the reference's names; with the overlap "Commit" / "Wire identity" / "Open" no longer cover the reference's
    # steps ("Wire identity" = the kernel phases of steps 2 and 4 and the enqueue of their passes, "Open" = the collection of both),
    # only "Distributed HyperPlonk" is comparable.
    # (with the one-batch schedule the pass is started right after the kernel batch of steps 2-4: the product tree and the batch then
    # run on an empty chip instead of queueing behind it, and the GPU works on the pass during the hand-offs; LATE_COMMIT = False: here)
    late = LATE_COMMIT and ONE_BATCH and dp.PIPELINE_MSM
    if not late:
        q.start()

    def finish_commit():
        q.finish()
        for name, cm in zip(names_c, f_c()):
            com[name] = cm
        for name, cm in zip(names_d, f_d()):
            com[name] = cm

    tm.end()

    # Step 3: gate identity (:223-260)
    tm.start("Gate identity")
    Ml = M // l
    sum_ab = be.fr_add(T["a_evals"], T["b_evals"], Ml)  # :233-238
    sum_ci = be.fr_sub(T["I"], T["c_evals"], Ml)  # -c + I  :251-256
    gate_pairs = [(T["eq"], T["S1"]), (T["S1"], sum_ab), (T["eq"], T["S2"]), (T["a_evals"], T["b_evals"]), (T["S2"], T["a_evals"]), (T["eq"], sum_ci)]
    if ONE_BATCH and dp.PIPELINE_MSM:
        # The sumcheck-family kernels of steps 2-4 as ONE batch (dp.ScQueue; zkhost/hyperplonk.hpp does the same): the six gate
        # sumchecks, everything of the wiring step, the fold rounds of the Open step.  None needs another's result on the device; as
        # ~10 blocking calls they were launch chains on a nearly empty chip.  Phase B of every primitive (the exchanges that need a
        # kernel result) follows in the reference's order; then the two long MSM passes start.
        sq = dp.ScQueue(be)
        a_gate = dp.c_sumcheck_product_many_sq(be, sq, gate_pairs, Ml, pk.challenge, pp, net)
        tm.end()
        tm.start("Wire identity")
        q_w, b_wiring = _wiring_enqueue(n, pk, pp, be, net, seed, data_parallel, local_s_p, local_s_l, eq_top, sq)
        q_o = dp.MsmQueue(be)
        a_co = dp.c_open_many_sq(be, sq, q_o, cc, [T[x] for x in names_c], [L[x] for x in names_c], [pk.challenge] * 3, pp, net)
        a_do = dp.d_open_many_sq(be, sq, q_o, dc, [T[x] for x in names_d], [L[x] for x in names_d], [pk.challenge] * 3, net)
        sq.run()
        if late:
            q.start()
        gate_proofs = a_gate()
        finalize_wiring = b_wiring()
        f_co, f_do = a_co(), a_do()
        q_w.start()
        q_o.start()
        finish_commit()
        tm.end()
        tm.start("Open")
        q_w.finish()
        wiring_proofs, wiring_commits, wiring_opens = finalize_wiring()
        q_o.finish()
        gate_commitments = [(com[name], op) for name, op in zip(names_c + names_d, list(f_co()) + list(f_do()))]
        tm.end()
        tm.end()
        return ((gate_proofs, gate_commitments), (wiring_proofs, wiring_commits, wiring_opens)), tm.t
    # the six sumchecks are independent: one batched phase 1, then the hand-offs in the reference's order
    gate_proofs = dp.c_sumcheck_product_many(be, gate_pairs, Ml, pk.challenge, pp, net)
    tm.end()

    # Step 2: wiring identity (shared with dpermcheck)
    tm.start("Wire identity")
    q_w, b_wiring = _wiring_enqueue(n, pk, pp, be, net, seed, data_parallel, local_s_p, local_s_l, eq_top)
    finalize_wiring = b_wiring()
    # The kernel phase of the Open step (:517-553: fold rounds and quotients of a, b, c, I, S1, S2) depends on nothing the wiring step
    # produces, so it runs BEFORE the wiring pass is started: behind a running pass its ~0.5 ms of kernels would wait for the pass to
    # drain (they share the runtime's hardware queues), the Open pass could only be enqueued after that, and the GPU would idle between
    # the two passes while the host prepares the second.  Both passes are now in flight back to back; outputs keep their positions.
    gate_commitments = []
    q_o = dp.MsmQueue(be)
    f_co = dp.c_open_many_q(be, q_o, cc, [T[x] for x in names_c], [L[x] for x in names_c], [pk.challenge] * 3, pp, net)
    f_do = dp.d_open_many_q(be, q_o, dc, [T[x] for x in names_d], [L[x] for x in names_d], [pk.challenge] * 3, net)
    q_w.start()
    q_o.start()
    finish_commit()  # (host: exchange + point combinations of step 1, beside the passes on the GPU)
    tm.end()

    # Open (:517-553): collection of both passes
    tm.start("Open")
    q_w.finish()
    wiring_proofs, wiring_commits, wiring_opens = finalize_wiring()
    q_o.finish()
    for name, op in zip(names_c, f_co()):
        gate_commitments.append((com[name], op))
    for name, op in zip(names_d, f_do()):
        gate_commitments.append((com[name], op))
    tm.end()
    tm.end()
    return ((gate_proofs, gate_commitments), (wiring_proofs, wiring_commits, wiring_opens)), tm.t


def dpermcheck(n: int, pk: PackedProvingParameters, pp: PackedSharingParams, be, net: Net, seed: int = 1):
    """hyperplonk/src/dhyperplonk.rs:962-1247: the distributed permutation check alone (= step 2 of dhyperplonk)"""
    l, npar = pp.l, net.n_parties
    M = 1 << n
    tm = Timers(net.is_leader)
    local_s_p, local_s_l, eq_top = _per_run_data(pk, pp, be, net, seed)
    net.sync()
    tm.start("Distributed Permcheck")
    res = _wiring_identity(n, pk, pp, be, net, seed, False, local_s_p, local_s_l, eq_top)
    tm.end()
    return res, tm.t


def cpermcheck(n: int, pk: PackedProvingParameters, pp: PackedSharingParams, be, net: Net, seed: int = 1):
    """
    hyperplonk/src/dhyperplonk.rs:1249-1385: the collaborative (packed) permutation check:
    num/den maps, c_commit/c_open of the public wires, and per polynomial the masked product tree
    (c_acc_product_and_share) with its commits, opens and three product sumchecks.
    """
    T = pk.tables
    l = pp.l
    G4 = 4 * ((1 << n) // l)  # gate_count * 4 with gate_count = 2^n / l (:1270)
    tm = Timers(net.is_leader)
    # masks (PackedProvingParameters::new :124-127), generated on first use
    for i, name in enumerate(("mask", "unmask0", "unmask1", "unmask2")):
        if name not in T:
            T[name] = be.to_device(random_fr(G4, seed * 977 + 50 + i))
            pk.lens[name] = G4
    net.sync()
    tm.start("Collaborative Permcheck")
    cc = pk.c_commitment
    num = be.fr_axpb(T["V"], T["sid"], pk.alpha, pk.beta, G4)  # :1277-1279
    den = be.fr_axpb(T["eq_r1"], T["ssigma"], pk.alpha, pk.beta, G4)  # :1280-1282
    # The reference runs 10 c_commit, 12 c_open and 6 c_sumcheck_product one after the other (:1289-1375); none feeds another on the
    # device -- only the two masked product trees produce tables the rest reads.  So: the trees first, then ALL commitments and quotient
    # commitments as ONE MSM pass and the fold rounds of the opens / the product sumchecks as one batched call each.  The second open
    # of num / den (:1371-1375) is the first one again (same table, same point): c_open_many_q runs it once.  Outputs keep the
    # reference's positions.  CPERM_SERIAL: the call-by-call form (A/B, same transcript).
    proofs, commits, opens = [], [], []
    if CPERM_SERIAL:
        ccommit = lambda tab: dp.c_commit(be, cc, [tab], [G4], pp, net)[0]
        copen = lambda tab: dp.c_open(be, cc, tab, G4, pk.challenge_r1, pp, net)
        for name in ("ssigma", "sid"):  # :1289-1308
            commits.append(ccommit(T[name]))
            opens.append(copen(T[name]))
        for ev in (num, den):
            (d0, n0), (d1, n1), (d2, n2) = dp.c_acc_product_and_share(be, ev, T["mask"], T["unmask0"], T["unmask1"], T["unmask2"], G4, pp, net)
            assert n0 == n1 == n2 == G4  # the three share vectors stay in HBM and feed the commits / opens / sumchecks below
            for tab in (ev, d0, d1, d2):  # :1324-1363
                commits.append(ccommit(tab))
                opens.append(copen(tab))
            csp = lambda f, g: dp.c_sumcheck_product(be, f, g, G4, pk.challenge_r1, pp, net)
            proofs.append(csp(T["eq_r1"], d2))  # :1365-1369
            proofs.append(csp(T["eq_r1"], d0))
            proofs.append(csp(d0, d1))
            opens.append(copen(ev))  # :1371-1375
    else:
        com_tabs, open_tabs, pairs = [T["ssigma"], T["sid"]], [T["ssigma"], T["sid"]], []  # :1289-1308
        for ev in (num, den):
            (d0, n0), (d1, n1), (d2, n2) = dp.c_acc_product_and_share(be, ev, T["mask"], T["unmask0"], T["unmask1"], T["unmask2"], G4, pp, net)
            assert n0 == n1 == n2 == G4  # the three share vectors stay in HBM and feed the commits / opens / sumchecks below
            com_tabs += [ev, d0, d1, d2]  # :1324-1363
            open_tabs += [ev, d0, d1, d2, ev]  # ... and :1371-1375
            pairs += [(T["eq_r1"], d2), (T["eq_r1"], d0), (d0, d1)]  # :1365-1369
        q = dp.MsmQueue(be)
        f_com = dp.c_commit_q(be, q, cc, com_tabs, [G4] * len(com_tabs), pp, net)
        f_open = dp.c_open_many_q(be, q, cc, open_tabs, [G4] * len(open_tabs), [pk.challenge_r1] * len(open_tabs), pp, net)
        proofs = dp.c_sumcheck_product_many(be, pairs, G4, pk.challenge_r1, pp, net)
        q.run()
        commits = list(f_com())
        opens = f_open()
    tm.end()
    return (proofs, commits, opens), tm.t
