"""
Party-axis exchange layer -- what replaces mpc-net's TCP star (mpc-net/src/lib.rs:35-287) and
the typed adapter dist-primitive/src/utils/serializing_net.rs.

Every exchange of the hot path is a star through party 0 whose leader function is a PUBLIC
linear map, so on one node it is re-expressed as ONE all-gather of raw limbs
followed by the same map computed by every party for its own slot.  Payloads are raw
Montgomery limbs (no compression on xGMI).

  TorchDistNet   one process per GPU, torch.distributed: backend "nccl" (= RCCL over xGMI) on
                 the GPU box, "gloo" in the CPU tests
  LocalTestNet   all parties as threads of one process (mirrors LocalTestNet::simulate_network_round,
                 mpc-net/src/multi.rs:268-362)
  LeaderEchoNet  the no-`comm` fake (serializing_net.rs:144-264): the leader sees n copies of
                 its own message; only party 0 is meaningful
"""
from __future__ import annotations

import threading
from typing import Callable, List

import numpy as np


def _at(buf, byte_off: int):
    """a device buffer (or raw address) advanced by byte_off"""
    return buf.at(byte_off) if hasattr(buf, "at") else buf + byte_off


class Net:
    n_parties: int
    party_id: int

    @property
    def is_leader(self) -> bool:
        return self.party_id == 0

    def all_gather(self, a: np.ndarray) -> List[np.ndarray]:
        """every party contributes `a` (same shape everywhere); returns the list ordered by party id"""
        raise NotImplementedError

    def all_to_all(self, chunks: List[np.ndarray], echo: str = "slot0") -> List[np.ndarray]:
        """
        party p sends chunks[q] to party q; returns [what party q sent to me for q < n].  Replaces the
        reference's loops of dynamic gathers / scatters over every root (dacc_product.rs:94-104,
        :155-203).  Default: all-gather of the stacked chunks, then select (backends override).
        """
        allc = self.all_gather(np.stack(chunks))
        return [allc[q][self.party_id] for q in range(self.n_parties)]

    def sync(self):
        self.all_gather(np.zeros(1, dtype=np.uint64))

    # device-buffer forms of the two exchanges.  `be` is the compute backend that owns the buffers (a zkhip.Ctx in
    # production).  These defaults stage through the numpy methods above (thread / gloo / echo nets of the tests);
    # RcclNet overrides them with collectives that never leave HBM.
    def all_gather_device(self, d_send, nbytes: int, d_recv=None, be=None):
        """-> device buffer holding n_parties * nbytes, ordered by party"""
        parts = self.all_gather(be.download_ptr(d_send, nbytes))
        out = d_recv or be.alloc(max(nbytes * self.n_parties, 1))
        be.upload_ptr(out, np.concatenate([np.asarray(x, dtype=np.uint8).reshape(-1) for x in parts]))
        return out

    def all_to_all_device(self, d_send, nbytes_per_peer: int, d_recv=None, be=None, echo: str = "slot0"):
        """party p sends bytes [q * nbytes_per_peer, (q + 1) * nbytes_per_peer) of d_send to party q; -> what every party sent to me, ordered by party"""
        a = be.download_ptr(d_send, nbytes_per_peer * self.n_parties).reshape(self.n_parties, nbytes_per_peer)
        got = self.all_to_all([a[q] for q in range(self.n_parties)], echo=echo)
        out = d_recv or be.alloc(max(nbytes_per_peer * self.n_parties, 1))
        be.upload_ptr(out, np.concatenate([np.asarray(x, dtype=np.uint8).reshape(-1) for x in got]))
        return out

    # byte accounting like MPCNet::get_comm (multi.rs:378-387): (up, down)
    def __init_counters(self):
        self.upload = 0
        self.download = 0

    def _count(self, nbytes: int):
        if not hasattr(self, "upload"):
            self.__init_counters()
        self.upload += nbytes * (self.n_parties - 1)
        self.download += nbytes * (self.n_parties - 1)


class LeaderEchoNet(Net):
    echo = True

    def __init__(self, n_parties: int = 8):
        self.n_parties, self.party_id = n_parties, 0

    def all_gather(self, a):
        self._count(a.nbytes)
        return [np.array(a, copy=True) for _ in range(self.n_parties)]

    def all_to_all(self, chunks, echo="slot0"):
        """
        the fake's two behaviours: a dynamic GATHER to self returns n copies of the own message
        (serializing_net.rs:159-162 -> echo="slot0"); for the looped dynamic SCATTERs the callers
        substitute their local data per destination (dacc_product.rs:194-202 -> echo="identity")
        """
        self._count(sum(c.nbytes for c in chunks) // max(len(chunks), 1))
        if echo == "identity":
            return [np.array(c, copy=True) for c in chunks]
        return [np.array(chunks[0], copy=True) for _ in range(self.n_parties)]

    # device forms: the fabricated copies are made in HBM (the leader's data never visits the host)
    def all_gather_device(self, d_send, nbytes: int, d_recv=None, be=None):
        self._count(nbytes)
        out = d_recv or be.alloc(max(nbytes * self.n_parties, 1))
        for q in range(self.n_parties):
            be.copy_d2d(_at(out, q * nbytes), d_send, nbytes)
        return out

    def all_to_all_device(self, d_send, nbytes_per_peer: int, d_recv=None, be=None, echo: str = "slot0"):
        self._count(nbytes_per_peer)
        out = d_recv or be.alloc(max(nbytes_per_peer * self.n_parties, 1))
        if echo == "identity":
            be.copy_d2d(out, d_send, nbytes_per_peer * self.n_parties)
        else:
            for q in range(self.n_parties):
                be.copy_d2d(_at(out, q * nbytes_per_peer), d_send, nbytes_per_peer)
        return out


class TorchDistNet(Net):
    """torch.distributed process group; device tensors for nccl, CPU tensors for gloo"""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.n_parties = dist.get_world_size(group)
        self.party_id = dist.get_rank(group)
        self.device = device

    def all_gather(self, a):
        import torch

        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy())
        self._count(a.nbytes)
        if self.device is not None:  # nccl: one device buffer for all parties, ONE copy back
            t = t.to(self.device)
            out = torch.empty(self.n_parties * t.numel(), dtype=torch.uint8, device=self.device)
            self.dist.all_gather_into_tensor(out, t, group=self.group)
            h = out.cpu().numpy().reshape(self.n_parties, -1)
            return [h[q].view(a.dtype).reshape(a.shape) for q in range(self.n_parties)]
        outs = [torch.empty_like(t) for _ in range(self.n_parties)]
        self.dist.all_gather(outs, t, group=self.group)
        return [o.numpy().view(a.dtype).reshape(a.shape) for o in outs]

    def all_to_all(self, chunks, echo="slot0"):
        import torch

        if self.dist.get_backend(self.group) != "nccl":
            return super().all_to_all(chunks)
        a = np.ascontiguousarray(np.stack(chunks))
        t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(self.device)
        out = torch.empty_like(t)
        self.dist.all_to_all_single(out, t, group=self.group)
        self._count(a.nbytes // self.n_parties)
        r = out.cpu().numpy().view(a.dtype).reshape(a.shape)
        return [r[q] for q in range(self.n_parties)]


class RcclNet(Net):
    """
    The C-ABI communicator (zk_comm_init / zk_allgather / zk_alltoall, include/zkhip.h): the exchanges run on
    the ctx stream over RCCL, buffers stay in HBM -- `all_gather_device` moves no byte over PCIe.  The
    numpy-typed methods of the Net interface stage through one cached device buffer.
    """

    def __init__(self, ctx, rank: int, world: int, unique_id: bytes = None):
        self.ctx, self.party_id, self.n_parties = ctx, rank, world
        if unique_id is not None:  # (None: the ctx already carries its communicator, see from_init_all)
            ctx.comm_init(rank, world, unique_id)

    @staticmethod
    def from_init_all(ctxs) -> list:
        """one process, a ctx per GPU (zk_comm_init_all -- the reference's task-per-party model, mpc-net/src/multi.rs:330-352):
        -> one net per party; each must then be driven from its own thread"""
        from .api import comm_init_all

        comm_init_all(ctxs)
        return [RcclNet(c, p, len(ctxs)) for p, c in enumerate(ctxs)]

    @staticmethod
    def from_torch_dist(ctx, group=None) -> "RcclNet":
        """bootstrap: rank 0 creates the RCCL id, torch.distributed (any backend) hands it to the others"""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return RcclNet(ctx, rank, world, box[0])

    def all_gather_device(self, d_send, nbytes: int, d_recv=None, be=None):
        """-> device buffer with n_parties * nbytes, ordered by party; stays in HBM (zk_allgather on the ctx stream)"""
        assert be is None or be is self.ctx, "the communicator lives in its own ctx"
        self._count(nbytes)
        return self.ctx.allgather(d_send, nbytes, d_recv)

    def all_to_all_device(self, d_send, nbytes_per_peer: int, d_recv=None, be=None, echo: str = "slot0"):
        """zk_alltoall on the ctx stream: HBM -> HBM over xGMI (the looped dynamic scatters of dacc_product.rs:94-104,155-203)"""
        assert be is None or be is self.ctx, "the communicator lives in its own ctx"
        self._count(nbytes_per_peer)
        return self.ctx.alltoall(d_send, nbytes_per_peer, d_recv)

    def all_gather(self, a):
        a = np.ascontiguousarray(a)
        nb = a.nbytes
        if nb == 0:
            return [a.copy() for _ in range(self.n_parties)]
        buf = self.ctx.temp(nb * (self.n_parties + 1), "rcclnet")
        buf.upload(a)
        self.all_gather_device(buf.ptr, nb, buf.at(nb))
        h = buf.download((self.n_parties, nb), dtype=np.uint8, offset=nb)
        return [h[q].view(a.dtype).reshape(a.shape) for q in range(self.n_parties)]

    def all_to_all(self, chunks, echo="slot0"):
        a = np.ascontiguousarray(np.stack(chunks))
        per = a.nbytes // self.n_parties
        if per == 0:
            return [a[q].copy() for q in range(self.n_parties)]
        buf = self.ctx.temp(2 * a.nbytes, "rcclnet")
        buf.upload(a)
        self.all_to_all_device(buf.ptr, per, buf.at(a.nbytes))
        r = buf.download(a.shape, dtype=a.dtype, offset=a.nbytes)
        return [r[q] for q in range(self.n_parties)]


class _LocalHub:
    def __init__(self, n):
        self.n = n
        self.slots = [None] * n
        self.barrier = threading.Barrier(n)


class LocalTestNet(Net):
    def __init__(self, hub: _LocalHub, party_id: int):
        self.hub, self.party_id, self.n_parties = hub, party_id, hub.n

    def all_gather(self, a):
        self.hub.slots[self.party_id] = np.array(a, copy=True)
        self.hub.barrier.wait()
        out = [np.array(s, copy=True) for s in self.hub.slots]
        self.hub.barrier.wait()
        self._count(a.nbytes)
        return out

    def all_to_all(self, chunks, echo="slot0"):
        self.hub.slots[self.party_id] = [np.array(c, copy=True) for c in chunks]
        self.hub.barrier.wait()
        out = [np.array(self.hub.slots[q][self.party_id], copy=True) for q in range(self.n_parties)]
        self.hub.barrier.wait()
        self._count(sum(c.nbytes for c in chunks) // max(len(chunks), 1))
        return out

    @staticmethod
    def simulate_network_round(n_parties: int, fn: Callable[["LocalTestNet"], object]) -> list:
        """run fn(net) for every party on its own thread; results ordered by party id"""
        hub = _LocalHub(n_parties)
        results: list = [None] * n_parties
        errors: list = []

        def run(p):
            try:
                results[p] = fn(LocalTestNet(hub, p))
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
                hub.barrier.abort()

        threads = [threading.Thread(target=run, args=(p,)) for p in range(n_parties)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results
