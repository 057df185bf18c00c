"""Message transport of the slab-decomposed paths (nns/slab.py) over a `torch.distributed` group, one process per GPU.

  * backend "nccl" (= RCCL over xGMI on a GPU node): device buffers go to the collective as they are; every call is
    enqueued stream-ordered and returns a handle whose wait() makes the CURRENT STREAM wait (no host block), so compute
    enqueued between the call and wait() overlaps the transfer.
  * backend "gloo" (CPU tests, and the two-ranks-on-one-GPU rehearsal of the HIP path on a one-GPU box): gloo moves host
    memory, so device tensors are staged through host buffers around a blocking collective; wait() is then a no-op.

xGMI is point-to-point: a halo exchange talks to the two ring neighbours only (their two direct links); the transpose is ONE
all-to-all (all seven links of a GPU at once).

STATUS of the "nccl" branch: a one-GPU lease cannot hold two RCCL ranks, so what has run on hardware is (a) the world-1 LOOPBACK
form (`Transport(loopback=True)` or NNS_COMM_LOOPBACK=1: a rank's messages to "its neighbours" and its all-to-all go through the
RCCL process group to itself -- device buffers, grouped send/recv, async all_to_all_single, stream-ordered wait(), the int-view
all-reduce -- tests/test_gpu_rccl_loopback.py) and (b) every multi-rank path on the gloo-staged branch.  Transfers between two
GPUs over xGMI, and the overlap they allow, are first exercised by the driver's multi-GPU bench.
"""
import os

import torch
import torch.distributed as dist


class _Done(object):
    def wait(self):
        return None


class _Works(object):
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


def _adjacent(a, b):
    """A flat view over a and b when b starts where a ends in the same storage (both contiguous), else None."""
    if a is None or b is None or a.dtype != b.dtype or a.device != b.device or not (a.is_contiguous() and b.is_contiguous()):
        return None
    if a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr() or b.storage_offset() != a.storage_offset() + a.numel():
        return None
    return torch.as_strided(a, (a.numel() + b.numel(),), (1,), a.storage_offset())


class Transport(object):
    def __init__(self, group=None, loopback=None):
        self.group = group
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else 'none'
        # loopback: with ONE rank, still send every message through the process group (to this rank itself) instead of the local copy
        if loopback is None:
            loopback = os.environ.get('NNS_COMM_LOOPBACK', '0') == '1'
        # (RCCL only: it carries a rank's messages to itself; gloo has no self-send -- a world-1 gloo group stays on local copies)
        self.loopback = bool(loopback) and self.P == 1 and dist.is_initialized() and self.backend == 'nccl'

    @property
    def local(self):
        """True when there is nobody to talk to (one rank, no loopback): messages are local copies."""
        return self.P == 1 and not self.loopback

    def _staged(self, t):
        return t.is_cuda and self.backend != 'nccl'

    # ------------------------------------------------------------------ point to point
    def sendrecv(self, sends, recvs):
        """sends / recvs: lists of (tensor, peer) in POSTING ORDER (two messages between one pair of ranks are matched in
        that order -- with two ranks both ring neighbours are the same peer).  Returns a handle; the receive buffers are
        valid after handle.wait()."""
        if self.local or not (sends or recvs):
            return _Done()
        if any(self._staged(t) for t, _ in sends + recvs):
            hs = [(t.cpu(), peer) for t, peer in sends]
            hr = [(torch.empty(t.shape, dtype=t.dtype), peer) for t, peer in recvs]
            reqs = [dist.isend(t, peer, group=self.group) for t, peer in hs] + [dist.irecv(t, peer, group=self.group) for t, peer in hr]
            for q in reqs:
                q.wait()
            for (dst, _), (src, _) in zip(recvs, hr):
                dst.copy_(src)
            return _Done()
        ops = [dist.P2POp(dist.isend, t, peer, self.group) for t, peer in sends] + [dist.P2POp(dist.irecv, t, peer, self.group) for t, peer in recvs]
        return _Works(dist.batch_isend_irecv(ops))

    def ring_exchange(self, first, last, from_down, from_up, wrap=True):
        """Every rank sends `first` to the rank before it and `last` to the rank after it; from_down receives the next
        rank's `first`, from_up the previous rank's `last`.  wrap=False: the ends of the chain have no neighbour there
        (pass None for the buffers that do not exist)."""
        P, r = self.P, self.rank
        if self.local:
            if wrap:
                from_up.copy_(last), from_down.copy_(first)
            return _Done()
        up, down = (r - 1) % P, (r + 1) % P
        has_up, has_down = wrap or r > 0, wrap or r < P - 1
        if wrap and up == down and not self.loopback:
            # two ranks on a periodic ring: both neighbours are the SAME peer.  Two messages each way would be told apart only by their posting order
            # (fine on gloo, documented for grouped NCCL point-to-point, never run between two GPUs here): when the buffers allow it -- `last` directly
            # behind `first` in memory, `from_up` behind `from_down` (SlabResidual allocates them so) -- ONE message each way carries both rows
            pair_s, pair_r = _adjacent(first, last), _adjacent(from_down, from_up)
            if pair_s is not None and pair_r is not None:
                return self.sendrecv([(pair_s, up)], [(pair_r, up)])           # the peer's (first, last) = my (row below, row above)
        sends, recvs = [], []
        if has_up:
            sends.append((first, up))
        if has_down:
            sends.append((last, down))
        # posting order of the receives = the peers' sending order (their `first`, then their `last`)
        if has_down:
            recvs.append((from_down, down))
        if has_up:
            recvs.append((from_up, up))
        return self.sendrecv(sends, recvs)

    # ------------------------------------------------------------------ collectives
    def all_to_all(self, recv, send):
        """recv[src] <- rank src's send[this rank] (equal splits along dim 0)."""
        if self.local:
            recv.copy_(send)
            return _Done()
        if self._staged(send):
            hs = send.cpu()
            hr = torch.empty_like(hs)
            dist.all_to_all_single(hr, hs, group=self.group)
            recv.copy_(hr)
            return _Done()
        w = dist.all_to_all_single(recv, send, group=self.group, async_op=True)
        return _Works([w])

    def all_reduce_(self, t, op):
        """In place, stream-ordered on nccl (no host synchronisation)."""
        if self.local:
            return t
        if self._staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
            return t
        dist.all_reduce(t, op=op, group=self.group)
        return t
