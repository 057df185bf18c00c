"""Data-parallel training over one process per GPU (BASELINE config 5: ensemble members /
initial conditions are sharded over ranks, parameters are replicated, gradients are summed).

xGMI is point-to-point: a ring all-reduce is bound by ONE link (~153 GB/s) and by per-collective latency, so all
parameter gradients travel as ONE flat bucket per step -- for cfg 5 that is ~24 k ODEFunc floats + K*3*nx*ny basis
floats (7.9 MB at K = 10, 256^2) in a single RCCL call instead of one call per tensor.  `backend='nccl'` is RCCL on ROCm;
the same code runs on gloo (tests/test_data_parallel_gloo.py)."""
import torch
import torch.distributed as dist


class FlatGradAllReduce(object):
    """Owns one flat buffer aliased by every parameter's .grad, so the reduction needs no packing copy."""

    def __init__(self, params, average=True):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        off = 0
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("parameters must share one device and dtype")
            p.grad = self.flat[off:off + p.numel()].view_as(p)          # .grad is a VIEW of the bucket
            off += p.numel()
        self.average = average

    def zero_(self):
        self.flat.zero_()

    def reduce_(self):
        """Sum (or average) the bucket over all ranks, in place; a no-op without a process group."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if self.average:
                self.flat.div_(dist.get_world_size())
        return self.flat


def broadcast_parameters(params, src=0):
    """Replicate rank `src`'s parameters (start of training)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for p in params:
            dist.broadcast(p.data, src)


def shard(tensor, dim=0):
    """This rank's contiguous share of `tensor` along `dim` (ensemble members / initial conditions)."""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    w, r = dist.get_world_size(), dist.get_rank()
    n = tensor.shape[dim]
    if n % w:
        raise ValueError("size %d along dim %d does not divide over %d ranks" % (n, dim, w))
    return tensor.narrow(dim, r * (n // w), n // w)
