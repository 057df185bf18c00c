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

    def __init__(self, params, average=True, extra=0):
        """extra: scalar slots appended to the bucket (`self.extra`, a view) that travel in the same all-reduce -- e.g. the
        local sum of squares of a norm loss (norm_loss_step)."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n + int(extra), dtype=dt, device=dev)
        self.extra = self.flat[n:]
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
            if self.flat.is_cuda and dist.get_backend() != 'nccl':          # gloo rehearsal on a one-GPU box: stage through the host
                h = self.flat.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                self.flat.copy_(h)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)                # RCCL: one ring all-reduce over xGMI, stream-ordered
            if self.average:
                self.flat.div_(dist.get_world_size())
        return self.flat


def norm_loss_step(model, bucket, optimizer, sumsq_fn):
    """One data-parallel optimiser step on the reference's objective  L = || pred - obs ||_2  over the WHOLE ensemble
    (src/neural_spectral/spectral_ode.py:178-190) when every rank holds a shard of it.  L is not additive over shards but
    L^2 is, and dL/dtheta = (1 / 2L) * sum_r d(ss_r)/dtheta: each rank back-propagates its local sum of squares ss_r =
    sumsq_fn() (e.g. PDEFunc.sumsq on its members), ONE all-reduce(SUM) carries the flat gradient bucket AND ss_r (an extra
    slot of the same buffer), and every rank applies the common factor 1 / (2 sqrt(sum_r ss_r)) on the device -- exactly the
    full-batch gradient, no host synchronisation.  `bucket` = FlatGradAllReduce(model.parameters(), average=False, extra=1).
    Returns the global loss (a device scalar)."""
    if bucket.average or bucket.extra.numel() < 1:
        raise ValueError("norm_loss_step needs FlatGradAllReduce(..., average=False, extra=1)")
    bucket.zero_()
    ss = sumsq_fn()
    ss.backward()
    bucket.extra[0] = ss.detach().to(bucket.flat.dtype)
    bucket.reduce_()
    loss = torch.sqrt(bucket.extra[0])
    bucket.flat[:bucket.flat.numel() - bucket.extra.numel()].mul_(0.5 / loss)
    optimizer.step()
    return loss


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
