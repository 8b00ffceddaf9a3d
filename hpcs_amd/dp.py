"""Data parallelism for the HPCS step: one process per GPU, one gradient all-reduce per step over RCCL/xGMI.

The reference has no distributed code at all (no strategy/devices on its Trainer); what its
Lightning stack would do with devices>1 is DDP without SyncBN, which is what this reproduces: clouds are sharded
across ranks, BatchNorm statistics and triplet mining stay per-replica, and only the 1.3 M-float gradient is
exchanged.  All .grad tensors are views into ONE flat fp32 bucket, so the collective needs no pack/unpack:
autograd accumulates straight into the bucket, the bucket is all-reduced in place and scaled by 1/world.
"""
import torch
import torch.distributed as dist

from . import _lib


class FlatGradBucket:
    def __init__(self, params, adjacent=()):
        """adjacent: pairs (p, q) of parameters whose gradients should lie back to back in the bucket (q right after p), so
        that one GEMM can write both -- the feature and direction weights of a VN layer are the two row blocks of ONE
        stacked weight-gradient GEMM (ops.VNLayerFn).  The layout of the bucket is free: the optimiser walks its own
        parameter list, the all-reduce sees one flat buffer either way."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        follower = {id(p): q for p, q in adjacent if p.requires_grad and q.requires_grad}
        followed = {id(q) for q in follower.values()}
        order = []
        for p in self.params:
            if id(p) in followed:
                continue
            order.append(p)
            if id(p) in follower:
                order.append(follower[id(p)])
        assert len(order) == len(self.params) and len({id(p) for p in order}) == len(order)
        off = 0
        for p in order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        self.numel = total

    def zero(self):
        self.flat.zero_()
        self.check_attached()

    def check_attached(self):
        for p in self.params:          # every .grad must still alias the flat bucket
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                    p.grad.data_ptr() >= self.flat.data_ptr() + 4 * self.numel:
                raise RuntimeError("a parameter's .grad no longer aliases the flat bucket (use bucket.zero(), not "
                                   "optimizer.zero_grad(set_to_none=True))")

    def all_reduce_mean(self, group=None):
        """Sum over ranks, then scale by 1/world (in place).  No-op when not distributed."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if self.flat.is_cuda:
            _lib.call("hpcs_scale_inplace", self.flat, self.numel, 1.0 / world)
        else:                              # gloo / CPU rehearsal of the collective path (tests only)
            self.flat.mul_(1.0 / world)


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, local_rank, world)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("HPCS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            # "nccl" is RCCL on ROCm; HPCS_DIST_BACKEND=gloo rehearses the multi-rank path on one GPU
        if backend == "nccl":
            # (HPCS_SINGLE_DEVICE_REHEARSAL: every rank on cuda:0 -- RCCL itself decides whether it accepts that)
            torch.cuda.set_device(0 if os.environ.get("HPCS_SINGLE_DEVICE_REHEARSAL") else local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def broadcast_parameters(module, src=0):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


def broadcast_buffers(module, src=0):
    """Re-synchronise the BatchNorm running statistics from rank `src` (what DDP's broadcast_buffers=True does every
    forward; the step driver keeps them per-replica between calls -- call this before a checkpoint or an evaluation)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in module.buffers():
        dist.broadcast(t.data, src=src)
