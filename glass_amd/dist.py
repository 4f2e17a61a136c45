"""Subgraph-batch data parallelism — the one multi-GPU scheme this path admits.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU
for tests).  The graph is replicated; each rank takes a disjoint slice of every subgraph batch,
labels / pools only its slice, runs the full-graph forward/backward, and the ranks exchange ONE
collective per step: an all-reduce of a single flat fp32 gradient bucket, then divide by world
size.  There is no activation or halo exchange (GraphNorm statistics are over the replicated
graph, identical in structure on every rank).

Semantics (stated wherever results are reported): k ranks x batch b  ==  the reference run with
batch b and gradients averaged over k consecutive steps — NOT the reference at batch k*b, because
max-zero-one labels are per batch (impl/utils.py:40-45).
"""
import torch
import torch.distributed as td


def is_distributed():
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def rank():
    return td.get_rank() if is_distributed() else 0


def world_size():
    return td.get_world_size() if is_distributed() else 1


def shard(index_batch):
    """This rank's slice of a batch of subgraph indices: perm[rank::world]."""
    if not is_distributed():
        return index_batch
    return index_batch[td.get_rank()::td.get_world_size()]


def broadcast_cpu(t):
    """Make rank 0's CPU tensor (a shuffle permutation) the one every rank uses."""
    if not is_distributed():
        return t
    if td.get_backend() == "nccl":
        d = t.to(torch.device("cuda", torch.cuda.current_device()))
        td.broadcast(d, 0)
        return d.cpu()
    t = t.clone()
    td.broadcast(t, 0)
    return t


class FlatGradBucket:
    """All parameter gradients as views into one contiguous fp32 buffer, so the step's only
    collective is a single all-reduce (payload 4 * #params bytes: ~0.2 MB at H=64, L=2)."""
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        first = self.params[0]
        self.flat = torch.zeros(total, dtype=first.dtype, device=first.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def attached(self):
        """True while every .grad still aliases the bucket (zero_grad(set_to_none=True) detaches)."""
        base = self.flat.untyped_storage().data_ptr()
        return all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in self.params)

    def zero(self):
        if not self.attached():
            self.__init__(self.params)
        self.flat.zero_()

    def all_reduce_mean(self):
        if not is_distributed():
            return
        if td.get_backend() == "nccl":
            td.all_reduce(self.flat, op=td.ReduceOp.AVG)  # RCCL averages in the collective: no extra kernel
        else:
            td.all_reduce(self.flat, op=td.ReduceOp.SUM)
            self.flat.div_(td.get_world_size())


def bucket_for(model):
    b = getattr(model, "_glass_grad_bucket", None)
    if b is None:
        b = FlatGradBucket(list(model.parameters()))
        model._glass_grad_bucket = b
    return b
