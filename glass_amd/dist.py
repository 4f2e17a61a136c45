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


class GradExchange:
    """The step's collective over a flat gradient buffer cut in two buckets:

      small = flat[0:big_start]       layer / head parameters (0.2 - 3 MB): all-reduce(mean) — latency-bound
      big   = flat[big_start:total]   embedding-sized gradients (`--use_nodeid`: the [N,H] table of
                                      GLASSTest.py:153-157 — 25.6 MB at C4, 1 GB at C5): reduce-scatter(mean) ->
                                      the optimizer updates only THIS rank's 1/world shard of the table -> all-gather
                                      of the updated parameter shards.  Against a full all-reduce that is the same
                                      bytes on each xGMI link for the gradient (reduce-scatter is its first half),
                                      half of them replaced by parameters, and 1/world of the Adam work and state.

    Works on any backend torch.distributed offers (RCCL in the product, gloo in the CPU tests).  `big` must be a multiple
    of the world size (ParamArena pads).  Elementwise optimizers (Adam) do not care that a shard cuts through a row."""
    def __init__(self, flat_grad, flat_param, big_start):
        self.grad, self.param = flat_grad, flat_param
        self.total = flat_grad.numel()
        self.big_start = int(big_start)
        self.world, self.rank = world_size(), rank()
        big = self.total - self.big_start
        if big % self.world:
            raise ValueError(f"big bucket ({big} elements) must be a multiple of the world size {self.world}")
        self.shard_len = big // self.world
        self.shard_lo = self.big_start + self.rank * self.shard_len
        self.shard_grad = torch.empty(self.shard_len, dtype=flat_grad.dtype, device=flat_grad.device) if big else None
        self._inplace = td.get_backend() == "nccl"  # RCCL supports output = slice of input; gloo gets a staging copy

    @property
    def has_big(self):
        return self.total > self.big_start

    def _mean(self, t):
        if td.get_backend() == "nccl":
            return td.ReduceOp.AVG, None
        return td.ReduceOp.SUM, self.world

    def reduce_small(self):
        if self.big_start == 0:
            return
        small = self.grad[:self.big_start]
        op, div = self._mean(small)
        td.all_reduce(small, op=op)
        if div:
            small.div_(div)

    def reduce_big(self):
        """-> shard_grad: this rank's slice of the mean gradient of the big bucket."""
        if not self.has_big:
            return None
        big = self.grad[self.big_start:]
        op, div = self._mean(big)
        td.reduce_scatter_tensor(self.shard_grad, big, op=op)
        if div:
            self.shard_grad.div_(div)
        return self.shard_grad

    def shard_views(self, *flats):
        """This rank's shard of other flat buffers laid out like the arena (parameters, optimizer state)."""
        return tuple(f[self.shard_lo:self.shard_lo + self.shard_len] for f in flats)

    def gather_params(self):
        """After the optimizer has updated this rank's parameter shard: every rank receives every shard."""
        if not self.has_big:
            return
        big = self.param[self.big_start:]
        mine = self.param[self.shard_lo:self.shard_lo + self.shard_len]
        td.all_gather_into_tensor(big, mine if self._inplace else mine.clone())

    def payload_bytes(self):
        es = self.grad.element_size()
        return {"small_allreduce": self.big_start * es, "big_reduce_scatter": (self.total - self.big_start) * es,
                "big_all_gather": (self.total - self.big_start) * es}


def bucket_for(model):
    b = getattr(model, "_glass_grad_bucket", None)
    if b is None:
        b = FlatGradBucket(list(model.parameters()))
        model._glass_grad_bucket = b
    return b


# ---- what the exchange should cost: a stated model, to hold the first multi-rank measurement against ------------------
# One node, N GPUs, every pair joined by one xGMI link (point to point, ~153 GB/s per direction per link; 7 links per GPU).
# RCCL runs small all-reduces as a ring (or a tree of similar depth) of 2 (N - 1) dependent steps.  The model:
#     t(bytes, N) = T_LAUNCH + steps(N) * T_HOP + wire_bytes(bytes, N) / (LINK_GBPS * LINK_EFF)
# with steps = 2 (N - 1) for all-reduce, (N - 1) for reduce-scatter / all-gather (the latency of a ring: the pessimistic
# choice), and — the mesh is point to point, so a rank's N - 1 links carry its N - 1 shards side by side — wire bytes per
# link of bytes / N per half: 2 bytes / N for an all-reduce, bytes / N for a reduce-scatter or all-gather.  T_LAUNCH (the collective's kernel launch + its
# flag handshake) and T_HOP (one dependent store -> remote poll -> reduce hop over xGMI) are ASSUMPTIONS taken from typical
# small-message all-reduce latencies on 8-GPU xGMI nodes (20-40 us below 1 MB); no run in this repository has measured
# them — they are printed with every prediction so that a SCALE run can replace them.
MODEL_T_LAUNCH_US = 12.0
MODEL_T_HOP_US = 1.5
MODEL_LINK_GBPS = 153.0
MODEL_LINK_EFF = 0.7


def predict_collective_us(payload_bytes, world):
    """{"small_allreduce", "big_reduce_scatter", "big_all_gather": bytes} -> predicted device time of the step's exchange at
    `world` ranks under the model above: per-collective times, their sum, and the model's parameters."""
    n = int(world)
    out = {"world": n, "model": {"t_launch_us": MODEL_T_LAUNCH_US, "t_hop_us": MODEL_T_HOP_US, "link_GBps": MODEL_LINK_GBPS,
                                "link_efficiency": MODEL_LINK_EFF,
                                "form": "t = t_launch + steps * t_hop + wire_bytes / (link_GBps * link_efficiency); all-reduce: 2(N-1) "
                                        "dependent steps, 2 * bytes / N per link (N-1 links side by side); reduce-scatter / all-gather "
                                        "half of both",
                                "status": "assumed constants, not measured on this pool"}}
    if n <= 1:
        out.update(small_allreduce_us=0.0, big_reduce_scatter_us=0.0, big_all_gather_us=0.0, total_us=0.0)
        return out
    bw = MODEL_LINK_GBPS * MODEL_LINK_EFF * 1e3  # bytes per us

    def t(nbytes, steps, wire):
        return 0.0 if nbytes <= 0 else MODEL_T_LAUNCH_US + steps * MODEL_T_HOP_US + wire * nbytes / bw

    small = t(payload_bytes.get("small_allreduce", 0), 2 * (n - 1), 2.0 / n)
    rs = t(payload_bytes.get("big_reduce_scatter", 0), n - 1, 1.0 / n)
    ag = t(payload_bytes.get("big_all_gather", 0), n - 1, 1.0 / n)
    out.update(small_allreduce_us=small, big_reduce_scatter_us=rs, big_all_gather_us=ag, total_us=small + rs + ag)
    return out


# The one-shot form of the small-bucket exchange (glass_amd/peer.py, csrc/peer_allreduce.hip): every rank reads its N - 1 peers'
# gradient arenas through xGMI peer mappings — N - 1 links side by side, ONE dependent hop (flag store -> remote poll) instead of
# a ring's 2 (N - 1) — reduces in fixed rank order and applies Adam in the same launch.  Same status: assumed constants.
MODEL_T_ONESHOT_FLAG_US = 4.0   # publish "my gradients are final" + see every peer's flag (one store + one polled read over xGMI)


def predict_oneshot_us(payload_bytes, world):
    """Predicted EXTRA device time of the one-shot fused all-reduce + Adam over the plain (1-rank) Adam launch it replaces."""
    n = int(world)
    if n <= 1:
        return 0.0
    nbytes = payload_bytes.get("small_allreduce", 0)
    bw = MODEL_LINK_GBPS * MODEL_LINK_EFF * 1e3
    return MODEL_T_ONESHOT_FLAG_US + nbytes / bw   # each peer's arena crosses its own link once


def predict_scaling(payload_bytes, step_ms_1gpu, overlaps_small=False, worlds=(1, 2, 4, 8)):
    """The forecast a SCALE run is to be held against (weak scaling, replicated graph: per-rank work is the 1-GPU step): per
    world size the exchange time under the ring model, what of it is exposed (all of it unless the small bucket overlaps the
    backward tail — embedding-sized bucket only), the predicted step time and efficiency = step / (step + exposed); the same
    for the one-shot fused form."""
    out = []
    for n in worlds:
        p = predict_collective_us(payload_bytes, n)
        exposed = p["total_us"] - (p["small_allreduce_us"] if overlaps_small else 0.0)
        one = predict_oneshot_us(payload_bytes, n) + p["big_reduce_scatter_us"] + p["big_all_gather_us"]
        step_us = step_ms_1gpu * 1e3
        out.append({"world": n, "ring_exchange_us": p["total_us"], "ring_exposed_us": exposed,
                    "ring_efficiency": step_us / (step_us + exposed) if step_us > 0 else None,
                    "oneshot_exposed_us": one, "oneshot_efficiency": step_us / (step_us + one) if step_us > 0 else None})
    return {"per_world": out, "step_ms_1gpu": step_ms_1gpu,
            "model": {"t_launch_us": MODEL_T_LAUNCH_US, "t_hop_us": MODEL_T_HOP_US, "link_GBps": MODEL_LINK_GBPS,
                      "link_efficiency": MODEL_LINK_EFF, "t_oneshot_flag_us": MODEL_T_ONESHOT_FLAG_US,
                      "status": "assumed constants, not measured on this pool (no multi-GPU run has been available)"}}
