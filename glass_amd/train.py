"""Epoch loops with the reference's interface (/root/reference/impl/train.py:4-34), plus the
data-parallel hook: when torch.distributed is initialised, gradients are averaged across ranks
through one flat bucket (glass_amd.dist) between backward() and optimizer.step()."""
import os

import torch

from . import dist as gdist

USE_GRAPH = os.environ.get("GLASS_TRAIN_GRAPH", "1") != "0"   # 0: the same training step, eager launches instead of a replay
USE_STEP = os.environ.get("GLASS_TRAIN_STEP", "1") != "0"     # 0: the plain per-batch loop below (autograd, optimizer.step())
USE_HEAD_LABELS = os.environ.get("GLASS_HEAD_LABELS", "1") != "0"  # 0: the label launch eager in front of every replay


def _graph_step(optimizer, model, dataloader, loss_fn):
    """The training step object (glass_amd.step.TrainStep: the step program, replayed from a hipGraph unless
    GLASS_TRAIN_GRAPH=0) when the epoch is graph-safe: a GLASS model on the GPU, ZGDataloader with z_fn = MaxZOZ and
    drop_last (fixed batch shape), and an optimizer whose step is capturable — optim.FlatAdam, or a plain
    `torch.optim.Adam(model.parameters(), lr)` as the reference driver builds it (GLASSTest.py:213), which is taken over
    in place (optim.adopt: flat parameter arena built on the spot, the torch optimizer's param_groups / state stay the
    interface, so `ReduceLROnPlateau(optimizer)` and `optimizer.state_dict()` keep working).  Cached on the model.
    None -> eager loop."""
    from . import utils
    from .SubGDataset import ZGDataloader
    from .models import GLASS
    from .optim import FlatAdam, adopt, adoptable
    if not (USE_STEP and isinstance(model, GLASS) and isinstance(dataloader, ZGDataloader) and
            dataloader.z_fn is utils.MaxZOZ and dataloader.drop_last and dataloader.Gdataset.x.is_cuda and
            len(dataloader) > 0):
        return None
    engine = optimizer
    if not isinstance(optimizer, FlatAdam):
        # a torch optimizer's Python-float lr would be baked into a captured graph and a scheduler's changes silently ignored:
        # the flat engine mirrors param_groups[0]["lr"] into device memory before every replay (sync_lr)
        if adoptable(optimizer, model) is not None:
            return None
        engine = adopt(optimizer, model)
    elif not optimizer.arena.attached():
        optimizer.arena.reattach()
    ds = dataloader.Gdataset
    key = (id(optimizer), id(engine), id(loss_fn), id(ds.x), id(ds.edge_index), dataloader.batch_size, gdist.world_size(), USE_GRAPH)
    cache = model.__dict__.setdefault("_glass_train_steps", utils.RuntimeCache())
    step = cache.get(key)
    if step is None:
        from .step import TrainStep
        step = TrainStep(model, engine, loss_fn, ds.x, ds.edge_index, ds.edge_attr, gdist.bucket_for(model),
                         use_graph=USE_GRAPH, warmup_iters=2, preserve_state=True)  # GLASS_TRAIN_GRAPH=0: the same step, eager launches
        cache.clear()  # one live graph per model
        cache[key] = step
    return step


def _epoch_loss(mean_loss, dataloader):
    """The value train() returns: with a sharded (data-parallel) loader the mean over ranks, so that the scheduler
    and the early-stop logic of the driver see the same number on every rank."""
    if gdist.is_distributed() and getattr(dataloader, "shard", False):
        import torch.distributed as td
        t = mean_loss.detach().clone().reshape(1)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        mean_loss = t[0] / gdist.world_size()
    return mean_loss.item()


def train(optimizer, model, dataloader, loss_fn):
    """One epoch; returns the mean per-step loss.  batch = (x, ei, ea, pos, [z,] y)."""
    model.train()
    step = _graph_step(optimizer, model, dataloader, loss_fn)
    if step is not None:
        # The loader's own iteration would select the batch (`pos[perm], y[perm]`: two index kernels) and label it (z_fn =
        # MaxZOZ) before the step copies and labels it again; the step does all of that in its one label launch from the index
        # batch (TrainStep.__call__(..., index)), and keeps the running loss sum itself — per step: that launch + one replay.
        from .SubGDataset import ZGDataloader
        ds = dataloader.Gdataset
        step.reset_loss_sum()
        n = 0
        if type(dataloader) is ZGDataloader:
            batches = list(dataloader._batches())
            # the whole epoch's index batches up front: the step's head launch then selects and labels its batch itself through
            # a device-resident cursor (TrainStep.begin_epoch: one launch less per step, nothing in front of the replay)
            if USE_HEAD_LABELS and batches and step.begin_epoch(ds.pos, ds.y, torch.stack(batches)):
                for _ in batches:
                    step.next_step()
                    n += 1
            else:
                for perm in batches:
                    step(ds.pos, ds.y, perm)
                    n += 1
        else:  # a subclass may select its batches differently: take them as it yields them
            for batch in dataloader:
                step(batch[3], batch[-1])
                n += 1
        if n == 0:
            return float("nan")
        out = _epoch_loss(step.loss_sum() / n, dataloader)   # (the epoch's one host sync)
        if hasattr(step.opt, "publish"):
            step.opt.publish()   # adopted torch.optim.Adam: state[p]["step"] follows the device counter
        return out
    total_loss = []
    bucket = gdist.bucket_for(model) if gdist.is_distributed() else None
    for batch in dataloader:
        if bucket is None:
            optimizer.zero_grad()
        else:
            bucket.zero()
        pred = model(*batch[:-1], id=0)
        loss = loss_fn(pred, batch[-1])
        loss.backward()
        if bucket is not None:
            bucket.all_reduce_mean()
        total_loss.append(loss.detach())
        optimizer.step()
    # one host sync per epoch instead of the reference's .item() per step (train.py:15)
    return _epoch_loss(torch.stack(total_loss).mean(), dataloader)


USE_EVAL_GRAPH = os.environ.get("GLASS_EVAL_GRAPH", "1") != "0"  # 0: one eager forward per evaluation batch
EVAL_PARALLEL = int(os.environ.get("GLASS_EVAL_PARALLEL", "8"))   # evaluation batches run side by side (evalstep.EvalGraph)


EVAL_GRAPH_MAX_BYTES = int(os.environ.get("GLASS_EVAL_GRAPH_MAX_MB", "8192")) << 20  # activation budget of one cached graph


def _eval_branches(model, n_nodes, k):
    """(parallel branches of one evaluation graph, estimated bytes per branch), bounded by memory: every branch keeps its own
    copy of the forward's activations (~ (4 L + 6) [N, H] fp32 buffers).  (All evaluation graphs of a device share one
    memory pool — evalstep._pool — so neither the up-to-4 cached graphs of a model nor the parked execs of dropped ones add
    up; evalstep.retired_bytes() reports what parked execs pin beyond that: nothing.)"""
    from . import evalstep
    try:
        emb = model.conv
        H = emb.input_emb.weight.shape[1]
        L = len(emb.convs)
    except AttributeError:
        return k, 0
    per_branch = 4 * n_nodes * H * (4 * L + 6)
    budget = EVAL_GRAPH_MAX_BYTES - evalstep.retired_bytes()
    return max(1, min(k, budget // max(per_branch, 1))), per_branch


def _eval_graph(model, batch, k):
    """The cached evalstep.EvalGraph for this model / graph tensors / batch shape, or None when the batch is not the plain
    GLASS evaluation call (x, ei, ea, pos, z) on the GPU — or when its capture failed before (remembered per key: the eager
    forward then serves that shape)."""
    from .models import GLASS
    from .utils import RuntimeCache
    if not (USE_EVAL_GRAPH and k > 1 and isinstance(model, GLASS) and len(batch) == 5 and batch[0].is_cuda and
            batch[3].dim() == 2 and batch[3].dtype == torch.int64):
        return None
    x, ei, ea, pos = batch[0], batch[1], batch[2], batch[3]
    k, per_branch = _eval_branches(model, x.shape[0], k)
    if k <= 1:
        return None
    key = (id(x), id(ei), id(ea), tuple(pos.shape), k)
    cache = model.__dict__.setdefault("_glass_eval_graphs", RuntimeCache())
    if key in cache:
        return cache[key]  # (None: a capture of this shape failed earlier)
    from .evalstep import EvalGraph
    if len(cache) >= 4:
        cache.clear()
    g = EvalGraph(model, x, ei, ea, pos.shape, k, est_bytes=k * per_branch)
    try:
        g.capture()
    except Exception as e:  # noqa: BLE001 — whatever the runtime or a non-capturable op refuses: the eager loop still works
        torch.cuda.synchronize()
        import warnings
        warnings.warn(f"glass_amd.train.test: evaluation graph not captured ({e!r}); eager forwards for batches of shape "
                      f"{tuple(pos.shape)}")
        g = None
    cache[key] = g
    return g


@torch.no_grad()
def test(model, dataloader, metrics, loss_fn):
    """Evaluate: returns (metric(pred, y), loss).  The batches are independent forward passes (no parameter changes in
    between, reference impl/train.py:20-34): on the GPU, runs of EVAL_PARALLEL equal-shaped batches replay ONE hipGraph whose
    branches are the batches (evalstep.EvalGraph: most of the chip idles during a single small forward); a ragged tail and
    anything else takes the plain per-batch forward.  Same kernels either way: bitwise the same predictions."""
    from . import utils
    model.eval()
    preds, ys = [], []
    pending = []  # equal-shaped batches waiting for a parallel replay
    # (the graph's branches label their batch with utils.MaxZOZ themselves: only loaders that do the same qualify)
    maxzoz = getattr(dataloader, "z_fn", None) is utils.MaxZOZ

    def flush():
        if not pending:
            return
        g = _eval_graph(model, pending[0][:-1], EVAL_PARALLEL) if (maxzoz and len(pending) > 1) else None
        if g is None:
            preds.extend(model(*b[:-1]) for b in pending)  # (b = (x, ei, ea, pos, z, y))
        else:
            for i in range(0, len(pending), g.k):
                grp = pending[i:i + g.k]
                preds.extend(o.clone() for o in g([b[3] for b in grp]))
        pending.clear()

    for batch in dataloader:
        ys.append(batch[-1])
        if pending and (not maxzoz or batch[3].shape != pending[0][3].shape or batch[0] is not pending[0][0]):
            flush()
        pending.append(batch)
        if len(pending) >= 4 * EVAL_PARALLEL:
            flush()
    flush()
    pred, y = torch.cat(preds, dim=0), torch.cat(ys, dim=0)
    return metrics(pred.cpu().numpy(), y.cpu().numpy()), loss_fn(pred, y)
