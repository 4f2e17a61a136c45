"""Epoch loops with the reference's interface (/root/reference/impl/train.py:4-34), plus the
data-parallel hook: when torch.distributed is initialised, gradients are averaged across ranks
through one flat bucket (glass_amd.dist) between backward() and optimizer.step()."""
import os

import torch

from . import dist as gdist

USE_GRAPH = os.environ.get("GLASS_TRAIN_GRAPH", "1") != "0"   # 0: the same training step, eager launches instead of a replay
USE_STEP = os.environ.get("GLASS_TRAIN_STEP", "1") != "0"     # 0: the plain per-batch loop below (autograd, optimizer.step())


def _graph_step(optimizer, model, dataloader, loss_fn):
    """The training step object (glass_amd.step.TrainStep: the step program, replayed from a hipGraph unless
    GLASS_TRAIN_GRAPH=0) when the epoch is graph-safe: a GLASS model on the GPU,
    ZGDataloader with z_fn = MaxZOZ and drop_last (fixed batch shape), and an optimizer whose step is capturable
    (FlatAdam: its learning rate lives in device memory, so schedulers keep working under replay).  Cached on the model.  None -> eager loop."""
    from . import utils
    from .SubGDataset import ZGDataloader
    from .models import GLASS
    from .optim import FlatAdam
    if not (USE_STEP and isinstance(model, GLASS) and isinstance(dataloader, ZGDataloader) and
            dataloader.z_fn is utils.MaxZOZ and dataloader.drop_last and dataloader.Gdataset.x.is_cuda and
            len(dataloader) > 0):
        return None
    if not isinstance(optimizer, FlatAdam):
        # FlatAdam mirrors param_groups[...]["lr"] into device memory before every replay (sync_lr); a torch optimizer's
        # Python-float lr would be baked into the captured graph and a scheduler's changes silently ignored
        return None
    ds = dataloader.Gdataset
    key = (id(optimizer), id(loss_fn), id(ds.x), id(ds.edge_index), dataloader.batch_size, gdist.world_size(), USE_GRAPH)
    cache = model.__dict__.setdefault("_glass_train_steps", {})
    step = cache.get(key)
    if step is None:
        from .step import TrainStep
        step = TrainStep(model, optimizer, loss_fn, ds.x, ds.edge_index, ds.edge_attr, gdist.bucket_for(model),
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
        total, n = None, 0
        for batch in dataloader:
            loss = step(batch[3], batch[-1])  # z is recomputed inside the captured step (MaxZOZ kernel)
            total = loss.clone() if total is None else total.add_(loss)
            n += 1
        return _epoch_loss(total / n, dataloader)
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


@torch.no_grad()
def test(model, dataloader, metrics, loss_fn):
    """Evaluate: returns (metric(pred, y), loss)."""
    model.eval()
    preds, ys = [], []
    for batch in dataloader:
        preds.append(model(*batch[:-1]))
        ys.append(batch[-1])
    pred, y = torch.cat(preds, dim=0), torch.cat(ys, dim=0)
    return metrics(pred.cpu().numpy(), y.cpu().numpy()), loss_fn(pred, y)
