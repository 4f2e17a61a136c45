"""Epoch loops with the reference's interface (/root/reference/impl/train.py:4-34), plus the
data-parallel hook: when torch.distributed is initialised, gradients are averaged across ranks
through one flat bucket (glass_amd.dist) between backward() and optimizer.step()."""
import torch

from . import dist as gdist


def train(optimizer, model, dataloader, loss_fn):
    """One epoch; returns the mean per-step loss.  batch = (x, ei, ea, pos, [z,] y)."""
    model.train()
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
    return torch.stack(total_loss).mean().item()


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
