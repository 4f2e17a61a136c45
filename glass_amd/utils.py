"""Label / index helpers with the reference's names (/root/reference/impl/utils.py:5-45)."""
import torch

from . import ops


def MaxZOZ(x, pos):
    """Max-zero-one labeling trick: z[n] = 1 iff node n belongs to any subgraph of the batch.
    `x` only supplies the node count and device; `pos` is the padded [B,Smax] node matrix (-1 pad).
    Runs the HIP kernel glass_maxzoz_i64 (memset + idempotent scatter)."""
    return ops.maxzoz(x.shape[0], pos)


def pad2batch(pad):
    """[[0,2,3],[1,4,5],[6,7,-1]] -> (batch=[0,0,0,1,1,1,2,2], pos=[0,2,3,1,4,5,6,7]): row-major
    flatten of the padded matrix with the -1 entries dropped."""
    rows = torch.arange(pad.shape[0], device=pad.device).unsqueeze(1).expand_as(pad)
    keep = pad >= 0
    return rows[keep], pad[keep]


def batch2pad(batch):
    """batch [0,1,0,0,1,1,2,2] -> pad [[0,2,3],[1,4,5],[6,7,-1]]: row i lists, in ascending order,
    the positions whose batch value is the i-th distinct non-negative value; -1 padding."""
    order = torch.argsort(batch, stable=True)
    order = order[batch[order] >= 0]
    vals, counts = torch.unique_consecutive(batch[order], return_counts=True)
    starts = torch.cumsum(counts, 0) - counts
    seg = torch.repeat_interleave(torch.arange(vals.shape[0], device=batch.device), counts)
    within = torch.arange(order.shape[0], device=batch.device) - starts[seg]
    pad = torch.full((vals.shape[0], int(counts.max())), -1, dtype=torch.int64, device=batch.device)
    pad[seg, within] = order
    return pad


class RuntimeCache(dict):
    """What the training / evaluation loops hang on a model (captured hipGraphs, the adopted optimizer engine): per-process
    runtime objects, not model state.  `copy.deepcopy(model)` (an early-stopping snapshot, a twin in a test) and pickling get
    an EMPTY cache instead of trying to copy device graphs — the copy simply builds its own on first use."""
    def __deepcopy__(self, memo):
        return RuntimeCache()

    def __reduce__(self):
        return (RuntimeCache, ())
