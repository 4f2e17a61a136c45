"""Model construction as the reference driver does it (GLASSTest.py:129-175 `buildModel`), through the
drop-in `impl.models` surface: EmbZGConv(GLASSConv layers, ELU, JK, GraphNorm) + nn.Linear head + pool."""
import functools

import torch.nn as nn


def build_glass(hidden, layers, max_deg, out_ch, aggr, pool, z_ratio, dropout=0.0, jk=True):
    from impl import models
    conv = models.EmbZGConv(hidden, hidden, layers, max_deg=max_deg, activation=nn.ELU(inplace=True), jk=jk,
                            dropout=dropout,
                            conv=functools.partial(models.GLASSConv, aggr=aggr, z_ratio=z_ratio, dropout=dropout),
                            gn=True)
    mlp = nn.Linear(hidden * layers if jk else hidden, out_ch)
    pools = {"mean": models.MeanPool, "max": models.MaxPool, "sum": models.AddPool, "size": models.SizePool}
    if pool not in pools:
        raise NotImplementedError
    return models.GLASS(conv, nn.ModuleList([mlp]), nn.ModuleList([pools[pool]()]))
