"""Model construction as the reference driver does it (GLASSTest.py:129-175 `buildModel`), through the
drop-in `impl.models` surface: EmbZGConv(GLASSConv layers, ELU, JK, GraphNorm) + nn.Linear head + pool."""
import functools

import torch.nn as nn


def build_glass(hidden, layers, max_deg, out_ch, aggr, pool, z_ratio, dropout=0.0, jk=True, pad_width=True):
    """pad_width: a hidden width no kernel family serves (33..63, 65..127, ...) is built at the next family width with
    zero padding — exact, same logical parameters, logical state_dict (glass_amd/widths.py); False keeps the width as it
    is (the per-op path with library GEMMs serves it)."""
    from impl import models
    pools = {"mean": models.MeanPool, "max": models.MaxPool, "sum": models.AddPool, "size": models.SizePool}
    if pool not in pools:
        raise NotImplementedError

    def build(h):
        conv = models.EmbZGConv(h, h, layers, max_deg=max_deg, activation=nn.ELU(inplace=True), jk=jk, dropout=dropout,
                                conv=functools.partial(models.GLASSConv, aggr=aggr, z_ratio=z_ratio, dropout=dropout),
                                gn=True)
        mlp = nn.Linear(h * layers if jk else h, out_ch)
        return models.GLASS(conv, nn.ModuleList([mlp]), nn.ModuleList([pools[pool]()]))

    if not pad_width:
        return build(hidden)
    from . import widths
    return widths.build_at_fused_width(hidden, build)
